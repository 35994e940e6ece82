"""MI355X drop-in for the reference's `cubemapencoder` package (submodules/cubemapencoder)."""
from .cubemap_encoder import CubemapEncoder, MipCubemapEncoder, cubemap_encode, _backend  # noqa: F401
