"""CPU: the surface-pass oracle (oracle/oracle_train.cpp: surface) against torch float64 autograd golden vectors of the
reference's op chain (tests/golden/surface_golden.npz, tests/golden/make_surface_golden.py)."""
import os
import numpy as np
import pytest

from oracle import oracle as orc

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "surface_golden.npz"))


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_surface_oracle_matches_torch_golden(tag):
    am, ray, ratio = G[f"{tag}_allmap"], G[f"{tag}_raymat"], float(G[f"{tag}_ratio"])
    sd, sn, gam = orc.surface_pass(am, ray, ratio, G[f"{tag}_g_sd"][0], G[f"{tag}_g_sn"], dtype=np.float64)
    assert np.allclose(sd, G[f"{tag}_surf_depth"][0], rtol=1e-12, atol=1e-12)
    assert np.abs(sn - G[f"{tag}_surf_normal"]).max() < 1e-9
    ref = G[f"{tag}_g_allmap"]
    # an infinite depth sum gives 0 * inf = NaN in the alpha gradient in torch as well: same places, same values elsewhere
    assert (np.isfinite(gam) == np.isfinite(ref)).all()
    ok = np.isfinite(ref)
    assert np.abs(gam[ok] - ref[ok]).max() <= 1e-9 * max(1.0, np.abs(ref[ok]).max())
    assert np.abs(gam[[2, 3, 4, 6, 7]]).max() == 0 and (~ok).sum() <= 2
