"""Build libgsr_hip.so (gfx950 only) in-tree with hipcc.

    python gaussian-splatting-reflection_amd/csrc/build.py [--force]

Each .hip translation unit is compiled to an object file in parallel (hipcc cross-compiles without a
GPU) and linked into gaussian-splatting-reflection_amd/libgsr_hip.so.  The .so is git-ignored but travels with
the tree to the GPU box.
"""
import concurrent.futures
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OUT = os.path.join(PKG, "libgsr_hip.so")
OBJ_DIR = os.path.join(HERE, "_obj")
SOURCES = ["gsr_common.hip", "gsr_gauss.hip", "gsr_surfel.hip", "gsr_cubemap.hip", "gsr_train.hip", "gsr_surface.hip", "gsr_densify.hip"]
HEADERS = ["gsr_internal.hpp", "gsr_math.hpp", "gsr_sort.hpp", os.path.join("..", "..", "include", "gsr_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fhip-fp32-correctly-rounded-divide-sqrt is hipcc's default; stated because parity of the integer outputs
# (radii, tile rects, sort keys) relies on IEEE division and square root in the per-Gaussian kernels.
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-fhip-fp32-correctly-rounded-divide-sqrt",
         "-Wall", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-but-set-variable"]


# Per-file flags.  The tile kernels keep their per-Gaussian record in SGPRs; clang's SLP vectoriser turns pairs of
# scalar FMAs into v_pk_* instructions whose SGPR operands must be even-aligned pairs, so it re-packs the freshly loaded
# record with ~25 s_mov per (wave, Gaussian) pair and waits for the scalar load immediately instead of one pair later.
# Scalar issue slots are as scarce as vector ones here, so SLP is off for both variants; the packed math that pays
# is written explicitly on <2 x float> against a record laid out in aligned pairs (gsr_surfel.hip).  Measured at 1 M
# Gaussians, variant G: fwd 0.27 / 0.30 ms, bwd 0.70 / 0.63 ms with / without SLP; variant S: see DESIGN.md.
EXTRA_FLAGS = {"gsr_surfel.hip": ["-fno-slp-vectorize"], "gsr_gauss.hip": ["-fno-slp-vectorize"]}
if os.environ.get("GSR_SLP") == "1":      # development switch for A/B measurements
    EXTRA_FLAGS = {}


def _digest():
    h = hashlib.sha256()
    for f in SOURCES + HEADERS + ["build.py"]:
        p = os.path.join(HERE, f)
        if os.path.exists(p):
            with open(p, "rb") as fh:
                h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(EXTRA_FLAGS.items())).encode())
    return h.hexdigest()


def _compile(src):
    obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
    cmd = [HIPCC] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(HERE, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force=False, verbose=True):
    os.makedirs(OBJ_DIR, exist_ok=True)
    stamp = os.path.join(OBJ_DIR, "digest.txt")
    dig = _digest()
    if not force and os.path.exists(OUT) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return OUT
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(HERE, s))]
    with concurrent.futures.ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(_compile, srcs))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as fh:
        fh.write(dig)
    if verbose:
        print(f"built {OUT}")
    return OUT


BINDING_OUT = os.path.join(PKG, "_gsr_C.so")


def build_binding(force=False, verbose=True):
    """Optional compiled torch/pybind binding (gsr_torch_binding.cpp -> _gsr_C.so next to libgsr_hip.so): marshaling only, links
    the C-ABI library.  The package works without it (ctypes, _gsr.py); GSR_BINDING=pybind selects it."""
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    build(verbose=verbose)
    src = os.path.join(HERE, "gsr_torch_binding.cpp")
    h = hashlib.sha256(open(src, "rb").read() + open(os.path.join(HERE, "..", "..", "include", "gsr_hip.h"), "rb").read() + torch.__version__.encode())
    stamp = os.path.join(OBJ_DIR, "binding_digest.txt")
    if not force and os.path.exists(BINDING_OUT) and os.path.exists(stamp) and open(stamp).read().strip() == h.hexdigest():
        return BINDING_OUT
    inc = ce.include_paths() + [sysconfig.get_paths()["include"], "/opt/rocm/include"]
    libdirs = ce.library_paths()
    cmd = [HIPCC, "-x", "c++", "-O2", "-fPIC", "-shared", "-std=c++17", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=_gsr_C",
           "-DTORCH_API_INCLUDE_EXTENSION_H", "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI), "-Wno-unused-result"]
    cmd += ["-I" + i for i in inc] + [src, "-o", BINDING_OUT] + ["-L" + d for d in libdirs] + ["-Wl,-rpath," + d for d in libdirs]
    cmd += ["-lc10", "-ltorch", "-ltorch_cpu", "-ltorch_python", "-lc10_hip", "-ltorch_hip", "-L" + PKG, "-lgsr_hip", "-Wl,-rpath,$ORIGIN"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"binding build failed:\n{r.stdout}\n{r.stderr[-4000:]}")
    os.makedirs(OBJ_DIR, exist_ok=True)
    with open(stamp, "w") as fh:
        fh.write(h.hexdigest())
    if verbose:
        print(f"built {BINDING_OUT}")
    return BINDING_OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    if "--binding" in sys.argv:
        build_binding(force="--force" in sys.argv)
