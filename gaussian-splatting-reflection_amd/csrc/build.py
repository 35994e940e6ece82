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


if __name__ == "__main__":
    build(force="--force" in sys.argv)
