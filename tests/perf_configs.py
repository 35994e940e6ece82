"""Per-stage timings (hipEvent, inside the library) for BASELINE configs other than the headline one.
   python tests/perf_configs.py  [--big]      (development aid / DESIGN.md numbers; not a test)"""
import sys
import time
import numpy as np
import torch
sys.path.insert(0, __file__.rsplit("/", 1)[0])
from helpers import HipGauss, HipSurfel, S, scene_kwargs
import _gsr


def run(variant, P, W, H, mu, sh_degree, aa=False, iters=5, seed=1002):
    kw, cam, sc = scene_kwargs(variant, P, W, H, seed, mu, sh_degree, (0, 0, 0))
    g = S.make_upstream_grads(H, W, seed)
    res = None
    for it in range(iters + 2):
        if it == 2:
            torch.cuda.synchronize()
            _gsr.profile_enable(True)
            t0 = time.perf_counter()
        if variant == "S":
            hip = HipSurfel(kw)
            hip.backward(g["dL_dcolor"], g["dL_dplanes"], g["dL_drefl"])
        else:
            hip = HipGauss(kw, antialiasing=aa)
            hip.backward(g["dL_dcolor"], g["dL_dinvdepth"], g["dL_dnormal"], g["dL_drefl"])
        res = hip.R
    torch.cuda.synchronize()
    st = _gsr.profile_collect()
    _gsr.profile_enable(False)
    ms = {k: round(v[0] / iters, 4) for k, v in st.items() if v[1]}
    tot = sum(ms.values())
    print(f"{variant} P={P} {W}x{H} SH{sh_degree} aa={aa} R={res} tiles/gauss={res/P:.2f}  kernels total {tot:.3f} ms  {ms}", flush=True)


if __name__ == "__main__":
    import os
    if os.environ.get("GSR_DEV"):
        _gsr.set_option("dev", int(os.environ["GSR_DEV"], 0))     # development A/B switches
    run("G", 100_000, 800, 800, -3.6, 3, aa=True)
    run("S", 100_000, 800, 800, -3.6, 3)
    run("G", 1_000_000, 1920, 1080, -4.75, 3, aa=True, seed=1003)
    if "--big" in sys.argv:
        run("G", 5_000_000, 1920, 1080, -5.3, 3, aa=True, iters=3, seed=1005)
        run("S", 5_000_000, 1920, 1080, -5.3, 3, iters=3, seed=1005)
