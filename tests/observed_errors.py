"""Developer aid (GPU box): prints the errors the parity tests bound — colour max-abs, share of pixels beyond 1e-5, gaussian_weights relative error,
n_contrib mismatches — at C1 / small / C2 sizes for both variants, so that the tolerances in tests/test_gpu_parity.py can be set to
~10x what is observed.  Usage: python tests/observed_errors.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import HipGauss, HipSurfel, scene_kwargs  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def main():
    cases = [("C1", 10_000, 256, 256, 1001, -3.0, 0, (0, 0, 0)), ("small", 5_000, 200, 136, 7, -3.0, 3, (1, 1, 1)),
             ("C2", 100_000, 800, 800, 1002, -3.6, 3, (0, 0, 0))]
    for name, P, W, H, seed, mu, deg, bg in cases:
        kw, _, _ = scene_kwargs("S", P, W, H, seed, mu, deg, bg)
        o = orc.SurfelOracle(np.float32)
        ref = o.forward(**kw)
        hip = HipSurfel(kw, requires_grad=False) if False else HipSurfel(kw)
        out = hip.out()
        dc = np.abs(out["color"] - ref["color"])
        gw, gr = out["gaussian_weights"], ref["gaussian_weights"]
        rel = np.abs(gw - gr) / np.maximum(np.abs(gr), 1e-12)
        big = gr > 1e-3
        nc_h, nc_o = hip.state("n_contrib").astype(np.uint32), o.state("n_contrib")
        planes = [float(np.abs(out["allmap"][p] - ref["allmap"][p]).max()) for p in range(8)]
        print(f"S {name}: colour max-abs {dc.max():.3e}, pixels > 1e-5: {(dc.max(axis=0) > 1e-5).mean():.3e}, > 1e-4: {(dc.max(axis=0) > 1e-4).mean():.3e}; "
              f"gaussian_weights max abs {np.abs(gw - gr).max():.3e}, max rel (w > 1e-3) {rel[big].max() if big.any() else 0:.3e}, "
              f"rows with rel > 1e-5: {(rel[big] > 1e-5).mean() if big.any() else 0:.3e}; n_contrib mismatches {int((nc_h[0] != nc_o[0]).sum())} / "
              f"{int((nc_h[1] != nc_o[1]).sum())} of {nc_o[0].size}; plane max-abs {['%.1e' % v for v in planes]}", flush=True)
        kw, _, _ = scene_kwargs("G", P, W, H, seed, mu, deg, bg)
        o = orc.GaussOracle(np.float32)
        ref = o.forward(antialiasing=True, **kw)
        hip = HipGauss(kw, antialiasing=True)
        out = hip.out()
        dc = np.abs(out["color"] - ref["color"])
        nc_h, nc_o = hip.state("n_contrib").astype(np.uint32)[0], o.state("n_contrib")
        print(f"G {name}: colour max-abs {dc.max():.3e}, pixels > 1e-5: {(dc.max(axis=0) > 1e-5).mean():.3e}; n_contrib mismatches {int((nc_h != nc_o).sum())} of {nc_o.size}",
              flush=True)


if __name__ == "__main__":
    main()
