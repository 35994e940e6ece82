"""Prints parity metrics (not asserts) for both variants; used while developing and for DESIGN.md numbers.
   python tests/gpu_probe.py [P W H]"""
import sys
import time
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 1)[0])
from helpers import HipGauss, HipSurfel, S, psnr, rel_maxnorm, scene_kwargs
from oracle import oracle as orc

P, W, H = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (10000, 256, 256)
mu = float(sys.argv[4]) if len(sys.argv) > 4 else -3.0
for variant in ("S", "G"):
    kw, cam, sc = scene_kwargs(variant, P, W, H, 1001, mu, 3, (0.2, 0.4, 0.6))
    o = (orc.SurfelOracle if variant == "S" else orc.GaussOracle)(np.float32)
    t = time.time()
    ref = o.forward(**kw) if variant == "S" else o.forward(antialiasing=True, **kw)
    print(f"[{variant}] oracle fwd {time.time()-t:.2f}s  R={ref['num_rendered']}")
    hip = HipSurfel(kw) if variant == "S" else HipGauss(kw, antialiasing=True)
    out = hip.out()
    print("  num_rendered", out["num_rendered"], ref["num_rendered"], "radii mismatches", int((out["radii"] != ref["radii"]).sum()))
    for name in ("tiles_touched", "point_offsets", "point_list", "ranges", "keys"):
        a, b = hip.state(name), o.state(name)
        print("  ", name, "mismatches", int((a.astype(b.dtype).reshape(b.shape) != b).sum()), "of", b.size)
    nc_h, nc_o = hip.state("n_contrib").astype(np.uint32), o.state("n_contrib")
    print("   n_contrib mismatch frac", float((nc_h.reshape(nc_o.shape) != nc_o).mean()))
    for k in ref:
        if isinstance(ref[k], np.ndarray) and ref[k].dtype == np.float32 and k in out:
            print(f"   {k}: psnr {psnr(out[k], ref[k], peak=max(1.0, float(np.abs(ref[k]).max()))):.1f} dB  maxabs {np.abs(out[k]-ref[k]).max():.3e}")
    g = S.make_upstream_grads(H, W, 1001)
    t = time.time()
    if variant == "S":
        gr = o.backward(dL_dcolor=g["dL_dcolor"], dL_dallmap=g["dL_dplanes"], dL_drefl_strength_map=g["dL_drefl"])
        gh = hip.backward(g["dL_dcolor"], g["dL_dplanes"], g["dL_drefl"])
    else:
        gr = o.backward(dL_dcolor=g["dL_dcolor"], dL_dinvdepth=g["dL_dinvdepth"], dL_dnormal_map=g["dL_dnormal"], dL_drefl_strength_map=g["dL_drefl"])
        gh = hip.backward(g["dL_dcolor"], g["dL_dinvdepth"], g["dL_dnormal"], g["dL_drefl"])
    print(f"  bwd done {time.time()-t:.2f}s")
    for k, v in gh.items():
        if v is not None and k in gr:
            print(f"   {k}: rel-maxnorm err {rel_maxnorm(v.reshape(gr[k].shape), gr[k]):.3e}  (max|ref| {np.abs(gr[k]).max():.3e})")
