"""One-off parity check at BASELINE config C5 (5 M Gaussians, 1080p, anti-aliasing + inverse-depth backward, variant G)
and the 5 M surfel equivalent.  Too slow for the regular suite (the oracle needs minutes); run by hand:
    python tests/parity_c5.py            (results recorded in DESIGN.md)"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import HipGauss, HipSurfel, S, psnr, rel_maxnorm, scene_kwargs
from oracle import oracle as orc

P, W, H = 5_000_000, 1920, 1080
g = S.make_upstream_grads(H, W, 1005)
for variant in ("G", "S"):
    kw, cam, sc = scene_kwargs(variant, P, W, H, 1005, -5.3, 3, (0, 0, 0))
    t = time.time()
    if variant == "G":
        o = orc.GaussOracle(np.float32); ref = o.forward(antialiasing=True, **kw)
        hip = HipGauss(kw, antialiasing=True)
    else:
        o = orc.SurfelOracle(np.float32); ref = o.forward(**kw)
        hip = HipSurfel(kw)
    out = hip.out()
    print(variant, "oracle forward %.1fs" % (time.time() - t), "num_rendered", out["num_rendered"], ref["num_rendered"], flush=True)
    assert out["num_rendered"] == ref["num_rendered"] and (out["radii"] == ref["radii"]).all()
    assert (hip.state("point_list").astype(np.uint32) == o.state("point_list")).all()
    nc = (hip.state("n_contrib").astype(np.int64) != o.state("n_contrib").astype(np.int64)).mean()
    print("  n_contrib mismatch fraction", nc, "PSNR color", psnr(out["color"], ref["color"]), flush=True)
    t = time.time()
    if variant == "G":
        gr = o.backward(dL_dcolor=g["dL_dcolor"], dL_dinvdepth=g["dL_dinvdepth"], dL_dnormal_map=g["dL_dnormal"], dL_drefl_strength_map=g["dL_drefl"])
        gh = hip.backward(g["dL_dcolor"], g["dL_dinvdepth"], g["dL_dnormal"], g["dL_drefl"])
    else:
        gr = o.backward(dL_dcolor=g["dL_dcolor"], dL_dallmap=g["dL_dplanes"], dL_drefl_strength_map=g["dL_drefl"])
        gh = hip.backward(g["dL_dcolor"], g["dL_dplanes"], g["dL_drefl"])
    print("  oracle backward %.1fs" % (time.time() - t), flush=True)
    for k in ("dL_dmeans3D", "dL_dsh", "dL_dopacity", "dL_dscales", "dL_drotations"):
        print("   ", k, "rel max-norm error %.2e" % rel_maxnorm(gh[k].reshape(gr[k].shape), gr[k]), flush=True)
    del o, hip
