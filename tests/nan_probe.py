import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import HipSurfel, S, scene_kwargs
kw, _, _ = scene_kwargs("S", 3000, 160, 120, 40, -2.8, 0, (0.1, 0.1, 0.1))
hip = HipSurfel(kw)
out = hip.out()
g = S.make_upstream_grads(120, 160, 3)
gh = hip.backward(g["dL_dcolor"], g["dL_dplanes"], g["dL_drefl"])
for k, v in gh.items():
    if v is None: continue
    v = np.asarray(v)
    bad = ~np.isfinite(v)
    print(k, v.shape, "nonfinite", int(bad.sum()), "rows", np.unique(np.argwhere(bad)[:, 0])[:10] if bad.any() else "")
for k, v in out.items():
    if isinstance(v, np.ndarray) and v.dtype.kind == "f":
        print("out", k, "nonfinite", int((~np.isfinite(v)).sum()))
