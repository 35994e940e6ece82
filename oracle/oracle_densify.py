"""ORACLE — TEST INFRASTRUCTURE ONLY.  Literal numpy restatement of the reference's adaptive density control
(scene/gaussian_model.py:403-584: _prune_optimizer, prune_points, cat_tensors_to_optimizer, densification_postfix,
densify_and_split, densify_and_clone, densify_and_prune, add_densification_stats; train.py:242-245), operation by
operation on per-group arrays — i.e. the many small boolean-index / concatenate passes the product path replaces by one
row map.  Parity status: "parity unpinned" by reference artefacts (the reference has no tests for it and its model class
needs CUDA-only packages to import); it is a line-by-line transcription checked by known-answer cases in
tests/test_oracle_densify.py."""
import numpy as np

GROUPS = ("means3D", "shs", "opacities", "scales", "rotations", "refl_strengths")


def add_densification_stats(stats, viewspace_grad, radii, render_weight):
    """train.py:242-245 + scene/gaussian_model.py:578-584.  stats: dict of float32 arrays (P,)."""
    vis = radii > 0
    stats["max_radii2D"][vis] = np.maximum(stats["max_radii2D"][vis], radii[vis].astype(np.float32))
    stats["xyz_gradient_accum"][vis] += np.sqrt((viewspace_grad[vis].astype(np.float32) ** 2).sum(-1, dtype=np.float32))
    stats["denom"][vis] += 1
    m = render_weight > 0.0
    stats["accum_w"][m] += render_weight[m]
    stats["denom_w"][m] += 1


def build_rotation(r):
    # utils/general_utils.py:78-99
    q = r / np.sqrt((r * r).sum(-1, keepdims=True))
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.zeros((q.shape[0], 3, 3), r.dtype)
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - w * z); R[:, 0, 2] = 2 * (x * z + w * y)
    R[:, 1, 0] = 2 * (x * y + w * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - w * x)
    R[:, 2, 0] = 2 * (x * z - w * y); R[:, 2, 1] = 2 * (y * z + w * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


class Model:
    """params / exp_avg / exp_avg_sq: dict group -> array with leading dimension P; stats: the five (P,) arrays."""

    def __init__(self, params, exp_avg, exp_avg_sq, stats, percent_dense=0.01):
        self.p = {k: np.array(v, np.float32) for k, v in params.items()}
        self.m = {k: np.array(v, np.float32) for k, v in exp_avg.items()}
        self.v = {k: np.array(v, np.float32) for k, v in exp_avg_sq.items()}
        self.s = {k: np.array(v, np.float32) for k, v in stats.items()}
        self.percent_dense = percent_dense

    def prune_points(self, mask):                      # :430-446 with _prune_optimizer :403-428
        valid = ~mask
        for k in GROUPS:
            self.p[k], self.m[k], self.v[k] = self.p[k][valid], self.m[k][valid], self.v[k][valid]
        for k in self.s:
            self.s[k] = self.s[k][valid]

    def densification_postfix(self, new):              # :486-506 with cat_tensors_to_optimizer :460-484
        for k in GROUPS:
            self.p[k] = np.concatenate([self.p[k], new[k]], 0)
            self.m[k] = np.concatenate([self.m[k], np.zeros_like(new[k])], 0)
            self.v[k] = np.concatenate([self.v[k], np.zeros_like(new[k])], 0)
        P = self.p["means3D"].shape[0]
        for k in self.s:
            self.s[k] = np.zeros(P, np.float32)

    def densify_and_split(self, grads, grad_threshold, scene_extent, noise, N=2):      # :508-534
        n_init = self.p["means3D"].shape[0]
        padded = np.zeros(n_init, np.float32)
        padded[:grads.shape[0]] = grads.reshape(-1)
        scal = np.exp(self.p["scales"])
        sel = (padded >= grad_threshold) & (scal.max(1) > self.percent_dense * scene_extent)
        stds = np.tile(scal[sel], (N, 1))
        stds3 = np.concatenate([stds, np.zeros_like(stds[:, :1])], -1) if stds.shape[1] == 2 else stds
        n3 = np.concatenate([noise, np.zeros_like(noise[:, :1])], -1) if noise.shape[1] == 2 else noise
        samples = stds3 * n3                                                           # torch.normal(0, stds)
        rots = np.tile(build_rotation(self.p["rotations"][sel]), (N, 1, 1))
        new = {"means3D": np.einsum("nij,nj->ni", rots, samples).astype(np.float32) + np.tile(self.p["means3D"][sel], (N, 1)),
               "scales": np.log(np.tile(scal[sel], (N, 1)) / (0.8 * N)).astype(np.float32)}
        for k in ("rotations", "shs", "opacities", "refl_strengths"):
            new[k] = np.tile(self.p[k][sel], (N,) + (1,) * (self.p[k].ndim - 1))
        self.densification_postfix(new)
        self.prune_points(np.concatenate([sel, np.zeros(N * int(sel.sum()), bool)]))
        return int(sel.sum())

    def densify_and_clone(self, grads, grad_threshold, scene_extent):                  # :536-546
        sel = (np.abs(grads.reshape(-1)) >= grad_threshold) & (np.exp(self.p["scales"]).max(1) <= self.percent_dense * scene_extent)
        self.densification_postfix({k: self.p[k][sel] for k in GROUPS})
        return int(sel.sum())

    def densify_and_prune(self, max_grad, min_opacity, mean, extent, max_screen_size, noise):   # :548-576
        with np.errstate(all="ignore"):
            accum_w = self.s["accum_w"] / self.s["denom_w"]
        accum_w[self.s["denom_w"] == 0] = 0.0
        self.prune_points(accum_w < 0.01)
        with np.errstate(all="ignore"):
            grads = self.s["xyz_gradient_accum"] / self.s["denom"]
        grads[np.isnan(grads)] = 0.0
        nc = self.densify_and_clone(grads, max_grad, extent)
        ns = self.densify_and_split(grads, max_grad, extent, noise)
        prune = np.zeros(self.p["means3D"].shape[0], bool)
        if max_screen_size:
            big_vs = self.s["max_radii2D"] > max_screen_size
            inside = ((self.p["means3D"] - mean[None]) ** 2).sum(-1) < extent ** 2
            ms = np.exp(self.p["scales"]).max(1)
            prune = prune | big_vs | ((ms > 0.1 * extent) & inside) | ((ms > 1.5 * extent) & ~inside)
        self.prune_points(prune)
        return nc, ns
