"""Adaptive density control on the flat training state (reference: scene/gaussian_model.py:403-584,
train.py:239-255).

The reference prunes / clones / splits by rebuilding every parameter tensor and both Adam moment tensors of every group
once per operation.  Here the masks (a handful of P-sized boolean vectors, evaluated with torch ops) are composed into
one row map per densification step and libgsr_hip.so applies it with one streaming gather per flat buffer
(gsr_gather_rows); the per-view statistics are one fused kernel (gsr_densification_stats) and the split children are
sampled by gsr_split_children.  Ordering of the surviving rows is the reference's: originals, then clones, then the two
copies of the split children, each in index order.
"""
import torch

from _gsr import GatherGroup, check, lib, ptr, stream_ptr
from gsr_train import FlatParams, GaussianTrainState

PER_GAUSSIAN = ("means3D", "shs", "opacities", "scales", "rotations", "refl_strengths")


class DensifyStats:
    """xyz_gradient_accum, denom, accum_w, denom_w (scene/gaussian_model.py:189-192) and max_radii2D (:183), P floats each."""

    def __init__(self, P, device):
        self.buf = torch.zeros(5, P, dtype=torch.float32, device=device)
        self.xyz_gradient_accum, self.denom, self.accum_w, self.denom_w, self.max_radii2D = self.buf

    def update(self, viewspace_grad, radii, gaussian_weights):
        """train.py:242-245: max_radii2D update + add_densification_stats, one kernel."""
        P = self.buf.shape[1]
        g = viewspace_grad.float().contiguous()
        r = radii.to(torch.int32).contiguous()
        w = gaussian_weights.float().contiguous()
        if g.shape != (P, 3) or r.numel() != P or w.numel() != P:
            raise ValueError("DensifyStats.update: shapes do not match the number of Gaussians")
        with torch.cuda.device(self.buf.device):
            check(lib.gsr_densification_stats(P, ptr(g), ptr(r), ptr(w), ptr(self.xyz_gradient_accum), ptr(self.denom), ptr(self.accum_w),
                                              ptr(self.denom_w), ptr(self.max_radii2D), stream_ptr(self.buf.device)), "gsr_densification_stats")


def _gather(src_flat, dst_flat, row_map, old, new, names):
    groups = (GatherGroup * len(names))()
    for i, k in enumerate(names):
        width = old.slices[k][1] - old.slices[k][0]
        rows_old = old.shapes[k][0]
        groups[i] = GatherGroup(old.slices[k][0], new.slices[k][0], width // rows_old if rows_old else 1)
    n_rows = int(row_map.numel())
    if n_rows == 0:
        return
    with torch.cuda.device(src_flat.device):
        check(lib.gsr_gather_rows(ptr(src_flat), ptr(dst_flat), ptr(row_map), n_rows, groups, len(names), stream_ptr(src_flat.device)),
              "gsr_gather_rows")


def _whole_moments(state, group):
    """Both Adam moment buffers over the WHOLE flat layout.  Unsharded state: the optimizer's own tensors.  Sharded state
    (GaussianTrainState(shard=(rank, N)): the optimizer holds total / N floats, the chunk this rank steps): the chunks of all ranks,
    all-gathered in rank order = buffer order — the row map below addresses rows by their offsets in the whole buffer, and a chunk
    border falls anywhere, also inside a parameter group."""
    import torch.distributed as dist
    opt, total = state.optimizer, state.params.total
    a, b = opt.owned
    if (a, b) == (0, total):
        return opt.exp_avg, opt.exp_avg_sq
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) * (b - a) != total:
        raise RuntimeError("densify_and_prune: the train state is sharded (%r) but there is no process group of that size to gather the Adam "
                           "moments from" % (state.shard,))
    out = []
    for part in (opt.exp_avg, opt.exp_avg_sq):
        if dist.get_backend(group) == "nccl":
            full = torch.empty(total, dtype=part.dtype, device=part.device)
            dist.all_gather_into_tensor(full, part.contiguous(), group=group)
        else:                               # gloo (the CPU-rank rehearsal of the tests): staged through host memory
            host = torch.empty(total, dtype=part.dtype)
            dist.all_gather_into_tensor(host, part.detach().cpu().contiguous(), group=group)
            full = host.to(part.device)
        out.append(full)
    return out[0], out[1]


def densify_and_prune(state, stats, max_grad, min_opacity, mean, extent, max_screen_size, percent_dense=0.01, N=2, noise=None, group=None):
    """GaussianModel.densify_and_prune (scene/gaussian_model.py:548-576) on a GaussianTrainState.  Returns
    (new_state, new_stats, info).  `noise`: optional standard-normal tensor (N*k, scale_dims) for the k split parents (the
    reference draws it with torch.normal); drawn with torch.randn when None.  min_opacity is unused, as in the reference
    (its opacity prune is commented out, :561-562).
    Sharded state (view-parallel training with gsr_dist.ShardedStep): every rank holds all parameters and, after
    gsr_dist.reduce_densification_stats, the same statistics, so every rank takes the same decisions — given the same `noise`, which the
    caller must then supply (e.g. drawn on rank 0 and broadcast) — and rebuilds the same parameter buffer; the Adam moments, of which a
    rank holds only its chunk, are all-gathered over `group`, permuted as a whole and re-cut into the new layout's chunks."""
    if state.shard is not None and state.shard[1] > 1 and noise is None:
        raise ValueError("densify_and_prune on a sharded state needs `noise` (identical on every rank): ranks drawing their own would split differently")
    old = state.params
    dev = old.flat.device
    names = [k for k in PER_GAUSSIAN if k in old.names]
    xyz, scaling, rotation = state.p["means3D"].detach(), state.p["scales"].detach(), state.p["rotations"].detach()
    P0, S = xyz.shape[0], scaling.shape[1]
    # -- a. prune by accumulated blend weight (:549-552)
    accum_w = stats.accum_w / stats.denom_w
    accum_w[stats.denom_w == 0] = 0.0
    idx_a = torch.nonzero(~(accum_w < 0.01), as_tuple=False).flatten()                 # old rows that survive
    grads = (stats.xyz_gradient_accum / stats.denom)[idx_a]                            # :555-556
    grads[grads.isnan()] = 0.0
    max_scale_a = torch.exp(scaling[idx_a]).max(dim=1).values
    # -- b. clone (:536-546): small surfels with a large view-space gradient
    sel_clone = (grads.abs() >= max_grad) & (max_scale_a <= percent_dense * extent)
    clones = idx_a[sel_clone]
    rows_b = torch.cat([idx_a, clones])                                                # old row of each row after the clone step
    # -- c. split (:508-534): large surfels; the clones enter with gradient 0 (padded_grad)
    padded_grad = torch.cat([grads, torch.zeros(clones.numel(), device=dev)])
    max_scale_b = torch.cat([max_scale_a, max_scale_a[sel_clone]])
    sel_split = (padded_grad >= max_grad) & (max_scale_b > percent_dense * extent)
    parents = rows_b[sel_split]                                                        # old rows of the split parents
    k = int(parents.numel())
    child_parent = parents.repeat(N).to(torch.int32)                                   # .repeat(N, 1): copy 1 of all, copy 2 of all
    if noise is None:
        noise = torch.randn(N * k, S, device=dev)
    noise = noise.float().contiguous()
    child_xyz = torch.empty(N * k, 3, device=dev)
    child_scaling = torch.empty(N * k, S, device=dev)
    if k:
        with torch.cuda.device(dev):
            check(lib.gsr_split_children(N * k, S, N, ptr(child_parent), ptr(xyz.contiguous()), ptr(scaling.contiguous()), ptr(rotation.contiguous()),
                                         ptr(noise), ptr(child_xyz), ptr(child_scaling), stream_ptr(dev)), "gsr_split_children")
    rows_c = torch.cat([rows_b[~sel_split], child_parent.long()])                      # after removing the parents (:533-534)
    is_child = torch.cat([torch.zeros(rows_b.numel() - k, dtype=torch.bool, device=dev), torch.ones(N * k, dtype=torch.bool, device=dev)])
    is_new = torch.cat([torch.zeros(idx_a.numel(), dtype=torch.bool, device=dev),
                        torch.ones(clones.numel(), dtype=torch.bool, device=dev)])[~sel_split]
    is_new = torch.cat([is_new, torch.ones(N * k, dtype=torch.bool, device=dev)])       # clones + children: fresh Adam state
    # -- d. prune big points (:559-572).  max_radii2D was reset to zero by densification_postfix (:502), so the
    # screen-size test of the reference can never fire here; it is evaluated on those zeros for fidelity.
    keep = torch.ones(rows_c.numel(), dtype=torch.bool, device=dev)
    if max_screen_size:
        cur_xyz = xyz[rows_c].clone()
        cur_xyz[is_child] = child_xyz
        cur_max_scale = torch.exp(scaling[rows_c]).max(dim=1).values
        cur_max_scale[is_child] = torch.exp(child_scaling).max(dim=1).values if k else cur_max_scale[is_child]
        big_points_vs = torch.zeros_like(keep)
        inside = ((cur_xyz - mean[None].to(dev)) ** 2).sum(dim=-1) < extent ** 2
        big_ws = (cur_max_scale > 0.1 * extent) & inside
        big_ws_far = (cur_max_scale > 1.5 * extent) & ~inside
        keep = ~(big_points_vs | big_ws | big_ws_far)
    final_rows = rows_c[keep]
    P1 = int(final_rows.numel())
    map_param = final_rows.to(torch.int32).contiguous()
    map_moment = torch.where(is_new[keep], torch.full_like(map_param, -1), map_param).contiguous()
    # -- one gather per flat buffer into a new store of P1 rows
    shapes = {kk: ((P1,) + old.shapes[kk][1:] if kk in names else old.shapes[kk]) for kk in old.names}
    world = 1 if state.shard is None else state.shard[1]
    new = FlatParams(None, dev, shapes=shapes, shards=world)             # (padded to `world` equal 16-byte-aligned chunks, as the old store was)
    new_state = GaussianTrainState(None, dev, spatial_lr_scale=state.spatial_lr_scale, lrs=state.lrs, _params=new, shard=state.shard)
    opt_old, opt_new = state.optimizer, new_state.optimizer
    old_avg, old_sq = _whole_moments(state, group)
    sharded = opt_new.owned != (0, new.total)
    new_avg = torch.zeros(new.total, dtype=torch.float32, device=dev) if sharded else opt_new.exp_avg
    new_sq = torch.zeros(new.total, dtype=torch.float32, device=dev) if sharded else opt_new.exp_avg_sq
    _gather(old.flat, new.flat, map_param, old, new, names)
    _gather(old_avg, new_avg, map_moment, old, new, names)
    _gather(old_sq, new_sq, map_moment, old, new, names)
    with torch.no_grad():
        kept_child = is_child[keep]
        if bool(kept_child.any()):
            child_keep = keep[is_child]
            new.p["means3D"][kept_child] = child_xyz[child_keep]
            new.p["scales"][kept_child] = child_scaling[child_keep]
        for kk in old.names:                                                            # the "env" group is left alone (:462)
            if kk not in names:
                a, b = old.slices[kk]
                c, d = new.slices[kk]
                new.flat[c:d].copy_(old.flat[a:b])
                new_avg[c:d].copy_(old_avg[a:b])
                new_sq[c:d].copy_(old_sq[a:b])
        if sharded:                                                                     # this rank's chunk of the new layout
            a2, b2 = opt_new.owned
            opt_new.exp_avg.copy_(new_avg[a2:b2])
            opt_new.exp_avg_sq.copy_(new_sq[a2:b2])
    opt_new.step_count = opt_old.step_count
    for kk in opt_old.groups:
        opt_new.groups[kk] = opt_old.groups[kk]
    info = dict(pruned_by_weight=P0 - int(idx_a.numel()), cloned=int(clones.numel()), split=k, pruned_big=int((~keep).sum()), before=P0, after=P1)
    return new_state, DensifyStats(P1, dev), info
