"""GPU parity of the densification bookkeeping through the C ABI against the literal numpy
restatement of the reference (oracle/oracle_densify.py): statistics bit-exact up to the fp32 norm (1e-6), surviving rows,
their order and the carried Adam moments bit-exact (pure data movement), split children 1e-5 relative."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(P, seed, mu):
    import gsr_synth as S
    from gsr_densify import DensifyStats
    from gsr_train import GaussianTrainState
    sc = S.make_scene(P, "S", seed=seed, mu=mu)
    tex, fail = S.make_cubemap(8, 3, seed)
    names = ["means3D", "shs", "opacities", "scales", "rotations", "refl_strengths"]
    tensors = {k: torch.from_numpy(sc[k]) for k in names}
    tensors["cubemap"], tensors["fail"] = torch.from_numpy(tex), torch.from_numpy(fail)
    st = GaussianTrainState(tensors, "cuda")
    g = torch.Generator(device="cpu").manual_seed(seed)
    st.optimizer.exp_avg.copy_(torch.randn(st.params.total, generator=g))
    st.optimizer.exp_avg_sq.copy_(torch.rand(st.params.total, generator=g))
    st.optimizer.step_count = 17
    return sc, st, DensifyStats(P, "cuda"), names


def _views(st, flat, names):
    return {k: st.params.view_of(flat, k).detach().cpu().numpy().copy() for k in names}


def test_stats_kernel_matches_oracle_over_several_views():
    from oracle import oracle_densify as od
    sc, st, stats, names = _setup(20000, 5, -3.0)
    P = 20000
    ref = {k: np.zeros(P, np.float32) for k in ("xyz_gradient_accum", "denom", "accum_w", "denom_w", "max_radii2D")}
    rs = np.random.RandomState(1)
    for _ in range(3):
        g = (rs.randn(P, 3) * 1e-3).astype(np.float32)
        radii = (rs.rand(P) < 0.6).astype(np.int32) * rs.randint(1, 40, P).astype(np.int32)
        w = (rs.rand(P) * (rs.rand(P) < 0.5)).astype(np.float32)
        stats.update(torch.from_numpy(g).cuda(), torch.from_numpy(radii).cuda(), torch.from_numpy(w).cuda())
        od.add_densification_stats(ref, g, radii, w)
    np.testing.assert_allclose(stats.xyz_gradient_accum.cpu().numpy(), ref["xyz_gradient_accum"], rtol=1e-6, atol=1e-12)
    for k in ("denom", "accum_w", "denom_w", "max_radii2D"):
        np.testing.assert_array_equal(getattr(stats, k).cpu().numpy(), ref[k])


@pytest.mark.parametrize("max_screen_size", [None, 20])
def test_densify_and_prune_matches_reference_sequence(max_screen_size):
    from gsr_densify import densify_and_prune
    from oracle import oracle_densify as od
    P = 30000
    sc, st, stats, names = _setup(P, 7, -3.2)
    rs = np.random.RandomState(3)
    # statistics as a few hundred training views would leave them: some never seen, some low blend weight
    denom = rs.randint(0, 5, P).astype(np.float32)
    stats.xyz_gradient_accum.copy_(torch.from_numpy((rs.rand(P) * 8e-4 * denom).astype(np.float32)))
    stats.denom.copy_(torch.from_numpy(denom))
    dw = rs.randint(0, 4, P).astype(np.float32)
    stats.denom_w.copy_(torch.from_numpy(dw))
    stats.accum_w.copy_(torch.from_numpy((rs.rand(P) * 0.05 * dw).astype(np.float32)))
    stats.max_radii2D.copy_(torch.from_numpy(rs.randint(0, 60, P).astype(np.float32)))
    extent = 3.0
    with torch.no_grad():                               # a spread of sizes around percent_dense * extent = 0.03, some huge
        st.p["scales"].copy_(torch.from_numpy(np.log(np.exp(rs.randn(P, 2) * 1.2) * 0.03).astype(np.float32)))
    mdl = od.Model(_views(st, st.params.flat, names), _views(st, st.optimizer.exp_avg, names), _views(st, st.optimizer.exp_avg_sq, names),
                   {k: getattr(stats, k).cpu().numpy() for k in ("xyz_gradient_accum", "denom", "accum_w", "denom_w", "max_radii2D")})
    env_before = st.p["cubemap"].detach().clone()
    # the number of split parents k sizes the noise (2k rows): evaluate the reference's masks once to get it
    big_noise = rs.randn(2 * P, 2).astype(np.float32)
    acc = mdl.s["accum_w"] / np.where(mdl.s["denom_w"] == 0, 1, mdl.s["denom_w"])
    acc[mdl.s["denom_w"] == 0] = 0
    keep_a = ~(acc < 0.01)
    with np.errstate(all="ignore"):
        gr = (mdl.s["xyz_gradient_accum"] / mdl.s["denom"])[keep_a]
    gr[np.isnan(gr)] = 0
    ms = np.exp(mdl.p["scales"][keep_a]).max(1)
    k = int(((gr >= 0.0002) & (ms > 0.01 * extent)).sum())
    assert k > 100 and int(((np.abs(gr) >= 0.0002) & (ms <= 0.01 * extent)).sum()) > 100 and (~keep_a).sum() > 100
    noise = big_noise[:2 * k]
    nc, ns = mdl.densify_and_prune(0.0002, 0.05, np.zeros(3, np.float32), extent, max_screen_size, noise)
    new_state, new_stats, info = densify_and_prune(st, stats, 0.0002, 0.05, torch.zeros(3), extent, max_screen_size,
                                                   noise=torch.from_numpy(noise).cuda())
    assert (info["cloned"], info["split"]) == (nc, ns) == (info["cloned"], k)
    P1 = mdl.p["means3D"].shape[0]
    assert info["after"] == P1 and new_state.p["means3D"].shape[0] == P1
    if max_screen_size:
        assert info["pruned_big"] > 0
    got_p = _views(new_state, new_state.params.flat, names)
    got_m = _views(new_state, new_state.optimizer.exp_avg, names)
    got_v = _views(new_state, new_state.optimizer.exp_avg_sq, names)
    for kk in ("shs", "opacities", "rotations", "refl_strengths"):
        np.testing.assert_array_equal(got_p[kk], mdl.p[kk].reshape(got_p[kk].shape))
    np.testing.assert_allclose(got_p["means3D"], mdl.p["means3D"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(got_p["scales"], mdl.p["scales"], rtol=1e-5, atol=1e-6)
    for kk in names:
        np.testing.assert_array_equal(got_m[kk], mdl.m[kk].reshape(got_m[kk].shape))
        np.testing.assert_array_equal(got_v[kk], mdl.v[kk].reshape(got_v[kk].shape))
    assert torch.equal(new_state.p["cubemap"].detach(), env_before) and new_state.optimizer.step_count == 17
    assert float(new_stats.buf.abs().max()) == 0 and new_stats.buf.shape == (5, P1)
    # the new state trains: one fused Adam step runs over the re-laid-out buffers
    new_state.grads.flat.normal_()
    new_state.optimizer.step()
    assert torch.isfinite(new_state.params.flat).all()


@pytest.mark.parametrize("world", [2, 3])
def test_densify_and_prune_on_a_sharded_state(world, monkeypatch):
    """A state built with shard=(rank, N) (gsr_dist.ShardedStep) holds 1 / N of the Adam moments; densification permutes rows of the whole
    moment buffers.  Every rank of a (here: emulated) group densifies its own state; the all-gather of the moment chunks is stood in for
    by concatenating the ranks' chunks.  Result: the same parameters as the unsharded state on every rank, and each rank's new moment
    chunk = that range of the unsharded state's new moments (the two layouts differ only in the padding at the very end)."""
    import gsr_densify
    from gsr_densify import DensifyStats, densify_and_prune
    from gsr_train import GaussianTrainState
    P = 9001                                            # odd: chunk borders fall inside parameter groups
    sc, plain, stats, names = _setup(P, 11, -3.2)
    tensors = {k: plain.p[k].detach().cpu() for k in plain.params.names}
    shards = [GaussianTrainState(tensors, "cuda", shard=(r, world)) for r in range(world)]
    total, n = shards[0].params.total, shards[0].params.total // world
    assert total % (4 * world) == 0 and total >= plain.params.total
    pad = total - plain.params.total
    for r, st in enumerate(shards):                     # the same moments as the unsharded state, cut into chunks
        whole = torch.cat([plain.optimizer.exp_avg, torch.zeros(pad, device="cuda")]), torch.cat([plain.optimizer.exp_avg_sq, torch.zeros(pad, device="cuda")])
        st.optimizer.exp_avg.copy_(whole[0][r * n:(r + 1) * n])
        st.optimizer.exp_avg_sq.copy_(whole[1][r * n:(r + 1) * n])
        st.optimizer.step_count = 17
        assert st.optimizer.exp_avg.numel() == n
    rs = np.random.RandomState(5)
    denom = rs.randint(0, 5, P).astype(np.float32)
    dw = rs.randint(0, 4, P).astype(np.float32)
    vals = dict(xyz_gradient_accum=(rs.rand(P) * 8e-4 * denom).astype(np.float32), denom=denom, denom_w=dw, accum_w=(rs.rand(P) * 0.05 * dw).astype(np.float32),
                max_radii2D=rs.randint(0, 60, P).astype(np.float32))
    scales = torch.from_numpy(np.log(np.exp(rs.randn(P, 2) * 1.2) * 0.03).astype(np.float32)).cuda()

    def fill(st):
        s = DensifyStats(P, "cuda")
        for k, v in vals.items():
            getattr(s, k).copy_(torch.from_numpy(v))
        with torch.no_grad():
            st.p["scales"].copy_(scales)
        return s
    noise = torch.from_numpy(rs.randn(2 * P, 2).astype(np.float32)).cuda()
    with pytest.raises(ValueError, match="noise"):
        densify_and_prune(shards[0], fill(shards[0]), 0.0002, 0.05, torch.zeros(3), 3.0, 20)
    # the unsharded reference first tells how many split parents there are (the noise has 2 k rows)
    _, _, info = densify_and_prune(plain, fill(plain), 0.0002, 0.05, torch.zeros(3), 3.0, 20)
    k = info["split"]
    ref, _, info = densify_and_prune(plain, fill(plain), 0.0002, 0.05, torch.zeros(3), 3.0, 20, noise=noise[:2 * k])
    assert info["split"] == k > 20 and info["cloned"] > 20
    monkeypatch.setattr(gsr_densify, "_whole_moments", lambda state, group: (torch.cat([s.optimizer.exp_avg for s in shards]),
                                                                             torch.cat([s.optimizer.exp_avg_sq for s in shards])))
    for r, st in enumerate(shards):
        new, new_stats, inf = densify_and_prune(st, fill(st), 0.0002, 0.05, torch.zeros(3), 3.0, 20, noise=noise[:2 * k])
        assert inf == info and new.shard == (r, world) and new.params.total % (4 * world) == 0
        for kk in new.params.names:
            assert torch.equal(new.p[kk].detach(), ref.p[kk].detach()), kk
        a, b = new.optimizer.owned
        assert b - a == new.params.total // world == new.optimizer.exp_avg.numel()
        hi = min(b, ref.params.total)
        assert torch.equal(new.optimizer.exp_avg[:max(0, hi - a)], ref.optimizer.exp_avg[a:hi])
        assert torch.equal(new.optimizer.exp_avg_sq[:max(0, hi - a)], ref.optimizer.exp_avg_sq[a:hi])
        assert float(new.optimizer.exp_avg[max(0, hi - a):].abs().max() if b > hi else 0.0) == 0.0
        # and it steps: this rank's chunk only
        new.grads.flat.normal_()
        before = new.params.flat.clone()
        new.optimizer.step()
        changed = before != new.params.flat
        assert bool(changed[a:b].any()) and not bool(changed[:a].any()) and not bool(changed[b:].any())
