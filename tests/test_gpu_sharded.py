"""GPU tests of the sharded optimizer step and of bench.py's own launcher:

  * gsr_adam_step_range: stepping the flat buffer in shards (any split at multiples of 4, moments held per shard) is bit-identical to one
    full-buffer step — the kernel indexes parameters, gradients and learning-rate segments by the GLOBAL element index;
  * gsr_dist.ShardedStep at world size 1 over RCCL (backend "nccl"): reduce-scatter -> shard Adam -> all-gather degenerates to the
    all-reduce -> Adam path, same bits;
  * `python bench.py --gpus 2` started WITHOUT torchrun (as the driver starts the N = 1 case): the launcher starts the ranks itself;
    rehearsed with the gloo backend, two ranks sharing this box's one GPU.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tensors(P, seed=3):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    return {"means3D": r(P, 3), "shs": r(P, 16, 3), "opacities": r(P, 1), "scales": r(P, 2), "rotations": r(P, 4), "refl_strengths": r(P, 1),
            "cubemap": r(6, 3, 8, 8), "fail": r(3)}


@pytest.mark.parametrize("P,world", [(1001, 2), (4099, 8), (250_000, 3)])
def test_adam_in_shards_equals_adam_over_the_whole_buffer(P, world):
    from gsr_train import GaussianTrainState
    full = GaussianTrainState(_tensors(P), "cuda", shard=(0, 1))
    # the same layout as the sharded states' (padded to 4 * world) for the full-buffer reference
    ref = GaussianTrainState(_tensors(P), "cuda", shard=(0, world))
    ref.optimizer.owned = (0, ref.params.total)
    ref.optimizer.exp_avg = torch.zeros(ref.params.total, device="cuda")
    ref.optimizer.exp_avg_sq = torch.zeros(ref.params.total, device="cuda")
    shards = [GaussianTrainState(_tensors(P), "cuda", shard=(r, world)) for r in range(world)]
    assert full.params.total <= ref.params.total and ref.params.total % (4 * world) == 0
    n = ref.params.total // world
    for step in range(3):
        g = torch.randn(ref.params.total, generator=torch.Generator().manual_seed(50 + step)).cuda()
        ref.grads.flat.copy_(g)
        ref.update_learning_rate(step + 1)
        ref.optimizer.step()
        for r, st in enumerate(shards):
            st.grads.flat.copy_(g)
            st.update_learning_rate(step + 1)
            st.optimizer.step()
            assert st.optimizer.exp_avg.numel() == n
    for r, st in enumerate(shards):
        a, b = r * n, (r + 1) * n
        assert torch.equal(st.params.flat[a:b], ref.params.flat[a:b]), r                       # its own shard: stepped, same bits
        assert torch.equal(st.optimizer.exp_avg, ref.optimizer.exp_avg[a:b]) and torch.equal(st.optimizer.exp_avg_sq, ref.optimizer.exp_avg_sq[a:b])
        other = torch.ones(ref.params.total, dtype=torch.bool, device="cuda")
        other[a:b] = False
        init = GaussianTrainState(_tensors(P), "cuda", shard=(r, world)).params.flat
        assert torch.equal(st.params.flat[other], init[other]), r                               # everything else: untouched
    assert not torch.equal(ref.params.flat, GaussianTrainState(_tensors(P), "cuda", shard=(0, world)).params.flat)


def test_sharded_step_world_size_one_over_rccl_equals_allreduce_then_adam():
    import torch.distributed as dist
    from gsr_dist import ShardedStep
    from gsr_train import GaussianTrainState
    P = 5000
    g = torch.randn(59 * P + 6 * 3 * 64 + 4, generator=torch.Generator().manual_seed(9)).cuda()
    plain = GaussianTrainState(_tensors(P), "cuda")
    plain.grads.flat.copy_(g[:plain.params.total])
    plain.grads.all_reduce()
    plain.optimizer.step()
    dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:29547", world_size=1, rank=0, device_id=torch.device("cuda", 0))
    try:
        st = GaussianTrainState(_tensors(P), "cuda", shard=(dist.get_rank(), dist.get_world_size()))
        st.grads.flat.copy_(g[:st.params.total])
        ShardedStep(st).step()
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    assert st.params.total == plain.params.total and torch.equal(st.params.flat, plain.params.flat)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no torchrun environment: the parent starts two rank processes before touching the GPU and relays
    rank 0's line.  gloo backend (GSR_BENCH_BACKEND) so that both ranks can share the one GPU of this box; a small scene."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["GSR_BENCH_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--gaussians", "20000", "--width", "320",
           "--height", "200", "--mu", "-3.2", "--cubemap", "16", "--no-cpu-baseline", "--no-c5", "--no-dropin", "--no-overlap-extra"]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks"]["world_size"] == 2 and out["ranks"]["backend"] == "gloo" and out["nccl_ranks"] == 0
    assert out["config"]["views_per_step"] == 8 and out["config"]["views_per_step_per_gpu"] == 4 and out["scaling"] == "strong"
    assert out["allreduce_ms"] > 0 and out["value"] > 0 and out["full_train_step"]["ms_per_step"] > 0
    assert out["ranks"]["ms_per_step_per_rank"]["min"] <= out["ranks"]["ms_per_step_per_rank"]["max"]
    # asking for N ranks under a launcher that started another number is an error, not a silent N = 1 run
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p2 = subprocess.run(cmd, env=env2, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert p2.returncode == 2 and "WORLD_SIZE=1" in p2.stderr and not p2.stdout.strip()
