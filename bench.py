#!/usr/bin/env python
"""Headline benchmark: one "step" = forward + backward of the hot path for one 1080p view of 1e6
Gaussians (BASELINE config C3): surfel rasterizer (variant S, the one gaussian_renderer calls) +
fused deferred reflection / cubemap lookup, then their backward with synthetic upstream gradients.

    python bench.py --gpus N --steps K --warmup W

N > 1 (launched by torch.distributed.run, one rank per GPU): every rank renders its own view of the same
1e6-Gaussian scene (weak scaling: per-GPU work fixed) and the per-Gaussian + cubemap gradients are summed
with ONE RCCL all-reduce over a flat pre-packed buffer.  value = views/s over all ranks.

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline      dominant kernel (tile-render backward): algorithmic bytes / hipEvent-measured launch time
  cpu_baseline  the CPU oracle (oracle/, "port") timed on this host on one full C3 step
and, beside them, `full_train_step`: the same step + L1/SSIM loss + gradient all-reduce + fused Adam (secondary figure,
never `value`; --no-full-step skips it).
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "gaussian-splatting-reflection_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def yaw_camera(S, W, H, deg):
    """Base C3 camera (R = I, T = 0) rotated about the y axis by `deg` degrees: rank r looks r*3 degrees to the side."""
    a = math.radians(deg)
    c2w = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]], dtype=np.float64)
    return S.make_camera(W, H, R=c2w, T=np.zeros(3))


class Scene:
    """Parameters of the synthetic scene as leaf tensors whose .grad are views into ONE flat buffer
    (gsr_dist.FlatGrads: the all-reduce payload, 59 floats per Gaussian + cubemap texels + fail value)."""

    def __init__(self, S, P, mu, L, device, seed):
        from gsr_dist import FlatGrads
        sc = S.make_scene(P, "S", seed=seed, mu=mu)
        tex, fail = S.make_cubemap(L, 3, seed)
        names = ["means3D", "shs", "opacities", "scales", "rotations", "refl_strengths"]
        src = {k: torch.from_numpy(sc[k]) for k in names}
        src["cubemap"] = torch.from_numpy(tex)
        src["fail"] = torch.from_numpy(fail)
        self.p = {k: v.to(device).requires_grad_(True) for k, v in src.items()}
        self.grads = FlatGrads(self.p)
        self.flat_grad = self.grads.flat
        self.mask = torch.from_numpy(sc["env_scope_mask"]).to(device)
        self.P = P

    def release(self):
        """Drop parameters and the flat gradient buffer (the end-to-end leg re-creates them inside its own flat store)."""
        for p in self.p.values():
            p.grad = None
        self.p, self.grads, self.flat_grad = {}, None, None


class EnvMap:
    def __init__(self, tex, fail):
        self.params = {"Cubemap_texture": tex, "Cubemap_failv": fail}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--mu", type=float, default=-4.75)
    ap.add_argument("--cubemap", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--trace-steps", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-full-step", action="store_true", help="skip the secondary end-to-end (loss + Adam) timing")
    ap.add_argument("--serial-allreduce", action="store_true", help="N > 1: all-reduce inside every step instead of overlapping it with the next one")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist_on = world > 1
    ndev = max(1, torch.cuda.device_count())
    dev_index = local_rank % ndev        # one rank per GPU on a full node; ranks share a GPU only in the gloo rehearsal below
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if dist_on:
        import torch.distributed as dist
        backend = os.environ.get("GSR_BENCH_BACKEND", "nccl")   # "nccl" is RCCL on ROCm; "gloo" only to rehearse N > 1 on one GPU
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    import gsr_synth as S
    import _gsr
    from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from gaussian_renderer import deferred_reflection

    if os.environ.get("GSR_DEV"):
        _gsr.set_option("dev", int(os.environ["GSR_DEV"], 0))   # development ablations only (tests/ablate.py)
    P, W, H = args.gaussians, args.width, args.height
    scene = Scene(S, P, args.mu, args.cubemap, dev, seed=1003)
    cam = yaw_camera(S, W, H, 3.0 * rank)
    ct = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in cam.items() if isinstance(v, np.ndarray)}
    bg = torch.zeros(3, device=dev)
    settings = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=bg,
                                             scale_modifier=1.0, viewmatrix=ct["viewmatrix"], projmatrix=ct["projmatrix"], sh_degree=3,
                                             campos=ct["campos"], prefiltered=False, debug=False)
    rasterizer = GaussianRasterizer(settings)
    env = EnvMap(scene.p["cubemap"], scene.p["fail"])
    HWK = (H, W, cam["K"])
    g = S.make_upstream_grads(H, W, 1003)
    g_final = torch.from_numpy(g["dL_dcolor"]).to(dev)
    g_allmap = torch.from_numpy(g["dL_dplanes"]).to(dev)
    means2D = torch.zeros(P, 3, device=dev, requires_grad=True)
    info = {}
    refl_sink = [None]    # gradient sink of the fused reflection op for the next forward (bound per call)

    def forward():
        base, radii, allmap, refl_map, gw = rasterizer(means3D=scene.p["means3D"], means2D=means2D, opacities=scene.p["opacities"],
                                                       shs=scene.p["shs"], refl_strengths=scene.p["refl_strengths"],
                                                       scales=scene.p["scales"], rotations=scene.p["rotations"],
                                                       env_scope_mask=scene.mask)
        final, refl_color, nrm = deferred_reflection(allmap[2:5], base, refl_map, env, ct["viewmatrix"], HWK, ct["R"], ct["T"],
                                                     grad_sink=refl_sink[0])
        if base.grad_fn is not None:
            info["R"] = base.grad_fn.num_rendered
        return final, allmap

    # the backward kernels write their parameter gradients straight into a flat all-reduce buffer (gradient sinks).  With
    # more than one rank there are two such buffers used alternately: the all-reduce of step k (RCCL, its own stream) runs
    # while step k+1 renders and writes the other buffer; a buffer is reused only after its all-reduce has completed.
    # Nothing consumes the reduced gradients in this leg, so the pipelining changes no result; the end-to-end leg below
    # (optimizer step after every all-reduce) has the strict dependency and reports the unhidden cost.
    overlap = dist_on and not args.serial_allreduce
    scene_payload_mb = scene.grads.flat.numel() * 4 / 1e6
    bufs = [scene.grads] + ([scene.grads.twin()] if overlap else [])
    sinks = [(b.sink(), b.sink(names=("cubemap", "fail"))) for b in bufs]
    pending = [None] * len(bufs)
    overlap_failed = []
    counter = [0]

    def step():
        k = counter[0] % len(bufs)
        counter[0] += 1
        if pending[k] is not None:
            pending[k].wait()              # the current stream waits for this buffer's previous all-reduce
            pending[k] = None
        rasterizer.set_grad_sink(sinks[k][0])
        refl_sink[0] = sinks[k][1]              # every gradient is sunk: nothing left for autograd to zero or accumulate
        means2D.grad = None
        final, allmap = forward()
        torch.autograd.backward([final, allmap], [g_final, g_allmap])
        if overlap and not overlap_failed:
            try:
                pending[k] = bufs[k].all_reduce_async()
            except Exception as ex:   # keep the run alive: fall back to the all-reduce inside every step
                overlap_failed.append(repr(ex))
                print("bench: asynchronous all-reduce failed (%r); continuing with the serial one" % (ex,), file=sys.stderr, flush=True)
                bufs[k].all_reduce()
        else:
            bufs[k].all_reduce()

    def drain():
        for k, w in enumerate(pending):
            if w is not None:
                w.wait()
                pending[k] = None

    def sync_all():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    drain()
    sync_all()
    _gsr.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    sync_all()
    dt = time.perf_counter() - t0
    stages = _gsr.profile_collect()
    _gsr.profile_enable(False)

    # N > 1, for transparency: the same steps with the all-reduce inside every step (what a strictly sequential loop pays)
    serial_ms = None
    if overlap:
        def serial_step():
            rasterizer.set_grad_sink(sinks[0][0])
            refl_sink[0] = sinks[0][1]
            means2D.grad = None
            final, allmap = forward()
            torch.autograd.backward([final, allmap], [g_final, g_allmap])
            bufs[0].all_reduce()
        for _ in range(2):
            serial_step()
        sync_all()
        ts = time.perf_counter()
        for _ in range(args.steps):
            serial_step()
        sync_all()
        tser = torch.tensor([time.perf_counter() - ts], device=dev, dtype=torch.float64)
        dist.all_reduce(tser, op=dist.ReduceOp.MAX)
        serial_ms = float(tser.item()) / args.steps * 1e3

    # forward-only render rate (render FPS @1080p), un-timed for the headline but reported
    rasterizer.set_grad_sink(None)
    refl_sink[0] = None
    with torch.no_grad():
        for _ in range(2):
            forward()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        nf = max(5, args.steps)
        for _ in range(nf):
            forward()
        torch.cuda.synchronize()
        fwd_ms = (time.perf_counter() - t1) / nf * 1e3

    # ---- secondary figure: the END-TO-END training step of the reference's loop (train.py:144-306):
    # render -> (1 - lambda) L1 + lambda (1 - SSIM) against a synthetic ground-truth image -> backward -> gradient
    # all-reduce -> Adam over all eight parameter groups.  Reported beside the headline, never as `value`.
    full = None
    if not args.no_full_step:
        from gsr_train import DEFAULT_LRS, GaussianTrainState
        from utils.loss_utils import photometric_loss
        tensors = {k: v.detach().clone() for k, v in scene.p.items()}
        scene.release()
        # all learning rates 0: Adam does its full arithmetic and memory traffic but the scene stays the C3 configuration
        # (with real rates the random target image changes opacities/scales within a few steps and the render cost drifts)
        st = GaussianTrainState(tensors, dev, lrs={k: 0.0 for k in DEFAULT_LRS})
        del tensors
        fsink = st.grads.sink()
        rasterizer.set_grad_sink(fsink)
        frsink = st.grads.sink(names=("cubemap", "fail"))
        fsunk = set(fsink) | set(frsink)
        fenv = EnvMap(st.p["cubemap"], st.p["fail"])
        gt_image = torch.rand(3, H, W, generator=torch.Generator(device="cpu").manual_seed(1003)).to(dev)

        def full_step(it):
            st.update_learning_rate(it)
            st.grads.zero_except_(fsunk)
            means2D.grad = None
            base, radii, allmap, refl_map, gw = rasterizer(means3D=st.p["means3D"], means2D=means2D, opacities=st.p["opacities"],
                                                           shs=st.p["shs"], refl_strengths=st.p["refl_strengths"], scales=st.p["scales"],
                                                           rotations=st.p["rotations"], env_scope_mask=scene.mask)
            final, _, _ = deferred_reflection(allmap[2:5], base, refl_map, fenv, ct["viewmatrix"], HWK, ct["R"], ct["T"], grad_sink=frsink)
            loss = photometric_loss(final, gt_image, 0.2)
            loss.backward()
            st.grads.all_reduce()
            st.optimizer.step()
            return loss

        for i in range(args.warmup):
            full_step(i + 1)
        sync_all()
        _gsr.profile_enable(True)
        t2 = time.perf_counter()
        for i in range(args.steps):
            loss = full_step(args.warmup + i + 1)
            if args.trace_steps:                      # development aid: per-step wall time (adds a sync per step)
                torch.cuda.synchronize()
                print("full step %d: %.3f ms (cumulative)" % (i, (time.perf_counter() - t2) * 1e3), file=sys.stderr, flush=True)
        sync_all()
        fdt = time.perf_counter() - t2
        fstages = _gsr.profile_collect()
        _gsr.profile_enable(False)
        if dist_on:
            tmax = torch.tensor([fdt], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            fdt = float(tmax.item())
        full = {"ms_per_step": round(fdt / args.steps * 1e3, 4), "views_per_s": round(world * args.steps / fdt, 3),
                "what": "render + L1/SSIM loss + backward + grad all-reduce + fused Adam (59 floats/Gaussian + cubemap), learning rates 0 so the workload stays C3",
                "final_loss": round(float(loss.item()), 6),
                "stage_ms_per_step": {k: round(v[0] / max(1, args.steps), 4) for k, v in fstages.items() if v[1] > 0}}

    if dist_on:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ms_per_step = dt / args.steps * 1e3
    value = world * args.steps / dt

    if rank == 0:
        R = info["R"]
        HW = W * H
        bwd_ms, bwd_n = stages["render_bwd"]
        # algorithmic bytes of ONE tile-render-backward launch (DESIGN.md §"Kernels"): per instance the 4-byte id, the
        # 80-byte render record and one 76-byte reduced gradient row; per pixel 64 bytes of upstream grads + saved state
        bytes_bwd = R * (4 + 80 + 76) + HW * 64
        achieved = bytes_bwd / (bwd_ms / max(1, bwd_n) * 1e-3) / 1e9 if bwd_ms > 0 else 0.0
        fwd_bytes = 347 * P + 257 * R + 68 * HW     # SURVEY.md §8d, variant S
        bwdall_bytes = 871 * P + 156 * R + 64 * HW
        refl_bytes = (64 + 112) * HW
        out = {
            "metric": "train_step_views_per_s (fwd+bwd, 1e6 Gaussians @1080p, surfel rasterizer + reflection path)",
            "value": round(value, 3), "unit": "views/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "C3: 1M Gaussians, 1920x1080, SH deg 3 + reflection/specular path (cubemap L=%d), fwd+bwd" % args.cubemap,
                       "gaussians": P, "width": W, "height": H, "num_rendered": R, "views_per_step_per_gpu": 1,
                       "parallelism": ("1 view per GPU + RCCL all-reduce of per-Gaussian grads" + (", overlapped with the next step (double-buffered)" if (overlap and not overlap_failed) else ", inside every step")) if dist_on else "single GPU"},
            "render_fps_forward_only": round(1e3 / fwd_ms, 2), "forward_ms": round(fwd_ms, 4),
            "stage_ms_per_step": {k: round(v[0] / max(1, args.steps), 4) for k, v in stages.items() if v[1] > 0},
            "step_algorithmic_GBps": round((fwd_bytes + bwdall_bytes + refl_bytes) / (ms_per_step * 1e-3) / 1e9, 1),
            "roofline": {"kernel": "surfel_render_bwd_wave_kernel", "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": pmc_traffic("surfel_render_bwd_wave_kernel", P, W, H),
                         "avg_launch_ms": round(bwd_ms / max(1, bwd_n), 4), "algorithmic_bytes_per_launch": bytes_bwd},
        }
        if serial_ms is not None:
            out["allreduce"] = {"payload_MB": round(scene_payload_mb, 1), "overlapped_ms_per_step": round(ms_per_step, 4),
                                "serial_ms_per_step": round(serial_ms, 4),
                                "what": "ms_per_step / value use the overlapped loop; serial = all-reduce inside every step"}
        if full is not None:
            out["full_train_step"] = full
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(S, P, W, H, args.mu, args.cubemap)
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic(kernel, P, W, H):
    """HBM bytes per launch of `kernel` from the committed PMC passes of this same command (profiles/r01_pmc_traffic.json,
    produced by tests/pmc_summary.py from two separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE --kernel-trace` runs).
    Counters are in KB; FETCH_SIZE is doubled (gfx950 counts 32-B fetches as half, MI355X_MICROARCH.md HBM section).  None when
    the file is absent or was collected on another configuration: the counters cannot be read from inside the timed process."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        d = json.load(f)
    cfg = d.get("_config", {})
    if (cfg.get("P"), cfg.get("W"), cfg.get("H")) != (P, W, H):
        return None
    for k, v in d.items():
        if kernel in k:
            return int((2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024)
    return None


def cpu_baseline(S, P, W, H, mu, L):
    """The CPU oracle (test infrastructure, oracle/) timed on this host: ONE full step of the same workload
    (rasterizer fwd+bwd + cubemap lookup fwd+bwd), OpenMP over all host cores."""
    from oracle import oracle as orc
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import scene_kwargs
    kw, cam, sc = scene_kwargs("S", P, W, H, 1003, mu, 3, (0, 0, 0))
    g = S.make_upstream_grads(H, W, 1003)
    tex, fail = S.make_cubemap(L, 3, 1003)
    o = orc.SurfelOracle(np.float32)
    o.forward(**dict(kw, means3D=kw["means3D"][:1000], opacities=kw["opacities"][:1000], shs=kw["shs"][:1000],
                     refl_strengths=kw["refl_strengths"][:1000], scales=kw["scales"][:1000], rotations=kw["rotations"][:1000],
                     env_scope_mask=kw["env_scope_mask"][:1000]))  # warm the library
    t = time.perf_counter()
    ref = o.forward(**kw)
    dirs = np.ascontiguousarray(np.moveaxis(ref["allmap"][2:5], 0, -1).reshape(-1, 3))
    c = orc.cubemap_forward(dirs, tex, fail)
    orc.cubemap_backward(np.ascontiguousarray(np.broadcast_to(g["dL_dcolor"].reshape(3, -1), c.shape)), dirs, tex)
    o.backward(dL_dcolor=g["dL_dcolor"], dL_dallmap=g["dL_dplanes"], dL_drefl_strength_map=g["dL_drefl"])
    dt = time.perf_counter() - t
    return {"value": round(1.0 / dt, 4), "unit": "views/s", "cores": os.cpu_count(), "kind": "port",
            "sample": "one full C3 step (1M Gaussians, 1920x1080, R=%d) through the CPU oracle (OpenMP, all host cores), %.1f s" % (
                ref["num_rendered"], dt)}


if __name__ == "__main__":
    main()
