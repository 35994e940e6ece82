#!/usr/bin/env python
"""Headline benchmark: one "step" = forward + backward of the hot path for the views of one training batch of the
1e6-Gaussian 1080p scene (surfel rasterizer — the one gaussian_renderer calls — + fused deferred reflection / cubemap lookup,
then their backward with synthetic upstream gradients).

    python bench.py --gpus N --steps K --warmup W [--views V]

N = 1 (default): BASELINE config C3 — ONE view per step, no collective.
N > 1 (launched by torch.distributed.run, one rank per GPU): BASELINE config C4 — a batch of V = 8 views per step sharded over
the ranks (rank r renders views r, r+N, ...; 8/N views each), every rank ACCUMULATES its views' per-Gaussian + cubemap gradients
on the device in one flat buffer (kernel `+=`, gradient sinks), then ONE RCCL all-reduce of that buffer INSIDE the step:
the next step starts only when the reduced gradients are there, as a training step (Adam) needs them.  Total work per step is
fixed (strong scaling).  value = views/s over all ranks.  `--views 8 --gpus 1` runs the same batch on one GPU.

Rank 0 prints ONE JSON line (contract in the task statement) with extra objects:
  roofline         dominant kernel (tile-render backward): algorithmic bytes / hipEvent-measured launch time (bound: hbm), and
                   bound2 = VALU issue (wave-instructions per launch from the committed PMC pass / launch time vs 1 per 2 cycles)
  cpu_baseline     the CPU oracle (oracle/, "port") on this host: one full C3 step on all cores + a bounded single-thread sample
  full_train_step  the same step + L1/SSIM loss + gradient all-reduce + fused Adam (secondary figure, never `value`)
  c5               BASELINE config C5 (5e6 Gaussians, 3DGS rasterizer, anti-aliasing + inverse-depth backward) on this GPU
  allreduce        (N > 1) payload, and the time of a loop that overlaps the all-reduce with the next step (labelled extra)
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

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
N_SIMDS = 256 * 4
VALU_PEAK_WAVE_INSTS_PER_S = N_SIMDS * 2.4e9 / 2.0   # 1024 SIMDs, one wave64 VALU instruction per 2 cycles at 2.4 GHz (same guide)
HBM_MEASURED_GBS = 6290.0    # same guide: the float4-copy ceiling measured on this part (SURVEY.md §8d asks for both)
PMC_FILE = os.path.join(ROOT, "profiles", "r04_pmc_summary.json")
PMC_FILE_C5 = os.path.join(ROOT, "profiles", "r04_c5_pmc_summary.json")
DIGEST_FILE = os.path.join(PKG, "csrc", "_obj", "digest.txt")      # written by csrc/build.py: sha256 over every kernel source + flags


class Scene:
    """Parameters of the synthetic scene as leaf tensors whose .grad are views into ONE flat buffer
    (gsr_dist.FlatGrads: the all-reduce payload, 59 floats per Gaussian + cubemap texels + fail value)."""

    def __init__(self, S, P, mu, L, device, seed, ball=False):
        from gsr_dist import FlatGrads
        sc = S.make_scene(P, "S", seed=seed, mu=mu, ball=ball)
        tex, fail = S.make_cubemap(L, 3, seed)
        names = ["means3D", "shs", "opacities", "scales", "rotations", "refl_strengths"]
        src = {k: torch.from_numpy(sc[k]) for k in names}
        src["cubemap"] = torch.from_numpy(tex)
        src["fail"] = torch.from_numpy(fail)
        self.p = {k: v.to(device).requires_grad_(True) for k, v in src.items()}
        self.grads = FlatGrads(self.p)
        self.mask = torch.from_numpy(sc["env_scope_mask"]).to(device)
        self.P = P
        self.env = EnvMap(self.p["cubemap"], self.p["fail"])
        self.means2D = torch.zeros(P, 3, device=device, requires_grad=True)

    def release(self):
        """Drop parameters and the flat gradient buffer (the end-to-end leg re-creates them inside its own flat store)."""
        for p in self.p.values():
            p.grad = None
        self.p, self.grads, self.env, self.means2D = {}, None, None, None


class EnvMap:
    def __init__(self, tex, fail):
        self.params = {"Cubemap_texture": tex, "Cubemap_failv": fail}


class View:
    """One camera of the batch with its own rasterizer instance (the gradient sink is per rasterizer)."""

    def __init__(self, S, index, W, H, dev, cam=None):
        from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
        cam = cam or S.yaw_camera(W, H, 3.0 * index)     # view v of the batch looks 3 v degrees to the side
        self.ct = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in cam.items() if isinstance(v, np.ndarray)}
        self.HWK = (H, W, cam["K"])
        settings = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"],
                                                 bg=torch.zeros(3, device=dev), scale_modifier=1.0, viewmatrix=self.ct["viewmatrix"],
                                                 projmatrix=self.ct["projmatrix"], sh_degree=3, campos=self.ct["campos"], prefiltered=False,
                                                 debug=False)
        self.rasterizer = GaussianRasterizer(settings)
        # (--unfused only) allmap[2:5] as an output tap: the reflection pass's normal gradient reaches the tile backward as its own pointer
        # instead of through autograd's zero-fill + slice copy + add over the 8-plane image
        self.rasterizer.set_output_taps(("normal_view",))
        # what gaussian_renderer.surface_pass reads from a camera
        self.world_view_transform, self.full_proj_transform = self.ct["viewmatrix"], self.ct["projmatrix"]
        self.image_width, self.image_height = W, H


def percentiles(ms):
    a = np.sort(np.asarray(ms, dtype=np.float64))
    if a.size == 0:
        return None
    q = lambda f: float(a[min(a.size - 1, int(round(f * (a.size - 1))))])
    return {"median": round(q(0.5), 4), "p10": round(q(0.1), 4), "p90": round(q(0.9), 4), "min": round(float(a[0]), 4), "max": round(float(a[-1]), 4)}


def settle(step, args, views_per_step=1):
    """Untimed steps in front of a loop's warm-up (see --settle-steps): a fixed count (every rank runs the same number of collectives), scaled
    down for loops whose step is a batch of views.  Returns how many."""
    n = -(-max(0, int(getattr(args, "settle_steps", 0))) // max(1, views_per_step))
    for _ in range(n):
        step()
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--settle-steps", type=int, default=24,
                    help="untimed steps IN FRONT of the --warmup steps of every timed loop: after an idle period (scene set-up, a host-side "
                         "collect) the first ~12 launches run up to 15 %% slower (profiles/r04_clock_ramp_tile_backward.txt: the tile backward takes "
                         "808, 843, 860, 816, 813, 804, 783, 767, 758, 754, 736, 736, 722 us, then 720-735); 0 = off.  Reported as `settle_steps`; "
                         "the timed region is still exactly --steps steps behind --warmup steps")
    ap.add_argument("--views", type=int, default=0, help="views per step over all ranks (default: 1 on one GPU = C3, 8 on several = C4)")
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--mu", type=float, default=-4.75)
    ap.add_argument("--heavy-mu", type=float, default=-4.4,
                    help="log-scale mean of the second, heavier C3-shaped line (c3_heavy): SURVEY.md 8d calibrates the scenes for ~6 tiles per Gaussian "
                         "(R ~ 6 M at 1 M Gaussians); mu = -4.75 gives 3.9")
    ap.add_argument("--no-heavy", action="store_true", help="N = 1: skip the c3_heavy object")
    ap.add_argument("--cubemap", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-full-step", action="store_true", help="skip the secondary end-to-end (loss + Adam) timing")
    ap.add_argument("--no-c5", action="store_true", help="skip the C5 (5e6 Gaussians, variant G) object")
    ap.add_argument("--no-c4", action="store_true", help="N = 1: skip the C4-on-one-GPU object (the 8-view batch)")
    ap.add_argument("--sync-reflection-tail", action="store_true",
                    help="keep the cubemap-gradient tail of the reflection backward on the main stream (default: side stream, joined by the all-reduce)")
    ap.add_argument("--no-overlap-extra", action="store_true", help="N > 1: skip the extra loop that overlaps the all-reduce with the next step")
    ap.add_argument("--only-c5", action="store_true", help="run only the C5 object (5e6 Gaussians, variant G) and print it: the command the C5 rocprofv3 passes wrap")
    ap.add_argument("--view-streams", type=int, default=1,
                    help="views of a rank's batch alternate over this many torch streams: the binning chain of view i+1 (launch-latency-bound) runs beside "
                         "the tile kernels of view i (VALU-bound); the backwards stay ordered (they add into one gradient buffer)")
    ap.add_argument("--no-dropin", action="store_true", help="skip the drop-in object (reference entry points with plain autograd)")
    ap.add_argument("--unfused", action="store_true",
                    help="rasterizer and deferred reflection as two autograd nodes / two more kernels per direction (rounds 1-3) instead of the fused "
                         "rasterize + reflect path (the reflection's pixel code inside the tile kernels)")
    ap.add_argument("--sharded-adam", action="store_true",
                    help="N > 1, full_train_step: reduce-scatter -> Adam on this rank's 1/N of the flat buffer -> all-gather instead of all-reduce -> full Adam")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench: --gpus must be >= 1")

    # ---- launcher.  `python bench.py --gpus N` with N > 1 and no torchrun environment starts its N ranks itself, BEFORE this process
    # makes any GPU call (fresh children through `python -m torch.distributed.run`; a process that has initialised the GPU is never
    # re-exec'ed), relays rank 0's JSON line and exits with the children's status.
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        # never report another N than the one asked for (round 2: `--gpus 8` without torchrun silently measured one GPU)
        if rank == 0:
            print("bench: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world), file=sys.stderr, flush=True)
        raise SystemExit(2)
    dist_on = world > 1
    ndev = max(1, torch.cuda.device_count())
    dev_index = local_rank % ndev        # one rank per GPU on a full node; ranks share a GPU only in the gloo rehearsal below
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    backend, ranks = None, 1
    if dist_on:
        import torch.distributed as dist
        backend = os.environ.get("GSR_BENCH_BACKEND", "nccl")   # "nccl" is RCCL on ROCm; "gloo" only to rehearse N > 1 on one GPU
        if backend == "nccl":
            if ndev < world:
                if rank == 0:
                    print("bench: %d ranks but %d visible GPUs: RCCL needs one GPU per rank (GSR_BENCH_BACKEND=gloo rehearses the N > 1 "
                          "code path on fewer GPUs)" % (world, ndev), file=sys.stderr, flush=True)
                raise SystemExit(2)
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)
        ranks = dist.get_world_size()
        if ranks != args.gpus:
            raise SystemExit("bench: process group has %d ranks, --gpus %d" % (ranks, args.gpus))

    import gsr_synth as S
    import _gsr
    from gaussian_renderer import deferred_reflection, rasterize_reflect
    from gsr_dist import shard_views

    if args.only_c5:
        if world != 1:
            raise SystemExit("bench: --only-c5 is a single-GPU run")
        print(json.dumps(c5_object(S, dev, steps=max(5, args.steps), settle_steps=args.settle_steps // 2)), flush=True)
        return

    if os.environ.get("GSR_DEV"):
        _gsr.set_option("dev", int(os.environ["GSR_DEV"], 0))   # development ablations only (tests/ablate.py)
    P, W, H = args.gaussians, args.width, args.height
    views_total = args.views if args.views > 0 else (8 if dist_on else 1)
    my_views = shard_views(views_total, rank, world)
    if not my_views:
        raise SystemExit("bench: fewer views per step (%d) than ranks (%d)" % (views_total, world))
    scene = Scene(S, P, args.mu, args.cubemap, dev, seed=1003)
    views = [View(S, v, W, H, dev) for v in my_views]
    env = scene.env
    g = S.make_upstream_grads(H, W, 1003)
    g_final = torch.from_numpy(g["dL_dcolor"]).to(dev)
    g_allmap = torch.from_numpy(g["dL_dplanes"]).to(dev)
    means2D = scene.means2D
    info = {}

    def render(view, refl_sink, accumulate, sc=None):
        sc = sc or scene
        ct = view.ct
        env, means2D = sc.env, sc.means2D
        kw = dict(means3D=sc.p["means3D"], means2D=means2D, opacities=sc.p["opacities"], shs=sc.p["shs"], refl_strengths=sc.p["refl_strengths"],
                  scales=sc.p["scales"], rotations=sc.p["rotations"], env_scope_mask=sc.mask)
        async_tail = refl_sink is not None and not args.sync_reflection_tail
        if args.unfused:
            base, radii, allmap, refl_map, gw, normal_view = view.rasterizer(**kw)
            final, refl_color, nrm = deferred_reflection(normal_view, base, refl_map, env, ct["viewmatrix"], view.HWK, ct["R"], ct["T"],
                                                         grad_sink=refl_sink, accumulate=accumulate, async_tail=async_tail)
        else:
            # rasterizer + deferred reflection in one pass over the pixels: the reflection's per-pixel code is the epilogue of the forward tile
            # kernel and the prologue of the backward one (gaussian_renderer.rasterize_reflect, csrc/gsr_refl.hpp)
            final, refl_color, nrm, base, radii, allmap, refl_map, gw = rasterize_reflect(view.rasterizer, env, ct["viewmatrix"], view.HWK, ct["R"], ct["T"],
                                                                                         refl_grad_sink=refl_sink, accumulate=accumulate,
                                                                                         async_tail=async_tail, **kw)
        if base.grad_fn is not None:
            info["R"] = base.grad_fn.num_rendered
        return final, allmap

    # The backward kernels write their parameter gradients straight into the flat all-reduce buffer (gradient sinks): the first
    # view of a step overwrites it, every further view of this rank adds to it on the device (accumulate mode) — no zero-fill,
    # no autograd accumulation passes.  Then ONE all-reduce, inside the step.  The part of the reflection backward that only
    # produces the cubemap gradient runs on the library's side stream beside the rasterizer backward (async_tail);
    # FlatGrads.all_reduce() makes the step's stream wait for it, so it is inside the timed region.
    ar_marks = []      # (start, end) event pairs around the all-reduce; filled only in the instrumented pass

    vstreams = [torch.cuda.Stream(device=dev) for _ in range(args.view_streams)] if args.view_streams > 1 else []

    def step_into(buf, reduce, timed=False, batch=None, sc=None):
        sink, rsink = buf.sink(), buf.sink(names=("cubemap", "fail"))
        vs = views if batch is None else batch
        means2D = (sc or scene).means2D
        if vstreams and len(vs) > 1:
            main = torch.cuda.current_stream()
            fork = torch.cuda.Event()
            fork.record(main)
            done = []
            for i, view in enumerate(vs):
                st = vstreams[i % len(vstreams)]
                if i < len(vstreams):
                    st.wait_event(fork)
                with torch.cuda.stream(st):
                    view.rasterizer.set_grad_sink(sink, accumulate=i > 0)
                    means2D.grad = None
                    final, allmap = render(view, rsink, i > 0, sc)
                    if done:
                        st.wait_event(done[-1])      # the backwards add into ONE buffer: view i's may not start before view i-1's is over
                    torch.autograd.backward([final, allmap], [g_final, g_allmap])
                    ev = torch.cuda.Event()
                    ev.record(st)
                    done.append(ev)
            main.wait_event(done[-1])
        else:
            for i, view in enumerate(vs):
                view.rasterizer.set_grad_sink(sink, accumulate=i > 0)
                means2D.grad = None
                final, allmap = render(view, rsink, i > 0, sc)
                torch.autograd.backward([final, allmap], [g_final, g_allmap])
        if not timed:
            return reduce(buf)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = reduce(buf)       # RCCL runs it on its own stream; the step's stream waits for the result, so e1 sees it complete
        e1.record()
        ar_marks.append((e0, e1))
        return out

    def step(timed=False):
        step_into(scene.grads, lambda b: b.all_reduce(), timed)

    def sync_all():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    settle_steps = settle(step, args, -(-views_total // world))     # (the same count on every rank: each step holds a collective)
    for _ in range(args.warmup):
        step()
    # the timed region: exactly K steps between two barrier + synchronize brackets, nothing else on the stream (no events, no
    # per-stage timers: every hipEventRecord costs a ~5-10 us bubble between two kernels, 0.1 ms per step when every stage has two)
    import gc as pygc
    gc_events, gc_t = [], [0.0]

    def on_gc(phase, info):    # collections of the Python garbage collector inside the timed region are reported, not hidden (`python_gc_in_timed_region`)
        if phase == "start":
            gc_t[0] = time.perf_counter()
        else:
            gc_events.append({"generation": info.get("generation"), "ms": round((time.perf_counter() - gc_t[0]) * 1e3, 2)})
    pygc.callbacks.append(on_gc)
    host_t = [0.0] * (args.steps + 1)     # host clock at every step boundary (a perf_counter call: ~50 ns; no GPU call, no synchronisation)
    sync_all()
    t0 = time.perf_counter()
    host_t[0] = t0
    for i in range(args.steps):
        step()
        host_t[i + 1] = time.perf_counter()
    sync_all()
    dt = time.perf_counter() - t0
    pygc.callbacks.remove(on_gc)
    host_step_ms = [(host_t[i + 1] - host_t[i]) * 1e3 for i in range(args.steps)]
    # the same K steps again, instrumented: one event per step (percentiles) and the library's per-stage hipEvent timers
    _gsr.profile_enable(True)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    for i in range(args.steps):
        marks[i].record()
        step(timed=True)
    marks[args.steps].record()
    sync_all()
    stages = _gsr.profile_collect()
    _gsr.profile_enable(False)
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    R_headline = info["R"]      # (later objects render other views: their instance counts must not leak into the headline's byte counts)
    n_color = int((scene.grads.view("shs")[:, 0, :] != 0).any(dim=1).sum().item()) if len(views) == 1 else None     # surfels with a colour gradient
    allreduce_ms = float(np.median([a.elapsed_time(b) for a, b in ar_marks])) if ar_marks else 0.0
    # the dominant kernel on its own: a few more steps with the reflection tail on the step's stream (nothing runs beside the tile backward)
    alone_ms = None
    if not args.sync_reflection_tail:
        args.sync_reflection_tail = True
        for _ in range(2):                     # (the first steps of the other mode size new scratch buffers)
            step()
        sync_all()
        settle(step, args, -(-views_total // world))   # (the collect above left the GPU idle: without this the five launches below are the slow ones of the ramp)
        _gsr.profile_enable(True)
        for _ in range(5):
            step()
        sync_all()
        a = _gsr.profile_collect()
        _gsr.profile_enable(False)
        args.sync_reflection_tail = False
        if a["render_bwd"][1] > 0:
            alone_ms = a["render_bwd"][0] / a["render_bwd"][1]

    # N > 1, labelled extra: the same steps with the all-reduce of step k (RCCL, its own stream) overlapped with step k+1, which
    # renders into a second buffer.  Nothing consumes the reduced gradients in that loop — a training step cannot do this
    # (Adam needs them before the next forward) — so it is NOT the headline.
    overlap_ms = None
    if dist_on and not args.no_overlap_extra:
        try:
            bufs = [scene.grads, scene.grads.twin()]
            pending = [None, None]

            def overlapped(k):
                if pending[k] is not None:
                    pending[k].wait()
                pending[k] = step_into(bufs[k], lambda b: b.all_reduce_async())
            for i in range(2):
                overlapped(i % 2)
            sync_all()
            ts = time.perf_counter()
            for i in range(args.steps):
                overlapped(i % 2)
            for w in pending:
                if w is not None:
                    w.wait()
            sync_all()
            tov = torch.tensor([time.perf_counter() - ts], device=dev, dtype=torch.float64)
            dist.all_reduce(tov, op=dist.ReduceOp.MAX)
            overlap_ms = float(tov.item()) / args.steps * 1e3
            del bufs
        except Exception as ex:     # keep the run alive: the extra is optional
            print("bench: overlapped extra loop failed (%r); skipped" % (ex,), file=sys.stderr, flush=True)

    # forward-only render rate (render FPS @1080p), un-timed for the headline but reported
    for view in views:
        view.rasterizer.set_grad_sink(None)
    with torch.no_grad():
        for _ in range(2):
            render(views[0], None, False)
        torch.cuda.synchronize()
        nf = max(5, args.steps)
        # wall clock, in five synchronised chunks, median chunk reported: this loop has one host synchronisation per iteration
        # (num_rendered) and nothing to hide a host hiccup behind — a single allocator / GC stall of a few ms used to move the figure by 10-20 %
        chunk, chunks = max(1, nf // 5), []
        for c in range(5):
            t1 = time.perf_counter()
            for i in range(chunk):
                render(views[0], None, False)
            torch.cuda.synchronize()
            chunks.append((time.perf_counter() - t1) / chunk * 1e3)
        fwd_ms = sorted(chunks)[2]
        fmarks = [torch.cuda.Event(enable_timing=True) for _ in range(nf + 1)]
        for i in range(nf):
            fmarks[i].record()
            render(views[0], None, False)
        fmarks[nf].record()
        torch.cuda.synchronize()
        fwd_step_ms = [fmarks[i].elapsed_time(fmarks[i + 1]) for i in range(nf)]

    def side_line(sc, batch, nb, what):
        """A reported (never `value`) line on another scene / batch of views: wall clock over nb synchronised steps, then one instrumented
        pass for the stage table (per view)."""
        for _ in range(2):
            step_into(sc.grads, lambda b: b.all_reduce(), batch=batch, sc=sc)
        settle(lambda: step_into(sc.grads, lambda b: b.all_reduce(), batch=batch, sc=sc), args, len(batch))
        torch.cuda.synchronize()
        tc = time.perf_counter()
        for _ in range(nb):
            step_into(sc.grads, lambda b: b.all_reduce(), batch=batch, sc=sc)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - tc) / nb * 1e3
        _gsr.profile_enable(True)
        for _ in range(nb):
            step_into(sc.grads, lambda b: b.all_reduce(), batch=batch, sc=sc)
        torch.cuda.synchronize()
        st = _gsr.profile_collect()
        _gsr.profile_enable(False)
        nvw = len(batch)
        out = {"workload": what, "ms_per_step": round(ms, 4), "ms_per_view": round(ms / nvw, 4), "views_per_s": round(nvw * 1e3 / ms, 2), "steps": nb,
               "num_rendered": info.get("R"), "stage_ms_per_view": {k: round(v[0] / max(1, nb * nvw), 4) for k, v in st.items() if v[1] > 0}}
        for view in batch:
            view.rasterizer.set_grad_sink(None)
        return out

    # BASELINE C4 on this one GPU (reported, never `value`), with the geometry SURVEY.md 8d fixes for it: 1e6 surfels in the ball of radius 2
    # about the origin, eight cameras on a circle of radius 5 looking at it — one step = the batch of 8 views, first view overwriting the flat
    # gradient buffer, seven adding to it on the device, as the N > 1 runs shard it over the ranks.  (The N > 1 headline keeps the C3 scene
    # with eight yawed cameras, so that the driver's efficiency = value(N) / (N value(1)) compares like with like.)
    c4 = heavy = None
    if world == 1 and views_total == 1 and not args.no_c4:
        ball = Scene(S, P, args.mu, args.cubemap, dev, seed=1004, ball=True)
        batch = [View(S, k, W, H, dev, cam=c) for k, c in enumerate(S.circle_cameras(W, H, 8))]
        c4 = side_line(ball, batch, max(3, args.steps // 4),
                       "C4 on one GPU (SURVEY.md 8d geometry): 1M surfels in the ball of radius 2, a batch of 8 views from cameras on a circle of radius 5 "
                       "per step, gradients accumulated on the device in the flat buffer")
        ball.release()
        del ball, batch
        torch.cuda.empty_cache()
    # A second, heavier C3-shaped line: the same step on a scene calibrated for ~6 tiles per Gaussian (every tuning decision of rounds 1-3 was
    # taken at 3.9).  Never `value`.
    if world == 1 and views_total == 1 and not args.no_heavy:
        hv = Scene(S, P, args.heavy_mu, args.cubemap, dev, seed=1003)
        heavy = side_line(hv, [View(S, 0, W, H, dev)], max(5, args.steps // 2),
                          "C3-shaped, heavier: 1M surfels with log-scale mean mu = %.2f (C3: %.2f), 1920x1080, SH 3 + reflection path, fwd+bwd, one view per step"
                          % (args.heavy_mu, args.mu))
        heavy["tiles_per_gaussian"] = round(heavy["num_rendered"] / P, 2)
        hv.release()
        del hv
        torch.cuda.empty_cache()
    info["R"] = None

    scene_payload_mb = scene.grads.flat.numel() * 4 / 1e6
    full = None if args.no_full_step else full_train_step(args, scene, views, S, dev, means2D, sync_all, dist_on, world, views_total)

    rank_ms = {"min": round(dt / args.steps * 1e3, 4), "max": round(dt / args.steps * 1e3, 4)}
    if dist_on:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        tmin = tmax.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        dt = float(tmax.item())
        rank_ms = {"min": round(float(tmin.item()) / args.steps * 1e3, 4), "max": round(dt / args.steps * 1e3, 4)}
        armax = torch.tensor([allreduce_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(armax, op=dist.ReduceOp.MAX)
        allreduce_ms = float(armax.item())
    ms_per_step = dt / args.steps * 1e3
    value = views_total * args.steps / dt

    if rank == 0:
        R = R_headline
        HW = W * H
        nv = len(views)
        bwd_ms, bwd_n = stages["render_bwd"]
        # algorithmic bytes of ONE tile-render-backward launch (DESIGN.md §"Kernels"): per instance the 4-byte id, the
        # 80-byte render record and one 76-byte reduced gradient row; per pixel 64 bytes of upstream grads + saved state
        bytes_bwd = R * (4 + 80 + 76) + HW * 64
        launch_s = bwd_ms / max(1, bwd_n) * 1e-3
        achieved = bytes_bwd / launch_s / 1e9 if bwd_ms > 0 else 0.0
        fwd_bytes = 347 * P + 257 * R + 68 * HW     # SURVEY.md §8d, variant S
        bwdall_bytes = 871 * P + 156 * R + 64 * HW
        refl_bytes = (64 + 112) * HW
        pmc = pmc_summary("surfel_render_bwd_rows_kernel", P, W, H)
        roof = {"kernel": "surfel_render_bwd_rows_kernel", "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "frac_of_measured_ceiling": round(achieved / HBM_MEASURED_GBS, 4),
                "measured_ceiling": HBM_MEASURED_GBS, "traffic": pmc.get("traffic"),
                "traffic_source": pmc.get("source"), "avg_launch_ms": round(launch_s * 1e3, 4), "algorithmic_bytes_per_launch": bytes_bwd}
        if alone_ms:
            # the same kernel without the reflection's run combine beside it (extra steps with the tail on the step's stream, outside the timed region)
            roof["avg_launch_ms_alone"] = round(alone_ms, 4)
            roof["frac_alone"] = round(bytes_bwd / (alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        if pmc.get("insts_valu"):
            rate = pmc["insts_valu"] / launch_s
            roof["bound2"] = {"bound": "valu_issue", "achieved": round(rate / 1e9, 1), "peak": round(VALU_PEAK_WAVE_INSTS_PER_S / 1e9, 1),
                              "unit": "G wave-instructions/s", "frac": round(rate / VALU_PEAK_WAVE_INSTS_PER_S, 4),
                              "wave_insts_valu_per_launch": pmc["insts_valu"], "wave_insts_salu_per_launch": pmc.get("insts_salu"),
                              "what": "SQ_INSTS_VALU of the committed PMC pass / the launch time measured in this run, against one wave64 VALU "
                                      "instruction per SIMD per 2 cycles; the kernel's mix (packed fp32, SGPR operands, DPP: ~4 cycles each, "
                                      "tests/microbench/inst_cost.hip) puts its own issue bound at ~1.0 of the measured time (DESIGN.md)"}
            mix = isa_mix("surfel_render_bwd_rows_kernel")
            if mix:
                # issue bound of THIS kernel's instruction mix: its VALU count (PMC) x the average issue cost of the hot loop's static mix
                # (tests/isa_mix.py: compiler assembly of this build, priced with the measured per-class costs) / the chip's 1024 SIMDs
                bound_ms = pmc["insts_valu"] * mix["avg_ns_per_valu"] * 1e-6 / N_SIMDS
                roof["bound2"]["issue_bound_of_this_mix"] = {
                    "avg_ns_per_valu_instruction": mix["avg_ns_per_valu"], "static_mix_of_the_hot_loop": mix["mix"], "bound_ms": round(bound_ms, 4),
                    "frac": round(bound_ms / (launch_s * 1e3), 4), "frac_alone": round(bound_ms / alone_ms, 4) if alone_ms else None,
                    "source": "profiles/r04_isa_mix.json (tests/isa_mix.py, same build digest) x SQ_INSTS_VALU of the PMC pass"}
        out = {
            "metric": "train_step_views_per_s (fwd+bwd, 1e6 Gaussians @1080p, surfel rasterizer + reflection path)",
            "value": round(value, 3), "unit": "views/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "settle_steps": settle_steps,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong" if dist_on else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("C4: batch of %d views of the C3 scene per step, sharded over the GPUs, one RCCL all-reduce of the per-Gaussian + "
                                    "cubemap gradients inside every step" % views_total) if (dist_on or views_total > 1) else
                                   "C3: 1M Gaussians, 1920x1080, SH deg 3 + reflection/specular path (cubemap L=%d), fwd+bwd" % args.cubemap,
                       "gaussians": P, "width": W, "height": H, "num_rendered": R, "views_per_step": views_total,
                       "views_per_step_per_gpu": nv,
                       "parallelism": ("views sharded %d per GPU, gradients accumulated on device, all-reduce inside the step" % nv) if dist_on
                       else "single GPU",
                       "reflection": "two nodes: rasterizer, then deferred_reflection (--unfused)" if args.unfused else
                                     "fused: the deferred reflection's pixel code runs inside the rasterizer's tile kernels (rasterize_reflect)"},
            "step_ms": percentiles(step_ms),
            "python_gc_in_timed_region": gc_events,
            "host_ms_per_step_in_timed_region": dict(percentiles(host_step_ms), what="host time inside each timed step's calls (every step has one "
                                                     "host-GPU rendezvous, the num_rendered read-back): a max far above the median is a step in which "
                                                     "the HOST thread was held up, and the GPU with it"),
            "render_fps_forward_only": round(1e3 / fwd_ms, 2), "forward_ms": round(fwd_ms, 4), "forward_step_ms": percentiles(fwd_step_ms),
            "stage_ms_per_view": {k: round(v[0] / max(1, args.steps * nv), 4) for k, v in stages.items() if v[1] > 0},
            "instrumented": "step_ms and stage_ms_per_view come from a second pass of the same K steps with one event per step and the library's "
                            "per-stage hipEvent timers on (each event record costs a 5-10 us bubble between kernels); value / ms_per_step are the "
                            "uninstrumented pass.  refl_bwd is the pixel kernel of the reflection backward; refl_bwd_tail (sort + run combine + unpack) is "
                            "timed on the stream it runs on: with the asynchronous tail that is the library's side stream, where it runs BESIDE the tile "
                            "backward and its events span that kernel too (0.8 ms for 0.13 ms of work) - it is not on the step's critical path",
            "step_algorithmic_GBps": round(nv * (fwd_bytes + bwdall_bytes + refl_bytes) / (ms_per_step * 1e-3) / 1e9, 1),
            "step_frac_of_hbm_peak": round(nv * (fwd_bytes + bwdall_bytes + refl_bytes) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "step_frac_of_measured_ceiling": round(nv * (fwd_bytes + bwdall_bytes + refl_bytes) / (ms_per_step * 1e-3) / 1e9 / HBM_MEASURED_GBS, 4),
            "roofline": roof,
        }
        if n_color is not None:
            # the other kernels of the step against the HBM roofline, with the bytes each MOVES for this input: the per-Gaussian backward skips the
            # 192-byte SH row of surfels without a colour gradient (SURVEY.md 8d's 871 B assume every surfel blends; 60 B of per-view outputs
            # nobody asked for are not written either: 811)
            st_ms = {k: v[0] / max(1, v[1]) for k, v in stages.items() if v[1] > 0}
            per_kernel = {}
            for name, key, upper, moved, why in (
                    ("surfel_preprocess_kernel", "preprocess", 347 * P, 339 * P, "339 B per surfel (SURVEY.md 8d's 347: inputs 232 + records and per-Gaussian state, less the 8-byte means2D array that is no longer kept beside the record)"),
                    ("surfel_preprocess_bwd_kernel", "preprocess_bwd", 871 * P, 369 * P + 192 * n_color,
                     "what this kernel reads and writes per surfel: 80 (accumulator row) + 12 + 4 + 1 + 16 + 8 (means, radii, clamp flags, rotation, scale) "
                     "read, 248 written (dL_dmean2D 12, dL_dmean3D 12, dL_dsh 192, dL_dscale 8, dL_drot 16, opacity 4, refl 4) = 369, + the 192-byte SH row "
                     "of the %d of %d surfels with a colour gradient; SURVEY.md 8d's 871 add 308 B of zero-initialised gradient tensors this design "
                     "does not have and outputs nobody asked for" % (n_color, P)),
                    ("surfel_render_fwd_wave_kernel", "render_fwd", 85 * R + (68 + (0 if args.unfused else 44)) * HW, 85 * R + (68 + (0 if args.unfused else 44)) * HW,
                     "85 B per instance + 68 B per pixel (SURVEY.md 8d)" + ("" if args.unfused else " + 44 B per pixel written by the reflection epilogue")),
                    ("deferred_refl_bwd_entries_kernel", "refl_bwd", 112 * HW, 112 * HW, "112 B per pixel (SURVEY.md 8d, reflection backward)")):
                if st_ms.get(key):
                    per_kernel[name] = roofline_entry(round(st_ms[key], 4), moved, upper, why, pmc_summary(name, P, W, H))
            out["roofline_other_kernels"] = per_kernel
        out["ranks"] = {"world_size": ranks, "backend": backend or "none", "ms_per_step_per_rank": rank_ms}
        if dist_on:
            out["nccl_ranks"] = ranks if backend == "nccl" else 0   # dist.get_world_size() of the RCCL process group (0: a gloo rehearsal)
            out["allreduce_ms"] = round(allreduce_ms, 4)            # median over the instrumented steps, max over ranks, events on the step's stream
            out["allreduce"] = {"payload_MB": round(scene_payload_mb, 1), "inside_step": True,
                                "overlapped_with_next_step_ms_per_step": None if overlap_ms is None else round(overlap_ms, 4),
                                "what": "value / ms_per_step: all-reduce inside every step (what a training step pays); the overlapped figure "
                                        "hides it behind the next step's rendering and is not achievable with an optimizer in the loop",
                                "xgmi_model_ms": xgmi_model_ms(scene_payload_mb, world)}
        if c4 is not None:
            out["c4_one_gpu"] = c4
        if heavy is not None:
            out["c3_heavy"] = heavy
        if full is not None:
            out["full_train_step"] = full
        if not args.no_c5 and world == 1:
            torch.cuda.empty_cache()
            out["c5"] = c5_object(S, dev, settle_steps=args.settle_steps // 2)
        if not args.no_dropin and world == 1:
            torch.cuda.empty_cache()
            out["dropin"] = dropin_object(args, S, dev)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(S, P, W, H, args.mu, args.cubemap)
            if "c5" in out:
                try:
                    out["c5"]["cpu_baseline"] = cpu_baseline_c5(S)
                except Exception as ex:
                    out["c5"]["cpu_baseline"] = {"error": repr(ex)}
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


def launch_ranks(n):
    """Parent side of `python bench.py --gpus N` (N > 1) outside torchrun: starts `python -m torch.distributed.run --nproc-per-node N` on this
    very script with the same arguments as a CHILD process (this process has made no GPU call and makes none), passes the children's
    stderr through, prints rank 0's JSON line once and returns the exit status.  A line whose n_gpus differs from N is an error."""
    import socket
    import subprocess
    with socket.socket() as sock:          # a free rendezvous port on the loopback interface (the container hostname may not resolve)
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        elif ln.strip():
            print(ln, file=sys.stderr)
    if proc.returncode != 0:
        print("bench: the rank processes exited with status %d" % proc.returncode, file=sys.stderr, flush=True)
        return proc.returncode or 1
    if line is None:
        print("bench: rank 0 printed no result line", file=sys.stderr, flush=True)
        return 3
    got = json.loads(line).get("n_gpus")
    if got != n:
        print("bench: asked for %d ranks, the result line reports %r" % (n, got), file=sys.stderr, flush=True)
        return 4
    print(line, flush=True)
    return 0


def xgmi_model_ms(payload_mb, n):
    """Analytic all-reduce time on MI355X's xGMI (7 point-to-point links x ~153 GB/s per GPU): a ring is bound
    by one link, 2 (n-1)/n S / 153 GB/s; a direct full-mesh reduce-scatter + all-gather uses n-1 links at once, 2 (S/n) / 153 GB/s."""
    S = payload_mb * 1e6
    return {"ring": round(2 * (n - 1) / n * S / 153e9 * 1e3, 3), "full_mesh": round(2 * (S / n) / 153e9 * 1e3, 3)}


def full_train_step(args, scene, views, S, dev, means2D, sync_all, dist_on, world, views_total):
    """Secondary figure: the END-TO-END training step of the reference's loop (train.py:144-306) for this
    rank's views: render -> (1 - lambda) L1 + lambda (1 - SSIM) against a synthetic ground-truth image -> backward (gradients
    accumulated on the device) -> gradient all-reduce -> Adam over all eight parameter groups.  Never `value`."""
    import _gsr
    from gaussian_renderer import deferred_reflection, rasterize_reflect, surface_pass
    from gsr_train import DEFAULT_LRS, GaussianTrainState
    from utils.loss_utils import normal_consistency_loss, photometric_loss
    H, W = args.height, args.width
    tensors = {k: v.detach().clone() for k, v in scene.p.items()}
    mask = scene.mask
    scene.release()
    # all learning rates 0: Adam does its full arithmetic and memory traffic but the scene stays the C3 configuration
    # (with real rates the random target image changes opacities/scales within a few steps and the render cost drifts)
    sharded = bool(args.sharded_adam and dist_on)
    rank = int(os.environ.get("RANK", "0"))
    st = GaussianTrainState(tensors, dev, lrs={k: 0.0 for k in DEFAULT_LRS}, shard=(rank, world) if sharded else None)
    del tensors
    closer = None
    if sharded:
        from gsr_dist import ShardedStep
        closer = ShardedStep(st)
    fsink, frsink = st.grads.sink(), st.grads.sink(names=("cubemap", "fail"))
    fenv = EnvMap(st.p["cubemap"], st.p["fail"])
    gt_image = torch.rand(3, H, W, generator=torch.Generator(device="cpu").manual_seed(1003)).to(dev)

    def full_step(it):
        st.update_learning_rate(it)
        loss = None
        for i, view in enumerate(views):
            view.rasterizer.set_grad_sink(fsink, accumulate=i > 0)
            means2D.grad = None
            kw = dict(means3D=st.p["means3D"], means2D=means2D, opacities=st.p["opacities"], shs=st.p["shs"], refl_strengths=st.p["refl_strengths"],
                      scales=st.p["scales"], rotations=st.p["rotations"], env_scope_mask=mask)
            if args.unfused:
                base, radii, allmap, refl_map, gw, normal_view = view.rasterizer(**kw)
                final, _, rend_normal = deferred_reflection(normal_view, base, refl_map, fenv, view.ct["viewmatrix"], view.HWK, view.ct["R"], view.ct["T"],
                                                            grad_sink=frsink, accumulate=i > 0, async_tail=not args.sync_reflection_tail)
            else:
                final, _, rend_normal, base, radii, allmap, refl_map, gw = rasterize_reflect(
                    view.rasterizer, fenv, view.ct["viewmatrix"], view.HWK, view.ct["R"], view.ct["T"], refl_grad_sink=frsink, accumulate=i > 0,
                    async_tail=not args.sync_reflection_tail, **kw)
            # the reference's iteration (train.py:144-196): surface pass of render(), photometric + normal-consistency loss
            surf_depth, surf_normal = surface_pass(allmap, view, 0.0)
            loss = photometric_loss(final, gt_image, 0.2) + normal_consistency_loss(rend_normal, surf_normal, 0.05)
            loss.backward()
        if closer is not None:
            closer.step()               # reduce-scatter -> Adam over this rank's 1/N of the flat buffers -> all-gather of the parameters
        else:
            st.grads.all_reduce()
            st.optimizer.step()
        return loss

    settle(lambda: full_step(1), args, -(-views_total // world))
    for i in range(args.warmup):
        full_step(i + 1)
    sync_all()
    t2 = time.perf_counter()
    for i in range(args.steps):
        loss = full_step(args.warmup + i + 1)
    sync_all()
    fdt = time.perf_counter() - t2
    _gsr.profile_enable(True)       # instrumented repeat (see main)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    for i in range(args.steps):
        marks[i].record()
        loss = full_step(args.warmup + args.steps + i + 1)
    marks[args.steps].record()
    sync_all()
    fstages = _gsr.profile_collect()
    _gsr.profile_enable(False)
    for view in views:
        view.rasterizer.set_grad_sink(None)
    if dist_on:
        import torch.distributed as dist
        tmax = torch.tensor([fdt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        fdt = float(tmax.item())
    nv = len(views)
    n_floats = st.params.total
    return {"ms_per_step": round(fdt / args.steps * 1e3, 4), "views_per_s": round(views_total * args.steps / fdt, 3),
            "exchange": ("reduce-scatter -> Adam over 1/%d of the flat buffer per rank -> all-gather (--sharded-adam)" % world) if sharded else
                        ("all-reduce -> the same Adam over all parameters on every rank" if dist_on else "none (one rank)"),
            # what the two forms of the exchange + optimizer cost by the xGMI model (bytes on the wire are the same; Adam moves 28 B per
            # parameter at the ~5.7 TB/s the fused kernel reaches): what --sharded-adam changes is the optimizer term
            "exchange_model_ms": {"all_reduce_then_adam": dict(xgmi_model_ms(n_floats * 4 / 1e6, world), adam=round(n_floats * 28 / 5.7e12 * 1e3, 3)),
                                  "reduce_scatter_sharded_adam_all_gather": dict(xgmi_model_ms(n_floats * 4 / 1e6, world),
                                                                                 adam=round(n_floats * 28 / 5.7e12 * 1e3 / world, 3))} if dist_on else None,
            "step_ms": percentiles([marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]),
            "what": "the reference's training iteration: render (rasterizer + surface pass + reflection chain) + L1/SSIM and normal-consistency losses + backward + grad all-reduce + fused Adam (59 floats/Gaussian + cubemap), learning rates 0 so the workload stays C3",
            "final_loss": round(float(loss.item()), 6),
            "stage_ms_per_view": {k: round(v[0] / max(1, args.steps * (1 if k == "adam" else nv)), 4) for k, v in fstages.items() if v[1] > 0}}


def dropin_object(args, S, dev):
    """The path a user of the reference gets by switching the imports and nothing else: C3 through `gaussian_renderer.render()` ->
    l1_loss / ssim / the normal-consistency expression as train.py:144-196 writes them -> `loss.backward()` with PLAIN autograd (no
    gradient sinks, no asynchronous reflection tail, parameters as ordinary leaf tensors, the environment map a CubemapEncoder module),
    and render FPS through `render_fast()` timed the way the reference's eval_fps.py:48-59 times it (time.time() around every call,
    no explicit synchronisation: the call's own num_rendered read-back is the only one) with the synchronised figure beside it."""
    from cubemapencoder import CubemapEncoder
    from gaussian_renderer import render, render_fast
    from utils.loss_utils import l1_loss, ssim
    P, W, H = args.gaussians, args.width, args.height
    sc = S.make_scene(P, "S", seed=1003, mu=args.mu)
    tex, fail = S.make_cubemap(args.cubemap, 3, 1003)
    cam = S.make_camera(W, H)
    ct = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in cam.items() if isinstance(v, np.ndarray)}
    t = {k: torch.from_numpy(sc[k]).to(dev).requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations", "refl_strengths")}
    env = CubemapEncoder(output_dim=3, resolution=args.cubemap).to(dev)
    with torch.no_grad():
        env.params["Cubemap_texture"].copy_(torch.from_numpy(tex))
        env.params["Cubemap_failv"].copy_(torch.from_numpy(fail))

    class View:          # what render() reads from a scene.cameras.Camera
        FoVx, FoVy, image_width, image_height = cam["FoVx"], cam["FoVy"], W, H
        world_view_transform, full_proj_transform, camera_center = ct["viewmatrix"], ct["projmatrix"], ct["campos"]
        HWK, R, T, znear, zfar = (H, W, cam["K"]), ct["R"], ct["T"], cam["znear"], cam["zfar"]

    class Pipe:          # arguments.PipelineParams defaults
        depth_ratio, compute_cov3D_python, convert_SHs_python, debug = 0.0, False, False, False

    class PC:            # what render() reads from a scene.gaussian_model.GaussianModel
        get_xyz, get_opacity, get_scaling, get_rotation, get_features, get_refl = (t["means3D"], t["opacities"], t["scales"], t["rotations"], t["shs"],
                                                                                   t["refl_strengths"])
        active_sh_degree, get_envmap = 3, env
    bg = torch.zeros(3, device=dev)
    gt_image = torch.rand(3, H, W, generator=torch.Generator(device="cpu").manual_seed(1003)).to(dev)
    leaves = list(t.values()) + list(env.parameters())

    def train_step():
        for x in leaves:
            x.grad = None
        pkg = render(View, PC, Pipe, bg)
        image = pkg["render"]
        Ll1 = l1_loss(image, gt_image)
        loss = (1.0 - 0.2) * Ll1 + 0.2 * (1.0 - ssim(image, gt_image))
        normal_error = (1 - (pkg["rend_normal"] * pkg["surf_normal"]).sum(dim=0))[None]
        loss = loss + 0.05 * normal_error.mean()
        loss.backward()
        return loss

    K, Wm = args.steps, args.warmup
    settle(train_step, args)
    for _ in range(Wm):
        train_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        loss = train_step()
    torch.cuda.synchronize()
    step_ms = (time.perf_counter() - t0) / K * 1e3
    with torch.no_grad():
        for _ in range(3):
            render_fast(View, PC, Pipe, bg)
        torch.cuda.synchronize()
        n = max(20, K)
        times = []
        t1 = time.perf_counter()
        for _ in range(n):
            a = time.time()
            res = render_fast(View, PC, Pipe, bg)
            rgb = res["render"]                      # (the reference applies its PPISP post-process here; out of scope)
            times.append(time.time() - a)
        torch.cuda.synchronize()
        fps_sync = n / (time.perf_counter() - t1)
    del t, env, leaves, rgb, res
    torch.cuda.empty_cache()
    return {"what": "reference entry points with plain autograd: gaussian_renderer.render() + l1_loss + ssim + normal-consistency expression + "
                    "loss.backward() (train.py:144-198; no sinks, no async tail, no fused loss), and render_fast() timed as eval_fps.py:48-59",
            "train_step_ms": round(step_ms, 4), "train_steps_per_s": round(1e3 / step_ms, 2), "final_loss": round(float(loss.item()), 6),
            "render_fast_fps_eval_fps_method": round(1.0 / float(np.mean(times)), 2), "render_fast_fps_synchronised": round(fps_sync, 2),
            "render_fast_calls": n}


def c5_object(S, dev, steps=20, settle_steps=12):
    """BASELINE config C5 on this GPU: 5e6 Gaussians, 1920x1080, SH 3, the 3DGS rasterizer (variant G) with anti-aliasing and the
    inverse-depth (depth-regularisation) backward enabled, forward + backward.  Round 4: the parameter gradients go through gradient
    sinks into one flat buffer as in the C3 step (variant G has them now); the plain-autograd form of rounds 1-3 is timed beside it
    with the caching allocator's counters per step, which is what names the 60-ms step a driver run of round 3 contained.  Per-stage
    hipEvent times and the HBM roofline of the two dominant backward kernels (algorithmic bytes: SURVEY.md §8d, variant G)."""
    import _gsr
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from gsr_dist import FlatGrads
    P, W, H = 5_000_000, 1920, 1080
    sc = S.make_scene(P, "G", seed=1005, mu=-5.3)
    cam = S.make_camera(W, H)
    names = ("means3D", "shs", "opacities", "scales", "rotations", "refl_strengths", "normals")
    t = {k: torch.from_numpy(sc[k]).to(dev).requires_grad_(True) for k in names}
    ct = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in cam.items() if isinstance(v, np.ndarray)}
    del sc
    rast = GaussianRasterizer(GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"],
                                                            bg=torch.zeros(3, device=dev), scale_modifier=1.0, viewmatrix=ct["viewmatrix"],
                                                            projmatrix=ct["projmatrix"], sh_degree=3, campos=ct["campos"], prefiltered=False,
                                                            antialiasing=True, debug=False))
    g = S.make_upstream_grads(H, W, 1005)
    gc, gi, gn, gr = (torch.from_numpy(g[k]).to(dev) for k in ("dL_dcolor", "dL_dinvdepth", "dL_dnormal", "dL_drefl"))
    means2D = torch.zeros(P, 3, device=dev, requires_grad=True)
    R = [0]

    plain = [True]

    def step():
        means2D.grad = None
        if plain[0]:            # plain autograd: the parameter gradients of the last step are dropped, new ones allocated by the backward
            for x in t.values():
                x.grad = None
        color, radii, invd, nmap, rmap = rast(means3D=t["means3D"], means2D=means2D, opacities=t["opacities"], shs=t["shs"], normals=t["normals"],
                                              refl_strengths=t["refl_strengths"], scales=t["scales"], rotations=t["rotations"])
        R[0] = color.grad_fn.num_rendered
        torch.autograd.backward([color, invd, nmap, rmap], [gc, gi, gn, gr])

    def timed(n):
        """n steps back to back between two synchronisations (ms_per_step: what a training loop pays; a loop that synchronises after every step
        adds the host's launch latency of the next forward, ~0.15 ms at this size, which no loop has to), one event per step for the per-step
        list, and the caching allocator's counters around every step (host-side reads, no synchronisation)."""
        import gc as pygc
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        allocs, host_ms, gc_events, cur = [], [], [], [0, 0.0]

        def on_gc(phase, info):       # a collection of the Python garbage collector inside the loop: which step, which generation, how long
            if phase == "start":
                cur[1] = time.perf_counter()
            else:
                gc_events.append({"step": cur[0], "generation": info.get("generation"), "ms": round((time.perf_counter() - cur[1]) * 1e3, 2),
                                  "collected": info.get("collected")})
        pygc.callbacks.append(on_gc)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        marks[0].record()
        for i in range(n):
            cur[0] = i
            m0 = torch.cuda.memory_stats(dev)
            h0 = time.perf_counter()
            step()
            host_ms.append(round((time.perf_counter() - h0) * 1e3, 3))
            marks[i + 1].record()
            m1 = torch.cuda.memory_stats(dev)
            d = {k: int(m1.get(k, 0) - m0.get(k, 0)) for k in ("num_alloc_retries", "num_device_alloc", "num_device_free")}
            d["allocated_MB"] = round((m1.get("allocated_bytes.all.allocated", 0) - m0.get("allocated_bytes.all.allocated", 0)) / 1e6, 1)
            allocs.append(d)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / n * 1e3
        pygc.callbacks.remove(on_gc)
        return wall, [marks[i].elapsed_time(marks[i + 1]) for i in range(n)], allocs, host_ms, gc_events

    # ---- plain autograd (rounds 1-3): ~2 GB of gradient tensors allocated per step, freed when the next step drops .grad
    for _ in range(2 + settle_steps):       # (settle_steps: main's --settle-steps, halved — a C5 step is 1.5 C3 steps long)
        step()
    torch.cuda.synchronize()
    plain_wall, plain_ms, plain_alloc, plain_host, plain_gc = timed(max(5, steps // 2))
    for x in t.values():
        x.grad = None
    plain[0] = False
    # ---- gradient sinks: the per-Gaussian backward writes the seven parameter gradients into one flat buffer
    fg = FlatGrads(t)
    rast.set_grad_sink(fg.sink(), accumulate=False)
    for _ in range(2 + settle_steps):
        step()
    torch.cuda.synchronize()
    ms, per_step, sink_alloc, host_ms, gc_events = timed(steps)           # no stage timers (each costs two event records: ~0.1 ms per step in all)
    _gsr.profile_enable(True)                         # the same steps again for the stage table
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    st = _gsr.profile_collect()
    _gsr.profile_enable(False)
    stage = {k: round(v[0] / steps, 4) for k, v in st.items() if v[1] > 0}
    HW = W * H
    # Gaussians whose colour gradient is non-zero: the per-Gaussian backward reads the 192-byte SH row only for those (it still writes the
    # zero rows of dL_dsh in overwrite mode), so the bytes it MOVES follow from the data, not from P alone
    n_color = int((fg.view("shs")[:, 0, :] != 0).any(dim=1).sum().item())
    kernels = {}
    for name, key, upper, moved, why in (
            ("gauss_preprocess_bwd_kernel", "preprocess_bwd", 639 * P, 381 * P + 192 * n_color,
             "what this kernel reads and writes per Gaussian: 64 (accumulator row) + 12 + 4 + 4 + 12 + 16 + 1 (means, radii, opacity, scale, "
             "rotation, clamp flags; the 3D covariance is recomputed, not read) read, 268 written (dL_dmean2D_pixels 12, dL_dnormal 12, opacity 4, refl 4, inverse depth 4, dL_dmean3D 12, dL_dsh 192, "
             "dL_dscale 12, dL_drot 16) = 381, + the 192-byte SH row of the %d of %d Gaussians with a colour gradient; SURVEY.md 8d's 639 count "
             "every SH row and outputs nobody asked for" % (n_color, P)),
            ("gauss_render_bwd_wave_kernel", "render_bwd", 124 * R[0] + 40 * HW, 124 * R[0] + 40 * HW, "124 B per instance + 40 B per pixel (SURVEY.md 8d)")):
        if stage.get(key):
            pmc = pmc_summary(name, P, W, H, PMC_FILE_C5)
            kernels[name] = roofline_entry(stage[key], moved, upper, why, pmc)
    del t, means2D, fg
    torch.cuda.empty_cache()
    return {"workload": "C5: 5M Gaussians, 1920x1080, SH deg 3, variant G, anti-aliasing + inverse-depth backward, fwd+bwd; parameter gradients through "
                        "gradient sinks into one flat buffer (as the C3 step)", "num_rendered": R[0],
            "ms_per_step": round(ms, 4), "ms_per_step_all": [round(x, 3) for x in per_step], "steps": steps,
            "timing": "ms_per_step: wall clock over the steps back to back between two synchronisations; ms_per_step_all: one event per step",
            "stage_ms_per_step": stage,
            "kernel_sum_ms": round(sum(stage.values()), 4), "step_over_kernel_sum": round(ms / max(1e-9, sum(stage.values())), 4),
            "allocator_per_step": {"what": "torch.cuda.memory_stats() deltas around each synchronised step (caching allocator): num_device_alloc > 0 or "
                                           "num_alloc_retries > 0 inside the loop means the step went to hipMalloc / freed cached blocks and retried",
                                   "sinks": alloc_summary(sink_alloc), "plain_autograd": alloc_summary(plain_alloc)},
            "host_ms_per_step": host_ms, "python_gc_during_timed_steps": gc_events,
            "outliers": {"what": "a step far above the others: host_ms_per_step says whether the HOST was held up inside that step's calls (then "
                                 "python_gc_during_timed_steps / allocator_per_step name the cause) or only the GPU time between its events grew",
                         "steps_above_1.5x_median": [i for i, x in enumerate(per_step) if x > 1.5 * sorted(per_step)[len(per_step) // 2]]},
            "plain_autograd": {"ms_per_step": round(plain_wall, 4), "ms_per_step_all": [round(x, 3) for x in plain_ms],
                               "what": "the same step with the ~2 GB of parameter gradients allocated by autograd every step (rounds 1-3)"},
            "roofline": kernels}


def alloc_summary(rows):
    """Per-step allocator deltas, compressed: the bytes served per step and the steps in which the caching allocator went to the device."""
    mb = sorted(r["allocated_MB"] for r in rows)
    return {"allocated_MB_per_step_median": mb[len(mb) // 2] if mb else None,
            "steps_with_device_alloc_free_or_retry": [dict(step=i, **{k: v for k, v in r.items() if k != "allocated_MB"}) for i, r in enumerate(rows)
                                                       if r["num_alloc_retries"] or r["num_device_alloc"] or r["num_device_free"]]}


def roofline_entry(ms, moved, upper, why, pmc):
    """HBM roofline of one kernel launch: `achieved` = the bytes the kernel MOVES by its algorithm for THIS input (where a kernel skips work
    by data, `moved` < `algorithmic_upper`, the SURVEY.md 8d figure for an input in which nothing is skipped) / its hipEvent-measured
    launch time; `traffic` = HBM bytes of the committed PMC pass (null when that pass belongs to another build) with the rate it implies.
    A fraction of the measured copy ceiling above 1 would mean the kernel is credited with bytes it does not move: flagged, never hidden."""
    gbs = moved / (ms * 1e-3) / 1e9
    e = {"bound": "hbm", "avg_launch_ms": ms, "algorithmic_bytes_per_launch": int(moved), "algorithmic_upper": int(upper), "bytes_note": why,
         "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
         "frac_of_measured_ceiling": round(gbs / HBM_MEASURED_GBS, 4), "traffic": pmc.get("traffic"), "traffic_source": pmc.get("source")}
    if pmc.get("traffic"):
        tg = pmc["traffic"] / (ms * 1e-3) / 1e9
        e["traffic_GBps"] = round(tg, 1)
        e["traffic_frac_of_peak"] = round(tg / HBM_PEAK_GBS, 4)
    if e["frac_of_measured_ceiling"] > 1.0:
        e["inconsistent"] = "achieved exceeds the measured copy ceiling: the byte count credits traffic the kernel does not move"
    return e


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_c5(S):
    """The CPU oracle (variant G, anti-aliasing, inverse-depth backward) on a bounded sample of C5 with all host cores: 1/4 of the step with the
    same per-tile statistics (Gaussians / 4, image / 2 per side, scales x 2), extrapolated x 4."""
    from oracle import oracle as orc
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import scene_kwargs
    f = 2
    P, W, H, mu = 5_000_000 // (f * f), 1920 // f, 1080 // f, -5.3 + math.log(f)
    kw, cam, sc = scene_kwargs("G", P, W, H, 1005, mu, 3, (0, 0, 0))
    g = S.make_upstream_grads(H, W, 1005)
    o = orc.GaussOracle(np.float32)
    t = time.perf_counter()
    ref = o.forward(antialiasing=True, **kw)
    o.backward(dL_dcolor=g["dL_dcolor"], dL_dinvdepth=g["dL_dinvdepth"], dL_dnormal_map=g["dL_dnormal"], dL_drefl_strength_map=g["dL_drefl"])
    dt = time.perf_counter() - t
    return {"value": round(1.0 / (dt * f * f), 4), "unit": "views/s", "cores": os.cpu_count(), "cpu_model": cpu_model(), "kind": "port",
            "sample": "1/%d of C5 with the same per-tile statistics (%d Gaussians, scales x %d, %dx%d, R=%d) through the CPU oracle (OpenMP, all host "
                      "cores), %.1f s, extrapolated x %d" % (f * f, P, f, W, H, ref["num_rendered"], dt, f * f)}


def isa_mix(kernel):
    """Static instruction mix of `kernel`'s hot loop (profiles/r04_isa_mix.json, tests/isa_mix.py), quoted only for the build it was taken on."""
    try:
        j = json.load(open(os.path.join(ROOT, "profiles", "r04_isa_mix.json")))
        built = open(DIGEST_FILE).read().strip()
    except (OSError, ValueError):
        return None
    if j.get("_config", {}).get("digest") != built:
        return None
    return j.get(kernel)


def pmc_summary(kernel, P, W, H, PMC_FILE=PMC_FILE):
    """Per-launch counters of `kernel` from the committed PMC passes of this same command (profiles/r02_pmc_summary.json, written
    by tests/pmc_summary.py from separate `rocprofv3 --pmc ... --kernel-trace` runs on the GPU box).  FETCH_SIZE / WRITE_SIZE are in
    KB; FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at 64, MI355X_MICROARCH.md HBM section).  The counters cannot be read
    from inside the timed process, so they are QUOTED (source says from where); empty when the file is absent or was collected on
    another configuration."""
    if not os.path.exists(PMC_FILE):
        return {"source": "none: %s is absent" % os.path.relpath(PMC_FILE, ROOT)}
    with open(PMC_FILE) as f:
        d = json.load(f)
    cfg = d.get("_config", {})
    if (cfg.get("P"), cfg.get("W"), cfg.get("H")) != (P, W, H):
        return {"source": "none: %s was collected on another configuration" % os.path.relpath(PMC_FILE, ROOT)}
    # the counters belong to ONE build of the kernels: the summary carries the build digest it was taken on (tests/pmc_summary.py) and is
    # quoted only while the library in this tree has that digest — after a kernel change the old counters are refused, not quoted stale
    built = open(DIGEST_FILE).read().strip() if os.path.exists(DIGEST_FILE) else None
    if not cfg.get("digest") or cfg.get("digest") != built:
        return {"source": "none: %s was collected on build %s, this library is build %s" % (os.path.relpath(PMC_FILE, ROOT), str(cfg.get("digest"))[:12],
                                                                                           str(built)[:12])}
    for k, v in d.items():
        if kernel in k:
            out = {"source": "quoted from %s (rocprofv3 --pmc passes of `%s`, %s, build %s, commit %s)" % (
                os.path.relpath(PMC_FILE, ROOT), cfg.get("cmd", "bench.py"), cfg.get("when", "this round"), str(cfg.get("digest"))[:12], cfg.get("commit", "?"))}
            if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
                out["traffic"] = int((2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024)
            if "SQ_INSTS_VALU" in v:
                out["insts_valu"] = int(v["SQ_INSTS_VALU"])
            if "SQ_INSTS_SALU" in v:
                out["insts_salu"] = int(v["SQ_INSTS_SALU"])
            return out
    return {"source": "none: %s has no row for %s" % (os.path.relpath(PMC_FILE, ROOT), kernel)}


def cpu_baseline(S, P, W, H, mu, L):
    """The CPU oracle (test infrastructure, oracle/) timed on this host: ONE full step of the same workload (rasterizer fwd+bwd +
    cubemap lookup fwd+bwd), OpenMP over all host cores; and a bounded single-thread sample: the same scene statistics at 1/4
    of the size (Gaussians / 4, image / 2 per side, scales x 2, hence the same tiles per Gaussian and list length per tile),
    OMP_NUM_THREADS = 1, extrapolated x 4.  (The oracle's backward accumulates with `omp atomic`: it scales poorly to hundreds
    of cores, which is why the single-thread figure is reported beside the all-core one.)"""
    import ctypes
    from oracle import oracle as orc
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import scene_kwargs

    def one_step(Pn, Wn, Hn, mun, seed):
        kw, cam, sc = scene_kwargs("S", Pn, Wn, Hn, seed, mun, 3, (0, 0, 0))
        g = S.make_upstream_grads(Hn, Wn, seed)
        tex, fail = S.make_cubemap(L, 3, seed)
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
        return time.perf_counter() - t, ref["num_rendered"]

    dt, R = one_step(P, W, H, mu, 1003)
    out = {"value": round(1.0 / dt, 4), "unit": "views/s", "cores": os.cpu_count(), "cpu_model": cpu_model(), "kind": "port",
           "sample": "one full C3 step (%d Gaussians, %dx%d, R=%d) through the CPU oracle (OpenMP, all host cores), %.1f s" % (P, W, H, R, dt)}
    # "C1 and C2 are timed fully on CPU"
    try:
        c1, c2 = S.CONFIGS[1], S.CONFIGS[2]
        kw, _, _ = scene_kwargs("S", c1["P"], c1["W"], c1["H"], 1001, c1["mu"], c1["sh_degree"], (0, 0, 0))
        o1 = orc.SurfelOracle(np.float32)
        t = time.perf_counter()
        r1 = o1.forward(**kw)
        d1 = time.perf_counter() - t
        d2, R2 = one_step(c2["P"], c2["W"], c2["H"], c2["mu"], 1002)
        out["c1"] = {"value": round(1.0 / d1, 3), "unit": "views/s", "cores": os.cpu_count(), "kind": "port",
                     "sample": "C1 in full: %d surfels, %dx%d, SH degree 0, forward only (R=%d), %.3f s" % (c1["P"], c1["W"], c1["H"], r1["num_rendered"], d1)}
        out["c2"] = {"value": round(1.0 / d2, 3), "unit": "views/s", "cores": os.cpu_count(), "kind": "port",
                     "sample": "C2 in full: %d surfels, %dx%d, SH degree 3, forward + backward + cubemap lookup both ways (R=%d), %.2f s" % (c2["P"], c2["W"], c2["H"], R2, d2)}
    except Exception as ex:     # the headline's baseline above must survive a failure here
        out["c1_c2_error"] = repr(ex)
    try:
        omp = ctypes.CDLL("libgomp.so.1")
        omp.omp_get_max_threads.restype = ctypes.c_int
        before = omp.omp_get_max_threads()
        omp.omp_set_num_threads(1)
        try:
            f = 2          # 1/4 of the step: ~10 s of single-thread work
            dts, Rs = one_step(max(1000, P // (f * f)), W // f, H // f, mu + math.log(f), 1003)
        finally:
            omp.omp_set_num_threads(before)
        out["single_thread"] = {"value": round(1.0 / (dts * f * f), 6), "unit": "views/s", "cores": 1, "kind": "port",
                                "sample": "1/%d of C3 with the same per-tile statistics (%d Gaussians, scales x %d, %dx%d, R=%d), "
                                          "OMP_NUM_THREADS=1, %.1f s, extrapolated x %d" % (f * f, max(1000, P // (f * f)), f, W // f, H // f, Rs, dts, f * f)}
    except OSError:
        pass
    return out


if __name__ == "__main__":
    main()
