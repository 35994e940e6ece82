"""View-parallel training support: frames shard one-per-GPU, all Gaussian parameters and the
cubemap are replicated, and the only exchange is ONE all-reduce (RCCL over xGMI on the GPU box; gloo in the CPU
tests) of a flat, pre-packed gradient buffer: 59 floats per Gaussian (xyz 3, SH 48, opacity 1, scale 2, rotation 4,
reflection strength 1) + the cubemap texels + the fail value.  The reference has no distributed code at all; this is
new code outside its API.

xGMI note: 7 point-to-point links x ~153 GB/s per GPU.  A ring all-reduce of S bytes is per-link bound
(~2*(7/8)*S / 153 GB/s = 2.7 ms for the 236 MB of 1 M Gaussians), a direct full-mesh reduce-scatter + all-gather
uses all links (~0.4 ms); the collective algorithm is RCCL's choice, the payload is kept in one contiguous buffer so
that either works on a single large message.
"""
import torch
import torch.distributed as dist


def shard_views(n_views, rank, world_size):
    """Views {rank, rank + world, ...} of a batch go to this rank (round-robin keeps the per-rank count within one)."""
    return list(range(rank, n_views, world_size))


class FlatGrads:
    """Owns one flat float32 buffer and makes every parameter's .grad a view into it, so that autograd accumulates
    straight into the all-reduce payload (no packing pass)."""

    def __init__(self, params):
        """params: dict name -> leaf tensor (requires_grad).  Order of the dict = order in the buffer."""
        self.params = params
        self.names = list(params.keys())
        # every slice starts on a 16-byte boundary (as gsr_train.FlatParams does): the per-Gaussian backward kernels store
        # dL_dsh and dL_drot rows as float4 into these views when they are gradient sinks, whatever P is
        first = next(iter(params.values()))
        self.slices = {}
        off = 0
        for k, p in params.items():
            if p.dtype != torch.float32:
                raise TypeError(f"{k}: FlatGrads packs float32 parameters only")
            n = p.numel()
            self.slices[k] = (off, off + n)
            off += (n + 3) // 4 * 4
        self.flat = torch.zeros(off, dtype=torch.float32, device=first.device)
        for k, p in params.items():
            a, b = self.slices[k]
            p.grad = self.flat[a:b].view(p.shape)

    @classmethod
    def mirroring(cls, flat_params):
        """Gradient buffer with exactly the layout of a gsr_train.FlatParams (same offsets incl. alignment padding), as
        the fused Adam kernel requires."""
        self = cls.__new__(cls)
        self.params = flat_params.p
        self.names = list(flat_params.names)
        self.flat = torch.zeros_like(flat_params.flat)
        self.slices = dict(flat_params.slices)
        for k, p in self.params.items():
            a, b = self.slices[k]
            p.grad = self.flat[a:b].view(p.shape)
        return self

    def twin(self):
        """A second gradient buffer with the same layout that does NOT take over the parameters' .grad views: the other
        half of a double buffer for gradient sinks, so that the all-reduce of one step can run while the next step's
        backward writes the other buffer."""
        other = FlatGrads.__new__(FlatGrads)
        other.params, other.names, other.slices = self.params, list(self.names), dict(self.slices)
        other.flat = torch.zeros_like(self.flat)
        return other

    def _join(self):
        """Work the HIP library still has on its side stream for this buffer (deferred_reflection(async_tail=True)) is ordered
        before whatever the current stream does next."""
        if self.flat.is_cuda:
            import _gsr
            _gsr.side_join(self.flat.device)

    def zero_(self):
        self._join()
        self.flat.zero_()

    def sink(self, names=("means3D", "shs", "opacities", "scales", "rotations", "refl_strengths", "normals")):
        """Views of the flat buffer keyed by the rasterizer's gradient names (those this buffer holds; `normals` is a parameter of
        variant G only), for GaussianRasterizer.set_grad_sink: the backward kernels then write straight into the all-reduce payload
        (no zero-fill, no accumulate pass).  Several views per step and rank: the first backward of the step with
        set_grad_sink(sink, accumulate=False) overwrites the buffer, every further one with accumulate=True adds to it on the device
        (kernel `+=`; bench.py's step_into, tests/test_gpu_api_paths.py::test_grad_sink_accumulates_views_on_the_device)."""
        return {k: self.view(k) for k in names if k in self.slices}

    def zero_except_(self, names):
        """Zero the slices that are still accumulated by autograd (everything not covered by a sink)."""
        for k, (a, b) in self.slices.items():
            if k not in names:
                self.flat[a:b].zero_()

    def view(self, name):
        a, b = self.slices[name]
        return self.flat[a:b].view(self.params[name].shape)

    def all_reduce(self, group=None, average=False):
        """Sum (or mean) of the per-rank gradients, in place.  No-op without an initialised process group."""
        self._join()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            if average:
                self.flat.div_(dist.get_world_size(group))
        return self.flat

    def all_reduce_async(self, group=None):
        """Start the sum of the per-rank gradients and return the work handle (None without a process group of more than
        one rank).  `handle.wait()` makes the CURRENT STREAM wait for the result (no host block with RCCL); until then
        nothing may write this buffer."""
        self._join()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=True)
        return None


class ShardedStep:
    """The exchange + optimizer part of a view-parallel training step in its sharded form (ZeRO-1 shape):

        reduce-scatter(flat gradients)  ->  Adam over THIS rank's 1/N of the flat buffers  ->  all-gather(flat parameters)

    instead of all-reduce(flat gradients) -> the same Adam over all parameters on every rank.  The bytes on the wire are the same
    (an all-reduce IS a reduce-scatter followed by an all-gather; on xGMI 2 (N-1)/N S per rank either way), but each rank runs
    the optimizer over 1/N of the 59 floats per Gaussian (0.30 -> 0.04 ms at N = 8, 10^6 Gaussians) and holds 1/N of the Adam
    moments.  `state` is a gsr_train.GaussianTrainState built with shard=(rank, world_size); with world_size 1 (or no process
    group) this is exactly all_reduce() + optimizer.step()."""

    def __init__(self, state, group=None):
        self.state, self.group = state, group
        on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if on else 1
        self.rank = dist.get_rank(group) if on else 0
        total = state.params.total
        if state.shard is not None and state.shard != (self.rank, self.world):
            raise ValueError(f"train state was built for shard {state.shard}, the process group says {(self.rank, self.world)}")
        if self.world > 1 and (state.shard is None or total % (4 * self.world)):
            raise ValueError("ShardedStep needs GaussianTrainState(..., shard=(rank, world_size)): equal, 16-byte-aligned chunks")
        n = total // self.world
        self.range = (self.rank * n, (self.rank + 1) * n)
        self.in_place = on and dist.get_backend(group) == "nccl"     # RCCL reduces / gathers in place when the shard is the rank's chunk of the buffer
        self._gather = None        # the parameter all-gather of the last step(async_gather=True), still in flight

    def wait(self):
        """Makes the CURRENT stream wait for the parameter all-gather a step(async_gather=True) left in flight (no host block with RCCL).
        Call it before the first kernel that reads the parameters — the next forward; step() calls it itself."""
        if self._gather is not None:
            work, keep = self._gather
            self._gather = None
            work.wait()
            del keep

    def step(self, async_gather=False):
        """async_gather=True: the all-gather of the updated parameters is only STARTED (on the collective's own stream, behind the optimizer
        kernel) and the call returns; whatever the caller enqueues next that does not read the parameters — the upload of the next
        ground-truth image, the densification statistics of this step's views, host work — runs beside it, and wait() orders the next
        forward behind it.  (The next forward itself cannot start earlier: its first kernel reads every parameter tensor.)"""
        self.wait()
        st = self.state
        g, p = st.grads.flat, st.params.flat
        a, b = self.range
        st.grads._join()
        if self.world > 1:
            if self.in_place:
                dist.reduce_scatter_tensor(g[a:b], g, op=dist.ReduceOp.SUM, group=self.group)
            else:                                   # (gloo on CPU ranks in the tests: no aliasing of input and output)
                out = torch.empty(b - a, dtype=g.dtype, device=g.device)
                dist.reduce_scatter_tensor(out, g, op=dist.ReduceOp.SUM, group=self.group)
                g[a:b].copy_(out)
        st.optimizer.step()                         # steps [a, b) only: its `owned` range
        if self.world > 1:
            with torch.no_grad():
                src = p[a:b] if self.in_place else p[a:b].clone()
                if async_gather:
                    self._gather = (dist.all_gather_into_tensor(p, src, group=self.group, async_op=True), src)
                else:
                    dist.all_gather_into_tensor(p, src, group=self.group)


def reduce_densification_stats(grad_norm_sum, visible_count, max_radii, group=None):
    """The densification side channels are not plain gradient sums (scene/gaussian_model.py:579-584, train.py:242-245
    of the reference): per-view ||viewspace grad|| accumulates (sum), the visibility counter accumulates (sum) and
    max_radii2D is a running maximum (max).  All three are reduced in place."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(grad_norm_sum, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(visible_count, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(max_radii, op=dist.ReduceOp.MAX, group=group)
    return grad_norm_sum, visible_count, max_radii
