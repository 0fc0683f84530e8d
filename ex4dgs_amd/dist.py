"""Multi-GPU layer of the hot path: frames/views shard embarrassingly across ranks (one process per GPU,
Gaussian parameters replicated), and the only collective is the sum all-reduce of the per-frame gradients
for the training step (the reference itself is single-process: utils/general_utils.py:161).

Backend "nccl" is RCCL on ROCm (xGMI inside a node); "gloo" is used by the CPU tests (world_size 2).
Large gradient tensors are reduced in place, small ones travel in one flat message (few, large collectives keep
every xGMI link busy), asynchronously on the communicator's stream.  What that overlaps with depends on the step
(DESIGN.md section 6): with an optimizer in the step (the default of the multi-GPU bench) the exchange of frame i
must finish before the optimizer update at the top of step i+1, so only the part issued before the attribute
backward runs beside compute; without one it hides behind the rasterization of frame i+1.

`force=True` (or EX4D_FORCE_COLLECTIVES=1) makes the exchange classes issue their collectives even in a process group
of ONE rank: the RCCL-native branches (reduce_scatter_tensor, all_gather_into_tensor, in-place all-gather) can then be
executed and checked on a single-GPU box (tests/test_gpu_dist.py).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract).
    Returns (rank, world_size, local_rank).  No-op for single-process runs."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def _forced(force):
    return bool(force) or os.environ.get("EX4D_FORCE_COLLECTIVES", "0") == "1"


def shard_views(num_views, rank, world_size):
    """Views i with i % world_size == rank (round-robin over the shuffled stack)."""
    return list(range(rank, num_views, world_size))


class GradBuckets:
    """Sum all-reduce of a fixed list of same-dtype gradient tensors, asynchronous.

    Large tensors (>= inplace_bytes, e.g. the [P,16,3] SH gradient: 192 MB at 1M Gaussians) are reduced IN PLACE,
    one collective each -- no packing copies, which would cost as much HBM traffic as a rasterizer stage.  Small
    ones are packed into flat buckets (one collective per bucket instead of many tiny ones).
    launch(tensors) first waits for the previous exchange, then starts the collectives on the process group's own
    stream; wait() blocks until they finished and unpacks the bucketed ones.  The caller must leave the tensors
    untouched between launch() and wait().
    """

    def __init__(self, shapes, dtype=torch.float32, device="cpu", bucket_bytes=64 << 20, inplace_bytes=4 << 20, group=None):
        self.group = group
        self.shapes = [tuple(s) for s in shapes]
        self.numels = [int(torch.Size(s).numel()) for s in self.shapes]
        esz = torch.empty(0, dtype=dtype).element_size()
        self.inplace = [n * esz >= inplace_bytes for n in self.numels]
        self.assign = []           # (bucket index, offset) per bucketed tensor, None for in-place ones
        sizes = []
        for n, ip in zip(self.numels, self.inplace):
            if ip:
                self.assign.append(None)
                continue
            if not sizes or ((sizes[-1] + n) * esz > bucket_bytes and sizes[-1] > 0):
                sizes.append(0)
            self.assign.append((len(sizes) - 1, sizes[-1]))
            sizes[-1] += n
        self.flat = [torch.zeros(max(s, 1), dtype=dtype, device=device) for s in sizes]
        self.pending = []
        self._tensors = None

    def _active(self):
        return dist.is_initialized() and dist.get_world_size(self.group) > 1

    def launch(self, tensors):
        assert len(tensors) == len(self.shapes)
        self.wait()
        self._tensors = list(tensors)
        for t, a, n in zip(tensors, self.assign, self.numels):
            if a is not None:
                self.flat[a[0]][a[1]:a[1] + n].copy_(t.reshape(-1))
        if self._active():
            self.pending = [dist.all_reduce(f, op=dist.ReduceOp.SUM, group=self.group, async_op=True) for f in self.flat]
            for t, a in zip(tensors, self.assign):
                if a is None:
                    assert t.is_contiguous()
                    self.pending.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        return self

    def wait(self):
        for w in self.pending:
            w.wait()
        self.pending = []
        if self._tensors is not None:
            for t, a, n in zip(self._tensors, self.assign, self.numels):
                if a is not None:
                    t.copy_(self.flat[a[0]][a[1]:a[1] + n].view(t.shape))
            self._tensors = None


def reduce_densification_stats(sums=(), maxima=(), minima=(), group=None):
    """In-place all-reduce of the running statistics the reference's densification reads every `densification_interval`
    iterations (scene/c_gaussian_model.py:1095-1145; train.py:210-211): SUM for the accumulators and their denominators
    (xyz_gradient_accum, denom, xyz_error_accum, xyz_ssim_error_accum, error_denom and the motion twins), MAX for
    max_radii2D, MIN for min_radii2D / *_error_min.  Same-op tensors travel as one flat message.
    No-op for single-process runs."""
    if not (dist.is_initialized() and dist.get_world_size(group) > 1):
        return
    for tensors, op in ((list(sums), dist.ReduceOp.SUM), (list(maxima), dist.ReduceOp.MAX), (list(minima), dist.ReduceOp.MIN)):
        by_dtype = {}
        for t in tensors:
            by_dtype.setdefault((t.dtype, t.device), []).append(t)
        for ts in by_dtype.values():
            flat = torch.cat([t.reshape(-1) for t in ts])
            dist.all_reduce(flat, op=op, group=group)
            off = 0
            for t in ts:
                t.copy_(flat[off:off + t.numel()].view(t.shape))
                off += t.numel()


def allreduce_max_scalar(value, device="cpu"):
    """max over ranks of a Python float (used for the max-over-ranks step time of bench.py)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ----------------------------------------------------------------------------------------------
# Model-parameter gradients: exchange + sharded optimizer
# ----------------------------------------------------------------------------------------------
def _backend(group=None):
    return dist.get_backend(group) if dist.is_initialized() else None


def shard_range(numel, rank, world, align=4):
    """Element range [lo, hi) of a tensor of `numel` elements owned by `rank`: equal slices of `align`-element granules, the last
    rank takes the remainder.  The first `world * slice` elements form the collective's payload; the tail (< world * align
    elements) travels separately."""
    s = (numel // (world * align)) * align
    lo = rank * s
    hi = numel if rank == world - 1 else lo + s
    return lo, hi, s


class ParamGradExchange:
    """Sum over ranks of the gradients of a fixed list of parameter tensors (the reference's 15 parameter groups,
    scene/c_gaussian_model.py:430-447), asynchronous, without packing copies for the large tensors.

    mode "allreduce": every rank ends up with the full summed gradient (replicated optimizer).
    mode "reduce_scatter": rank r ends up with the summed gradient of ITS element range of every tensor (shard_range) -- half the
        bytes per link of the all-reduce; the sharded optimizer (ShardedRAdam) updates that range and all-gathers the parameters.
    Tensors under `small_bytes` (and the un-sharded tails) are packed into one flat message and all-reduced.
    launch(grads) starts the collectives (on the process group's stream; the caller's current stream is waited for); wait() blocks
    the CALLER'S STREAM on them (async_op work handles: no host synchronisation with NCCL/RCCL).  bytes_on_wire() reports the
    per-rank payload of one exchange."""

    def __init__(self, shapes, device, mode="allreduce", group=None, small_bytes=1 << 20, force=False):
        assert mode in ("allreduce", "reduce_scatter")
        self.mode, self.group, self.device = mode, group, device
        self.force = _forced(force) and dist.is_initialized()
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.numels = [int(torch.Size(s).numel()) for s in shapes]
        self.small = [n * 4 < small_bytes for n in self.numels]
        self.ranges = [shard_range(n, self.rank, self.world) for n in self.numels]
        # flat message: small tensors whole; in reduce_scatter mode also the tails of the large ones
        self.flat_slices = []                    # (tensor index, lo, hi, offset in flat)
        off = 0
        for i, (n, sm) in enumerate(zip(self.numels, self.small)):
            if sm:
                self.flat_slices.append((i, 0, n, off)); off += n
            elif mode == "reduce_scatter":
                tail_lo = self.world * self.ranges[i][2]
                if tail_lo < n:
                    self.flat_slices.append((i, tail_lo, n, off)); off += n - tail_lo
        self.flat = torch.zeros(max(off, 1), dtype=torch.float32, device=device)
        self.pending, self._grads = [], None
        # gloo (CPU tests) has no reduce_scatter: all-reduce + slice gives the same sums
        self._native_rs = _backend(group) == "nccl"

    def active(self):
        return self.world > 1 or self.force

    def bytes_on_wire(self):
        """Payload bytes one rank contributes to one exchange (what a ring moves is 2 (W-1)/W of it for all-reduce, (W-1)/W for
        reduce-scatter)."""
        big = sum(n for n, sm in zip(self.numels, self.small) if not sm)
        return 4 * (big + self.flat.numel())

    def launch(self, grads):
        assert len(grads) == len(self.numels)
        self.wait()
        self._grads = [g.view(-1) for g in grads]
        for i, lo, hi, off in self.flat_slices:
            self.flat[off:off + hi - lo].copy_(self._grads[i][lo:hi])
        if not self.active():
            return self
        self.pending = [dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)]
        for i, (g, sm) in enumerate(zip(self._grads, self.small)):
            if sm:
                continue
            if self.mode == "allreduce" or not self._native_rs:
                self.pending.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            else:
                lo, hi, s = self.ranges[i]
                if s > 0:       # in place: the rank's own slice of the payload receives the sum
                    self.pending.append(dist.reduce_scatter_tensor(g[self.rank * s:(self.rank + 1) * s], g[:self.world * s],
                                                                   op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        return self

    def wait(self):
        for w in self.pending:
            w.wait()
        self.pending = []
        if self._grads is not None:
            for i, lo, hi, off in self.flat_slices:
                self._grads[i][lo:hi].copy_(self.flat[off:off + hi - lo])
            self._grads = None


class SliceGather:
    """Exchange of the keyframe-gradient SLICES of ex4d_attributes_backward_sliced: every rank contributes its [Nd, count, C] window
    (16 MB at 0.2 M dynamic Gaussians instead of a 196 MB dense all-reduce) plus its first keyframe index; after wait() every rank holds
    all W windows and their positions and feeds them to ex4d_radam_step_sliced, which adds them per element in rank order -- the same
    sum as the dense all-reduce.  (The union of W random timestamps covers most of the K keyframes, so a dense "union" tensor would
    save little; W small windows do.)"""

    def __init__(self, shape, device, group=None, local_only=False, force=False, native=None):
        """local_only: a single local window, no collective (exchange mode "none": nothing is summed over ranks).
        native: None = all_gather_into_tensor on RCCL, the list all-gather on gloo; False (or EX4D_SLICE_GATHER_LIST=1) = the list
        all-gather on any backend (the validated fallback of ADVICE r03)."""
        self.group = group
        self.local_only = bool(local_only)
        self.force = _forced(force) and dist.is_initialized() and not local_only
        self.world = dist.get_world_size(group) if (dist.is_initialized() and not local_only) else 1
        self.rank = dist.get_rank(group) if (dist.is_initialized() and not local_only) else 0
        self.all = torch.zeros((self.world,) + tuple(shape), dtype=torch.float32, device=device)
        self.first = torch.zeros(self.world, dtype=torch.int32, device=device)
        self.first_local = torch.zeros(1, dtype=torch.int32, device=device)
        self.pending = []
        self._keep = None
        self._native = (_backend(group) == "nccl") if native is None else bool(native)
        if os.environ.get("EX4D_SLICE_GATHER_LIST", "0") == "1":
            self._native = False

    def launch(self, window, first):
        self.wait()
        self._first_host = int(first)
        if self.world == 1 and not self.force:
            self.all[0].copy_(window)
            return self
        if self._native:
            # RCCL: the send buffers are the caller's window and a one-element tensor of their own -- never views of the receive
            # buffers (no reliance on in-place all-gather semantics).  Executed on hardware by tests/test_gpu_dist.py (one rank, forced)
            src = window if window.is_contiguous() else window.contiguous()
            self.first_local.fill_(int(first))
            self._keep = src
            self.pending = [dist.all_gather_into_tensor(self.all, src, group=self.group, async_op=True),
                            dist.all_gather_into_tensor(self.first, self.first_local, group=self.group, async_op=True)]
        else:
            self.all[self.rank].copy_(window)
            self.first[self.rank] = int(first)
            self.pending = [dist.all_gather([self.all[r] for r in range(self.world)], self.all[self.rank].clone(), group=self.group, async_op=True),
                            dist.all_gather([self.first[r:r + 1] for r in range(self.world)], self.first[self.rank:self.rank + 1].clone(), group=self.group, async_op=True)]
        return self

    def wait(self):
        for w in self.pending:
            w.wait()
        self.pending = []
        self._keep = None

    def windows(self, count):
        """[(first keyframe, count, device pointer of the [Nd, count, C] block)] for all ranks (call after wait()).  With more than
        one rank the positions of the other ranks' windows exist on the DEVICE only (first_device_ptr(): the optimizer kernel reads
        them there -- no device -> host round trip in the step); the host values returned for them are placeholders."""
        return [(self._first_host if r == self.rank else None, int(count), self.all[r].data_ptr()) for r in range(self.world)]

    def first_device_ptr(self):
        """Device int32[world] of the windows' first keyframes (None when no collective ran: the host value is exact).  The host
        positions windows() returns for OTHER ranks are None: a caller that does not hand this pointer to the optimizer kernel
        fails loudly (optim.radam_step_sliced_raw) instead of applying remote windows at keyframe 0."""
        return self.first.data_ptr() if (self.world > 1 or self.force) else None

    def bytes_on_wire(self):
        return 4 * self.all[0].numel()


class ShardedRAdam:
    """RAdam over a fixed list of parameter tensors with the update and the optimizer state sharded over ranks
    ("fused RAdam tied to the reduce-scatter of 8e").  Per step and tensor:
        reduce-scatter(grad)  ->  ex4d_radam_step on the rank's element range (exp_avg / exp_avg_sq exist only for it)
        ->  all-gather(param).
    RAdam is element-wise with per-tensor scalars (lr, step), so the parameters after a step are BIT-IDENTICAL to the replicated
    dense update of the summed gradient (tested: world-2 gloo on CPU with an injected step function, 2 processes on a GPU).
    1/W of the optimizer's 28 B/element HBM traffic per rank; the same bytes per link as the all-reduce it replaces.

    params: list of contiguous float32 tensors (updated in place); lrs: per-tensor learning rates.
    step_fn(items, betas, eps, device): defaults to the fused HIP launch (optim.radam_step_raw); the CPU tests inject the oracle."""

    def __init__(self, params, lrs, betas=(0.9, 0.999), eps=1e-8, group=None, step_fn=None, small_bytes=1 << 20, nan_to_num=None, force=False):
        """nan_to_num: per-tensor flags -- the gradient of a flagged tensor is read through torch.nan_to_num like the replicated
        paths do for _opacity_duration_var (train.py:244-247; ADVICE r03: without it one non-finite gradient poisons the sharded
        parameter and its moments for good, and "bit-identical to the replicated update" stops holding)."""
        self.params = list(params)
        self.lrs = [float(x) for x in lrs]
        self.nan_to_num = [0] * len(self.params) if nan_to_num is None else [int(bool(x)) for x in nan_to_num]
        assert len(self.nan_to_num) == len(self.params)
        self.betas, self.eps, self.group = betas, eps, group
        self.device = self.params[0].device
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.force = _forced(force) and dist.is_initialized()
        self.exchange = ParamGradExchange([p.shape for p in self.params], self.device, mode="reduce_scatter", group=group, small_bytes=small_bytes,
                                          force=force)
        if step_fn is None:
            from .optim import radam_step_raw
            step_fn = radam_step_raw
        self.step_fn = step_fn
        self.steps = [0] * len(self.params)
        # owned element ranges: large tensors -> shard_range (+ the tail for the last rank, which shard_range already includes);
        # small tensors are updated redundantly by every rank (their summed gradient is all-reduced): no all-gather needed
        self.owned = []
        for p, sm in zip(self.params, self.exchange.small):
            n = p.numel()
            self.owned.append((0, n) if (sm or self.world == 1) else shard_range(n, self.rank, self.world)[:2])
        self.exp_avg = [torch.zeros(hi - lo, dtype=torch.float32, device=self.device) for lo, hi in self.owned]
        self.exp_avg_sq = [torch.zeros(hi - lo, dtype=torch.float32, device=self.device) for lo, hi in self.owned]
        self._native = _backend(group) == "nccl"

    def state_bytes(self):
        return 8 * sum(hi - lo for lo, hi in self.owned)

    def launch_exchange(self, grads):
        """Start the gradient reduce-scatter (asynchronous); step() waits for it."""
        self._grads = [g for g in grads]
        self.exchange.launch(self._grads)

    def step(self, grads=None):
        if grads is not None:
            self.launch_exchange(grads)
        self.exchange.wait()
        items = []
        for i, (p, g) in enumerate(zip(self.params, self._grads)):
            self.steps[i] += 1
            lo, hi = self.owned[i]
            if hi > lo:
                pv, gv = p.view(-1)[lo:hi], g.view(-1)[lo:hi]
                items.append((pv.data_ptr() if pv.is_cuda else pv, gv.data_ptr() if gv.is_cuda else gv,
                              self.exp_avg[i].data_ptr() if pv.is_cuda else self.exp_avg[i],
                              self.exp_avg_sq[i].data_ptr() if pv.is_cuda else self.exp_avg_sq[i], hi - lo, self.lrs[i], self.steps[i],
                              self.nan_to_num[i]))
        self.step_fn(items, self.betas, self.eps, self.device)
        if self.device.type == "cuda":
            torch.autograd.graph.increment_version(self.params)        # written through raw pointers
        if self.world > 1 or self.force:
            works = []
            for p, sm in zip(self.params, self.exchange.small):
                if sm:
                    continue
                flat = p.view(-1)
                lo, hi, s = shard_range(flat.numel(), self.rank, self.world)
                if s > 0:
                    if self._native:        # in place: every rank's slice lands at its offset
                        works.append(dist.all_gather_into_tensor(flat[:self.world * s], flat[self.rank * s:(self.rank + 1) * s], group=self.group, async_op=True))
                    else:
                        outs = [flat[r * s:(r + 1) * s] for r in range(self.world)]
                        works.append(dist.all_gather(outs, flat[self.rank * s:(self.rank + 1) * s].clone(), group=self.group, async_op=True))
                if self.world * s < flat.numel():   # tail: owned (and updated) by the last rank
                    works.append(dist.broadcast(flat[self.world * s:], src=self.world - 1, group=self.group, async_op=True))
            for w in works:
                w.wait()
