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


class ViewAccumulator:
    """k views per rank and optimizer step (round 6): the gradients of a rank's views 1 .. k-1 are added up locally in
    persistent accumulators, the k-th view's add is followed by ONE exchange of the sums -- the bytes a rank puts on the wire per VIEW
    drop k-fold, and with them the exposed exchange per view (the exchange cannot hide behind the next frame while the optimizer steps
    every iteration: DESIGN.md section 6).  The batch of an optimizer step becomes N k views; the sum over ranks and views is the sum the
    k single-view exchanges would have delivered (addition order: views first, then ranks).

    add(grads) -> True when this was the group's last view: the exchange is on its way (exchange.wait() blocks on it) and `sums` holds /
    will hold the result; False: accumulated, nothing launched.  exchange: ParamGradExchange | None (one rank: plain accumulation)."""

    def __init__(self, shapes, device, views_per_step, exchange=None):
        assert views_per_step >= 1
        self.k, self.exchange = int(views_per_step), exchange
        self.sums = [torch.zeros(s, dtype=torch.float32, device=device) for s in shapes]
        self.count = 0

    def add(self, grads):
        assert len(grads) == len(self.sums)
        if self.count == 0:
            if self.exchange is not None:
                self.exchange.wait()                  # the previous group's collectives own the accumulators until here
            for a, g in zip(self.sums, grads):
                a.copy_(g.view_as(a))
        else:
            torch._foreach_add_(self.sums, [g.view_as(a) for a, g in zip(self.sums, grads)])
        self.count += 1
        if self.count < self.k:
            return False
        self.count = 0
        if self.exchange is not None:
            self.exchange.launch(self.sums)
        return True

    def bytes_on_wire_per_view(self):
        return 0 if self.exchange is None else self.exchange.bytes_on_wire() / self.k


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


class SliceRowExchange:
    """Keyframe-gradient windows for the SHARDED optimizer (round 5): the keyframe tensors [rows, K, C] are sharded by ROWS (rank r owns
    shard_range(rows * K * C, r, W, align = K * C), i.e. whole rows), and every rank sends every owner just the rows of its
    [rows, count, C] window that the owner owns -- one all-to-all.  After wait() rank r holds W windows [rows_r, count, C] of ITS rows
    and their W first-keyframe indices, which ex4d_radam_step_sliced adds per element in rank order: the same sum as the dense
    reduce-scatter, but (W - 1) / W x 16 MB per rank on the wire at 0.2 M dynamic Gaussians instead of 196 MB (dense reduce-scatter) or
    (W - 1) x 16 MB received (SliceGather's all-gather for the replicated optimizer)."""

    def __init__(self, rows, K, C, count, device, group=None, force=False):
        self.group = group
        self.rows, self.K, self.C, self.count = int(rows), int(K), int(C), int(count)
        self.force = _forced(force) and dist.is_initialized()
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        per = self.K * self.C
        self.row_ranges = []
        for r in range(self.world):
            lo, hi, _ = shard_range(self.rows * per, r, self.world, align=per)
            self.row_ranges.append((lo // per, hi // per))
        self.my_rows = self.row_ranges[self.rank][1] - self.row_ranges[self.rank][0]
        self.recv = torch.zeros((self.world * self.my_rows, self.count, self.C), dtype=torch.float32, device=device)
        self.first = torch.zeros(self.world, dtype=torch.int32, device=device)
        self.first_local = torch.zeros(1, dtype=torch.int32, device=device)
        self.pending, self._keep = [], None

    def active(self):
        return self.world > 1 or self.force

    def launch(self, window, first):
        self.wait()
        assert tuple(window.shape) == (self.rows, self.count, self.C), (tuple(window.shape), (self.rows, self.count, self.C))
        self._first_host = int(first)
        if not self.active():
            self.recv.copy_(window)
            self.first[0] = int(first)
            return self
        src = window if window.is_contiguous() else window.contiguous()
        self.first_local.fill_(int(first))
        self._keep = src
        send = [hi - lo for lo, hi in self.row_ranges]
        self.pending = [dist.all_to_all_single(self.recv, src, output_split_sizes=[self.my_rows] * self.world, input_split_sizes=send,
                                               group=self.group, async_op=True)]
        if _backend(self.group) == "nccl":
            self.pending.append(dist.all_gather_into_tensor(self.first, self.first_local, group=self.group, async_op=True))
        else:
            self.pending.append(dist.all_gather([self.first[r:r + 1] for r in range(self.world)], self.first_local.clone(), group=self.group, async_op=True))
        return self

    def wait(self):
        for w in self.pending:
            w.wait()
        self.pending, self._keep = [], None

    def windows(self):
        """[(first keyframe or None, count, [my_rows, count, C] tensor)] in rank order (call after wait()); the first keyframes of
        other ranks' windows exist in `first` (device) only."""
        return [(self._first_host if (r == self.rank or not self.active()) else None, self.count, self.recv[r * self.my_rows:(r + 1) * self.my_rows])
                for r in range(self.world)]

    def bytes_on_wire(self):
        """Bytes one rank SENDS per exchange (the rows of its window it does not own)."""
        return 4 * (self.rows - self.my_rows) * self.count * self.C


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

    def __init__(self, params, lrs, betas=(0.9, 0.999), eps=1e-8, group=None, step_fn=None, small_bytes=1 << 20, nan_to_num=None, force=False,
                 sliced=None, sliced_step_fn=None):
        """nan_to_num: per-tensor flags -- the gradient of a flagged tensor is read through torch.nan_to_num like the replicated
        paths do for _opacity_duration_var (train.py:244-247; ADVICE r03: without it one non-finite gradient poisons the sharded
        parameter and its moments for good, and "bit-identical to the replicated update" stops holding).
        sliced: {tensor index: count} for keyframe tensors [rows, K, C] whose gradient arrives as a [rows, count, C] WINDOW plus its
        first keyframe (launch_exchange(..., windows=...)): sharded by whole rows, exchanged by SliceRowExchange, updated by
        ex4d_radam_step_sliced on the owned rows (sliced_step_fn: defaults to optim.radam_step_sliced_raw; the CPU tests inject an
        oracle).  At most optim.MAX_WINDOWS ranks (one window per rank reaches the kernel)."""
        self.params = list(params)
        self.sliced = {int(i): int(c) for i, c in (sliced or {}).items()}
        self.lrs = [float(x) for x in lrs]
        self.nan_to_num = [0] * len(self.params) if nan_to_num is None else [int(bool(x)) for x in nan_to_num]
        assert len(self.nan_to_num) == len(self.params)
        self.betas, self.eps, self.group = betas, eps, group
        self.device = self.params[0].device
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.force = _forced(force) and dist.is_initialized()
        # the dense tensors travel by reduce-scatter; the sliced ones by SliceRowExchange (their windows never become dense tensors)
        self.dense_idx = [i for i in range(len(self.params)) if i not in self.sliced]
        self.exchange = ParamGradExchange([self.params[i].shape for i in self.dense_idx], self.device, mode="reduce_scatter", group=group,
                                          small_bytes=small_bytes, force=force)
        if step_fn is None:
            from .optim import radam_step_raw
            step_fn = radam_step_raw
        self.step_fn = step_fn
        if self.sliced and sliced_step_fn is None:
            from .optim import radam_step_sliced_raw, MAX_WINDOWS
            if self.world > MAX_WINDOWS:
                raise ValueError(f"sliced keyframe gradients take one window per rank, at most {MAX_WINDOWS} (use dense keyframe gradients beyond)")
            sliced_step_fn = radam_step_sliced_raw
        self.sliced_step_fn = sliced_step_fn
        self.steps = [0] * len(self.params)
        # per tensor: is it small (updated redundantly by every rank from the all-reduced flat message) and its shard granule
        self.small = [False] * len(self.params)
        self.align = [4] * len(self.params)
        for j, i in enumerate(self.dense_idx):
            self.small[i] = self.exchange.small[j]
        self.row_exchange = {}
        for i, count in self.sliced.items():
            rows, K, Cc = self.params[i].shape
            self.align[i] = K * Cc
            self.row_exchange[i] = SliceRowExchange(rows, K, Cc, count, self.device, group=group, force=force)
        # owned element ranges: large tensors -> shard_range (+ the tail for the last rank, which shard_range already includes);
        # small tensors are updated redundantly by every rank (their summed gradient is all-reduced): no all-gather needed
        self.owned = []
        for i, p in enumerate(self.params):
            n = p.numel()
            self.owned.append((0, n) if (self.small[i] or self.world == 1) else shard_range(n, self.rank, self.world, align=self.align[i])[:2])
        self.exp_avg = [torch.zeros(hi - lo, dtype=torch.float32, device=self.device) for lo, hi in self.owned]
        self.exp_avg_sq = [torch.zeros(hi - lo, dtype=torch.float32, device=self.device) for lo, hi in self.owned]
        self._native = _backend(group) == "nccl"

    def state_bytes(self):
        return 8 * sum(hi - lo for lo, hi in self.owned)

    def launch_exchange(self, grads, windows=None):
        """Start the gradient reduce-scatter (asynchronous); step() waits for it.  grads: one entry per parameter (the entries of
        sliced tensors are ignored); windows: {tensor index: (window [rows, count, C], first keyframe)} for the sliced tensors."""
        self._grads = [g for g in grads]
        self.exchange.launch([self._grads[i] for i in self.dense_idx])
        for i, ex in self.row_exchange.items():
            w, first = windows[i]
            ex.launch(w, first)

    def step(self, grads=None, windows=None):
        if grads is not None:
            self.launch_exchange(grads, windows)
        self.exchange.wait()
        for ex in self.row_exchange.values():
            ex.wait()
        items, sliced_items = [], []
        for i, (p, g) in enumerate(zip(self.params, self._grads)):
            self.steps[i] += 1
            lo, hi = self.owned[i]
            if i in self.sliced:
                rows_total, K, Cc = p.shape
                rows = (hi - lo) // (K * Cc)
                if rows > 0:
                    ex = self.row_exchange[i]
                    pv = p.view(-1)[lo:hi]
                    cuda = pv.is_cuda
                    wins = [(f, c, (t.data_ptr() if cuda else t)) for f, c, t in ex.windows()]
                    if not cuda:                # CPU tests: the positions are host-readable
                        firsts = ex.first.tolist()
                        wins = [(int(firsts[r]) if f is None else f, c, t) for r, (f, c, t) in enumerate(wins)]
                    sliced_items.append((pv.data_ptr() if cuda else pv, self.exp_avg[i].data_ptr() if cuda else self.exp_avg[i],
                                         self.exp_avg_sq[i].data_ptr() if cuda else self.exp_avg_sq[i], rows, K, Cc, self.lrs[i], self.steps[i], wins,
                                         (ex.first.data_ptr() if (cuda and ex.active()) else None)))
                continue
            if hi > lo:
                pv, gv = p.view(-1)[lo:hi], g.view(-1)[lo:hi]
                items.append((pv.data_ptr() if pv.is_cuda else pv, gv.data_ptr() if gv.is_cuda else gv,
                              self.exp_avg[i].data_ptr() if pv.is_cuda else self.exp_avg[i],
                              self.exp_avg_sq[i].data_ptr() if pv.is_cuda else self.exp_avg_sq[i], hi - lo, self.lrs[i], self.steps[i],
                              self.nan_to_num[i]))
        self.step_fn(items, self.betas, self.eps, self.device)
        if sliced_items:
            self.sliced_step_fn(sliced_items, self.betas, self.eps, self.device)
        if self.device.type == "cuda":
            torch.autograd.graph.increment_version(self.params)        # written through raw pointers
        if self.world > 1 or self.force:
            works = []
            for i, p in enumerate(self.params):
                if self.small[i]:
                    continue
                flat = p.view(-1)
                lo, hi, s = shard_range(flat.numel(), self.rank, self.world, align=self.align[i])
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
