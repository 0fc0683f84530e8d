"""Multi-GPU layer of the hot path: frames/views shard embarrassingly across ranks (one process per GPU,
Gaussian parameters replicated), and the only collective is the sum all-reduce of the per-frame gradients
for the training step (the reference itself is single-process: utils/general_utils.py:161).

Backend "nccl" is RCCL on ROCm (xGMI inside a node); "gloo" is used by the CPU tests (world_size 2).
Gradients are packed into a few large flat buckets (one all-reduce per bucket keeps every xGMI link busy
with large messages instead of 5-15 small tensors) and reduced asynchronously so the exchange of frame
i overlaps the rasterization of frame i+1.
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
