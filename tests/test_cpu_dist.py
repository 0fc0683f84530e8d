"""CPU tests of the multi-rank layer (gloo, world size 2): the model-parameter gradient exchange and the sharded optimizer of
ex4dgs_amd/dist.py.  The HIP RAdam launch is replaced by the numpy oracle through ShardedRAdam's step_fn hook (test
infrastructure standing in for the kernel; the sharding / collectives / ownership logic under test is the product's)."""
import os
import subprocess
import sys

import pytest

from tests import helpers as h

_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
from ex4dgs_amd import dist as xd
from oracle import optim_oracle

rank, world, local = xd.init_from_env(backend="gloo")
assert world == 2
# a miniature of the reference's 15 parameter groups: sizes that are not multiples of world * 4, one tensor below the packing threshold
shapes = [(1003, 3), (1003, 16, 3), (257, 35, 4), (10, 1), (4099,), (7,)]
lrs = [1.6e-4, 2.5e-3, 1e-3, 5e-2, 5e-3, 1e-3]
g0 = torch.Generator().manual_seed(7)
init = [torch.randn(*s, generator=g0) for s in shapes]

NAN_TENSOR = 4            # stands for _opacity_duration_var: its gradient goes through nan_to_num (train.py:244-247)

def grads_of(r, step):
    g = torch.Generator().manual_seed(1000 * step + r)
    out = [torch.randn(*s, generator=g) * (0.1 + step) for s in shapes]
    out[3] = torch.zeros(*shapes[3])                 # an all-zero gradient (RAdam still moves the parameter by its momentum)
    if step == 2 and r == 0:                         # one rank's frame produces non-finite gradients in both halves of the tensor
        out[NAN_TENSOR][5] = float("nan"); out[NAN_TENSOR][3000] = float("inf"); out[NAN_TENSOR][4098] = float("-inf")
    return out

def oracle_step(items, betas, eps, device):
    for p, g, m, v, n, lr, step, sanitize in items:  # tensors (CPU path of ShardedRAdam)
        pn, mn, vn = p.numpy(), m.numpy(), v.numpy()
        gn = g.numpy().copy()
        if sanitize:
            gn = np.nan_to_num(gn)                   # numpy's defaults are torch.nan_to_num's: NaN -> 0, +-inf -> +-FLT_MAX
        optim_oracle.radam_step(pn, gn, mn, vn, step, lr, betas[0], betas[1], eps)

# --- sharded: every rank holds all parameters, updates only its element ranges
params = [x.clone() for x in init]
opt = xd.ShardedRAdam(params, lrs, step_fn=oracle_step, small_bytes=256, nan_to_num=[i == NAN_TENSOR for i in range(len(shapes))])
assert opt.exchange.small == [False, False, False, True, False, True]
K = 8
for step in range(1, K + 1):
    g = grads_of(rank, step)
    opt.launch_exchange(g)        # asynchronous reduce-scatter ...
    opt.step()                    # ... waited for here; sharded update; all-gather
# --- replicated dense reference: the same oracle on the summed gradients, whole tensors
ref = [x.clone() for x in init]
m = [torch.zeros_like(x) for x in init]; v = [torch.zeros_like(x) for x in init]
for step in range(1, K + 1):
    ga, gb = grads_of(0, step), grads_of(1, step)
    for i in range(len(ref)):
        gsum = (ga[i] + gb[i]).numpy()
        if i == NAN_TENSOR:
            gsum = np.nan_to_num(gsum)               # the replicated path: nan_to_num on the summed gradient, then the dense step
        optim_oracle.radam_step(ref[i].numpy(), gsum, m[i].numpy(), v[i].numpy(), step, lrs[i])
for i, (a, b) in enumerate(zip(params, ref)):
    assert torch.equal(a, b), (i, float((a - b).abs().max()))          # bit-identical, including the un-sharded tails
    assert bool(torch.isfinite(a).all()), i                             # the non-finite gradient of step 2 poisoned nothing
# optimizer state is really sharded: about half of the replicated state per rank
full = 8 * sum(x.numel() for x in init)
assert opt.state_bytes() < 0.56 * full + 8 * (10 + 7) , (opt.state_bytes(), full)
# bytes one rank puts on the wire per exchange = all large tensors + the packed small ones and tails
assert opt.exchange.bytes_on_wire() == 4 * (sum(x.numel() for i, x in enumerate(init) if i not in (3, 5)) + opt.exchange.flat.numel())

# --- plain all-reduce mode: every rank gets the full sum
ex = xd.ParamGradExchange(shapes, "cpu", mode="allreduce", small_bytes=256)
g = grads_of(rank, 3)
ex.launch(g); ex.wait()
ga, gb = grads_of(0, 3), grads_of(1, 3)
for i in range(len(g)):
    assert torch.equal(g[i], ga[i] + gb[i]), i
torch.distributed.barrier(); torch.distributed.destroy_process_group()
print("OK", rank)
"""


def test_sharded_radam_and_gradient_exchange_gloo_world2(tmp_path):
    script = tmp_path / "worker_dist.py"
    script.write_text(_WORKER)
    port = 31500 + (os.getpid() % 2000)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), h.ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"OK {r}" in o, o


_WORKER4 = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
from ex4dgs_amd import dist as xd
from oracle import optim_oracle

rank, world, local = xd.init_from_env(backend="gloo")
assert world == 4
# two keyframe tensors [rows, K, C] (rows not a multiple of the world size) whose gradients are WINDOWS of 4 / 2 time slices, and dense tensors
K = 35
shapes = [(1003, 3), (259, K, 3), (1003, 16, 3), (259, K, 4), (7,), (4099,)]
SLICED = {1: 4, 3: 2}
lrs = [1.6e-4, 1.6e-4, 2.5e-3, 1e-3, 1e-3, 5e-3]
g0 = torch.Generator().manual_seed(11)
init = [torch.randn(*s, generator=g0) for s in shapes]

def frame(r, step):
    # (dense gradients, {i: (window, first keyframe)}) of rank r's frame at this step; windows of different ranks overlap or not
    g = torch.Generator().manual_seed(1000 * step + r)
    dense = [torch.randn(*s, generator=g) * (0.1 + step) for s in shapes]
    wins = {}
    for i, count in SLICED.items():
        first = (5 * step + 3 * r + i) % (K - count + 1) if step != 3 else 7          # step 3: all four windows coincide
        wins[i] = (torch.randn(shapes[i][0], count, shapes[i][2], generator=g), first)
    return dense, wins

def dense_of(i, w, first):
    d = torch.zeros(*shapes[i])
    d[:, first:first + w.shape[1], :] = w
    return d

def oracle_step(items, betas, eps, device):
    for p, g, m, v, n, lr, step, sanitize in items:
        optim_oracle.radam_step(p.numpy(), g.numpy().copy(), m.numpy(), v.numpy(), step, lr, betas[0], betas[1], eps)

def oracle_sliced(items, betas, eps, device):
    # what ex4d_radam_step_sliced does: every element of the owned rows is updated, the gradient = the windows added in rank order
    for p, m, v, rows, Kk, C, lr, step, wins, first_dev in items:
        g = np.zeros((rows, Kk, C), np.float32)
        assert len(wins) == world
        for first, count, t in wins:
            assert tuple(t.shape) == (rows, count, C)
            g[:, first:first + count, :] += t.numpy()
        optim_oracle.radam_step(p.numpy(), g.reshape(-1), m.numpy(), v.numpy(), step, lr, betas[0], betas[1], eps)

params = [x.clone() for x in init]
opt = xd.ShardedRAdam(params, lrs, step_fn=oracle_step, small_bytes=256, sliced=SLICED, sliced_step_fn=oracle_sliced)
# the keyframe tensors are sharded by whole rows
for i in SLICED:
    lo, hi = opt.owned[i]
    per = shapes[i][1] * shapes[i][2]
    assert lo % per == 0 and hi % per == 0 and (hi - lo) // per in (259 // 4, 259 - 3 * (259 // 4))
STEPS = 5
for step in range(1, STEPS + 1):
    dense, wins = frame(rank, step)
    opt.launch_exchange(dense, windows=wins)
    opt.step()
# replicated dense reference: the dense gradients the windows stand for, summed in rank order, whole tensors
ref = [x.clone() for x in init]
m = [torch.zeros_like(x) for x in init]; v = [torch.zeros_like(x) for x in init]
for step in range(1, STEPS + 1):
    frames = [frame(r, step) for r in range(world)]
    for i in range(len(ref)):
        if i in SLICED:
            # the windows are added per element in RANK ORDER (what ex4d_radam_step_sliced does): the reference is that sum, dense
            gs = [dense_of(i, *frames[r][1][i]) for r in range(world)]
            gsum = gs[0]
            for r in range(1, world):
                gsum = gsum + gs[r]
        else:
            # dense tensors: the sum IS the collective's (float addition is not associative: with more than two ranks the reference
            # is the same collective on whole tensors -- what the replicated optimizer would be handed)
            gsum = frames[rank][0][i].clone()
            torch.distributed.all_reduce(gsum)
        optim_oracle.radam_step(ref[i].numpy().reshape(-1), gsum.numpy().reshape(-1), m[i].numpy().reshape(-1), v[i].numpy().reshape(-1), step, lrs[i])
for i, (a, b) in enumerate(zip(params, ref)):
    assert torch.equal(a, b), (i, float((a - b).abs().max()))
# wire volume of the windows: every rank sends only the rows it does not own
for i, count in SLICED.items():
    ex = opt.row_exchange[i]
    assert ex.bytes_on_wire() == 4 * (259 - ex.my_rows) * count * shapes[i][2]
# optimizer state of the keyframe tensors is sharded too
assert opt.exp_avg[1].numel() == (opt.owned[1][1] - opt.owned[1][0]) < init[1].numel() // 3
torch.distributed.barrier(); torch.distributed.destroy_process_group()
print("OK", rank)
"""


def test_sharded_radam_with_row_sharded_keyframe_windows_gloo_world4(tmp_path):
    """the sharded optimizer with SLICED keyframe gradients (row-sharded windows, one all-to-all) equals the dense
    replicated update bit for bit, in a world of four ranks (gloo, the oracle injected for the two HIP step functions)."""
    script = tmp_path / "worker_dist4.py"
    script.write_text(_WORKER4)
    port = 33500 + (os.getpid() % 2000)
    procs = []
    for r in range(4):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="4", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), h.ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"OK {r}" in o, o


def test_shard_ranges_partition_every_tensor():
    from ex4dgs_amd.dist import shard_range
    for n in (0, 1, 7, 8, 31, 32, 33, 1000003, 9000000):
        for world in (1, 2, 3, 8):
            covered = 0
            for r in range(world):
                lo, hi, s = shard_range(n, r, world)
                assert lo == covered and hi >= lo and s % 4 == 0
                covered = hi
            assert covered == n


_WORKER_VIEWS = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import torch
from ex4dgs_amd import dist as xd

rank, world, local = xd.init_from_env(backend="gloo")
assert world == 2
shapes = [(1003, 3), (259, 35, 3), (1003, 16, 3), (7,), (4099,)]
K = 3

def grads_of(r, view):
    # exactly representable values (multiples of 1/8 below 2^10): every addition order gives the same float
    g = torch.Generator().manual_seed(100 * view + r)
    return [torch.randint(-4096, 4096, s, generator=g).float() / 8 for s in shapes]

# k views accumulated locally + ONE exchange ...
ex = xd.ParamGradExchange(shapes, "cpu", mode="allreduce", small_bytes=256)
acc = xd.ViewAccumulator(shapes, "cpu", K, exchange=ex)
for v in range(K):
    last = acc.add(grads_of(rank, v))
    assert last == (v == K - 1)
ex.wait()
# ... equals the sum of the k single-view exchanges
ex1 = xd.ParamGradExchange(shapes, "cpu", mode="allreduce", small_bytes=256)
ref = [torch.zeros(s) for s in shapes]
for v in range(K):
    g = grads_of(rank, v)
    ex1.launch(g); ex1.wait()
    for a, b in zip(ref, g):
        a += b
for i, (a, b) in enumerate(zip(acc.sums, ref)):
    assert torch.equal(a, b), (i, float((a - b).abs().max()))
    want = sum((grads_of(r, v)[i] for r in range(2) for v in range(K)), torch.zeros(shapes[i]))
    assert torch.equal(a, want), i
assert acc.bytes_on_wire_per_view() * K == ex.bytes_on_wire()
# a second group reuses the accumulators (the first view of a group overwrites them)
for v in range(K):
    acc.add(grads_of(rank, 10 + v))
ex.wait()
want = sum((grads_of(r, 10 + v)[2] for r in range(2) for v in range(K)), torch.zeros(shapes[2]))
assert torch.equal(acc.sums[2], want)
torch.distributed.barrier(); torch.distributed.destroy_process_group()
print("OK", rank)
"""


def test_views_per_step_accumulation_equals_single_view_exchanges_gloo_world2(tmp_path):
    """k views accumulated per rank + one exchange == the sum of k single-view exchanges (dist.ViewAccumulator, what
    FrameTrainer(views_per_step=k) does with its persistent accumulators)."""
    script = tmp_path / "worker_views.py"
    script.write_text(_WORKER_VIEWS)
    port = 33500 + (os.getpid() % 2000)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), h.ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"OK {r}" in o, o
