"""GPU tests of the multi-rank training path (run with `-m gpu` on an MI355X; a 1-GPU box is enough: two ranks share the device
and talk over gloo, and the RCCL-native collectives run in a one-rank "nccl" group with the exchange forced)."""
import json
import os
import subprocess
import sys

import pytest
import torch

from tests import helpers as h

pytestmark = pytest.mark.gpu


def _fixed_upstream(H, W, seed=5):
    w = torch.rand(3, H, W, generator=torch.Generator().manual_seed(seed)).cuda()
    return lambda out: ([out["render"]], [w])


def test_frame_trainer_matches_the_autograd_path(hip_lib):
    """FrameTrainer (manual attribute forward / rasterizer autograd / attribute backward into persistent buffers, side stream)
    produces the gradients of render() + autograd on the fused model."""
    from ex4dgs_amd.scene import make_scene
    from ex4dgs_amd.render import render
    from ex4dgs_amd.trainer import FrameTrainer
    model, cam, bg = make_scene("cfg3", P=20000, device="cuda", fused=True)
    H, W = cam.image_height, cam.image_width
    up = _fixed_upstream(H, W)
    tr = FrameTrainer(model, exchange="none")
    for t in (0, 137, 299):
        out = tr.step(cam, bg, t, up)
        tr.flush()
        got = {k: v.clone() for k, v in tr.grads().items()}
        for p in model.parameters():
            p.requires_grad_(True); p.grad = None
        ref_out = render(cam, model, None, bg, timestamp=t, near=4.0, far=300.0)
        assert torch.equal(ref_out["render"], out["render"])
        (ref_out["render"] * up(ref_out)[1][0]).sum().backward()
        for name in model.PARAM_NAMES:
            g, r = got[name], getattr(model, name).grad
            tol = 2e-5 * float(r.abs().max()) + 1e-12          # float atomics: equal to rounding
            assert float((g - r).abs().max()) <= tol, (name, t, float((g - r).abs().max()), tol)
        for p in model.parameters():
            p.requires_grad_(False); p.grad = None


def test_views_per_step_accumulates_and_steps_once(hip_lib):
    """FrameTrainer(views_per_step=k): the gradients of k views are added up in persistent accumulators and handed over (to the exchange /
    the optimizer) once, after the k-th view -- equal to the sum of the k single-view gradients; with an optimizer the parameters move once
    per k views."""
    from ex4dgs_amd.scene import make_scene
    from ex4dgs_amd.trainer import FrameTrainer
    model, cam, bg = make_scene("cfg3", P=20000, device="cuda", fused=True)
    H, W = cam.image_height, cam.image_width
    up = _fixed_upstream(H, W)
    stamps = (0, 137, 299)
    solo = FrameTrainer(model, exchange="none")
    acc = None
    for t in stamps:
        solo.step(cam, bg, t, up); solo.flush()
        g = {k: v.clone() for k, v in solo.grads().items()}
        acc = g if acc is None else {k: acc[k] + g[k] for k in g}
    tr = FrameTrainer(model, exchange="none", views_per_step=3)
    for i, t in enumerate(stamps):
        tr.step(cam, bg, t, up); tr.flush()
        assert (tr.grads() is None) == (i < 2), "gradients are handed over after the k-th view only"
    for name, a in tr.grads().items():
        b = acc[name]
        tol = 2e-5 * float(b.abs().max()) + 1e-12          # float atomics in the rasterizer backward: equal to rounding
        assert float((a - b).abs().max()) <= tol, (name, float((a - b).abs().max()), tol)
    # with the optimizer: one RAdam step per k views
    model2, _, _ = make_scene("cfg3", P=20000, device="cuda", fused=True)
    tro = FrameTrainer(model2, exchange="none", optimizer=True, views_per_step=2)
    before = model2._xyz.clone()
    tro.step(cam, bg, 0, up); tro.flush()
    assert torch.equal(model2._xyz, before), "no optimizer step after the first of two views"
    tro.step(cam, bg, 137, up); tro.flush()
    assert not torch.equal(model2._xyz, before) and tro.steps == 1
    with pytest.raises(ValueError):
        FrameTrainer(model2, exchange="none", optimizer=True, views_per_step=2, sliced=True)


_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import torch
from ex4dgs_amd import _C, dist as xd
from ex4dgs_amd.scene import make_scene
from ex4dgs_amd.trainer import FrameTrainer
from ex4dgs_amd.optim import radam_step_raw
from ex4dgs_amd.attributes import PARAM_ORDER

os.environ["LOCAL_RANK"] = "0"                       # both ranks on the box's one GPU
rank, world, local = xd.init_from_env(backend="gloo")
torch.cuda.set_device(0); _C.load()
model, cam, bg = make_scene("cfg3", P=20000, device="cuda", fused=True)
H, W = cam.image_height, cam.image_width
w = torch.rand(3, H, W, generator=torch.Generator().manual_seed(5)).cuda()
up = lambda out: ([out["render"]], [w])
stamps = [0, 100, 200, 299, 41, 88]

# (1) all-reduce of the 15 model-parameter gradients: every rank ends up with the sum over the ranks' frames
tr = FrameTrainer(model, exchange="allreduce")
tr.step(cam, bg, stamps[rank], up); tr.flush()
summed = [g.clone() for g in tr.grads().values()]
solo = FrameTrainer(model, exchange="none")
acc = None
for r in range(world):
    solo.step(cam, bg, stamps[r], up); solo.flush()
    g = [x.clone() for x in solo.grads().values()]
    acc = g if acc is None else [a + b for a, b in zip(acc, g)]
for name, a, b in zip(PARAM_ORDER, summed, acc):
    tol = 2e-5 * float(b.abs().max()) + 1e-12
    assert float((a - b).abs().max()) <= tol, (name, float((a - b).abs().max()), tol)
assert tr.exchange_bytes_on_wire() == 4 * sum(p.numel() for p in model.parameters())   # dense: every parameter gradient travels

# (2) sharded optimizer == replicated optimizer, bit for bit, over K steps driven by real rasterizer gradients
lrs = [1e-3 * (1 + i) for i in range(15)]
A = [p.clone() for p in model.parameters()]          # replicated dense update of the summed gradient
mA = [torch.zeros_like(p) for p in A]; vA = [torch.zeros_like(p) for p in A]
B = [p.clone() for p in model.parameters()]          # sharded: reduce-scatter -> update of the own range -> all-gather
sh = xd.ShardedRAdam(B, lrs, small_bytes=1 << 14)      # at 20 k Gaussians most tensors are under the default 1 MB packing threshold
for k in range(1, 5):
    solo.step(cam, bg, stamps[(2 * k + rank) % len(stamps)], up); solo.flush()
    local_g = [g.clone() for g in solo.grads().values()]
    dense_g = [g.clone() for g in local_g]
    for g in dense_g:
        torch.distributed.all_reduce(g)
    radam_step_raw([(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, k) for p, g, m, v, lr in zip(A, dense_g, mA, vA, lrs)],
                   (0.9, 0.999), 1e-8, torch.device("cuda", 0))
    sh.step(local_g)
    torch.cuda.synchronize()
    for name, a, b in zip(PARAM_ORDER, A, B):
        assert torch.equal(a, b), (name, k, float((a - b).abs().max()))
assert sh.state_bytes() < 0.6 * 8 * sum(p.numel() for p in A)

# (3) the trainer's own sharded mode runs end to end and keeps the ranks' parameters identical
m2, _, _ = make_scene("cfg3", P=20000, device="cuda", fused=True)
tr2 = FrameTrainer(m2, exchange="sharded", lrs={n: 1e-6 for n in PARAM_ORDER})    # unrectified first RAdam steps move by lr * gradient
# round 5: the sharded optimizer takes SLICED keyframe gradients too -- the two keyframe tensors are sharded by rows, the ranks' windows
# travel by one all-to-all each (dist.SliceRowExchange) and never become dense tensors
assert tr2.sliced and sorted(tr2.opt.row_exchange) == sorted(tr2.kf_idx) and len(tr2.kf_idx) == 2 and not tr2.kf_gather
p2_before = [p.clone() for p in m2.parameters()]
for k in range(3):
    tr2.step(cam, bg, stamps[(2 * k + rank) % len(stamps)], up)
tr2.flush(); torch.cuda.synchronize()
for name, p in zip(PARAM_ORDER, m2.parameters()):
    q = p.clone()
    torch.distributed.broadcast(q, src=0)
    assert torch.equal(p, q), (name, float((p - q).abs().max()), int((p != q).sum()), p.numel(), tr2.exchange.small[list(PARAM_ORDER).index(name)])
    assert torch.isfinite(p).all()

# (4) replicated optimizer with SLICED keyframe gradients: the ranks' windows are all-gathered and summed inside ex4d_radam_step_sliced
m3, _, _ = make_scene("cfg3", P=20000, device="cuda", fused=True)
tr3 = FrameTrainer(m3, exchange="allreduce", optimizer=True, lrs={n: 1e-6 for n in PARAM_ORDER})
assert tr3.sliced and len(tr3.kf_gather) == 2
for k in range(3):
    tr3.step(cam, bg, stamps[(2 * k + rank) % len(stamps)], up)
tr3.flush(); torch.cuda.synchronize()
for name, p in zip(PARAM_ORDER, m3.parameters()):
    q = p.clone()
    torch.distributed.broadcast(q, src=0)
    assert torch.equal(p, q) and torch.isfinite(p).all(), name
# the sharded + sliced run of (3) and this replicated + sliced run saw the same frames with the same learning rates: the same parameter
# movement up to the order of the rasterizer's float atomics
for name, a0, a, b in zip(PARAM_ORDER, p2_before, m2.parameters(), m3.parameters()):
    da, db = a - a0, b - a0
    scale = float(db.abs().max())
    assert float((da - db).abs().max()) <= 2e-3 * scale + 1e-12, (name, float((da - db).abs().max()), scale)
assert tr2.exchange_bytes_on_wire() < tr3.exchange_bytes_on_wire() + 1
assert tr3.kf_gather[0].bytes_on_wire() == 4 * m3.num_dynamic * 12 and tr3.exchange.bytes_on_wire() < 4 * sum(p.numel() for p in m3.parameters()) - 4 * m3.num_dynamic * 35 * 7 + 1
torch.distributed.barrier(); torch.distributed.destroy_process_group()
print("OK", rank)
"""


def test_two_ranks_exchange_and_sharded_optimizer(hip_lib, tmp_path):
    script = tmp_path / "dist_worker.py"
    script.write_text(_WORKER)
    port = 30700 + (os.getpid() % 2000)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), h.ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=900)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"OK {r}" in o, o[-3000:]


_RCCL_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import torch
import torch.distributed as dist
from ex4dgs_amd import _C, dist as xd
from ex4dgs_amd.scene import make_scene
from ex4dgs_amd.trainer import FrameTrainer, NAN_TO_NUM
from ex4dgs_amd.optim import radam_step_raw
from ex4dgs_amd.attributes import PARAM_ORDER

torch.cuda.set_device(0); _C.load()
dist.init_process_group(backend="nccl", rank=0, world_size=1)          # "nccl" IS RCCL on ROCm
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
dev = torch.device("cuda", 0)
model, cam, bg = make_scene("cfg3", P=20000, device="cuda", fused=True)
H, W = cam.image_height, cam.image_width
w = torch.rand(3, H, W, generator=torch.Generator().manual_seed(5)).cuda()
up = lambda out: ([out["render"]], [w])
stamps = [0, 100, 200, 299, 41, 88]
solo = FrameTrainer(model, exchange="none")
solo.step(cam, bg, stamps[0], up); solo.flush()
g = [x.clone() for x in solo.grads().values()]
shapes = [x.shape for x in g]

# (1) ParamGradExchange, both modes, collectives FORCED in the one-rank group: reduce_scatter_tensor (in place, dist.py) / all_reduce
#     over RCCL; the sum over one rank is the input, bit for bit, in the rank's range AND outside it (nothing else may be touched)
for mode in ("reduce_scatter", "allreduce"):
    ex = xd.ParamGradExchange(shapes, dev, mode=mode, force=True, small_bytes=1 << 14)
    assert ex.active() and ex._native_rs and not all(ex.small)
    g2 = [x.clone() for x in g]
    ex.launch(g2); ex.wait(); torch.cuda.synchronize()
    for name, a, b in zip(PARAM_ORDER, g, g2):
        assert torch.equal(a, b), (mode, name)

# (2) SliceGather, native all_gather_into_tensor from send buffers of its own
win = torch.randn(model.num_dynamic, 4, 3, device=dev)
sg = xd.SliceGather(win.shape, dev, force=True)
assert sg._native and sg.force
sg.launch(win, 7); sg.wait(); torch.cuda.synchronize()
assert torch.equal(sg.all[0], win) and int(sg.first[0]) == 7 and sg.first_device_ptr() is not None
assert sg.windows(4)[0][0] == 7

# (3) ShardedRAdam over RCCL (reduce-scatter -> update of the own range -> in-place all_gather_into_tensor) == dense replicated update,
#     bit for bit, over 4 steps of real rasterizer gradients; a NaN in the flagged tensor's gradient goes through nan_to_num on both sides
lrs = [1e-3 * (1 + i) for i in range(15)]
flags = [n in NAN_TO_NUM for n in PARAM_ORDER]
A = [p.clone() for p in model.parameters()]
mA = [torch.zeros_like(p) for p in A]; vA = [torch.zeros_like(p) for p in A]
B = [p.clone() for p in model.parameters()]
sh = xd.ShardedRAdam(B, lrs, small_bytes=1 << 14, nan_to_num=flags, force=True)
assert sh._native and sh.force
for k in range(1, 5):
    solo.step(cam, bg, stamps[k], up); solo.flush()
    gk = [x.clone() for x in solo.grads().values()]
    if k == 2:
        gk[flags.index(True)].view(-1)[3] = float("nan")
    ga = [x.clone() for x in gk]
    radam_step_raw([(p.data_ptr(), x.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, k, int(f)) for p, x, m, v, lr, f in zip(A, ga, mA, vA, lrs, flags)],
                   (0.9, 0.999), 1e-8, dev)
    sh.step(gk)
    torch.cuda.synchronize()
    for name, a, b in zip(PARAM_ORDER, A, B):
        assert torch.equal(a, b) and bool(torch.isfinite(a).all()), (name, k)

# (4) the trainer with its collectives forced == the trainer without an exchange (one rank: the sums are the rank's own gradients).
#     Replicated optimizer + sliced keyframe gradients (all-reduce x2, SliceGather x2, positions read from the gathered device array)
#     and the sharded optimizer; the two runs differ only by the order of the rasterizer's float atomics
def run(exchange, force):
    m, _, _ = make_scene("cfg3", P=20000, device="cuda", fused=True)
    p0 = [p.clone() for p in m.parameters()]
    tr = FrameTrainer(m, exchange=exchange, optimizer=True, lrs={n: 1e-4 for n in PARAM_ORDER}, force_collectives=force)
    for k in range(4):
        tr.step(cam, bg, stamps[k], up)
    tr.flush(); torch.cuda.synchronize()
    return tr, [p - q for p, q in zip(m.parameters(), p0)]
ref_tr, ref = run("none", False)
for exchange in ("allreduce", "sharded"):
    tr, got = run(exchange, True)
    assert tr.mode == exchange
    if exchange == "allreduce":
        assert tr.sliced and tr.exchange.active() and tr.exchange_feat.active() and all(gth.force and gth._native for gth in tr.kf_gather)
    for name, a, b in zip(PARAM_ORDER, got, ref):
        scale = float(b.abs().max())
        assert scale > 0 or float(a.abs().max()) == 0, name
        assert float((a - b).abs().max()) <= 2e-3 * scale + 1e-12, (exchange, name, float((a - b).abs().max()), scale)
dist.barrier(); dist.destroy_process_group()
print("OK rccl")
"""


def test_rccl_native_collectives_run_in_a_one_rank_group(hip_lib, tmp_path):
    """The RCCL-native branches of ex4dgs_amd/dist.py -- reduce_scatter_tensor, all_gather_into_tensor from separate send buffers, the
    in-place all-gather of the sharded optimizer -- executed on hardware in a process group of ONE rank with the collectives forced
    (until now they had never run anywhere; an 8-GPU job must not be their first execution)."""
    script = tmp_path / "rccl_worker.py"
    script.write_text(_RCCL_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(32700 + (os.getpid() % 2000)), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script), h.ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0 and "OK rccl" in r.stdout, r.stdout[-4000:]


@pytest.mark.parametrize("config", ["cfg3", "cfg4"])
def test_bench_spawns_its_ranks_itself(hip_lib, config):
    """`bench.py --gpus 2` without torchrun: the script starts both ranks (sharing the box's GPU over gloo), rank 0 prints ONE JSON
    line whose n_gpus is what was asked for, the collective carries the 15 model-parameter gradients."""
    cmd = [sys.executable, os.path.join(h.ROOT, "bench.py"), "--gpus", "2", "--share-device", "--backend", "gloo", "--config", config,
           "--points", "20000", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"] + (["--train-leg"] if config == "cfg3" else [])
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["multi_gpu"]["ranks_seen"] == 2
    assert d["multi_gpu"]["collective_tensors"] == 15
    assert d["multi_gpu"]["exchange_bytes_per_rank"] > 0 and d["value"] > 0
    assert d["scaling"] == "weak" and d["config"]["frames_per_step"] == 2
    if config == "cfg3":
        # round 6: the default N > 1 step is the N = 1 step (rasterizer forward + backward, every rank its own views, no collective) --
        # the training-iteration core with the exchange is the secondary leg under multi_gpu
        assert "no collective" in d["config"]["parallelism"] and "GaussianRasterizer forward + backward" in d["config"]["step"]
        assert d["multi_gpu"]["allreduce_probe_ms"] > 0 and d["multi_gpu"]["allreduce_probe_bytes"] == 4 * int(64.8 * 20000)      # the default wire probe
        assert d["multi_gpu"]["train_core_ms_per_step"] > 0 and "gradient exchange" in d["multi_gpu"]["train_core_step"]                 # --train-leg
        assert d["roofline"] is not None and d["roofline"]["kernel"] in d["roofline"]["stage_ms"]
    else:
        assert "gradient exchange" in d["config"]["step"]          # BASELINE config 4 is quoted on the training core


def test_bench_refuses_a_rank_count_it_cannot_start(hip_lib):
    cmd = [sys.executable, os.path.join(h.ROOT, "bench.py"), "--gpus", str(torch.cuda.device_count() + 1), "--points", "2000", "--steps", "1", "--warmup", "0"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0 and not any(l.startswith("{") for l in r.stdout.splitlines())
