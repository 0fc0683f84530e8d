#!/usr/bin/env python
"""bench.py -- fwd+bwd ms/frame of the MI355X-native rasterizer on BASELINE.json's headline workload.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg3|cfg4|...]

N = 1, default config (cfg3: 1.0M static+dynamic Gaussians, K=35 keyframes, 1352x1014, SH degree 3, inputs resident in HBM):
    one "step" = GaussianRasterizer.forward + .backward on one synthetic frame, boundary to boundary (the reference's
    _RasterizeGaussians autograd surface; model getters and the loss are outside).  This is BASELINE.json's metric.
N > 1, or --config cfg4 (2.0M Gaussians x 300 timestamps) at any N:
    one "step" = the training-iteration core of ONE view per rank (ex4dgs_amd/trainer.py): fused attribute evaluation at the view's
    timestamp -> rasterizer forward+backward -> fused attribute backward -> asynchronous sum of the 15 MODEL-PARAMETER gradients over
    the ranks (RCCL over xGMI, overlapping the next frame).  Views / timestamps shard round-robin over the ranks (i = r mod N),
    parameters are replicated; per-rank work is fixed as N grows (weak scaling), `value` = ms per frame of the whole job.
`--gpus N` without a torchrun environment starts the N ranks itself (one process per GPU).  Rank 0 prints ONE JSON line.
Extra legs outside the timed region: per-stage hipEvent timing (roofline) and, on rank 0 at N=1, the CPU baselines (the C oracle on
BASELINE config 2 at full size and on a bounded sample of config 3, the pure-PyTorch rasterize of config 1).
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from ex4dgs_amd import _C, build as hip_build, dist as xdist                      # noqa: E402
from ex4dgs_amd.diff_gaussian_rasterization_df import GaussianRasterizationSettings, rasterize_gaussians   # noqa: E402
from ex4dgs_amd.scene import CONFIGS, make_scene                                   # noqa: E402

HBM_PEAK_GBPS = 8000.0     # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
PMC_FILE = os.path.join("profiles", "r06_pmc_traffic.json")
SIMDS, CLOCK_HZ = 1024, 2.4e9        # 256 CUs x 4 SIMDs; a SIMD issues one wave64 VALU instruction per 4 cycles (MI355X_MICROARCH.md)
RASTER_SOURCES = ("ex4d_preprocess.hip", "ex4d_binning.hip", "ex4d_rowsort.hip", "ex4d_composite.hip", "ex4d_api.hip", "ex4d_internal.h")


# which source files a stage's kernels are compiled from (plus the shared internal header)
STAGE_SOURCES = {"composite_fwd": "ex4d_composite.hip", "composite_bwd": "ex4d_composite.hip", "preprocess_fwd": "ex4d_preprocess.hip",
                 "preprocess_bwd": "ex4d_preprocess.hip", "depth_sort": "ex4d_binning.hip", "tile_sort": ("ex4d_rowsort.hip", "ex4d_binning.hip"),
                 "scan_tiles": "ex4d_binning.hip", "duplicate": "ex4d_binning.hip"}


def pmc_counters_current(pmc, stage=None):
    """The committed counter file is only quoted for a kernel while the sources that kernel is compiled from (its .hip file and the
    shared internal header) still have the hashes the counters were collected from; stage=None checks every source."""
    import hashlib
    want = pmc.get("source_sha16")
    if not want:
        return False
    d = os.path.join(ROOT, "ex4dgs_amd", "csrc")
    files = [f for f in sorted(os.listdir(d)) if f.endswith((".hip", ".h"))]
    if stage == "frame":
        files = list(RASTER_SOURCES)
    elif stage is not None:
        if stage not in STAGE_SOURCES:
            return False
        src = STAGE_SOURCES[stage]
        files = list(src if isinstance(src, tuple) else (src,)) + ["ex4d_internal.h"]
    return all(want.get(f) == hashlib.sha256(open(os.path.join(d, f), "rb").read()).hexdigest()[:16] for f in files)


def frame_inputs(model, t, device):
    with torch.no_grad():
        ins = [model.get_xyz_at_t(t), model.get_features(), model.get_opacity_at_t(t), model.get_scaling(), model.get_rotation_at_t(t)]
    return [x.detach().to(device).contiguous().requires_grad_(True) for x in ins]


def algorithmic_bytes(P, V, R, HW, T, D=3, passes=6):
    """SURVEY.md 8(d) / BASELINE.md: algorithmic bytes per frame and per stage."""
    Ma = (D + 1) ** 2
    st = {
        "preprocess_fwd": 12 * P + V * (32 + 12 * Ma) + 8 * P + 48 * V,
        "scan_tiles": 8 * P,
        "duplicate": 12 * R,
        "sort": 24 * passes * R,
        "tile_ranges": 8 * R + 8 * T,
        "composite_fwd": 56 * R + 52 * HW,
        "composite_bwd": 44 * R + 56 * HW + 104 * R,
        "preprocess_bwd": 579 * V + 312 * P,
    }
    A_fwd = sum(st[k] for k in ("preprocess_fwd", "scan_tiles", "duplicate", "sort", "tile_ranges", "composite_fwd"))
    A_bwd = st["composite_bwd"] + st["preprocess_bwd"]
    return st, A_fwd, A_bwd


def cpu_baseline(cfg_name, sample_P, t):
    """Oracle (scalar C port, 1 core) fwd+bwd on the generator of `cfg_name` at `sample_P` Gaussians."""
    from oracle import oracle
    model, cam, bg = make_scene(cfg_name, P=sample_P)
    cfg = CONFIGS[cfg_name]
    with torch.no_grad():
        xyz, shs, opa, scl, rot = model.get_xyz_at_t(t), model.get_features(), model.get_opacity_at_t(t), model.get_scaling(), model.get_rotation_at_t(t)
    H, W = cam.image_height, cam.image_width
    kw = dict(bg=bg, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, campos=cam.camera_center,
              image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), kernel_size=0.1,
              sh_degree=3, min_depth=cfg.min_depth, max_depth=cfg.max_depth, want_fragile=False)
    g = torch.Generator().manual_seed(0)
    t0 = time.time()
    f = oracle.forward(xyz, torch.zeros_like(xyz), opa, shs=shs, scales=scl, rotations=rot, **kw)
    t1 = time.time()
    gc = torch.randn(3, H, W, generator=g); gd = 0.1 * torch.randn(1, H, W, generator=g)
    gf = torch.rand(3, H, W, generator=g); ga = torch.zeros(1, H, W)
    t2 = time.time()
    oracle.backward(f, gc, gd, gf, ga, want_sums=False)
    t3 = time.time()
    ms = 1e3 * ((t1 - t0) + (t3 - t2))
    R = f["num_rendered"]
    frac = sample_P / CONFIGS[cfg_name].P
    return {"value": round(ms, 1), "unit": "ms/frame", "cores": 1, "kind": "port",
            "sample": f"{cfg_name} generator at P={sample_P} ({'full size' if frac == 1 else f'{frac:.2f}x Gaussians'}), {W}x{H}, 1 frame fwd+bwd, "
                      f"R={R} instances, oracle/ex4d_oracle.c scalar C, fwd {1e3 * (t1 - t0):.0f} ms + bwd {1e3 * (t3 - t2):.0f} ms",
            "pair_evals_per_s": round(2 * R * 256 / (ms / 1e3), 0)}


def cpu_torch_baseline():
    """BASELINE.json configs[0]: pure-PyTorch CPU rasterize of 256 Gaussians @256x256 on all host cores."""
    from oracle import oracle_torch
    ncpu = os.cpu_count() or 1
    model, cam, bg = make_scene("cfg1")
    leaf = lambda x: x.detach().clone().requires_grad_(True)
    xyz, rot, opa, scl, shs = [leaf(x) for x in (model.get_xyz_at_t(0), model.get_rotation_at_t(0), model.get_opacity_at_t(0), model.get_scaling(), model.get_features())]
    H, W = cam.image_height, cam.image_width
    kw = dict(bg=bg, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, campos=cam.camera_center,
              image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), kernel_size=0.1,
              sh_degree=3, min_depth=4.0, max_depth=300.0)
    # a dense [65 536 x ~230] problem oversubscribes a 256-thread host (21 s on 256 threads against 9 s on 8):
    # the best of {8, 32, all} threads is reported, with every timing stated
    tried = {}
    prev = torch.get_num_threads()
    for n in sorted({min(8, ncpu), min(32, ncpu), ncpu}):
        torch.set_num_threads(n)
        for x in (xyz, rot, opa, scl, shs):
            x.grad = None
        t0 = time.time()
        out = oracle_torch.rasterize(xyz, torch.zeros_like(xyz), opa, shs, scl, rot, **kw)
        t1 = time.time()
        out["color"].sum().backward()
        t2 = time.time()
        tried[n] = (1e3 * (t2 - t0), 1e3 * (t1 - t0), 1e3 * (t2 - t1))
    torch.set_num_threads(prev)
    n = min(tried, key=lambda k: tried[k][0])
    return {"value": round(tried[n][0], 1), "unit": "ms/frame", "cores": n, "kind": "port", "host_cores": ncpu,
            "ms_by_threads": {str(k): round(v[0], 1) for k, v in tried.items()},
            "sample": f"cfg1: 256 static Gaussians, 256x256, oracle/oracle_torch.py dense pixels x Gaussians, fwd {tried[n][1]:.0f} ms + autograd bwd {tried[n][2]:.0f} ms "
                      f"on {n} threads (best of {sorted(tried)} threads on a {ncpu}-core host)"}


def model_step_timing(cfg_name, dev, grads, steps=20, warmup=5, points=None):
    """End-to-end training-step core on one GPU: per-frame attribute evaluation + rasterizer forward +
    backward down to the model parameters, with the reference's torch getters vs the fused HIP op."""
    from ex4dgs_amd.render import render
    cfg = CONFIGS[cfg_name]
    out = {}
    for mode in ("torch_getters", "fused_getters"):
        model, cam, bg = make_scene(cfg_name, P=points, device=dev, fused=(mode == "fused_getters"))
        cam = cam.to(dev); bg = bg.to(dev)
        for p in model.parameters():
            p.requires_grad_(True)
        stamps = [0, 137, 299]

        def step(i):
            for p in model.parameters():
                p.grad = None
            o = render(cam, model, None, bg, timestamp=stamps[i % 3], near=cfg.min_depth, far=cfg.max_depth, sync=False)
            torch.autograd.backward([o["render"], o["depth"], o["opticalflow"], o["acc"]], grads)
        for i in range(warmup):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        torch.cuda.synchronize()
        out[mode + "_ms_per_frame"] = round(1e3 * (time.perf_counter() - t0) / steps, 4)

        # the loss-driven iteration of train.py:126-153: getters -> render -> L1+SSIM loss (+ error maps) -> backward
        from ex4dgs_amd.loss import l1_ssim_loss, l1_ssim_loss_unfused
        loss_fn = l1_ssim_loss if mode == "fused_getters" else l1_ssim_loss_unfused
        gt = torch.rand(3, cfg.height, cfg.width, device=dev)

        def train_iter(i):
            for p in model.parameters():
                p.grad = None
            o = render(cam, model, None, bg, timestamp=stamps[i % 3], near=cfg.min_depth, far=cfg.max_depth, sync=False)
            loss, _l1e, _sse = loss_fn(o["render"], gt, 0.2)
            loss.backward()
        for i in range(warmup):
            train_iter(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            train_iter(i)
        torch.cuda.synchronize()
        tag = "fused" if mode == "fused_getters" else "torch"
        out[tag + "_host_side_train_iter_ms"] = round(1e3 * (time.perf_counter() - t0) / steps, 4)

        # ... plus the optimizer step of train.py:250-251 (RAdam over the 15 groups, tiny lr so the scene stays put)
        from ex4dgs_amd.optim import FusedRAdam
        groups = [{"params": [p], "lr": 1e-7, "name": str(i)} for i, p in enumerate(model.parameters())]
        opt = FusedRAdam(groups, lr=0.001) if mode == "fused_getters" else torch.optim.RAdam(groups, lr=0.001)

        def full_iter(i):
            o = render(cam, model, None, bg, timestamp=stamps[i % 3], near=cfg.min_depth, far=cfg.max_depth, sync=False)
            loss, _l1e, _sse = loss_fn(o["render"], gt, 0.2)
            loss.backward()
            opt.step()
            opt.zero_grad(set_to_none=True)
        for i in range(warmup + 3):
            full_iter(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            full_iter(i)
        torch.cuda.synchronize()
        out[tag + "_host_side_full_iter_with_radam_ms"] = round(1e3 * (time.perf_counter() - t0) / steps, 4)
        del model, opt
    # the same full iteration through the compiled host path (include/ex4d_trainer.h: one C++ call per iteration, persistent workspace,
    # keyframe gradients as slices), at this size and at 300 k Gaussians where the Python host work used to bound the iteration
    try:
        from ex4dgs_amd.native_trainer import NativeTrainer
        from ex4dgs_amd.trainer import FrameTrainer
        from ex4dgs_amd.loss import l1_ssim_loss as _loss
        for tag, npts in (("", points), ("_300k", 300_000)):
            model, cam, bg = make_scene(cfg_name, P=npts, device=dev, fused=True)
            cam = cam.to(dev); bg = bg.to(dev)
            gt = torch.rand(3, cfg.height, cfg.width, device=dev)
            lrs = {n: 1e-7 for n in model.PARAM_NAMES}
            nt = NativeTrainer(model, cam, optimizer=True, lrs=lrs, near=cfg.min_depth, far=cfg.max_depth)
            stamps = [0, 137, 299]
            for i in range(warmup + 3):
                nt.step(cam, bg, stamps[i % 3], gt)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                nt.step(cam, bg, stamps[i % 3], gt)
            t_host = time.perf_counter() - t0          # time the host needed to enqueue the iterations (the GPU runs behind)
            torch.cuda.synchronize()
            out["native_host_full_iter_with_radam_ms" + tag] = round(1e3 * (time.perf_counter() - t0) / steps, 4)
            out["native_host_enqueue_ms" + tag] = round(1e3 * t_host / steps, 4)
            out["native_host_workspace_MB" + tag] = round(nt.bytes() / 2 ** 20, 1)
            nt.close()
            if tag:
                ft = FrameTrainer(model, optimizer=True, lrs=lrs)
                upg = lambda o: ([_loss(o["render"], gt, 0.2)[0]], [None])
                for i in range(warmup + 3):
                    ft.step(cam, bg, stamps[i % 3], upg, near=cfg.min_depth, far=cfg.max_depth)
                ft.flush(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(steps):
                    ft.step(cam, bg, stamps[i % 3], upg, near=cfg.min_depth, far=cfg.max_depth)
                ft.flush(); torch.cuda.synchronize()
                out["python_host_full_iter_with_radam_ms" + tag] = round(1e3 * (time.perf_counter() - t0) / steps, 4)
                del ft
            del model, nt
    except Exception as e:      # reported, not fatal for the headline line
        out["native_host_error"] = repr(e)
    out["what"] = ("native_host_*: the full iteration (getters, render, L1+SSIM, backward, RAdam) through ex4d_trainer_step, "
                   "python_host_*: the same kernels sequenced by trainer.FrameTrainer; "
                   "*_getters_ms_per_frame: getters (xyz/rotation/opacity/scaling/features at t) + rasterizer fwd+bwd to the model "
                   "parameters; *_train_iter_ms: the same plus the L1+SSIM loss and error maps of train.py:144-151 "
                   "(torch = the reference's op composition, fused = ex4d_attributes + ex4d_l1_ssim); *_full_iter_with_radam_ms: plus "
                   "optimizer.step() + zero_grad (torch.optim.RAdam vs FusedRAdam); 1 GPU, same HIP rasterizer in both")
    return out


# ------------------------------------------------------------------------------------------------------
def spawn_ranks(args):
    """`--gpus N` outside a torchrun environment: start the N ranks ourselves (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* like
    torchrun), forward rank 0's JSON line, fail if any rank fails."""
    n = args.gpus
    have = torch.cuda.device_count()
    if not args.share_device and have < n:
        print(f"bench.py: --gpus {n} but only {have} GPU(s) visible (use --share-device --backend gloo to debug on fewer)", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    if torch.cuda.is_available():
        hip_build.build()                       # once, before the ranks race for it
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=None, text=True))
    out0 = procs[0].communicate()[0]
    rcs = [p.wait() for p in procs]
    if any(rcs):
        print(f"bench.py: rank exit codes {rcs}", file=sys.stderr)
        return 1
    lines = [l for l in out0.splitlines() if l.startswith("{")]
    if len(lines) != 1 or json.loads(lines[0]).get("n_gpus") != n:
        print(f"bench.py: expected one JSON line with n_gpus={n}, got: {out0[-500:]}", file=sys.stderr)
        return 1
    print(lines[0], flush=True)
    return 0


def percentiles(ms):
    s = sorted(ms)
    pick = lambda q: s[min(len(s) - 1, int(round(q * (len(s) - 1))))]
    return {"mean": round(sum(s) / len(s), 4), "p50": round(pick(0.5), 4), "p95": round(pick(0.95), 4), "min": round(s[0], 4), "max": round(s[-1], 4),
            "n": len(s), "how": "hipEvent pairs around every step of a second loop of the same K steps, behind the timed region (which holds nothing but the steps)"}


def timed_loop(step, steps, warmup, sync_all, finish=None):
    """W warm-up steps, then EXACTLY K timed steps between barrier + synchronize pairs -- nothing but the steps inside the timed region
    (round 6: the per-step hipEvents of rounds 2-5 sat inside it, one marker packet per step on the op's stream).  The per-step
    distribution (`step_ms`) comes from a SECOND loop of the same K steps with an event recorded after every step, outside the timed
    region."""
    for i in range(warmup):
        step(i)
    if finish:
        finish()
    sync_all()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    if finish:
        finish()
    sync_all()
    t1 = time.perf_counter()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    ev[0].record()
    for i in range(steps):
        step(i)
        ev[i + 1].record()
    if finish:
        finish()
    sync_all()
    per_step = [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]
    return 1e3 * (t1 - t0) / max(steps, 1), per_step


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="cfg3", choices=sorted(CONFIGS))
    ap.add_argument("--points", type=int, default=None, help="override the Gaussian count (parity/debug only)")
    ap.add_argument("--fwd-asm", type=int, default=None, choices=[0, 1], help="tuning: compositing forward with the hand-scheduled entry walk (1, default) or the compiler's loop (0)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-allreduce", action="store_true", help="N > 1 without the gradient exchange (pure replicas)")
    ap.add_argument("--forward-only", action="store_true", help="render only (BASELINE config 5 is quoted as forward-only FPS); not the headline metric")
    ap.add_argument("--no-model-step", action="store_true", help="skip the training-iteration timings (profiling runs)")
    ap.add_argument("--train-core", action="store_true", help="N = 1: time the training-iteration core (what N > 1 and cfg4 time) instead of the rasterizer alone")
    ap.add_argument("--optimizer", default=None, choices=["none", "replicated", "sharded"],
                    help="training-core steps (N > 1, cfg4, --train-core): the optimizer step of the iteration.  Default: replicated -- the "
                         "reference steps its optimizer every iteration (train.py:250); 'none' times the gradient-only core")
    ap.add_argument("--dense-keyframe-grads", action="store_true", help="training-core steps with the replicated optimizer: dense keyframe gradients instead of the 4 / 2 touched time slices")
    ap.add_argument("--cpu-sample", type=int, default=250_000)
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL); gloo for debugging")
    ap.add_argument("--share-device", action="store_true", help="debug: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--async-frames", action="store_true", help="N = 1 rasterizer steps with the ASYNCHRONOUS forward (no instance-count read-back: Ex4dParams.instance_capacity)")
    ap.add_argument("--graph", action="store_true", help="N = 1: forward + backward (asynchronous forward, raw C-ABI mirror calls) captured into ONE hipGraph and replayed per step")
    ap.add_argument("--bwd-variant", type=int, default=None, help="tuning: compositing-backward kernel (include/ex4d_rasterizer.h: ex4d_set_option)")
    ap.add_argument("--sync-forward", action="store_true", help="(the default since round 6; kept for old command lines) training-core steps: the trainer's rasterizer forward with the instance-count read-back")
    ap.add_argument("--async-forward", action="store_true", help="training-core steps on one rank without an exchange: the trainer's asynchronous rasterizer forward (no read-back, a frame that overflows its capacity is re-run)")
    ap.add_argument("--views-per-rank", type=int, default=1, metavar="K",
                    help="training-core steps: every rank renders K views per optimizer step, accumulates their gradients locally and exchanges ONCE "
                         "(FrameTrainer(views_per_step=K): batch = N K views, wire time per view 1/K); a timed step stays one view")
    ap.add_argument("--train-leg", action="store_true", help="N > 1 rasterizer steps: also time a short leg of the training-iteration core with its gradient exchange (reported under multi_gpu; "
                                                             "the default N > 1 run prices the wire with a plain all-reduce of the exchange's byte count only)")
    ap.add_argument("--set", action="append", default=[], metavar="OPTION=VALUE", help="tuning / A-B runs: any library option of ex4d_set_option, e.g. --set depth_sort_msd=0 (the 3-pass LSD depth sort)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the rasterizer has no CPU fallback")
    if args.share_device:
        os.environ["LOCAL_RANK"] = "0"
    rank, world, local = xdist.init_from_env(backend=args.backend)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: refusing to print a line for a different rank count")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if rank == 0:
        hip_build.build()            # one builder; the other ranks wait and then only dlopen
    if world > 1:
        torch.distributed.barrier()
    _C.load()
    if args.fwd_asm is not None:
        _C.set_option("composite_fwd_asm", args.fwd_asm)
    if args.bwd_variant is not None:
        _C.set_option("composite_bwd_variant", args.bwd_variant)
    for kv in args.set:
        name, _, value = kv.partition("=")
        _C.set_option(name, int(value))

    cfg = CONFIGS[args.config]
    # N > 1 (round 6): the timed step is the SAME step as at N = 1 -- GaussianRasterizer forward + backward on resident frames, every rank on
    # its own views, no collective in the data path (the metric BASELINE.json names; `value` at N ranks compares like with like with N = 1).
    # The training-iteration core with the RCCL gradient exchange -- what BASELINE config 4 quotes -- is `--train-core` / `--config cfg4`
    # (its N = 1 counterpart: `--train-core` at N = 1).  Beside the headline a default N > 1 run prices the WIRE alone under `multi_gpu`: a
    # plain all-reduce of the replicated exchange's byte count (the one collective a barrier needs anyway -- nothing that could take the
    # headline down with it); `--train-leg` adds a short leg of the training core itself with its exchange.
    train_mode = (args.config == "cfg4" or args.train_core) and not args.forward_only
    train_leg = world > 1 and not train_mode and not args.forward_only and args.train_leg
    wire_probe = world > 1 and not train_mode and not args.forward_only
    if args.optimizer is None:
        # N >= 4: reduce-scatter + sharded RAdam (row-sharded keyframe windows) + all-gather -- half the bytes per link of the all-reduce
        # and 1/N of the optimizer's HBM stream per rank; N < 4: the replicated optimizer
        args.optimizer = ("sharded" if world >= 4 else "replicated") if (train_mode or train_leg) else "none"
    H, W = cfg.height, cfg.width
    g = torch.Generator().manual_seed(1000 + rank)
    grads = [torch.randn(3, H, W, generator=g).to(dev), (0.1 * torch.randn(1, H, W, generator=g)).to(dev),
             torch.rand(3, H, W, generator=g).to(dev), torch.zeros(1, H, W, device=dev)]

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def run_train_core(steps, warmup):
        """The training-iteration core, one view per rank per step (N > 1: with the gradient exchange); returns its timings and, for the
        statistics pass of a run whose PRIMARY step it is, the frames it rendered."""
        multi = None
        # ---------------- training-iteration core, one view per rank per step ----------------
        from ex4dgs_amd.trainer import FrameTrainer
        model, cam, bg = make_scene(args.config, P=args.points, device=dev, fused=True)
        cam = cam.to(dev); bg = bg.to(dev)
        P = model.num_static + model.num_dynamic
        n_stamps = 300 if args.config == "cfg4" else 8
        all_stamps = list(range(300)) if args.config == "cfg4" else [0, 137, 299, 41, 203, 88, 266, 171]
        my_stamps = [all_stamps[i] for i in xdist.shard_views(n_stamps, rank, world)]      # i = rank (mod world)
        upstream = lambda out: ([out["render"], out["depth"], out["opticalflow"], out["acc"]], grads)
        exchange = "none" if (world == 1 or args.no_allreduce) else ("sharded" if args.optimizer == "sharded" else "allreduce")
        if world == 1 and args.optimizer == "sharded":
            exchange = "sharded"
        kv = max(1, args.views_per_rank)
        tr = FrameTrainer(model, exchange=exchange, optimizer=(args.optimizer != "none"), sliced=(False if (args.dense_keyframe_grads or kv > 1) else None),
                          lrs={n: 1e-7 for n in model.PARAM_NAMES},       # tiny learning rates: the synthetic scene stays put
                          async_forward=(True if (args.async_forward and kv == 1 and exchange == "none") else False), views_per_step=kv)

        def step(i):
            return tr.step(cam, bg, my_stamps[i % len(my_stamps)], upstream, near=cfg.min_depth, far=cfg.max_depth)["render"]

        ms_wall, per_step = timed_loop(step, steps, warmup, sync_all, finish=tr.flush)
        parallelism = f"views/timestamps sharded i = r (mod {world}), parameters replicated" + (
            "" if tr.exchange is None else (" + reduce-scatter / sharded RAdam / all-gather" if exchange == "sharded" else " + async RCCL all-reduce of the 15 model-parameter gradients"))
        step_what = (("" if kv == 1 else f"[{kv} views per rank and optimizer step, gradients accumulated locally, one exchange per {kv} views] ") +
                     "training-iteration core of one view per rank: fused attribute evaluation -> rasterizer forward+backward -> attribute "
                     "backward" + ("" if tr.exchange is None else " -> gradient exchange") + ("" if args.optimizer == "none" else f" -> {args.optimizer} RAdam step"))
        secondary = {}
        if args.optimizer != "none":
            # secondary number: the same step without the optimizer (the mode in which the exchange can hide behind the next frame)
            trn = FrameTrainer(model, exchange=("none" if exchange == "sharded" else exchange), optimizer=False)
            stepn = lambda i: trn.step(cam, bg, my_stamps[i % len(my_stamps)], upstream, near=cfg.min_depth, far=cfg.max_depth)
            ms_noopt, _ = timed_loop(stepn, steps, max(2, warmup // 2), sync_all, finish=trn.flush)
            secondary["ms_per_step_without_optimizer"] = round(xdist.allreduce_max_scalar(ms_noopt, device=dev), 4)
            del trn
        if world > 1:
            # the same loop (same optimizer) without the exchange (exposed communication = difference) and the exchange alone (its full length)
            tr0 = FrameTrainer(model, exchange="none", optimizer=(args.optimizer != "none"), sliced=(False if args.dense_keyframe_grads else None),
                               lrs={n: 1e-7 for n in model.PARAM_NAMES})
            step0 = lambda i: tr0.step(cam, bg, my_stamps[i % len(my_stamps)], upstream, near=cfg.min_depth, far=cfg.max_depth)
            ms_noex, _ = timed_loop(step0, steps, max(2, warmup // 2), sync_all, finish=tr0.flush)
            ex_alone = None
            if tr.exchange is not None and exchange == "allreduce":
                trg = FrameTrainer(model, exchange="none")
                trg.step(cam, bg, my_stamps[0], upstream, near=cfg.min_depth, far=cfg.max_depth); trg.flush()
                gd = trg.grads()
                gfeat = [gd[tr.names[i]] for i in tr.feat_pos]
                grest = [gd[tr.names[i]] for i in tr.rest_pos]

                def ex_only(i):
                    tr.exchange_feat.launch(gfeat); tr.exchange.launch(grest); tr.exchange_feat.wait(); tr.exchange.wait()
                ex_alone, _ = timed_loop(ex_only, max(4, steps // 4), 2, sync_all)
            ms_noex_local = ms_noex
            ms_noex = xdist.allreduce_max_scalar(ms_noex, device=dev)
            # ranks that really took part: a sum all-reduce of ones over the process group (RCCL on the GPU backend), not the configured size
            ones = torch.ones(1, device=dev)
            torch.distributed.all_reduce(ones)
            per_rank = torch.zeros(world, dtype=torch.float64, device=dev)
            per_rank[rank] = max(0.0, ms_wall - ms_noex_local)          # this rank's own exposed exchange (its step with - without the exchange)
            torch.distributed.all_reduce(per_rank)
            multi = {"ranks_seen": int(round(float(ones.item()))), "world_size": torch.distributed.get_world_size(), "backend": torch.distributed.get_backend(),
                     "exposed_exchange_ms_per_rank": [round(float(x), 4) for x in per_rank.tolist()],
                     "optimizer": args.optimizer,
                     "collective_tensors": len(model.PARAM_NAMES) if tr.exchange is not None else 0,
                     "exchange_bytes_per_rank": tr.exchange_bytes_on_wire(),
                     "sliced_keyframe_gradients": bool(tr.sliced),
                     "ms_per_step_without_exchange": round(ms_noex, 4),
                     "allreduce_ms_per_step": None if ex_alone is None else round(xdist.allreduce_max_scalar(ex_alone, device=dev), 4),
                     "share_device": bool(args.share_device)}
            multi.update(secondary)
        ms_step = xdist.allreduce_max_scalar(ms_wall, device=dev)
        if multi is not None:
            multi["exposed_exchange_ms_per_step"] = round(max(0.0, ms_step - multi["ms_per_step_without_exchange"]), 4)
            if multi["allreduce_ms_per_step"] is not None:
                multi["overlapped_exchange_ms_per_step"] = round(max(0.0, multi["allreduce_ms_per_step"] - multi["exposed_exchange_ms_per_step"]), 4)
        model.fused = False           # plain getters for the statistics pass below (one [P,16,3] SH tensor instead of the SplitSH)
        frames = [frame_inputs(model, t, dev) for t in my_stamps[:2]]
        settings = tr._settings(cam, bg, cfg.min_depth, cfg.max_depth)
        return dict(ms_wall=ms_wall, ms_step=ms_step, per_step=per_step, multi=multi, secondary=secondary, parallelism=parallelism, step_what=step_what,
                    frames=frames, settings=settings, P=P, my_stamps=my_stamps)


    multi = None
    if not train_mode:
        # ---------------- rasterizer alone on resident frames (BASELINE metric) ----------------
        model, cam, bg = make_scene(args.config, P=args.points)
        cam = cam.to(dev); bg = bg.to(dev)
        stamps = [0, 137, 299, 41, 203, 88, 266, 171]
        my_stamps = [stamps[(i * world + rank) % len(stamps)] for i in range(3)]
        frames = [frame_inputs(model, t, dev) for t in my_stamps]
        del model
        P = frames[0][0].shape[0]
        settings = GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), kernel_size=0.1,
            subpixel_offset=torch.zeros(H, W, 2, device=dev), bg=bg, scale_modifier=1.0, viewmatrix=cam.world_view_transform,
            projmatrix=cam.full_proj_transform, sh_degree=3, campos=cam.camera_center, prefiltered=False,
            min_depth=cfg.min_depth, max_depth=cfg.max_depth, debug=False)
        empty = torch.Tensor([])
        means2D = [torch.zeros(P, 3, device=dev, requires_grad=True) for _ in frames]
        dir3D = [torch.zeros(P, 3, device=dev, requires_grad=True) for _ in frames]

        def step(i):
            f = i % len(frames)
            xyz, shs, opa, scl, rot = frames[f]
            for t in frames[f] + [means2D[f], dir3D[f]]:
                t.grad = None
            if args.forward_only:
                with torch.no_grad():
                    return rasterize_gaussians(xyz, means2D[f], dir3D[f], shs, empty, opa, scl, rot, empty, settings)[0]
            color, radii, depth, flow, acc, idx = rasterize_gaussians(xyz, means2D[f], dir3D[f], shs, empty, opa, scl, rot, empty, settings)
            torch.autograd.backward([color, depth, flow, acc], [grads[0], grads[1], grads[2], grads[3]])
            return color

        mode_note = ""
        if args.graph:
            # forward + backward of frame 0 captured once (asynchronous forward: host-constant grids, no host wait), replayed per step
            from ex4dgs_amd.diff_gaussian_rasterization_df import async_frames
            xyz, shs, opa, scl, rot = [t.detach() for t in frames[0]]
            zero_dir = torch.zeros(P, 3, device=dev)
            fargs = (settings.bg, xyz, zero_dir, empty, opa, scl, rot, 1.0, empty, settings.viewmatrix, settings.projmatrix, settings.tanfovx,
                     settings.tanfovy, 0.1, settings.subpixel_offset, H, W, shs, 3, settings.campos, False, settings.min_depth, settings.max_depth, False)
            probe = _C.rasterize_gaussians(*fargs)
            cap = int(1.25 * int(probe[0])) + 4096
            del probe

            def raw(cap):
                f = _C.rasterize_gaussians(*fargs, prepare_backward=not args.forward_only, instance_capacity=cap, assume_no_flow=True)
                if args.forward_only:
                    return f, None
                b = _C.rasterize_gaussians_backward(settings.bg, xyz, f[2], empty, scl, rot, f[6], f[7], settings.min_depth, settings.max_depth, 1.0, empty,
                                                    settings.viewmatrix, settings.projmatrix, settings.tanfovx, settings.tanfovy, 0.1, settings.subpixel_offset,
                                                    grads[0], grads[1], grads[2], grads[3], shs, 3, settings.campos, f[3], f[0], f[4], f[5], False,
                                                    need_colors=False, need_cov3D=False, prepared=True)
                return f, b
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                raw(cap)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                gf, gb = raw(cap)

            def step(i):
                graph.replay()
                return gf[1]
            ms_wall, per_step = timed_loop(step, args.steps, args.warmup, sync_all)
            assert gf[0].wait().valid, "the captured frame overflowed its capacity"
            mode_note = f" replayed from one hipGraph (asynchronous forward, capacity {cap} for {gf[0].num_rendered} instances)"
        else:
            if args.async_frames:
                from ex4dgs_amd.diff_gaussian_rasterization_df import async_frames
                async_frames.enable(headroom=1.25, strict=False)
            ms_wall, per_step = timed_loop(step, args.steps, args.warmup, sync_all)
            if args.async_frames:
                async_frames.drain()
                assert async_frames.invalid_frames == 0, "an asynchronous frame overflowed inside the timed region"
                mode_note = f" (asynchronous forward: no instance-count read-back, capacity {async_frames.capacity})"
                async_frames.enabled = False
        parallelism = f"frame-sharded x{world}" + ("" if world == 1 else " (no collective)")
        step_what = "GaussianRasterizer forward" + ("" if args.forward_only else " + backward") + " on a resident frame" + mode_note


    secondary = {}
    if train_mode:
        tc = run_train_core(args.steps, args.warmup)
        ms_wall, per_step, multi, secondary = tc["ms_wall"], tc["per_step"], tc["multi"], tc["secondary"]
        parallelism, step_what, frames, settings, P, my_stamps = tc["parallelism"], tc["step_what"], tc["frames"], tc["settings"], tc["P"], tc["my_stamps"]
        empty = torch.Tensor([])
        means2D = [torch.zeros(P, 3, device=dev, requires_grad=True) for _ in frames]
        dir3D = [torch.zeros(P, 3, device=dev, requires_grad=True) for _ in frames]
    elif train_leg:
        # secondary leg of an N > 1 run: the training-iteration core with the RCCL exchange, a short loop -- NOT the line's `value`
        tc = run_train_core(min(args.steps, 12), min(args.warmup, 3))
        multi = tc["multi"] or {}
        multi["train_core_ms_per_step"] = round(tc["ms_step"], 4)
        multi["train_core_ms_per_frame"] = round(tc["ms_step"] / world, 4)
        multi["train_core_step"] = tc["step_what"]
        multi["train_core_parallelism"] = tc["parallelism"]
        multi["note"] = ("secondary leg (min(steps, 12) timed steps): the training-iteration core of BASELINE config 4's kind on this config, one view per rank and step, "
                         "with the gradient exchange over the process group; the line's `value` is the rasterizer step above, without any collective")
        multi.update(tc["secondary"])
        del tc
    if wire_probe:
        # what the replicated gradient exchange would put on the wire per step (15 model-parameter gradients with the keyframe tensors as
        # 4 / 2 touched time slices: ex4dgs_amd/dist.py), as ONE flat all-reduce: ring time of that byte count over this node's links
        # (259.2 bytes per Gaussian at the 20 % dynamic share of configs 3 / 4: FrameTrainer.exchange_bytes_on_wire() reports 259 200 004 at 1.0 M)
        nbytes = 4 * int(64.8 * P)
        buf = torch.zeros(nbytes // 4, device=dev)
        for _ in range(2):
            torch.distributed.all_reduce(buf)
        sync_all()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            torch.distributed.all_reduce(buf)
        sync_all()
        wire_ms = xdist.allreduce_max_scalar(1e3 * (time.perf_counter() - t0) / reps, device=dev)
        ones = torch.ones(1, device=dev)
        torch.distributed.all_reduce(ones)
        probe = {"ranks_seen": int(round(float(ones.item()))), "world_size": torch.distributed.get_world_size(), "backend": torch.distributed.get_backend(),
                 "allreduce_probe_bytes": int(nbytes), "allreduce_probe_ms": round(wire_ms, 4),
                 "allreduce_probe_busbw_GBps": round(2.0 * (world - 1) / world * nbytes / (wire_ms * 1e-3) / 1e9, 1),
                 "allreduce_probe_note": "one flat all-reduce of the replicated gradient exchange's byte count, outside the timed region: the wire time a training step adds "
                                         "when nothing overlaps it (DESIGN.md section 6); `--train-leg` / `--train-core` time the training core itself",
                 "share_device": bool(args.share_device)}
        del buf
        multi = dict(probe, **(multi or {}))

    ms_per_step = xdist.allreduce_max_scalar(ms_wall, device=dev)

    # ---- per-stage hipEvent timing of the rasterizer (outside the timed region), scene statistics ----------------
    _C.profile_enable(True)
    agg = {}
    nprof = 12
    for i in range(nprof):
        f = i % len(frames)
        xyz, shs, opa, scl, rot = frames[f]
        for t in frames[f] + [means2D[f], dir3D[f]]:
            t.grad = None
        outs = rasterize_gaussians(xyz, means2D[f], dir3D[f], shs, empty, opa, scl, rot, empty, settings)
        if not args.forward_only:
            torch.autograd.backward([outs[0], outs[2], outs[3], outs[4]], grads)
        torch.cuda.synchronize()
        for which in (0, 1):
            for name, ms in _C.profile_read(which):
                agg[name] = agg.get(name, 0.0) + ms / nprof
    _C.profile_enable(False)
    last = _C.rasterize_gaussians(settings.bg, frames[0][0].detach(), empty, empty, frames[0][2].detach(), frames[0][3].detach(),
                                  frames[0][4].detach(), 1.0, empty, settings.viewmatrix, settings.projmatrix, settings.tanfovx,
                                  settings.tanfovy, 0.1, settings.subpixel_offset, H, W, frames[0][1].detach(), 3, settings.campos,
                                  False, settings.min_depth, settings.max_depth, False)
    R = int(last[0])
    V = int((last[2] > 0).sum().item())
    T = ((W + 15) // 16) * ((H + 15) // 16)
    stage_bytes, A_fwd, A_bwd = algorithmic_bytes(P, V, R, H * W, T)
    A = A_fwd + (0 if args.forward_only else A_bwd)
    kernel_ms = sum(agg.values())
    dom = max(agg, key=agg.get) if agg else None
    dom_bytes = stage_bytes["sort"] if dom in ("tile_sort", "depth_sort") else stage_bytes.get(dom)
    roof = None
    traffic = valu_busy = valu_insts = frame_pmc = None
    try:   # HBM bytes per launch from the rocprofv3 --pmc passes of this command, collected separately and committed (tools/pmc.sh)
        with open(os.path.join(ROOT, PMC_FILE)) as fh:
            pmc = json.load(fh)
        pmc_ok = pmc_counters_current(pmc, dom)
        if pmc_ok and args.config == "cfg3" and args.points is None and dom in pmc["kernels"]:
            traffic = pmc["kernels"][dom]["hbm_bytes_per_launch"]
            valu_busy = pmc["kernels"][dom].get("valu_busy_frac")
            valu_insts = pmc["kernels"][dom].get("SQ_INSTS_VALU")
        if pmc.get("frame") and pmc_counters_current(pmc, "frame") and args.config == "cfg3" and args.points is None and not args.forward_only:
            frame_pmc = pmc["frame"]
    except Exception:
        traffic, pmc_ok = None, False
    if dom is not None and dom_bytes:
        achieved = dom_bytes / (agg[dom] * 1e-3) / 1e9
        rast_ms = ms_per_step if not train_mode else kernel_ms
        # What bounds the dominant kernel, from ITS counters: it is priced against HBM by the contract
        # (`achieved` / `frac` below use the section-8(d) bytes of the reference's algorithm), but when its real HBM traffic is a small
        # fraction of what the HBM could move in its duration and its SIMDs are busy issuing VALU instructions, the bound is the
        # instruction issue: floor = wave-instructions x 4 cycles / (1024 SIMDs x 2.4 GHz)
        issue_floor_ms = None if not valu_insts else valu_insts * 4.0 / (SIMDS * CLOCK_HZ) * 1e3
        frac_traffic_dom = None if traffic is None else traffic / (agg[dom] * 1e-3) / 1e9 / HBM_PEAK_GBPS
        bound = "hbm"
        if valu_busy is not None and frac_traffic_dom is not None and valu_busy > 0.6 and frac_traffic_dom < 0.4:
            bound = "valu-issue"
        roof = {"bound": bound, "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                "frac_traffic": None if frac_traffic_dom is None else round(frac_traffic_dom, 4),
                "valu_insts": valu_insts, "issue_floor_ms": None if issue_floor_ms is None else round(issue_floor_ms, 4),
                "frac_of_issue_floor": None if issue_floor_ms is None else round(issue_floor_ms / agg[dom], 4),
                "note": ("achieved / frac: SURVEY 8(d) algorithmic bytes of the REFERENCE's algorithm / this kernel's duration (the contract's index); "
                         "traffic / frac_traffic: this kernel's real HBM bytes per launch from the counters and their share of the 8 TB/s peak; "
                         "bound: what its own counters say limits it"),
                "traffic_source": (PMC_FILE + " (separate rocprofv3 --pmc passes of this command, committed with the hashes of the kernel sources; not re-measured in this run)")
                                  if traffic is not None else (PMC_FILE + " is missing or was collected from other kernel sources (hash mismatch): no counter figure is quoted"),
                "kernel_ms": round(agg[dom], 4), "algorithmic_bytes": int(dom_bytes),
                # the kernel the contract prices against HBM is VALU-issue bound in practice: SQ counters of the committed PMC pass
                "valu_busy_frac_pmc": valu_busy,
                "frame": {"A_fwd_bytes": int(A_fwd), "A_bwd_bytes": int(A_bwd), "A_bytes": int(A),
                          "achieved_GBps_walltime": round(A / (rast_ms * 1e-3) / 1e9, 1),
                          "frac_walltime": round(A / (rast_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                          "achieved_GBps_kernels": round(A / (kernel_ms * 1e-3) / 1e9, 1) if kernel_ms > 0 else None,
                          # the frame's REAL HBM traffic: counter bytes of every launch of one forward + backward, summed
                          "hbm_traffic_bytes_pmc": None if frame_pmc is None else frame_pmc["hbm_bytes_per_frame"],
                          "frac_traffic": None if frame_pmc is None else round(frame_pmc["hbm_bytes_per_frame"] / (rast_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                          "launches_pmc": None if frame_pmc is None else frame_pmc["launches_per_frame"],
                          "valu_insts_pmc": None if frame_pmc is None else frame_pmc["SQ_INSTS_VALU_per_frame"],
                          "issue_floor_ms": None if frame_pmc is None else round(frame_pmc["SQ_INSTS_VALU_per_frame"] * 4.0 / (SIMDS * CLOCK_HZ) * 1e3, 4)},
                "stage_ms": {k: round(v, 4) for k, v in agg.items()},
                "pair_evals_per_s_upper": round((1 if args.forward_only else 2) * 256.0 * R / (rast_ms * 1e-3), 0)}

    if rank == 0:
        headline = args.config == "cfg3" and args.points is None and not train_mode and not args.forward_only
        line = {
            "metric": ("fwd+bwd ms/frame @1M Gaussians 1352x1014; achieved HBM GB/s vs peak" if headline or (args.config == "cfg3" and args.points is None and train_mode)
                       else f"forward-only ms/frame ({cfg.name})" if args.forward_only else f"fwd+bwd ms/frame ({cfg.name})"),
            "value": round(ms_per_step / world, 4), "unit": "ms/frame", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": False, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "frames_per_s": round(1e3 * world / ms_per_step, 2),
            "step_ms": percentiles(per_step),
            "config": {"workload": cfg.name if args.points is None else f"{cfg.name} [P overridden to {P}]", "P": P, "V": V, "R": R,
                       "HW": H * W, "tiles": T, "sh_degree": 3, "frames_per_step": world, "step": step_what, "parallelism": parallelism},
            "roofline": roof,
        }
        if multi is not None:
            line["multi_gpu"] = multi
        elif train_mode and secondary:
            line["train_core"] = dict(optimizer=args.optimizer, **secondary)
        if world == 1 and not args.no_model_step and not train_mode:
            try:
                line["model_step"] = model_step_timing(args.config, dev, grads, points=args.points)
            except Exception as e:
                line["model_step"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                # the scalar C oracle on a bounded sample of this run's own workload ...
                line["cpu_baseline"] = cpu_baseline(args.config, min(args.cpu_sample, P), my_stamps[0])
                # ... SURVEY.md 8(d)(ii): on BASELINE config 2 at its full size (finishes in seconds), and config 1 on the pure-PyTorch
                # rasterizer (all host cores)
                line["cpu_oracle_cfg2_full"] = cpu_baseline("cfg2", CONFIGS["cfg2"].P, 0)
                line["cpu_torch_baseline"] = cpu_torch_baseline()
            except Exception as e:   # the checker must never take the measurement down
                line["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
