#!/usr/bin/env python
"""bench.py -- fwd+bwd ms/frame of the MI355X-native rasterizer on BASELINE.json's headline workload.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg3]

One "step" = GaussianRasterizer.forward + .backward (boundary-to-boundary: the reference's
_RasterizeGaussians autograd surface; model getters and the loss are outside) on one
synthetic frame per rank: 1.0M static+dynamic Gaussians (K=35 keyframes), 1352x1014, SH degree 3, inputs
resident in HBM.  N > 1: one process per GPU (torchrun contract), frames sharded round-robin over ranks,
rasterizer-input gradients sum-all-reduced over RCCL/xGMI asynchronously (overlapping the next frame).
Rank 0 prints ONE JSON line.  Extra legs outside the timed region: per-stage hipEvent timing (roofline) and,
on rank 0 at N=1, the CPU baseline (the C oracle on a bounded sample + the pure-PyTorch tiny-scene rasterize).
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from ex4dgs_amd import _C, build as hip_build, dist as xdist                      # noqa: E402
from ex4dgs_amd.diff_gaussian_rasterization_df import GaussianRasterizationSettings, rasterize_gaussians   # noqa: E402
from ex4dgs_amd.scene import CONFIGS, make_scene                                   # noqa: E402

HBM_PEAK_GBPS = 8000.0     # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


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
    """Oracle (scalar C port, 1 core) fwd+bwd on a bounded sample of the same generator."""
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
    return {"value": round(ms, 1), "unit": "ms/frame", "cores": 1, "kind": "port",
            "sample": f"{cfg_name} generator at P={sample_P} ({sample_P / CONFIGS[cfg_name].P:.2f}x Gaussians), {W}x{H}, 1 frame fwd+bwd, "
                      f"R={R} instances, oracle/ex4d_oracle.c scalar C, fwd {1e3 * (t1 - t0):.0f} ms + bwd {1e3 * (t3 - t2):.0f} ms",
            "pair_evals_per_s": round(2 * R * 256 / (ms / 1e3), 0)}


def cpu_torch_baseline():
    """BASELINE.json configs[0]: pure-PyTorch CPU rasterize of 256 Gaussians @256x256 on all host cores."""
    from oracle import oracle_torch
    n = os.cpu_count() or 1
    torch.set_num_threads(n)
    model, cam, bg = make_scene("cfg1")
    leaf = lambda x: x.detach().clone().requires_grad_(True)
    xyz, rot, opa, scl, shs = [leaf(x) for x in (model.get_xyz_at_t(0), model.get_rotation_at_t(0), model.get_opacity_at_t(0), model.get_scaling(), model.get_features())]
    H, W = cam.image_height, cam.image_width
    kw = dict(bg=bg, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, campos=cam.camera_center,
              image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), kernel_size=0.1,
              sh_degree=3, min_depth=4.0, max_depth=300.0)
    t0 = time.time()
    out = oracle_torch.rasterize(xyz, torch.zeros_like(xyz), opa, shs, scl, rot, **kw)
    t1 = time.time()
    out["color"].sum().backward()
    t2 = time.time()
    return {"value": round(1e3 * (t2 - t0), 1), "unit": "ms/frame", "cores": n, "kind": "port",
            "sample": f"cfg1: 256 static Gaussians, 256x256, oracle/oracle_torch.py dense pixels x Gaussians, fwd {1e3 * (t1 - t0):.0f} ms + autograd bwd {1e3 * (t2 - t1):.0f} ms"}


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
    out["what"] = ("*_getters_ms_per_frame: getters (xyz/rotation/opacity/scaling/features at t) + rasterizer fwd+bwd to the model "
                   "parameters; *_train_iter_ms: the same plus the L1+SSIM loss and error maps of train.py:144-151 "
                   "(torch = the reference's op composition, fused = ex4d_attributes + ex4d_l1_ssim); *_full_iter_with_radam_ms: plus "
                   "optimizer.step() + zero_grad (torch.optim.RAdam vs FusedRAdam); 1 GPU, same HIP rasterizer in both")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="cfg3", choices=sorted(CONFIGS))
    ap.add_argument("--points", type=int, default=None, help="override the Gaussian count (parity/debug only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-allreduce", action="store_true")
    ap.add_argument("--forward-only", action="store_true", help="render only (BASELINE config 5 is quoted as forward-only FPS); not the headline metric")
    ap.add_argument("--no-model-step", action="store_true", help="skip the training-iteration timings (profiling runs)")
    ap.add_argument("--cpu-sample", type=int, default=250_000)
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL); gloo for debugging")
    ap.add_argument("--share-device", action="store_true", help="debug: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--bwd-variant", type=int, default=None, help="tuning: compositing-backward kernel (include/ex4d_rasterizer.h: ex4d_set_option)")
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the rasterizer has no CPU fallback")
    if args.share_device:
        os.environ["LOCAL_RANK"] = "0"
    rank, world, local = xdist.init_from_env(backend=args.backend)
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if rank == 0:
        hip_build.build()            # one builder; the other ranks wait and then only dlopen
    if world > 1:
        torch.distributed.barrier()
    _C.load()
    if args.bwd_variant is not None:
        _C.set_option("composite_bwd_variant", args.bwd_variant)

    cfg = CONFIGS[args.config]
    model, cam, bg = make_scene(args.config, P=args.points)
    cam = cam.to(dev); bg = bg.to(dev)
    H, W = cam.image_height, cam.image_width
    # three resident frames per rank (timestamps of SURVEY.md 8d), sharded round-robin over ranks
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
    g = torch.Generator().manual_seed(1000 + rank)
    grads = [torch.randn(3, H, W, generator=g).to(dev), (0.1 * torch.randn(1, H, W, generator=g)).to(dev),
             torch.rand(3, H, W, generator=g).to(dev), torch.zeros(1, H, W, device=dev)]
    means2D = [torch.zeros(P, 3, device=dev, requires_grad=True) for _ in frames]
    dir3D = [torch.zeros(P, 3, device=dev, requires_grad=True) for _ in frames]

    buckets = None
    if world > 1 and not args.no_allreduce and not args.forward_only:
        buckets = xdist.GradBuckets([x.shape for x in frames[0]], device=dev)
    info = {}

    def step(i):
        f = i % len(frames)
        xyz, shs, opa, scl, rot = frames[f]
        for t in frames[f] + [means2D[f], dir3D[f]]:
            t.grad = None
        if args.forward_only:
            with torch.no_grad():
                color, radii, depth, flow, acc, idx = rasterize_gaussians(xyz, means2D[f], dir3D[f], shs, empty, opa, scl, rot, empty, settings)
            info["radii"] = radii
            return color
        color, radii, depth, flow, acc, idx = rasterize_gaussians(xyz, means2D[f], dir3D[f], shs, empty, opa, scl, rot, empty, settings)
        torch.autograd.backward([color, depth, flow, acc], [grads[0], grads[1], grads[2], grads[3]])
        if buckets is not None:
            buckets.launch([t.grad for t in frames[f]])      # waits for the previous frame's exchange first
        info["radii"] = radii
        return color

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    sync_all()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    if buckets is not None:
        buckets.wait()
    sync_all()
    t1 = time.perf_counter()
    ms_per_step = xdist.allreduce_max_scalar(1e3 * (t1 - t0) / max(args.steps, 1), device=dev)

    # ---- per-stage hipEvent timing (outside the timed region), scene statistics ------------------
    _C.profile_enable(True)
    agg = {}
    nprof = 12
    R = 0
    for i in range(nprof):
        f = i % len(frames)
        xyz, shs, opa, scl, rot = frames[f]
        for t in frames[f] + [means2D[f], dir3D[f]]:
            t.grad = None
        outs = rasterize_gaussians(xyz, means2D[f], dir3D[f], shs, empty, opa, scl, rot, empty, settings)
        torch.autograd.backward([outs[0], outs[2], outs[3], outs[4]], grads)
        torch.cuda.synchronize()
        for which in (0, 1):
            for name, ms in _C.profile_read(which):
                agg[name] = agg.get(name, 0.0) + ms / nprof
    _C.profile_enable(False)
    radii = info["radii"]
    V = int((radii > 0).sum().item())
    # instance count of the last profiled frame via the C ABI return value
    last = _C.rasterize_gaussians(settings.bg, frames[0][0].detach(), empty, empty, frames[0][2].detach(), frames[0][3].detach(),
                                  frames[0][4].detach(), 1.0, empty, settings.viewmatrix, settings.projmatrix, settings.tanfovx,
                                  settings.tanfovy, 0.1, settings.subpixel_offset, H, W, frames[0][1].detach(), 3, settings.campos,
                                  False, settings.min_depth, settings.max_depth, False)
    R = int(last[0])
    V = int((last[2] > 0).sum().item())
    T = ((W + 15) // 16) * ((H + 15) // 16)
    stage_bytes, A_fwd, A_bwd = algorithmic_bytes(P, V, R, H * W, T)
    A = A_fwd + A_bwd
    kernel_ms = sum(agg.values())
    dom = max(agg, key=agg.get) if agg else None
    dom_key = {"depth_sort": "sort", "tile_sort": "sort"}.get(dom, dom)
    dom_bytes = stage_bytes.get(dom_key)
    if dom == "tile_sort" or dom == "depth_sort":
        dom_bytes = stage_bytes["sort"]
    roof = None
    traffic = None
    valu_busy = None
    try:   # HBM bytes per launch from the PMC passes committed under profiles/ (collected with tools/pmc.sh, not in this run)
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as fh:
            pmc = json.load(fh)
        if args.config == "cfg3" and args.points is None and dom in pmc["kernels"]:
            traffic = pmc["kernels"][dom]["hbm_bytes_per_launch"]
            valu_busy = pmc["kernels"][dom].get("valu_busy_frac")
    except Exception:
        traffic = None
    if dom is not None and dom_bytes:
        achieved = dom_bytes / (agg[dom] * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                "kernel_ms": round(agg[dom], 4), "algorithmic_bytes": int(dom_bytes),
                # the kernel the contract prices against HBM is VALU-issue bound in practice: SQ counters of the committed PMC pass
                "valu_busy_frac_pmc": valu_busy,
                "frame": {"A_fwd_bytes": int(A_fwd), "A_bwd_bytes": int(A_bwd), "A_bytes": int(A),
                          "achieved_GBps_walltime": round(A / (ms_per_step * 1e-3) / 1e9, 1),
                          "frac_walltime": round(A / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                          "achieved_GBps_kernels": round(A / (kernel_ms * 1e-3) / 1e9, 1) if kernel_ms > 0 else None},
                "stage_ms": {k: round(v, 4) for k, v in agg.items()},
                "pair_evals_per_s_upper": round(2 * 256.0 * R / (ms_per_step * 1e-3), 0)}

    if rank == 0:
        line = {
            "metric": (f"forward-only ms/frame ({cfg.name})" if args.forward_only else
                       "fwd+bwd ms/frame @1M Gaussians 1352x1014; achieved HBM GB/s vs peak" if args.config == "cfg3" and args.points is None
                       else f"fwd+bwd ms/frame ({cfg.name})"),
            "value": round(ms_per_step / world, 4), "unit": "ms/frame", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": False, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "frames_per_s": round(1e3 * world / ms_per_step, 2),
            "config": {"workload": cfg.name if args.points is None else f"{cfg.name} [P overridden to {P}]", "P": P, "V": V, "R": R,
                       "HW": H * W, "tiles": T, "sh_degree": 3, "frames_per_step": world,
                       "parallelism": f"frame-sharded x{world}" + ("" if world == 1 else (" + async RCCL grad all-reduce" if buckets is not None else " (no collective)"))},
            "roofline": roof,
        }
        if world == 1 and not args.no_model_step:
            try:
                line["model_step"] = model_step_timing(args.config, dev, grads, points=args.points)
            except Exception as e:
                line["model_step"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(args.config, min(args.cpu_sample, P), my_stamps[0])
                line["cpu_torch_baseline"] = cpu_torch_baseline()
            except Exception as e:   # the checker must never take the measurement down
                line["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
