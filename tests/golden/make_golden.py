"""Generates tests/golden/*.npz|json by importing the REFERENCE's Python (read-only at /root/reference)
in the build container.  The reference never travels to the GPU box; only these small data fixtures do.

What is captured:
  model_getters.npz  CGaussianModel per-frame getters (scene/c_gaussian_model.py:170-215,330-375) + autograd grads
  model_getters_1k.npz  the same for the seeded 1k static / 1k dynamic case (parameters regenerated from tests/golden/param_gen.py)
  training_args.json the 15 optimizer groups / learning rates CGaussianModel.training_setup builds from the default OptimizationParams,
                     near / far defaults (arguments/__init__.py)
  sh_eval.npz        utils/sh_utils.eval_sh outputs (independent check of the SH->RGB restatement)
  cameras.npz        getWorld2View2 / getProjectionMatrix / getProjectionMatrixCV / Cameravideo matrix block
  loss_l1_ssim.npz   utils/loss_utils.l1_loss + ssim (train.py:144-151 combination) outputs + autograd grads
  radam.npz          torch.optim.RAdam (c_gaussian_model.py:449 call, per-group lr) parameter/state trajectories
  marshalling.json   positional-argument order the reference wrapper hands to _C (DGR/py:64-89, :120-149)

Run:  python tests/golden/make_golden.py      (needs /root/reference; third-party deps are stubbed)
"""
import json
import math
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub_modules():
    for name in ("plyfile", "cv2", "kornia", "natsort", "simple_knn", "simple_knn._C"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            sys.modules[name] = m
    sys.modules["plyfile"].PlyData = object
    sys.modules["plyfile"].PlyElement = object
    sys.modules["simple_knn._C"].distCUDA2 = lambda *a, **k: None
    sys.modules["kornia"].create_meshgrid = lambda *a, **k: None
    sys.modules["natsort"].natsorted = sorted
    torch.Tensor.cuda = lambda self, *a, **k: self          # no GPU in the build container
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "submodules", "diff_gaussian_rasterization_df"))


def golden_model_getters():
    from scene.c_gaussian_model import CGaussianModel
    out = {}
    for tag, (Ns, Nd) in {"small": (6, 7), "staticonly": (5, 0)}.items():
        g = torch.Generator().manual_seed(1234 + Ns)
        pc = CGaussianModel(3, 300, 10, 2, interp_type="cube", rot_interp_type="slerp")
        K = math.ceil((300 + pc.time_shift + 2 * pc.time_pad + 1) / pc.interval) + 3
        assert pc.time_shift == 12 and K == 35
        R = lambda *s: torch.randn(*s, generator=g)
        P = dict(
            _xyz=R(Ns, 3), _xyz_disp=0.1 * R(Ns, 3), _rotation=R(Ns, 4), _opacity=R(Ns, 1), _scaling=0.3 * R(Ns, 3) - 2,
            _features_dc=R(Ns, 1, 3), _features_rest=0.2 * R(Ns, 15, 3),
            _xyz_motion=torch.cumsum(0.2 * R(Nd, K, 3), 1), _rotation_motion=R(Nd, K, 4), _opacity_motion=R(Nd, 1),
            _opacity_duration_center=torch.sort(2 + torch.rand(Nd, 2, 1, generator=g) * (K - 5), dim=1)[0],
            _opacity_duration_var=R(Nd, 2, 1), _scaling_motion=0.3 * R(Nd, 3) - 2,
            _features_dc_motion=R(Nd, 1, 3), _features_rest_motion=0.2 * R(Nd, 15, 3))
        if Nd == 0:
            for k in list(P):
                if "motion" in k or "duration" in k:
                    P[k] = torch.empty(0)
        for k, v in P.items():
            v = v.clone().requires_grad_(v.numel() > 0)
            setattr(pc, k, v)
            out[f"{tag}/param/{k}"] = v.detach().numpy()
        pc.active_sh_degree = 3
        N = Ns + Nd
        wts = dict(xyz=R(N, 3), rot=R(N, 4), opa=R(N, 1), scl=R(N, 3), fea=R(N, 16, 3))
        for k, v in wts.items():
            out[f"{tag}/weight/{k}"] = v.numpy()
        for t in (0, 7, 137, 290, 299):
            vals = dict(xyz=pc.get_xyz_at_t(t), rot=pc.get_rotation_at_t(t), opa=pc.get_opacity_at_t(t),
                        scl=pc.get_scaling(), fea=pc.get_features())
            loss = sum((vals[k] * wts[k]).sum() for k in vals)
            names = [k for k, v in P.items() if v.numel() > 0]
            grads = torch.autograd.grad(loss, [getattr(pc, k) for k in names], allow_unused=True)
            for k, v in vals.items():
                out[f"{tag}/t{t}/{k}"] = v.detach().numpy()
            for k, gr in zip(names, grads):
                out[f"{tag}/t{t}/grad/{k}"] = (torch.zeros_like(getattr(pc, k)) if gr is None else gr).numpy()
    np.savez_compressed(os.path.join(OUT, "model_getters.npz"), **out)
    return len(out)


def golden_model_getters_1k():
    """The 1k static / 1k dynamic case of SURVEY 8(a): parameters regenerated from a seed (tests/golden/param_gen.py), so the fixture
    holds the reference's OUTPUTS and gradients only; keyframe gradients are stored as their non-zero time slices."""
    from scene.c_gaussian_model import CGaussianModel
    sys.path.insert(0, OUT)
    from param_gen import seeded_params, checksum
    Ns = Nd = 1000
    pc = CGaussianModel(3, 300, 10, 2, interp_type="cube", rot_interp_type="slerp")
    K = math.ceil((300 + pc.time_shift + 2 * pc.time_pad + 1) / pc.interval) + 3
    P, Wt = seeded_params(Ns, Nd, K, seed=4321)
    out = {"checksum": np.array(checksum(P)), "K": K, "seed": 4321}
    for k, v in P.items():
        setattr(pc, k, v.clone().requires_grad_(True))
    pc.active_sh_degree = 3
    for t in (0, 137, 299):
        vals = dict(xyz=pc.get_xyz_at_t(t), rot=pc.get_rotation_at_t(t), opa=pc.get_opacity_at_t(t), scl=pc.get_scaling())
        loss = sum((vals[k] * Wt[k]).sum() for k in vals)
        names = [k for k in P if "features" not in k]
        grads = torch.autograd.grad(loss, [getattr(pc, k) for k in names], allow_unused=True)
        for k, v in vals.items():
            out[f"t{t}/{k}"] = v.detach().numpy()
        for k, gr in zip(names, grads):
            gr = torch.zeros_like(getattr(pc, k)) if gr is None else gr
            if k in ("_xyz_motion", "_rotation_motion"):
                nz = (gr.abs().sum(dim=(0, 2)) > 0).nonzero().flatten()
                out[f"t{t}/grad_slices/{k}"] = nz.numpy()
                out[f"t{t}/grad/{k}"] = gr[:, nz].numpy()
            else:
                out[f"t{t}/grad/{k}"] = gr.numpy()
    fea = pc.get_features()
    assert torch.equal(fea, torch.cat([torch.cat([P["_features_dc"], P["_features_rest"]], 1), torch.cat([P["_features_dc_motion"], P["_features_rest_motion"]], 1)], 0))
    np.savez_compressed(os.path.join(OUT, "model_getters_1k.npz"), **out)
    return len(out)


def golden_training_args():
    """The 15 optimizer groups exactly as CGaussianModel.training_setup builds them (c_gaussian_model.py:430-447) from the reference's
    default OptimizationParams (arguments/__init__.py:93-110), for two values of spatial_lr_scale; near / far defaults of ModelParams."""
    import argparse
    from arguments import OptimizationParams, ModelParams
    from scene.c_gaussian_model import CGaussianModel
    parser = argparse.ArgumentParser()
    mp, op = ModelParams(parser), OptimizationParams(parser)
    real = {n: getattr(torch, n) for n in ("zeros", "ones")}
    strip = lambda f: (lambda *a, **k: f(*a, **{kk: vv for kk, vv in k.items() if kk != "device"}))
    info = {"near": mp.near, "far": mp.far, "lambda_dssim": op.lambda_dssim, "groups": {}}
    try:
        torch.zeros, torch.ones = strip(real["zeros"]), strip(real["ones"])
        for scale in (1.0, 3.7):
            pc = CGaussianModel(3, 300, 10, 2, interp_type="cube", rot_interp_type="slerp")
            sys.path.insert(0, OUT)
            from param_gen import seeded_params
            P, _ = seeded_params(3, 2, 35, seed=1)
            for k, v in P.items():
                setattr(pc, k, torch.nn.Parameter(v))
            pc.spatial_lr_scale = scale
            pc.training_setup(op)
            info["groups"][str(scale)] = {g["name"]: g["lr"] for g in pc.optimizer.param_groups}
            info["optimizer"] = type(pc.optimizer).__name__
            d = pc.optimizer.defaults
            info["optimizer_defaults"] = {"betas": list(d["betas"]), "eps": d["eps"], "weight_decay": d["weight_decay"]}
    finally:
        torch.zeros, torch.ones = real["zeros"], real["ones"]
    with open(os.path.join(OUT, "training_args.json"), "w") as f:
        json.dump(info, f, indent=1)
    return len(info["groups"]["1.0"])


def golden_sh():
    from utils.sh_utils import eval_sh
    g = torch.Generator().manual_seed(7)
    N = 64
    sh = torch.randn(N, 3, 16, generator=g)            # eval_sh layout [..., C, (deg+1)^2]
    dirs = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)
    out = dict(sh=sh.numpy(), dirs=dirs.numpy())
    for deg in range(4):
        out[f"rgb_deg{deg}"] = eval_sh(deg, sh[..., : (deg + 1) ** 2], dirs).numpy()
    np.savez_compressed(os.path.join(OUT, "sh_eval.npz"), **out)
    return len(out)


def golden_cameras():
    from utils.graphics_utils import getWorld2View2, getProjectionMatrix, getProjectionMatrixCV
    out = {}
    poses = {
        "identity": (np.eye(3), np.zeros(3), 0.0, 0.0),
        "posed": (np.array([[0.9362934, -0.2896295, 0.1986693], [0.3129918, 0.9447025, -0.0978434], [-0.1593451, 0.1537920, 0.9751703]]),
                  np.array([0.3, -0.2, 1.5]), 0.02, -0.01),
    }
    for name, (R, T, cxr, cyr) in poses.items():
        FoVx, FoVy = 2 * math.atan(1352 / (2 * 730.0)), 2 * math.atan(1014 / (2 * 730.0))
        wvt = torch.tensor(getWorld2View2(R, T, np.array([0.0, 0.0, 0.0]), 1.0), dtype=torch.float32).transpose(0, 1)
        if cyr != 0.0:
            proj = getProjectionMatrixCV(znear=0.01, zfar=100.0, fovX=FoVx, fovY=FoVy, cx=cxr, cy=cyr).transpose(0, 1)
        else:
            proj = getProjectionMatrix(znear=0.01, zfar=100.0, fovX=FoVx, fovY=FoVy).transpose(0, 1)
        full = wvt.unsqueeze(0).bmm(proj.unsqueeze(0)).squeeze(0)
        out[f"{name}/R"] = R; out[f"{name}/T"] = T; out[f"{name}/cxcy"] = np.array([cxr, cyr])
        out[f"{name}/fov"] = np.array([FoVx, FoVy])
        out[f"{name}/world_view_transform"] = wvt.numpy()
        out[f"{name}/projection_matrix"] = proj.numpy()
        out[f"{name}/full_proj_transform"] = full.numpy()
        out[f"{name}/camera_center"] = wvt.inverse()[3, :3].numpy()
    np.savez_compressed(os.path.join(OUT, "cameras.npz"), **out)
    return len(out)


def golden_loss():
    from utils.loss_utils import l1_loss, ssim
    rng = np.random.default_rng(11)
    out = {}
    cases = {"noise": (3, 37, 53), "smooth": (3, 48, 40), "tiny": (3, 7, 9)}
    for name, (Cn, H, W) in cases.items():
        if name == "smooth":                      # low-variance image: the C2-stabilised branch of the SSIM ratio
            yy, xx = np.meshgrid(np.linspace(0, 1, H), np.linspace(0, 1, W), indexing="ij")
            gt = np.stack([0.5 + 0.3 * np.sin(3 * xx + c) * np.cos(2 * yy) for c in range(Cn)]).astype(np.float32)
            img = (gt + 0.02 * rng.standard_normal(gt.shape)).astype(np.float32)
        else:
            gt = rng.random((Cn, H, W), dtype=np.float32)
            img = np.clip(gt + 0.2 * rng.standard_normal(gt.shape), 0, 1).astype(np.float32)
        for lam in (0.2, 0.5):
            x = torch.tensor(img, requires_grad=True)
            y = torch.tensor(gt)
            Ll1 = l1_loss(x, y)
            loss = (1.0 - lam) * Ll1 + lam * (1.0 - ssim(x, y))              # train.py:144-145
            loss.backward()
            with torch.no_grad():
                l1e = (x - y).abs().mean(dim=0)                              # train.py:149
                sse = ssim(x, y, reduce=False).mean(dim=0)                   # train.py:150
            k = f"{name}/lam{lam}/"
            out[k + "loss"] = np.float32(loss.item()); out[k + "Ll1"] = np.float32(Ll1.item())
            out[k + "grad"] = x.grad.numpy().copy(); out[k + "l1_errors"] = l1e.numpy(); out[k + "ssim_errors"] = sse.numpy()
        from utils.image_utils import psnr
        with torch.no_grad():
            xi, yi = torch.tensor(img), torch.tensor(gt)
            out[name + "/ssim"] = np.float32(ssim(xi.unsqueeze(0), yi.unsqueeze(0)).item())        # render.py:77 call form
            out[name + "/ssim_map"] = ssim(xi, yi, reduce=False).numpy()                              # train.py:150 call form
            out[name + "/psnr"] = psnr(xi.unsqueeze(0), yi.unsqueeze(0)).numpy()                      # render.py:76
            out[name + "/Ll1"] = np.float32(l1_loss(xi, yi).item())
        out[name + "/image"] = img; out[name + "/gt"] = gt
    np.savez_compressed(os.path.join(OUT, "loss_l1_ssim.npz"), **out)
    return len(out)


def golden_radam():
    """The optimizer the reference constructs (c_gaussian_model.py:430-449): RAdam, defaults, per-group lr that changes every
    step (update_learning_rate), one group without gradient on some steps.  torch is the reference's dependency, not its code."""
    g = torch.Generator().manual_seed(5)
    shapes = [(37, 3), (5, 35, 4), (4099,), (3, 1)]
    params = [torch.nn.Parameter(torch.randn(*s, generator=g)) for s in shapes]
    groups = [{"params": [p], "lr": 1e-3 * (i + 1), "name": str(i)} for i, p in enumerate(params)]
    opt = torch.optim.RAdam(groups, lr=0.001)
    out = {"n_steps": 12, "n_params": len(params)}
    for i, p in enumerate(params):
        out[f"p{i}/init"] = p.detach().numpy().copy()
    for it in range(12):
        for i, (p, grp) in enumerate(zip(params, opt.param_groups)):
            grp["lr"] = 1e-3 * (i + 1) * (0.9 ** it)
            skip = (i == 3 and it in (2, 3))                               # a tensor whose grad is None keeps its step count
            p.grad = None if skip else (torch.randn(p.shape, generator=g) * (10.0 ** (i - 2))) * (1.0 if it != 7 else 0.0)
            out[f"p{i}/lr{it}"] = np.float64(grp["lr"])
            out[f"p{i}/has_grad{it}"] = not skip
            if not skip:
                out[f"p{i}/grad{it}"] = p.grad.numpy().copy()
        opt.step()
        for i, p in enumerate(params):
            out[f"p{i}/after{it}"] = p.detach().numpy().copy()
    for i, p in enumerate(params):
        st = opt.state[p]
        out[f"p{i}/exp_avg"] = st["exp_avg"].numpy().copy(); out[f"p{i}/exp_avg_sq"] = st["exp_avg_sq"].numpy().copy()
        out[f"p{i}/step"] = np.float32(st["step"].item())
    np.savez_compressed(os.path.join(OUT, "radam.npz"), **out)
    return len(out)


def golden_marshalling():
    """Drive the reference's autograd wrapper with a recording stub `_C` and store which input lands in
    which positional slot (by tagging each tensor with a unique first element)."""
    rec = {}
    stub = types.ModuleType("diff_gaussian_rasterization_df._C")

    def rasterize_gaussians(*args):
        rec["fwd"] = args
        P, H, W = args[1].shape[0], args[15], args[16]
        z = torch.zeros
        return (17, z(3, H, W), z(P, dtype=torch.int32), z(11, dtype=torch.uint8), z(12, dtype=torch.uint8),
                z(13, dtype=torch.uint8), z(1, H, W), z(1, H, W), z(3, H, W), z(1, H, W, dtype=torch.int32))

    def rasterize_gaussians_backward(*args):
        rec["bwd"] = args
        P = args[1].shape[0]
        M = args[22].shape[1]
        z = torch.zeros
        return tuple(torch.full(s, float(i + 1)) for i, s in enumerate(
            [(P, 3), (P, 3), (P, 1), (P, 3), (P, 6), (P, M, 3), (P, 3), (P, 4), (P, 3)]))

    stub.rasterize_gaussians = rasterize_gaussians
    stub.rasterize_gaussians_backward = rasterize_gaussians_backward
    stub.mark_visible = lambda *a: rec.setdefault("mark", a)
    pkg = types.ModuleType("diff_gaussian_rasterization_df")
    pkg.__path__ = [os.path.join(REF, "submodules", "diff_gaussian_rasterization_df", "diff_gaussian_rasterization_df")]
    sys.modules["diff_gaussian_rasterization_df"] = pkg
    sys.modules["diff_gaussian_rasterization_df._C"] = stub
    pkg._C = stub
    src = open(os.path.join(pkg.__path__[0], "__init__.py")).read()
    exec(compile(src, "ref_dgr_init", "exec"), pkg.__dict__)

    P, H, W = 4, 8, 12
    tag = lambda v, *s: torch.full(s, float(v))
    settings = pkg.GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=0.5, tanfovy=0.25, kernel_size=0.1, subpixel_offset=tag(101, H, W, 2),
        bg=tag(102, 3), scale_modifier=1.5, viewmatrix=tag(103, 4, 4), projmatrix=tag(104, 4, 4), sh_degree=2,
        campos=tag(105, 3), prefiltered=False, min_depth=0.25, max_depth=77.0, debug=False)
    ins = dict(means3D=tag(1, P, 3), means2D=tag(2, P, 3), dir3D=tag(3, P, 3), opacities=tag(4, P, 1), shs=tag(5, P, 16, 3),
               scales=tag(6, P, 3), rotations=tag(7, P, 4))
    for v in ins.values():
        v.requires_grad_(True)
    outs = pkg.GaussianRasterizer(settings)(**ins)
    sum(o.float().sum() for o in outs if o.dtype.is_floating_point).backward()

    def describe(a):
        if isinstance(a, torch.Tensor):
            return {"tensor": list(a.shape), "tag": (float(a.reshape(-1)[0]) if a.numel() else None), "dtype": str(a.dtype)}
        return {"value": a, "type": type(a).__name__}

    info = dict(
        settings_fields=list(pkg.GaussianRasterizationSettings._fields),
        fwd_args=[describe(a) for a in rec["fwd"]], bwd_args=[describe(a) for a in rec["bwd"]],
        n_outputs=len(outs), output_shapes=[list(o.shape) for o in outs],
        input_grad_tags={k: (None if v.grad is None else float(v.grad.reshape(-1)[0])) for k, v in ins.items()})
    with open(os.path.join(OUT, "marshalling.json"), "w") as f:
        json.dump(info, f, indent=1)
    return len(info)


if __name__ == "__main__":
    _stub_modules()
    torch.set_default_dtype(torch.float32)
    print("model_getters:", golden_model_getters())
    print("model_getters_1k:", golden_model_getters_1k())
    print("training_args:", golden_training_args())
    print("sh_eval:", golden_sh())
    print("cameras:", golden_cameras())
    print("loss:", golden_loss())
    print("radam:", golden_radam())
    print("marshalling:", golden_marshalling())
