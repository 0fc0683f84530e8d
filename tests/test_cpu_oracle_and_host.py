"""CPU-only tests (`-m "not gpu"`): the oracle against its independent cross-checks and the golden vectors
captured from the reference's Python; the host-side logic (scene getters, cameras, argument marshalling,
frame sharding + gradient all-reduce over gloo); and that the C-ABI library builds, loads and exports every
symbol declared in include/ex4d_rasterizer.h (no compute calls without a GPU)."""
import ctypes
import json
import math
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests import helpers as h

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ------------------------------------------------------------------ oracle vs independent formulations
def _small_scene(P=96, size=96):
    cfg = h.SceneConfig("tiny", P, size, size, 60.0, z_lo=4.5, z_hi=25.0, sigma_px_med=5.0, seed=21)
    return h.scene_inputs(cfg)


def test_oracle_forward_matches_pure_torch_rasterizer():
    from oracle import oracle_torch
    ins, st = _small_scene()
    o = h.oracle_forward(ins, st)
    kw = {k: st[k] for k in ("bg", "viewmatrix", "projmatrix", "campos", "image_height", "image_width", "tanfovx", "tanfovy",
                             "kernel_size", "sh_degree", "min_depth", "max_depth")}
    g = oracle_torch.rasterize(ins["means3D"], ins["dir3D"], ins["opacities"], ins["shs"], ins["scales"], ins["rotations"], **kw)
    assert o["num_rendered"] == g["num_rendered"] and o["num_rendered"] > 500
    assert np.array_equal(o["radii"], g["radii"].numpy())
    assert np.array_equal(o["tiles_touched"].astype(np.int64), g["tiles_touched"].numpy().astype(np.int64))
    solid = o["fragile"] > 1e-4
    for k in ("color", "depth", "acc", "flow"):
        err = np.abs(o[k] - g[k].numpy()) / np.maximum(1.0, np.abs(o[k]))
        assert err[:, solid].max() < 1e-5, k
    assert np.array_equal(o["idx"][0][solid], g["idx"].numpy()[0][solid])
    assert np.array_equal(o["n_contrib"].astype(np.int64)[solid], g["n_contrib"].numpy().astype(np.int64)[solid])
    assert np.abs(o["final_T"] - g["final_T"].numpy())[solid].max() < 1e-5


def test_oracle_backward_matches_autograd_on_true_gradient_subset():
    """Colour-path gradients of the restated analytical backward (CR/backward.cu) == torch autograd of the
    independent formulation, for every term that IS a true derivative (SURVEY.md 8a-8 lists the others)."""
    from oracle import oracle, oracle_torch
    ins, st = _small_scene()
    leaf = {k: v.clone().requires_grad_(True) for k, v in ins.items()}
    o = h.oracle_forward(ins, st)
    H, W = st["image_height"], st["image_width"]
    g0 = torch.Generator().manual_seed(5)
    solid = torch.from_numpy(o["fragile"] > 1e-4)
    gc = torch.randn(3, H, W, generator=g0) * solid[None]
    z1, z3 = torch.zeros(1, H, W), torch.zeros(3, H, W)
    b = oracle.backward(o, gc, z1, z3, z1)
    kw = {k: st[k] for k in ("bg", "viewmatrix", "projmatrix", "campos", "image_height", "image_width", "tanfovx", "tanfovy",
                             "kernel_size", "sh_degree", "min_depth", "max_depth")}
    g = oracle_torch.rasterize(leaf["means3D"], leaf["dir3D"], leaf["opacities"], leaf["shs"], leaf["scales"], leaf["rotations"], **kw)
    (g["color"] * gc).sum().backward()

    def close(name, a, bb, rtol=2e-4):
        a = np.asarray(a, np.float64); bb = bb.detach().numpy().astype(np.float64)
        scale = max(1.0, np.abs(bb).max())
        assert np.abs(a - bb).max() <= rtol * scale, (name, np.abs(a - bb).max(), scale)
    close("dL_dcolors", b["dL_dcolors"], g["rgb"].grad)
    close("dL_dsh", b["dL_dsh"], leaf["shs"].grad)
    close("dmean2D.x", b["dL_dmeans2D"][:, 0], g["means2D_pix"].grad[:, 0] * 0.5 * W)
    close("dmean2D.y", b["dL_dmeans2D"][:, 1], g["means2D_pix"].grad[:, 1] * 0.5 * H)
    close("dconic.x", b["dL_dconic"][:, 0], g["conic"].grad[:, 0])
    close("dconic.y", b["dL_dconic"][:, 1], g["conic"].grad[:, 1] * 0.5)      # half the off-diagonal derivative (CR/backward.cu:674, :236)
    close("dconic.w", b["dL_dconic"][:, 3], g["conic"].grad[:, 2])
    close("dopacity", b["dL_dopacity"][:, 0], g["w"].grad)                     # w.r.t. opacity*coef, not rescaled (8a-8 i)
    close("dscales", b["dL_dscales"], leaf["scales"].grad)
    close("drotations", b["dL_drotations"], leaf["rotations"].grad)            # raw-quaternion gradient (8a-8 vi)
    close("dmeans3D", b["dL_dmeans3D"], leaf["means3D"].grad)                  # projection + SH path only (8a-8 ix)
    assert np.abs(b["sum13"][:, 0] - b["dL_dmeans2D"][:, 0]).max() < 1e-3
    # flow channel: dL_ddir = sum alpha*T*dL_dflow/acc (CR/backward.cu:640-642)
    for v in leaf.values():
        v.grad = None
    g = oracle_torch.rasterize(leaf["means3D"], leaf["dir3D"], leaf["opacities"], leaf["shs"], leaf["scales"], leaf["rotations"], **kw)
    gf = torch.randn(3, H, W, generator=g0) * solid[None]
    (g["flow"] * gf).sum().backward()
    b2 = oracle.backward(o, z3, z1, gf, z1)
    close("dL_ddir", b2["dL_ddir"], leaf["dir3D"].grad)


def test_oracle_backward_quirk_terms_match_independent_closed_form():
    """The NON-derivative terms of the reference backward (depth term + `dep > min_depth` gate CR/backward.cu:603-622, dL_dacc
    compounding :649-650,:679, alpha-clamp pass-through :588, flow channel :640-647, pre-scaling by acc :505-535) have no autograd
    counterpart.  oracle/quirk_closed_form.py states them as dense float64 algebra from SURVEY Appendix A.4 -- independently of
    oracle/ex4d_oracle.c's sequential replay -- and all 13 accumulators of the C oracle must agree with it, with every upstream
    gradient non-zero, with a Gaussian clamped at alpha = 0.99 and with Gaussians on both sides of the min_depth gate."""
    from oracle import oracle, oracle_torch, quirk_closed_form
    ins, st = _small_scene()
    # a large, opaque, near Gaussian in the image centre: alpha reaches the 0.99 clamp on many pixels
    ins["means3D"][3] = torch.tensor([0.0, 0.0, 5.0])
    ins["scales"][3] = torch.tensor([0.6, 0.6, 0.6])
    ins["opacities"][3] = 1.6          # (the API takes any value; sigmoid outputs stay below 1, where opacity * coef only grazes the clamp)
    o = h.oracle_forward(ins, st)
    H, W = st["image_height"], st["image_width"]
    kw = {k: st[k] for k in ("bg", "viewmatrix", "projmatrix", "campos", "image_height", "image_width", "tanfovx", "tanfovy",
                             "kernel_size", "sh_degree", "min_depth", "max_depth")}
    g = oracle_torch.rasterize(ins["means3D"], ins["dir3D"], ins["opacities"], ins["shs"], ins["scales"], ins["rotations"], return_dense=True, **kw)
    d = g["dense"]
    assert int(((d["alpha"] == 0.99) & d["use"]).sum()) > 50, "the scene must exercise the alpha clamp"
    solid = torch.from_numpy(o["fragile"] > 1e-4)
    assert bool((torch.from_numpy(o["n_contrib"].astype(np.int64))[solid] == g["n_contrib"][solid]).all())
    g0 = torch.Generator().manual_seed(17)
    gc = torch.randn(3, H, W, generator=g0) * solid[None]
    gd = 0.3 * torch.randn(1, H, W, generator=g0) * solid[None]
    gf = torch.randn(3, H, W, generator=g0) * solid[None]
    ga = torch.randn(1, H, W, generator=g0) * solid[None]
    depths_vis = np.sort(o["depths"][o["radii"] > 0])
    for gate_depth in (st["min_depth"], float(depths_vis[len(depths_vis) // 2])):     # the second run closes the gate for half the Gaussians
        fwd = dict(o)
        fwd["_inputs"] = dict(o["_inputs"], min_depth=gate_depth)
        b = oracle.backward(fwd, gc, gd, gf, ga)
        ref = quirk_closed_form.accumulators(d, final_T=o["final_T"], acc=o["acc"], final_depth=o["depth"], bg=st["bg"].numpy(), grad_color=gc.numpy(),
                                             grad_depth=gd.numpy(), grad_flow=gf.numpy(), grad_acc=ga.numpy(), W=W, H=H, min_depth=gate_depth,
                                             n_visible_total=o["P"])
        names = ["dmean2D.x", "dmean2D.y", "dmean2D.z", "dconic.x", "dconic.y", "dconic.w", "dopacity", "dcolor.r", "dcolor.g", "dcolor.b", "ddir.x", "ddir.y", "ddir.z"]
        for k, name in enumerate(names):
            a, r = b["sum13"][:, k], ref[:, k]
            scale = max(1.0, np.abs(r).max())
            assert np.abs(a - r).max() <= 2e-4 * scale, (name, gate_depth, np.abs(a - r).max(), scale)
            assert np.abs(r).max() > 0, name
        if gate_depth != st["min_depth"]:
            closed = (o["depths"] <= gate_depth) & (o["radii"] > 0)
            assert closed.sum() > 10 and np.all(b["sum13"][closed, 2] == 0.0) and np.abs(b["sum13"][~closed, 2]).max() > 0
    # each quirk on its own is visible: dropping grad_acc / grad_depth changes dL_dopacity
    b_no_acc = oracle.backward(o, gc, gd, gf, torch.zeros(1, H, W))
    b_all = oracle.backward(o, gc, gd, gf, ga)
    assert np.abs(b_all["sum13"][:, 6] - b_no_acc["sum13"][:, 6]).max() > 1e-3


def test_oracle_state_sensitivity_bounds_a_perturbed_forward_state():
    """oracle.backward(state_delta=...) -> state13: the first-order bound on what a difference in the forward's per-pixel state
    (out_depth, out_acc, final_T) moves in each accumulator -- the modelled part of the END-TO-END gradient bar (the 1/acc
    conditioning of dL_ddepth / acc, CR/backward.cu:535-541, :603-613).  Checked directly: run the backward on the state and on
    randomly perturbed states (relative 3e-7 .. 3e-5, the size forward rounding has), the double-precision sums must move by at most
    state13 (+ second order), and for the ill-conditioned depth channel the bound must not be slack by more than ~an order."""
    from oracle import oracle
    ins, st = _small_scene(P=300, size=64)
    ins["opacities"] = ins["opacities"] * 0.012         # faint Gaussians: many pixels with acc ~ 0.01 .. 0.1
    o = h.oracle_forward(ins, st)
    H, W = st["image_height"], st["image_width"]
    assert 0 < np.median(o["acc"][o["acc"] > 0]) < 0.3
    g0 = torch.Generator().manual_seed(3)
    solid = torch.from_numpy(o["fragile"] > 1e-4)
    gc = torch.randn(3, H, W, generator=g0) * solid[None]
    gd = torch.randn(1, H, W, generator=g0) * solid[None]
    gf = torch.randn(3, H, W, generator=g0) * solid[None]
    ga = torch.randn(1, H, W, generator=g0) * solid[None]
    base = oracle.backward(o, gc, gd, gf, ga, stage=False)
    rng = np.random.default_rng(0)
    for rel in (3e-7, 3e-6, 3e-5):
        pert = dict(o)
        for k in ("depth", "acc", "final_T"):
            pert[k] = (o[k].astype(np.float64) * (1.0 + rel * rng.uniform(-1, 1, o[k].shape))).astype(np.float32)
        delta = [np.abs(pert[k].astype(np.float64) - o[k].astype(np.float64)).reshape(H, W) for k in ("depth", "acc", "final_T")]
        b0 = oracle.backward(o, gc, gd, gf, ga, stage=False, state_delta=delta)
        b1 = oracle.backward(pert, gc, gd, gf, ga, stage=False)
        moved = np.abs(b1["sum13"] - base["sum13"])
        bound = b0["state13"]
        assert np.array_equal(b0["sum13"], base["sum13"])            # the instrumentation does not change the result
        # first order + float32 evaluation noise of the two runs (a few ulp of the sum of |terms|) + second order
        slack = 1.05 * bound + 8 * 2.0 ** -24 * base["abs13"] + 1e-9
        assert (moved <= slack).all(), (rel, float((moved / slack).max()), np.unravel_index((moved / slack).argmax(), moved.shape))
        # not vacuous: on the accumulators dL_dalpha feeds, the worst row moves by a sizeable share of its bound
        live = bound[:, 1] > 0
        assert live.sum() > 50 and (moved[live, 1] / bound[live, 1]).max() > 0.05, rel
        # and the amplification is real: the y-gradient moves by far more than `rel` of its magnitude somewhere
        assert (moved[:, 1] / (rel * (np.abs(base["sum13"][:, 1]) + 1e-12)))[live].max() > 3.0


def test_oracle_sh_matches_reference_eval_sh_golden():
    """SH->RGB of the oracle (CR/forward.cu:20-71 restated) vs outputs of the reference's utils/sh_utils.eval_sh."""
    from oracle import oracle
    z = np.load(os.path.join(GOLD, "sh_eval.npz"))
    sh, dirs = z["sh"], z["dirs"]                      # sh [N,3,16] (eval_sh layout), dirs [N,3] unit
    N = sh.shape[0]
    means = (dirs * 10.0).astype(np.float32)           # camera at the origin -> direction = mean / |mean|
    vm = np.eye(4, dtype=np.float32)
    proj = h.make_scene("cfg1")[1].full_proj_transform.numpy()
    for deg in range(4):
        o = oracle.forward(means, None, np.ones(N, np.float32), shs=np.ascontiguousarray(sh.transpose(0, 2, 1)),
                           scales=np.full((N, 3), 0.05, np.float32), rotations=np.tile(np.array([1, 0, 0, 0], np.float32), (N, 1)),
                           bg=np.zeros(3, np.float32), viewmatrix=vm, projmatrix=proj, campos=np.zeros(3, np.float32),
                           image_height=64, image_width=64, tanfovx=5.0, tanfovy=5.0, kernel_size=0.1, sh_degree=deg,
                           min_depth=-1e9, max_depth=1e9)
        vis = o["radii"] > 0
        assert vis.sum() >= N // 4
        ref = np.maximum(z[f"rgb_deg{deg}"] + 0.5, 0.0)
        assert np.abs(o["rgb"][vis] - ref[vis]).max() < 2e-6, deg
        assert np.array_equal(o["clamped"][vis].astype(bool), (z[f"rgb_deg{deg}"] + 0.5 < 0)[vis])


def test_oracle_binning_invariants_and_msb():
    from oracle import oracle
    ins, st = _small_scene(P=300, size=120)
    o = h.oracle_forward(ins, st)
    R = o["num_rendered"]
    assert R == int(o["tiles_touched"].sum()) == int(o["point_offsets"][-1])
    ks = o["keys_sorted"]
    assert np.all(ks[1:] >= ks[:-1])
    assert sorted(o["point_list"].tolist()) == sorted(o["values_unsorted"].tolist())
    T = o["ranges"].shape[0]
    counts = np.bincount((ks >> np.uint64(32)).astype(np.int64), minlength=T)
    assert np.array_equal(o["ranges"][:, 1].astype(np.int64) - o["ranges"][:, 0].astype(np.int64), counts)
    # getHigherMsb (CR/rasterizer_impl.cu:35-50): 13 bits for 1352x1014, 14 for 2048x1088, 9 for 256x256 tiles=256
    assert oracle.get_higher_msb(85 * 64) == 13 and oracle.get_higher_msb(128 * 68) == 14 and oracle.get_higher_msb(256) == 9
    assert np.array_equal(o["acc"][0] == 0, o["idx"][0] == -1)
    assert np.abs(o["acc"][0] + o["final_T"] - 1.0).max() < 2e-5


# ------------------------------------------------------------------ host side vs reference goldens
def test_model_getters_match_reference_golden():
    """ex4dgs_amd.scene.DynamicGaussians == CGaussianModel getters (scene/c_gaussian_model.py:170-215,330-375) + grads."""
    from ex4dgs_amd.scene import DynamicGaussians
    z = np.load(os.path.join(GOLD, "model_getters.npz"))
    for tag in ("small", "staticonly"):
        params = {n: torch.tensor(z[f"{tag}/param/{n}"]).requires_grad_(z[f"{tag}/param/{n}"].size > 0) for n in DynamicGaussians.PARAM_NAMES}
        m = DynamicGaussians(params, duration=300, interval=10, time_pad=2)
        assert m.time_shift == 12 and DynamicGaussians.keyframe_count(300, 10, 2) == 35
        wts = {k: torch.tensor(z[f"{tag}/weight/{k}"]) for k in ("xyz", "rot", "opa", "scl", "fea")}
        for t in (0, 7, 137, 290, 299):
            vals = dict(xyz=m.get_xyz_at_t(t), rot=m.get_rotation_at_t(t), opa=m.get_opacity_at_t(t), scl=m.get_scaling(), fea=m.get_features())
            for k, v in vals.items():
                ref = z[f"{tag}/t{t}/{k}"]
                assert v.shape == ref.shape and np.abs(v.detach().numpy() - ref).max() <= 1e-6, (tag, t, k)
            names = [n for n in DynamicGaussians.PARAM_NAMES if params[n].numel() > 0]
            grads = torch.autograd.grad(sum((vals[k] * wts[k]).sum() for k in vals), [params[n] for n in names], allow_unused=True)
            for n, gr in zip(names, grads):
                ref = z[f"{tag}/t{t}/grad/{n}"]
                got = np.zeros_like(ref) if gr is None else gr.numpy()
                assert np.abs(got - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), (tag, t, n)


def test_model_getters_match_reference_golden_1k_static_1k_dynamic():
    """The 1k / 1k case of SURVEY 8(a): parameters regenerated from the seed the golden was made with (tests/golden/param_gen.py),
    outputs and autograd gradients of the imported CGaussianModel at t in {0, 137, 299}; keyframe gradients are stored as their
    non-zero time slices (4 position / 2 rotation keyframes) and must be exactly zero elsewhere."""
    from ex4dgs_amd.scene import DynamicGaussians
    sys.path.insert(0, GOLD)
    from param_gen import seeded_params, checksum
    z = np.load(os.path.join(GOLD, "model_getters_1k.npz"))
    P, Wt = seeded_params(1000, 1000, int(z["K"]), seed=int(z["seed"]))
    assert np.allclose(checksum(P), z["checksum"], rtol=1e-12), "torch's CPU generator no longer reproduces the stored parameter set"
    params = {n: P[n].clone().requires_grad_(True) for n in DynamicGaussians.PARAM_NAMES}
    m = DynamicGaussians(params, duration=300, interval=10, time_pad=2)
    for t in (0, 137, 299):
        vals = dict(xyz=m.get_xyz_at_t(t), rot=m.get_rotation_at_t(t), opa=m.get_opacity_at_t(t), scl=m.get_scaling())
        for k, v in vals.items():
            ref = z[f"t{t}/{k}"]
            assert v.shape == ref.shape and np.abs(v.detach().numpy() - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max()), (t, k)
        names = [n for n in DynamicGaussians.PARAM_NAMES if "features" not in n]
        grads = torch.autograd.grad(sum((vals[k] * Wt[k]).sum() for k in vals), [params[n] for n in names], allow_unused=True)
        for n, gr in zip(names, grads):
            ref = z[f"t{t}/grad/{n}"]
            got = torch.zeros_like(params[n]) if gr is None else gr
            if n in ("_xyz_motion", "_rotation_motion"):
                sl = torch.from_numpy(z[f"t{t}/grad_slices/{n}"])
                assert len(sl) == (4 if n == "_xyz_motion" else 2)
                mask = torch.ones(got.shape[1], dtype=torch.bool); mask[sl] = False
                assert float(got[:, mask].abs().max()) == 0.0, (t, n)
                got = got[:, sl]
            assert np.abs(got.numpy() - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), (t, n)
    assert torch.equal(m.get_features(), torch.cat([torch.cat([P["_features_dc"], P["_features_rest"]], 1),
                                                    torch.cat([P["_features_dc_motion"], P["_features_rest_motion"]], 1)], 0))


def test_trainer_learning_rates_and_depth_range_match_reference_golden():
    """The optimizer groups the trainers build by default == what the imported CGaussianModel.training_setup built from the reference's
    default OptimizationParams (tests/golden/training_args.json), for two spatial_lr_scale values; near / far defaults == dataset.near /
    dataset.far; the gradient the reference sanitises with nan_to_num is flagged."""
    import inspect
    from ex4dgs_amd import trainer, native_trainer
    gold = json.load(open(os.path.join(GOLD, "training_args.json")))
    assert gold["optimizer"] == "RAdam" and gold["optimizer_defaults"] == {"betas": [0.9, 0.999], "eps": 1e-08, "weight_decay": 0}
    for scale, groups in gold["groups"].items():
        ours = trainer.reference_lrs(float(scale))
        assert set(trainer.REFERENCE_GROUP_NAMES) == set(groups) and len(groups) == 15
        for gname, lr in groups.items():
            assert ours[trainer.REFERENCE_GROUP_NAMES[gname]] == pytest.approx(lr, rel=1e-12), (scale, gname)
    assert trainer.DEFAULT_LRS == trainer.reference_lrs(1.0)
    for fn in (trainer.FrameTrainer.step, native_trainer.NativeTrainer.__init__):
        sig = inspect.signature(fn)
        assert sig.parameters["near"].default == gold["near"] and sig.parameters["far"].default == gold["far"]
    assert inspect.signature(native_trainer.NativeTrainer.__init__).parameters["lambda_dssim"].default == gold["lambda_dssim"]
    assert trainer.NAN_TO_NUM == ("_opacity_duration_var",)


def test_cameras_match_reference_golden():
    from ex4dgs_amd.scene import make_camera
    z = np.load(os.path.join(GOLD, "cameras.npz"))
    for name in ("identity", "posed"):
        cx, cy = z[f"{name}/cxcy"]
        fovx, fovy = z[f"{name}/fov"]
        cam = make_camera(1352, 1014, fovx, fovy, R=z[f"{name}/R"], T=z[f"{name}/T"], znear=0.01, zfar=100.0, cxr=float(cx), cyr=float(cy))
        assert np.abs(cam.world_view_transform.numpy() - z[f"{name}/world_view_transform"]).max() < 1e-6
        assert np.abs(cam.full_proj_transform.numpy() - z[f"{name}/full_proj_transform"]).max() < 1e-5
        assert np.abs(cam.camera_center.numpy() - z[f"{name}/camera_center"]).max() < 1e-5


def test_wrapper_marshalling_matches_reference_golden(monkeypatch):
    """Our autograd surface hands `_C` the same 24 / 30 positional arguments, in the same order, as the reference
    wrapper did when driven with tagged tensors (tests/golden/marshalling.json), and routes gradients identically."""
    import ex4dgs_amd.diff_gaussian_rasterization_df as dgr
    gold = json.load(open(os.path.join(GOLD, "marshalling.json")))
    assert list(dgr.GaussianRasterizationSettings._fields) == gold["settings_fields"]
    rec = {}

    def fwd(*args, **extension):      # prepare_backward: keyword-only extension, the positional arguments are the reference's
        assert set(extension) <= {"prepare_backward", "instance_capacity", "assume_no_flow"} and not extension.get("instance_capacity")
        rec["fwd"] = args
        P, H, W = args[1].shape[0], args[15], args[16]
        z = torch.zeros
        return (17, z(3, H, W), z(P, dtype=torch.int32), z(11, dtype=torch.uint8), z(12, dtype=torch.uint8), z(13, dtype=torch.uint8),
                z(1, H, W), z(1, H, W), z(3, H, W), z(1, H, W, dtype=torch.int32))

    def bwd(*args, **extension):      # need_colors / need_cov3D: keyword-only extension, the 30 positional arguments are the reference's
        assert set(extension) <= {"need_colors", "need_cov3D", "prepared"}
        rec["bwd"] = args
        P, M = args[1].shape[0], args[22].shape[1]
        return tuple(torch.full(s, float(i + 1)) for i, s in enumerate([(P, 3), (P, 3), (P, 1), (P, 3), (P, 6), (P, M, 3), (P, 3), (P, 4), (P, 3)]))
    monkeypatch.setattr(dgr._C, "rasterize_gaussians", fwd)
    monkeypatch.setattr(dgr._C, "rasterize_gaussians_backward", bwd)
    P, H, W = 4, 8, 12
    tag = lambda v, *s: torch.full(s, float(v))
    settings = dgr.GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=0.5, tanfovy=0.25, kernel_size=0.1, subpixel_offset=tag(101, H, W, 2), bg=tag(102, 3),
        scale_modifier=1.5, viewmatrix=tag(103, 4, 4), projmatrix=tag(104, 4, 4), sh_degree=2, campos=tag(105, 3), prefiltered=False,
        min_depth=0.25, max_depth=77.0, debug=False)
    ins = dict(means3D=tag(1, P, 3), means2D=tag(2, P, 3), dir3D=tag(3, P, 3), opacities=tag(4, P, 1), shs=tag(5, P, 16, 3),
               scales=tag(6, P, 3), rotations=tag(7, P, 4))
    for v in ins.values():
        v.requires_grad_(True)
    outs = dgr.GaussianRasterizer(settings)(**ins)
    assert len(outs) == gold["n_outputs"] and [list(o.shape) for o in outs] == gold["output_shapes"]
    sum(o.float().sum() for o in outs if o.dtype.is_floating_point).backward()

    def describe(a):
        if isinstance(a, torch.Tensor):
            return {"tensor": list(a.shape), "tag": (float(a.detach().reshape(-1)[0]) if a.numel() else None), "dtype": str(a.dtype)}
        return {"value": a, "type": type(a).__name__}
    assert [describe(a) for a in rec["fwd"]] == gold["fwd_args"]
    got_bwd = [describe(a) for a in rec["bwd"]]
    assert len(got_bwd) == len(gold["bwd_args"]) == 30
    for i, (a, b) in enumerate(zip(got_bwd, gold["bwd_args"])):
        assert a == b, (i, a, b)
    assert {k: (None if v.grad is None else float(v.grad.reshape(-1)[0])) for k, v in ins.items()} == gold["input_grad_tags"]


def test_model_oracle_matches_reference_golden():
    """oracle/model_oracle.py (numpy restatement of the getters + hand-derived backward) vs the reference's outputs and autograd grads."""
    from oracle import model_oracle as mo
    from ex4dgs_amd.scene import DynamicGaussians
    z = np.load(os.path.join(GOLD, "model_getters.npz"))
    shapes = {"_xyz_motion": (0, 35, 3), "_rotation_motion": (0, 35, 4), "_opacity_motion": (0, 1), "_opacity_duration_center": (0, 2, 1),
              "_opacity_duration_var": (0, 2, 1), "_scaling_motion": (0, 3), "_features_dc_motion": (0, 1, 3), "_features_rest_motion": (0, 15, 3)}
    for tag in ("small", "staticonly"):
        p = {}
        for n in DynamicGaussians.PARAM_NAMES:
            a = z[f"{tag}/param/{n}"].astype(np.float32)
            p[n] = a if a.size else np.zeros(shapes[n], np.float32)
        w = dict(means3D=z[f"{tag}/weight/xyz"], rotations=z[f"{tag}/weight/rot"], opacities=z[f"{tag}/weight/opa"],
                 scales=z[f"{tag}/weight/scl"], shs=z[f"{tag}/weight/fea"])
        for t in (0, 7, 137, 290, 299):
            o = mo.forward(p, t)
            for k, a in (("xyz", "means3D"), ("rot", "rotations"), ("opa", "opacities"), ("scl", "scales"), ("fea", "shs")):
                assert np.abs(o[a] - z[f"{tag}/t{t}/{k}"]).max() <= 1e-6, (tag, t, k)
            g = mo.backward(p, t, w)
            for n in DynamicGaussians.PARAM_NAMES:
                key = f"{tag}/t{t}/grad/{n}"
                if key in z.files:
                    assert np.abs(g[n] - z[key]).max() <= 1e-5 * max(1.0, np.abs(z[key]).max()), (tag, t, n)


def test_scene_generator_is_deterministic_and_in_spec():
    from ex4dgs_amd.scene import make_scene, CONFIGS
    a, cam, bg = make_scene("cfg3", P=5000)
    b, _, _ = make_scene("cfg3", P=5000)
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.equal(pa, pb)
    assert a.num_static == 4000 and a.num_dynamic == 1000 and a._xyz_motion.shape == (1000, 35, 3)
    assert cam.image_width == 1352 and cam.image_height == 1014
    assert abs(math.tan(cam.FoVx / 2) - 1352 / (2 * 730.0)) < 1e-6
    assert CONFIGS["cfg5"].width == 2048 and CONFIGS["cfg5"].height == 1088 and CONFIGS["cfg5"].min_depth == 0.01
    xyz = a.get_xyz_at_t(137)
    assert xyz.shape == (5000, 3) and torch.isfinite(xyz).all()
    assert abs(float(a.get_rotation_at_t(299)[4000:].norm(dim=-1).mean()) - 1.0) < 1e-5       # slerp output is normalised


# ------------------------------------------------------------------ 8f-2: L1 + SSIM loss oracle pinned by reference goldens
def test_loss_oracle_matches_reference_goldens():
    from oracle import loss_oracle
    from ex4dgs_amd import loss as loss_mod
    g = np.load(os.path.join(h.ROOT, "tests", "golden", "loss_l1_ssim.npz"))
    assert np.array_equal(loss_mod.gaussian_window(), loss_oracle.window_1d().numpy())
    for name in ("noise", "smooth", "tiny"):
        for lam in (0.2, 0.5):
            k = f"{name}/lam{lam}/"
            # float32 evaluation of the oracle = the reference's arithmetic: tight; float64 = what the GPU is held to
            r32 = loss_oracle.l1_ssim(g[name + "/image"], g[name + "/gt"], lam, dtype=torch.float32)
            r64 = loss_oracle.l1_ssim(g[name + "/image"], g[name + "/gt"], lam)
            assert abs(r32["loss"] - g[k + "loss"]) < 2e-7
            np.testing.assert_allclose(r32["grad"], g[k + "grad"], rtol=0, atol=1e-8 + 1e-5 * np.abs(g[k + "grad"]).max())
            np.testing.assert_allclose(r32["l1_errors"], g[k + "l1_errors"], rtol=0, atol=1e-7)
            np.testing.assert_allclose(r32["ssim_errors"], g[k + "ssim_errors"], rtol=0, atol=2e-5)
            # the reference's own float32 rounding against exact arithmetic: SSIM in flat regions is ill-conditioned (1/C2)
            assert abs(r64["loss"] - g[k + "loss"]) < 2e-6
            np.testing.assert_allclose(r64["grad"], g[k + "grad"], rtol=0, atol=1e-3 * np.abs(g[k + "grad"]).max())
            np.testing.assert_allclose(r64["ssim_errors"], g[k + "ssim_errors"], rtol=0, atol=5e-4)


def test_loss_refuses_cpu_tensors():
    from ex4dgs_amd.loss import l1_ssim_loss
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        l1_ssim_loss(torch.zeros(3, 8, 8), torch.zeros(3, 8, 8), 0.2)


# ------------------------------------------------------------------ 8f-3: RAdam oracle pinned by torch.optim.RAdam trajectories
def radam_golden_replay(step_fn):
    """Replays tests/golden/radam.npz; step_fn(i, it, grad or None, lr) performs one update; returns the golden dict."""
    g = np.load(os.path.join(h.ROOT, "tests", "golden", "radam.npz"))
    for it in range(int(g["n_steps"])):
        for i in range(int(g["n_params"])):
            has = bool(g[f"p{i}/has_grad{it}"])
            step_fn(i, it, g[f"p{i}/grad{it}"] if has else None, float(g[f"p{i}/lr{it}"]))
    return g


def test_radam_oracle_matches_torch_trajectories():
    from oracle import optim_oracle
    g0 = np.load(os.path.join(h.ROOT, "tests", "golden", "radam.npz"))
    n = int(g0["n_params"])
    P = [g0[f"p{i}/init"].copy() for i in range(n)]
    M = [np.zeros_like(p) for p in P]; V = [np.zeros_like(p) for p in P]; steps = [0] * n
    seen_both = set()

    def step(i, it, grad, lr):
        if grad is not None:
            steps[i] += 1
            seen_both.add(bool(optim_oracle.radam_step(P[i], grad, M[i], V[i], steps[i], lr)))
        if i == n - 1:
            for j in range(n):
                ref = g0[f"p{j}/after{it}"]
                np.testing.assert_allclose(P[j], ref, rtol=0, atol=1e-6 * max(1.0, np.abs(ref).max()))
    radam_golden_replay(step)
    assert seen_both == {True, False}                   # the trajectory crosses the rho_t > 5 switch
    for j in range(n):
        assert steps[j] == int(g0[f"p{j}/step"])
        np.testing.assert_allclose(M[j], g0[f"p{j}/exp_avg"], rtol=2e-6, atol=2e-6 * np.abs(M[j]).max())
        np.testing.assert_allclose(V[j], g0[f"p{j}/exp_avg_sq"], rtol=2e-6, atol=2e-6 * np.abs(V[j]).max())


def test_fused_radam_refuses_cpu_and_weight_decay():
    from ex4dgs_amd.optim import FusedRAdam
    p = torch.nn.Parameter(torch.zeros(4))
    with pytest.raises(ValueError):
        FusedRAdam([p], weight_decay=0.1)
    opt = FusedRAdam([{"params": [p], "lr": 0.1, "name": "xyz"}], lr=0.001)
    assert opt.param_groups[0]["betas"] == (0.9, 0.999) and opt.param_groups[0]["eps"] == 1e-8 and opt.param_groups[0]["name"] == "xyz"
    p.grad = torch.ones(4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        opt.step()


# ------------------------------------------------------------------ 8f-4: distCUDA2 oracle, two independent formulations
def knn_point_sets():
    rng = np.random.default_rng(21)
    uniform = rng.random((4000, 3), dtype=np.float32) * 10
    clustered = np.concatenate([c + 0.05 * rng.standard_normal((500, 3)) for c in rng.standard_normal((8, 3)) * 5]).astype(np.float32)
    dup = uniform[:1500].copy(); dup[100:110] = dup[100]; dup[500] = dup[499]          # coincident points count as neighbours
    plane = uniform[:2000].copy(); plane[:, 2] = 1.5                                    # zero extent on one axis
    line = np.zeros((700, 3), np.float32); line[:, 0] = np.sort(rng.random(700)).astype(np.float32)
    return dict(uniform=uniform, clustered=clustered, dup=dup, plane=plane, line=line, same=np.ones((70, 3), np.float32))


def test_knn_oracle_formulations_agree():
    from oracle import knn_oracle
    for name, pts in knn_point_sets().items():
        a, b = knn_oracle.dist2_bruteforce(pts), knn_oracle.dist2_kdtree(pts, k_search=16 if name != "same" else 69)
        assert np.array_equal(a, b), name
    assert np.all(knn_oracle.dist2_bruteforce(knn_point_sets()["same"]) == 0)
    tiny = knn_point_sets()["uniform"]
    assert np.isinf(knn_oracle.dist2_bruteforce(tiny[:1])).all() and np.isinf(knn_oracle.dist2_bruteforce(tiny[:2])).all()
    assert (knn_oracle.dist2_bruteforce(tiny[:3]) > 1e37).all() and np.isfinite(knn_oracle.dist2_bruteforce(tiny[:4])).all()


# ------------------------------------------------------------------ 8f-4: the two-file PLY checkpoint format
def test_ply_checkpoint_format_round_trip_and_layout(tmp_path):
    """save_ply/load_ply against an independent structured-array formulation of what the reference does through plyfile
    (c_gaussian_model.py:514-547: one named float32 record per Gaussian; :560-666: column-by-column reads by name)."""
    from ex4dgs_amd import ply_io
    from ex4dgs_amd.scene import make_scene
    model, _, _ = make_scene("cfg3", P=1200, device="cpu")
    path = str(tmp_path / "point_cloud" / "iteration_30000" / "point_cloud.ply")
    ply_io.save_ply(model, path)
    dyn_path = path.replace("point_cloud.ply", "dynamic_point_cloud.ply")
    assert os.path.exists(dyn_path)

    def parse(pth):                                  # minimal PLY reader: header -> numpy structured dtype -> records
        raw = open(pth, "rb").read()
        head, body = raw.split(b"end_header\n", 1)
        lines = head.decode().split("\n")
        assert lines[0] == "ply" and lines[1] == "format binary_little_endian 1.0"
        n = int(lines[2].split()[2]); assert lines[2].startswith("element vertex ")
        names = [l.split()[2] for l in lines[3:] if l.startswith("property")]
        assert all(l.split()[1] == "float" for l in lines[3:] if l.startswith("property"))
        rec = np.frombuffer(body, dtype=[(nm, "<f4") for nm in names])
        assert rec.shape[0] == n and len(body) == n * 4 * len(names)
        return names, rec
    names, rec = parse(path)
    K = model._xyz_motion.shape[1]
    assert names == ["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(45)] + \
        ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3", "xyz_disp_0", "xyz_disp_1", "xyz_disp_2"]
    g = lambda n: getattr(model, n).numpy()
    assert np.array_equal(rec["y"], g("_xyz")[:, 1]) and np.all(rec["nx"] == 0)
    assert np.array_equal(rec["f_dc_2"], g("_features_dc")[:, 0, 2])
    assert np.array_equal(rec["f_rest_17"], g("_features_rest")[:, 2, 1])          # channel-major: 17 = channel 1 * 15 + coeff 2
    assert np.array_equal(rec["rot_3"], g("_rotation")[:, 3]) and np.array_equal(rec["xyz_disp_1"], g("_xyz_disp")[:, 1])
    dnames, drec = parse(dyn_path)
    assert dnames[:3] == ["motion_xyz_0_0", "motion_xyz_0_1", "motion_xyz_0_2"] and dnames[-1] == f"motion_rot_{K - 1}_3"
    assert len(dnames) == K * 3 + 3 + 45 + 3 + 1 + 2 + 2 + K * 4
    assert np.array_equal(drec["motion_xyz_12_1"], g("_xyz_motion")[:, 12, 1])
    assert np.array_equal(drec["motion_rot_30_2"], g("_rotation_motion")[:, 30, 2])
    assert np.array_equal(drec["motion_opacity_v_1"], g("_opacity_duration_var")[:, 1, 0])
    assert np.array_equal(drec["motion_f_rest_44"], g("_features_rest_motion")[:, 14, 2])
    out = ply_io.load_ply(path, device="cpu")
    for n in model.PARAM_NAMES:
        assert out[n].dtype == torch.float32 and out[n].is_contiguous() and torch.equal(out[n], getattr(model, n)), n
    # properties are found by name: a file with permuted columns (and an extra one) loads identically
    perm = np.random.default_rng(0).permutation(len(dnames))
    pn = [dnames[i] for i in perm] + ["extra"]
    mat = np.stack([drec[nm] for nm in pn[:-1]] + [np.zeros(len(drec), np.float32)], 1)
    ply_io._write(dyn_path, pn, mat)
    out2 = ply_io.load_ply(path, device="cpu")
    for n in model.PARAM_NAMES:
        assert torch.equal(out2[n], out[n]), n
    # static-only model (config 2): zero dynamic rows
    smodel, _, _ = make_scene("cfg2", P=300, device="cpu")
    spath = str(tmp_path / "s" / "point_cloud.ply")
    ply_io.save_ply(smodel, spath)
    sout = ply_io.load_ply(spath, device="cpu")
    assert sout["_xyz_motion"].shape == smodel._xyz_motion.shape and torch.equal(sout["_xyz"], smodel._xyz)


# ------------------------------------------------------------------ INTEGRATION.md route A: the drop-in import names resolve
def test_drop_in_packages_import_under_the_reference_names():
    """With ex4dgs_amd/dropin in front on sys.path, the reference's own import lines (gaussian_renderer/__init__.py:15,
    scene/c_gaussian_model.py:20) resolve to this repository's packages -- and nothing else of the reference is shadowed."""
    code = ("from diff_gaussian_rasterization_df import GaussianRasterizationSettings, GaussianRasterizer, SplitSH\n"
            "from simple_knn._C import distCUDA2\n"
            "import diff_gaussian_rasterization_df as d, simple_knn._C as k\n"
            "print(d.__file__); print(k.__file__); print(len(GaussianRasterizationSettings._fields))\n"
            "import importlib.util as u\n"
            "print([n for n in ('scene', 'utils', 'render', 'train', 'arguments', 'gaussian_renderer', 'loss', 'optim') if u.find_spec(n)])")
    env = dict(os.environ, PYTHONPATH=os.path.join(h.ROOT, "ex4dgs_amd", "dropin"))
    out = subprocess.run([sys.executable, "-c", code], cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    assert lines[0].startswith(os.path.join(h.ROOT, "ex4dgs_amd", "dropin", "diff_gaussian_rasterization_df"))
    assert lines[1].startswith(os.path.join(h.ROOT, "ex4dgs_amd", "dropin", "simple_knn")) and lines[2] == "16"
    assert lines[3] == "[]", lines[3]            # module names of the reference stay free


def test_build_refuses_kernels_with_vector_register_spills(tmp_path):
    """ex4dgs_amd.build parses the compiler's resource remarks of every object: a kernel with VGPR spills fails the build (every kernel
    is laid out for its register budget), and the remarks do not drown the compiler's real diagnostics."""
    from ex4dgs_amd import build
    ok = ("a.hip:3:1: remark: Function Name: k_fine [-Rpass-analysis=kernel-resource-usage]\n"
          "a.hip:3:1: remark:     VGPRs Spill: 0 [-Rpass-analysis=kernel-resource-usage]\n")
    build._no_vgpr_spills("a.hip", ok, str(tmp_path / "a.o"))
    obj = tmp_path / "b.o"
    obj.write_bytes(b"x")
    bad = ok + ("a.hip:9:1: remark: Function Name: k_spills [-Rpass-analysis=kernel-resource-usage]\n"
                "a.hip:9:1: remark:     SGPRs Spill: 4 [-Rpass-analysis=kernel-resource-usage]\n"
                "a.hip:9:1: remark:     VGPRs Spill: 3 [-Rpass-analysis=kernel-resource-usage]\n    9 | {\n      | ^\n")
    with pytest.raises(RuntimeError, match="k_spills"):
        build._no_vgpr_spills("a.hip", bad, str(obj))
    assert not obj.exists()
    text = bad + "a.hip:20:5: warning: unused variable 'x' [-Wunused-variable]\n   20 |     int x;\n      |         ^\n2 remarks generated.\n"
    kept = build._without_remarks(text)
    assert "unused variable" in kept and "int x;" in kept and "remark" not in kept and "Spill" not in kept


def test_build_refuses_64bit_shifts_by_the_last_vector_register(tmp_path):
    """gfx950: a 64-bit shift whose shift amount sits in the last register of the wave's allocation shifts by VGPR0 in waves that share
    their SIMD (tools/dev/micro/topreg_probe.hip, profiles/r04_topreg_probe.txt): this, not its spills, made the 128-register build of
    the per-Gaussian backward wrong.  The build disassembles every object and refuses such a kernel: a kernel that does it on purpose is
    found, the same kernel with the amount one register lower is not, and none of the library's own objects holds one."""
    import subprocess
    from ex4dgs_amd import build
    src = """#include <hip/hip_runtime.h>
__global__ __launch_bounds__(256, 4) void k(unsigned long long *out, unsigned long long need, unsigned x)
{
    unsigned long long d;
    asm volatile("v_mov_b32_e32 REG, %1\\n\\ts_nop 1\\n\\tv_lshrrev_b64 %0, REG, %2" : "=&v"(d) : "v"(x + threadIdx.x), "s"(need) : "REG", "v127");
    out[blockIdx.x * 256 + threadIdx.x] = d;
}
"""
    for reg, expect in (("v127", 1), ("v126", 0)):
        hip, obj = tmp_path / f"k_{reg}.hip", tmp_path / f"k_{reg}.o"
        hip.write_text(src.replace("REG", reg))
        subprocess.check_call([build._hipcc(), "-O3", f"--offload-arch={build.ARCH}", "-c", str(hip), "-o", str(obj)])
        found = build.shift_amount_in_last_vgpr(str(obj))
        assert len(found) == expect, (reg, found)
        if expect:
            assert "v_lshrrev_b64" in found[0][1] and "v127" in found[0][1]
            with pytest.raises(RuntimeError, match="last vector register"):
                build._no_shift_amount_in_last_vgpr("k.hip", str(obj))
            assert not obj.exists()
    build.build()
    for f in build.SOURCES:
        assert build.shift_amount_in_last_vgpr(os.path.join(build.CSRC, f.replace(".hip", ".o"))) == [], f


def test_wait_state_check_of_the_machine_code(tmp_path):
    """ex4dgs_amd.isa_check.wait_state_violations: the software wait states that matter to the library's inline assembly (a DPP read
    2 states behind a VALU write, a transcendental result 1 state, a VALU-written SGPR read as a constant 2 states) are checked on the
    disassembly of every object.  A kernel that breaks each rule inside asm strings is reported rule by rule, the same kernel with the
    pads in place is clean, and the library's own objects -- some 3 000 DPP instructions and the hand-scheduled forward walk among
    them -- have no finding."""
    import subprocess
    from ex4dgs_amd import build, isa_check
    src = """#include <hip/hip_runtime.h>
__global__ void k(float *out, const float *in, unsigned *mask)
{
    float x = in[threadIdx.x], y, z;
    unsigned long long m, w;
    asm volatile("v_mov_b32 %0, %1\\n\\tPAD1v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "=&v"(y) : "v"(x));
    asm volatile("v_exp_f32 %0, %1\\n\\tPAD0v_add_f32 %0, %0, %0" : "=&v"(z) : "v"(x));
    asm volatile("v_cmp_gt_f32 %0, %2, %3\\n\\tPAD1v_lshlrev_b64 %1, 1, %0" : "=&s"(m), "=&v"(w) : "v"(x), "v"(y) : "vcc");
    out[threadIdx.x] = y + z;
    mask[threadIdx.x] = (unsigned)w;
}
"""
    rules = {}
    for name, pad1, pad0 in (("bad", "", ""), ("good", "s_nop 1\\n\\t", "s_nop 0\\n\\t")):
        hip, obj = tmp_path / f"w_{name}.hip", tmp_path / f"w_{name}.o"
        hip.write_text(src.replace("PAD1", pad1).replace("PAD0", pad0))
        subprocess.check_call([build._hipcc(), "-O3", f"--offload-arch={build.ARCH}", "-c", str(hip), "-o", str(obj)])
        rules[name] = sorted(r for _, r, _, _ in isa_check.wait_state_violations(str(obj)))
    assert rules["good"] == [], rules["good"]
    assert len(rules["bad"]) == 3 and "DPP" in rules["bad"][0] and "SGPR" in rules["bad"][1] and "transcendental" in rules["bad"][2], rules["bad"]
    build.build()
    seen = dict(dpp=0, trans=0, valu_sgpr_writes=0, instructions=0)
    for f in build.SOURCES:
        obj = os.path.join(build.CSRC, f.replace(".hip", ".o"))
        assert isa_check.wait_state_violations(obj) == [], f
        for k, v in isa_check.coverage(obj).items():
            seen[k] += v
    assert seen["dpp"] > 2000 and seen["trans"] > 200 and seen["valu_sgpr_writes"] > 1000 and seen["instructions"] > 50000, seen


def test_msd_depth_sort_index_arithmetic_restated_with_numpy():
    """The optional MSD depth sort (ex4d_binning.hip: dls_range_kernel / dls_digit / depth_local_sort_kernel, duplicate_kernel's bucket
    form) restated with numpy, so that a mistake in its arithmetic shows up without a GPU: the top digit cut from the occupied key range
    (invisible key -> last digit), a stable partition by digit, every bucket ordered by sorting the words `low key bits << 12 | arrival
    index`, the tile scan as bucket-local inclusive scans + bucket sums -- against a stable argsort of the keys and a plain exclusive
    scan in sorted order.  Key distributions: spread, a narrow band, heavy ties, no visible Gaussian, one Gaussian."""
    rng = np.random.default_rng(5)
    BINS, IDX_BITS = 1024, 12

    def run(keys, counts, inv_key):
        keys = keys.astype(np.uint64); n = len(keys)
        vis = keys != inv_key
        kmax = int(keys[vis].max()) if vis.any() else 0
        kmin = int(keys[vis].min()) if vis.any() else 0
        shift = 0
        while ((kmax - kmin) >> shift) > BINS - 2:
            shift += 1
        digit = np.where(vis, (keys - kmin) >> shift, BINS - 1).astype(np.int64)
        assert digit[vis].max(initial=0) <= BINS - 2
        order = np.argsort(digit, kind="stable")                  # the partition (stable: ties keep ascending ids)
        starts = np.concatenate([[0], np.cumsum(np.bincount(digit, minlength=BINS))])
        ids = order.copy()
        local_incl = np.zeros(n, np.int64); sums = np.zeros(BINS, np.int64)
        for b in range(BINS - 1):                                 # (the invisible bucket is left as the partition wrote it)
            s0, s1 = starts[b], starts[b + 1]
            if s1 - s0 == 0:
                continue
            assert s1 - s0 <= 1 << IDX_BITS
            kb = keys[order[s0:s1]]
            words = (((kb - kmin) & ((1 << shift) - 1)) << IDX_BITS) | np.arange(s1 - s0, dtype=np.uint64)
            assert int(words.max()) < 1 << 32
            perm = np.argsort(words >> IDX_BITS, kind="stable")   # the LSD passes sort on the key bits, stably
            assert np.array_equal(perm, np.argsort(words))        # ... which is the order of the whole words (the index breaks ties)
            ids[s0:s1] = order[s0:s1][(words[perm] & ((1 << IDX_BITS) - 1)).astype(np.int64)]
            c = counts[ids[s0:s1]]
            local_incl[s0:s1] = np.cumsum(c); sums[b] = c.sum()
        ref = np.argsort(keys, kind="stable")
        assert np.array_equal(ids, ref)
        base = np.concatenate([[0], np.cumsum(sums)])[:-1]
        c_sorted = counts[ids]
        d_sorted = digit[ids]
        off = base[d_sorted] + local_incl - c_sorted              # duplicate_kernel: bucket base + own inclusive value - own count
        ref_off = np.concatenate([[0], np.cumsum(c_sorted)])[:-1]
        live = c_sorted > 0
        assert np.array_equal(off[live], ref_off[live]) and int(sums.sum()) == int(counts.sum())

    inv = (1 << 26) + 5
    for n, lo, hi, ties in ((20000, 1, 1 << 26, 0), (20000, 5_000_000, 5_000_900, 0), (30000, 1000, 40_000_000, 3000), (500, 1, 1 << 26, 0), (1, 7, 8, 0)):
        keys = rng.integers(lo, hi, n).astype(np.uint64)
        if ties:                                                   # thousands of exactly equal keys (one bucket, still inside its LDS capacity)
            keys[rng.permutation(n)[:ties]] = np.uint64((lo + hi) // 2)
        invisible = rng.random(n) < 0.2
        keys[invisible] = inv
        counts = np.where(invisible, 0, rng.integers(1, 40, n)).astype(np.int64)
        run(keys, counts, inv)
    run(np.full(300, inv, np.uint64), np.zeros(300, np.int64), inv)            # nothing visible


def test_machine_code_checks_fail_closed(tmp_path, monkeypatch):
    """the checks must not turn into a silent pass when they cannot be made -- an object the tools cannot read, a kernel
    whose register metadata the parser does not find, LLVM tools that are not where ROCm keeps them: each raises IsaCheckError (the
    build stops); a host-only object has no device code and is simply skipped."""
    import subprocess
    from ex4dgs_amd import build, isa_check
    junk = tmp_path / "junk.o"
    junk.write_bytes(b"not an object file")
    with pytest.raises(isa_check.IsaCheckError):
        isa_check.shift_amount_in_last_vgpr(str(junk))
    with pytest.raises(isa_check.IsaCheckError):
        isa_check.wait_state_violations(str(junk))
    host = tmp_path / "h.c"
    host.write_text("int f(int x) { return x + 1; }\n")
    subprocess.check_call(["gcc", "-c", str(host), "-o", str(tmp_path / "h.o")])
    assert isa_check.shift_amount_in_last_vgpr(str(tmp_path / "h.o")) == []          # host-only: nothing to check
    build.build()
    obj = os.path.join(build.CSRC, "ex4d_binning.o")
    monkeypatch.setattr(isa_check, "kernel_registers", lambda co: {})                 # a metadata layout the parser does not understand
    with pytest.raises(isa_check.IsaCheckError, match="register metadata"):
        isa_check.shift_amount_in_last_vgpr(obj)
    monkeypatch.undo()
    # the tool directory follows the ROCm installation (ROCM_PATH / the resolved hipcc), and a missing one is an error, not a skip
    assert os.path.exists(os.path.join(isa_check._llvm_dir(), "llvm-objdump"))
    monkeypatch.setenv("ROCM_PATH", str(tmp_path))
    monkeypatch.setattr(isa_check.shutil, "which", lambda name: None)
    monkeypatch.setattr(isa_check.os.path, "exists", lambda path: False)
    with pytest.raises(isa_check.IsaCheckError, match="LLVM tools"):
        isa_check._llvm_dir()


# ------------------------------------------------------------------ C ABI library: builds, loads, exports
def test_c_abi_library_builds_loads_and_exports_declared_symbols():
    from ex4dgs_amd import build, _C
    lib = build.build()
    assert os.path.exists(lib)
    hdr = open(os.path.join(h.ROOT, "include", "ex4d_rasterizer.h")).read()
    body = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(ex4d_[a-z0-9_]+)\s*\(", body)) - {"ex4d_alloc_fn"}
    assert declared == set(_C.EXPORTS), declared ^ set(_C.EXPORTS)
    from ex4dgs_amd import attributes
    hdr2 = re.sub(r"/\*.*?\*/", "", open(os.path.join(h.ROOT, "include", "ex4d_attributes.h")).read(), flags=re.S)
    declared2 = set(re.findall(r"\b(ex4d_[a-z0-9_]+)\s*\(", hdr2))
    assert declared2 == set(attributes.EXPORTS), declared2 ^ set(attributes.EXPORTS)
    assert ctypes.sizeof(attributes.Ex4dAttrParams) == 13 * 4
    from ex4dgs_amd import loss as loss_mod
    hdr3 = re.sub(r"/\*.*?\*/", "", open(os.path.join(h.ROOT, "include", "ex4d_loss.h")).read(), flags=re.S)
    declared3 = set(re.findall(r"\b(ex4d_[a-z0-9_]+)\s*\(", hdr3))
    assert declared3 == set(loss_mod.EXPORTS), declared3 ^ set(loss_mod.EXPORTS)
    from ex4dgs_amd import optim as optim_mod
    hdr4 = re.sub(r"/\*.*?\*/", "", open(os.path.join(h.ROOT, "include", "ex4d_optim.h")).read(), flags=re.S)
    declared4 = set(re.findall(r"\b(ex4d_[a-z0-9_]+)\s*\(", hdr4))
    assert declared4 == set(optim_mod.EXPORTS), declared4 ^ set(optim_mod.EXPORTS)
    assert ctypes.sizeof(optim_mod.Ex4dRadamTensor) == 64
    from ex4dgs_amd.simple_knn import _C as knn_mod
    hdr5 = re.sub(r"/\*.*?\*/", "", open(os.path.join(h.ROOT, "include", "ex4d_knn.h")).read(), flags=re.S)
    declared5 = set(re.findall(r"\b(ex4d_[a-z0-9_]+)\s*\(", hdr5))
    assert declared5 == set(knn_mod.EXPORTS), declared5 ^ set(knn_mod.EXPORTS)
    from ex4dgs_amd import native_trainer as nt_mod
    hdr6 = re.sub(r"/\*.*?\*/", "", open(os.path.join(h.ROOT, "include", "ex4d_trainer.h")).read(), flags=re.S)
    declared6 = set(re.findall(r"\b(ex4d_[a-z0-9_]+)\s*\(", hdr6))
    assert declared6 == set(nt_mod.EXPORTS), declared6 ^ set(nt_mod.EXPORTS)
    assert ctypes.sizeof(nt_mod.Ex4dTrainerConfig) == 280 and nt_mod.Ex4dTrainerConfig.optimizer.offset == 272
    declared |= declared2 | declared3 | declared4 | declared5 | declared6
    handle = ctypes.CDLL(lib)
    for name in declared:
        assert hasattr(handle, name), name
    l = _C.load()
    assert l.ex4d_abi_version() == 5 and l.ex4d_target_arch() == b"gfx950"
    # size / layout queries are pure host code
    P = 1000
    lay = _C.GeomLayout(); l.ex4d_geom_layout(P, ctypes.byref(lay))
    assert lay.total == l.ex4d_geom_bytes(P) and lay.cov3D >= 64 * P and lay.cov3D % 256 == 0 and lay.records == 0
    assert l.ex4d_binning_bytes(0, 64, 64) > 0 and l.ex4d_img_bytes(1352, 1014) >= 1352 * 1014 * 8 + 5440 * 8
    assert l.ex4d_backward_scratch_bytes(P) >= P * 64
    assert ctypes.sizeof(_C.Ex4dParams) == 17 * 4
    # library options are host state: the depth sort's default is "auto" (3), values beyond it and unknown names are refused
    assert _C.get_option("depth_sort_msd") == 3 and _C.get_option("depth_sort_hold") == 0 and _C.get_option("depth_sort_trips") == 0
    for v in (0, 1, 2, 3):
        _C.set_option("depth_sort_msd", v)
        assert _C.get_option("depth_sort_msd") == v
    with pytest.raises(RuntimeError):
        _C.set_option("depth_sort_msd", 4)
    with pytest.raises(RuntimeError):
        _C.set_option("depth_sort_hold", 1)              # read-only
    assert _C.get_option("no_such_option") == -1
    # the kernels are gfx950 code objects
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o", f"--input={lib}"], capture_output=True, text=True)
    if out.returncode == 0 and out.stdout.strip():
        assert "gfx950" in out.stdout


def test_product_path_has_no_cpu_fallback_and_never_imports_the_oracle():
    from ex4dgs_amd.diff_gaussian_rasterization_df import GaussianRasterizer, GaussianRasterizationSettings
    ins, st = _small_scene(P=16, size=32)
    s = h.gpu_settings(st, "cpu")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        GaussianRasterizer(s)(ins["means3D"], None, ins["dir3D"], ins["opacities"], shs=ins["shs"], scales=ins["scales"], rotations=ins["rotations"])
    pkg = os.path.join(h.ROOT, "ex4dgs_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dp, f)
                assert "libex4d_oracle" not in src and "/root/reference" not in src.replace("under /root/reference", ""), os.path.join(dp, f)


# ------------------------------------------------------------------ multi-GPU logic on CPU (gloo, world_size 2)
_WORKER = r"""
import os, sys, torch
sys.path.insert(0, sys.argv[1])
from ex4dgs_amd import dist as xd
rank, world, local = xd.init_from_env(backend="gloo")
assert world == 2
views = xd.shard_views(7, rank, world)
assert views == list(range(rank, 7, 2))
shapes = [(50, 3), (50, 16, 3), (50, 1), (50, 3), (50, 4)]
g = torch.Generator().manual_seed(100 + rank)
mine = [torch.randn(*s, generator=g) for s in shapes]
g0, g1 = torch.Generator().manual_seed(100), torch.Generator().manual_seed(101)
want = [torch.randn(*s, generator=g0) + torch.randn(*s, generator=g1) for s in shapes]
b = xd.GradBuckets(shapes, bucket_bytes=2048, inplace_bytes=9000)      # several buckets + one in-place tensor ([50,16,3])
assert len(b.flat) > 1 and b.inplace == [False, True, False, False, False]
b.launch(mine); b.wait()
for a, w in zip(mine, want):
    assert torch.allclose(a, w, atol=1e-6), (a - w).abs().max()
m = xd.allreduce_max_scalar(1.0 + rank)
assert m == 2.0
# densification statistics at their cadence: SUM / MAX / MIN over ranks, several tensors per message
acc = torch.full((40, 1), 1.0 + rank); den = torch.full((40, 1), 2.0 * (rank + 1)); rad = torch.arange(40.) * (1 if rank else -1)
emin = torch.full((40, 1), 5.0 - rank); mx_i = torch.tensor([3 + rank, 9 - rank], dtype=torch.int32)
xd.reduce_densification_stats(sums=[acc, den], maxima=[rad, mx_i], minima=[emin])
assert torch.equal(acc, torch.full((40, 1), 3.0)) and torch.equal(den, torch.full((40, 1), 6.0))
assert torch.equal(rad, torch.arange(40.)) and torch.equal(emin, torch.full((40, 1), 4.0)) and mx_i.tolist() == [4, 9]
torch.distributed.barrier(); torch.distributed.destroy_process_group()
print("OK", rank)
"""


def test_frame_sharding_and_gradient_allreduce_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    port = 29500 + (os.getpid() % 2000)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), h.ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"OK {r}" in o, o


def test_integer_decisions_are_insensitive_to_fma_contraction():
    """The parity oracle is a NO-FMA evaluation (-ffp-contract=off); the reference binary is built by nvcc with contraction on
    (DGR/setup.py:21-29).  The same oracle source with gcc's contraction (-ffp-contract=fast -mfma) must take the same cull decisions
    and (all but a few in 1e5 of) the same radii / tile counts -- the error bar on "bit-exact against the reference binary".
    Full corpus: tools/fma_sensitivity.py -> profiles/archive/r02_fma_sensitivity.json (0 cull flips, 6 radii, 1 tile count in 1.05 M visible)."""
    import runpy
    mod = runpy.run_path(os.path.join(h.ROOT, "tools", "fma_sensitivity.py"), run_name="fma_sensitivity")
    from oracle import oracle
    plain, fma = oracle.lib(), mod["load_fma"]()
    tot = dict(vis=0, cull=0, radii=0, tiles=0, same_list=0, cases=0)
    try:
        for name, cfg, P, t, deg in list(mod["corpus"](12, 0))[:12] + [("cfg2@20k", "cfg2", 20000, 0, 3), ("cfg3@20k", "cfg3", 20000, 137, 3)]:
            ins, st = h.scene_inputs(cfg, P=P, t=t, sh_degree=deg)
            a, b = mod["run"](plain, ins, st), mod["run"](fma, ins, st)
            va, vb = a["radii"] > 0, b["radii"] > 0
            tot["vis"] += int(va.sum()); tot["cull"] += int((va != vb).sum()); tot["radii"] += int((a["radii"] != b["radii"]).sum())
            tot["tiles"] += int((a["tiles_touched"] != b["tiles_touched"]).sum()); tot["cases"] += 1
            tot["same_list"] += int(a["num_rendered"] == b["num_rendered"] and np.array_equal(a["point_list"], b["point_list"]))
    finally:
        oracle._LIB = plain
    assert tot["cull"] == 0, tot
    assert tot["radii"] <= 1e-4 * tot["vis"] + 1 and tot["tiles"] <= 1e-4 * tot["vis"] + 1, tot
    assert tot["same_list"] >= tot["cases"] - 1, tot


def test_msd_tile_sort_index_arithmetic_against_a_stable_sort():
    """The index arithmetic of the MSD-first tile sort (ex4d_binning.hip: pass A packed words, ts_locate_block, the bucket-relative offsets
    of pass B and the tile ranges that fall out of them), restated with numpy and checked against a stable argsort -- including empty
    buckets, buckets larger than one 4096-item block, a last block that is not full, and tile counts that need 9 .. 16 bits."""
    rng = np.random.default_rng(7)
    CH = 4096

    def msd_sort(tiles, ids, tile_bits):
        R = len(tiles)
        low = (tile_bits + 1) // 2
        high = tile_bits - low
        nbuckets = 1 << high
        # ---- pass A: stable partition by the high digit, one packed word per item
        hd = tiles >> low
        order_a = np.argsort(hd, kind="stable")
        packed = ((tiles[order_a] & ((1 << low) - 1)).astype(np.uint64) << np.uint64(32 - low) | ids[order_a].astype(np.uint64)).astype(np.uint32)
        totals = np.bincount(hd, minlength=nbuckets)
        # ---- ts_locate_block: first block / first position of every bucket, then (start, count, bucket) of every block
        nblk = (totals + CH - 1) // CH
        fb = np.concatenate([[0], np.cumsum(nblk)])
        st = np.concatenate([[0], np.cumsum(totals)])
        nb_max = (R + CH - 1) // CH + nbuckets + 1
        blocks = []
        for b in range(nb_max):
            if b >= fb[-1]:
                blocks.append((0, 0, 0)); continue
            h = 0
            step = 128
            fbp = np.concatenate([fb, np.full(257 - len(fb), fb[-1])])        # fb[nbuckets ..] = number of blocks
            while step:
                if h + step <= 255 and fbp[h + step] <= b:
                    h += step
                step >>= 1
            within = (b - fb[h]) * CH
            blocks.append((st[h] + within, min(CH, totals[h] - within), h))
        assert sum(c for _, c, _ in blocks) == R
        # ---- pass B: block histograms by the low digit, row scan over ALL blocks, bucket-relative offsets
        nd = 1 << low
        hist = np.zeros((nd, nb_max), np.int64)
        for b, (s0, c, h) in enumerate(blocks):
            hist[:, b] = np.bincount(packed[s0:s0 + c] >> np.uint32(32 - low), minlength=nd)
        excl = np.concatenate([np.zeros((nd, 1), np.int64), np.cumsum(hist, axis=1)], axis=1)        # excl[d, b] = items of digit d in blocks < b
        out = np.full(R, -1, np.int64)
        ranges = np.zeros((1 << tile_bits, 2), np.int64)
        for b, (s0, c, h) in enumerate(blocks):
            if c == 0:
                continue
            pf, pn = excl[:, fb[h]], excl[:, fb[h + 1]]
            cnt = pn - pf                                              # items of (bucket h, digit d)
            start = st[h] + np.concatenate([[0], np.cumsum(cnt)[:-1]])
            base = start + excl[:, b] - pf
            w = packed[s0:s0 + c]
            d = (w >> np.uint32(32 - low)).astype(np.int64)
            rank = np.zeros(c, np.int64)                               # stable rank inside the block per digit
            seen = np.zeros(nd, np.int64)
            for i in range(c):
                rank[i] = seen[d[i]]; seen[d[i]] += 1
            out[base[d] + rank] = (w & np.uint32((1 << (32 - low)) - 1)).astype(np.int64)
            if b == fb[h]:
                nz = cnt > 0
                ranges[(h << low) + np.nonzero(nz)[0]] = np.stack([start[nz], start[nz] + cnt[nz]], 1)
        return out, ranges

    for tile_bits, T, R in ((9, 300, 5000), (13, 5440, 30000), (14, 8704, 9000), (16, 33124, 12000), (12, 3000, 4096), (10, 600, 1)):
        # skewed tile distribution: a few hot tiles (buckets larger than a block), many empty ones
        hot = rng.integers(0, T, 5)
        tiles = np.where(rng.random(R) < 0.6, rng.choice(hot, R), rng.integers(0, T, R)).astype(np.int64)
        ids = rng.integers(0, 1 << 20, R).astype(np.int64)
        got, ranges = msd_sort(tiles, ids, tile_bits)
        ref = np.argsort(tiles, kind="stable")
        assert np.array_equal(got, ids[ref]), (tile_bits, T, R)
        sorted_tiles = tiles[ref]
        for t in np.unique(tiles):
            lo, hi = np.searchsorted(sorted_tiles, t, "left"), np.searchsorted(sorted_tiles, t, "right")
            assert tuple(ranges[t]) == (lo, hi)
        assert int((ranges[:, 1] - ranges[:, 0]).sum()) == R


def test_scan_backward_reformulations_against_the_oracle_sums():
    """tests/scan_backward_replay.py replays the arithmetic of composite_bwd_scan_kernel in float32 numpy (16-lane scans with carries,
    collapsed dL_dalpha, background term folded into the carry, hoisted exponent, per-parity moment sums) and must land inside the
    accumulator tolerance against the oracle's double-precision sums -- with and without an upstream dL_dacc."""
    from tests import scan_backward_replay as replay
    assert replay.run("cfg1", 200) <= 1.0
    assert replay.run("cfg1", 200, grad_acc_zero=True) <= 1.0


def test_compiled_trainer_time_scalars_match_the_python_arithmetic():
    """ex4d_trainer_time_scalars (C++: CPython's float divmod restated, pow) against attributes.time_scalars (the Python numbers of
    c_gaussian_model.py:184-186, :364 and interpolations.py:83-86): all 13 fields of Ex4dAttrParams bit-identical, for integer and
    fractional timestamps, negative ones, several intervals."""
    from ex4dgs_amd import attributes as attr
    from ex4dgs_amd import native_trainer as nt
    lib = nt._lib()
    rng = np.random.default_rng(11)
    stamps = list(range(0, 300, 7)) + [299, 0.5, 13.25, 299.999, -3, -0.75, 1e-9] + [float(x) for x in rng.uniform(-20, 320, 60)]
    for interval, shift, duration, var_pad in ((10, 12, 300, 3), (7.5, 9.5, 120, 2), (3, 5, 0, 1), (10, 12.000000001, 300, 3)):
        cfg = nt.Ex4dTrainerConfig()
        cfg.Ns, cfg.Nd, cfg.K = 5, 7, 35
        cfg.duration, cfg.interval, cfg.time_shift, cfg.var_pad = max(duration, 1), interval, shift, var_pad
        for t in stamps:
            got = attr.Ex4dAttrParams()
            lib.ex4d_trainer_time_scalars(ctypes.byref(cfg), float(t), ctypes.byref(got))
            ref = attr.time_scalars(t, 5, 7, 35, max(duration, 1), interval, shift, var_pad)
            for name, _ in attr.Ex4dAttrParams._fields_:
                a, b = getattr(got, name), getattr(ref, name)
                assert a == b or (a != a and b != b), (interval, shift, t, name, a, b)


def test_compiled_trainer_rejects_bad_configurations_without_touching_a_device():
    """ex4d_trainer_create validates its configuration before the first HIP call: wrong sizes come back as NULL + message (host code)."""
    from ex4dgs_amd import native_trainer as nt
    lib = nt._lib()
    ptrs = (ctypes.c_void_p * 15)()

    def create(**kw):
        cfg = nt.Ex4dTrainerConfig()
        cfg.Ns, cfg.Nd, cfg.K, cfg.W, cfg.H, cfg.sh_degree, cfg.interval = 10, 4, 35, 64, 48, 3, 10.0
        for k, v in kw.items():
            setattr(cfg, k, v)
        h = lib.ex4d_trainer_create(ctypes.byref(cfg), ptrs)
        return h, lib.ex4d_trainer_last_error().decode()
    for bad in (dict(Ns=0, Nd=0), dict(W=0), dict(K=3), dict(sh_degree=4), dict(interval=0.0), dict(Ns=-1)):
        h, msg = create(**bad)
        assert not h and "out of range" in msg, (bad, msg)
    h, msg = create()                      # sizes fine, but the parameter pointers are NULL
    assert not h and "NULL" in msg
    assert lib.ex4d_trainer_bytes(None) == 0 and lib.ex4d_trainer_output(None, 0) is None


def test_async_frames_policy_capacity_growth_and_reporting():
    """Host logic of the asynchronous forward's opt-in policy (diff_gaussian_rasterization_df.async_frames), no device involved: the
    first frame runs synchronously and seeds the capacity, later frames get headroom x the largest count seen, a frame above its
    capacity is reported at the next forward (strict) or counted while the capacity grows, and the flow-free kernel is chosen from the
    previous frame's flow flag."""
    from ex4dgs_amd import _C
    from ex4dgs_amd.diff_gaussian_rasterization_df import AsyncFrames

    class Fake(_C.PendingFrame):
        def __init__(self, capacity, count, flow=False, assumed_no_flow=False, ready=True):
            self.capacity, self.assumed_no_flow, self._count, self._flow, self._ready = capacity, assumed_no_flow, count, flow, ready
            self._status = self._event = self._pool = None

        def done(self):
            return self._ready

        def wait(self):
            self._ready = True
            return self

        num_rendered = property(lambda self: self._count)
        has_flow = property(lambda self: self._flow)
        prefilter_violation = property(lambda self: False)

    pol = AsyncFrames()
    assert pol.next_call() == (0, False)                       # disabled: synchronous, reference behaviour
    pol.enable(headroom=1.25)
    assert pol.next_call() == (0, False)                       # nothing known yet: the first frame runs synchronously ...
    pol.record(1000)                                           # ... and seeds the capacity
    cap, no_flow = pol.next_call()
    assert cap == int(1.25 * 1000) + 4096 and not no_flow
    pol.record(Fake(cap, 1200, flow=False))
    cap2, no_flow = pol.next_call()                            # settled: it fitted (the capacity follows the largest count seen); no flow -> flow-free kernel next
    assert cap2 == int(1.25 * 1200) + 4096 and no_flow and pol.invalid_frames == 0
    pol.record(Fake(cap2, 9000, ready=False))                  # its status has not arrived yet: not looked at, nothing blocks
    assert pol.next_call()[0] == cap2 and len(pol.pending) == 1
    pol.pending[0]._ready = True
    with pytest.raises(RuntimeError, match="exceed the capacity"):
        pol.next_call()                                        # strict: the truncated frame is reported at the next forward
    assert pol.capacity == int(1.25 * 9000) + 4096 and pol.invalid_frames == 1
    pol.enable(headroom=1.5, capacity=100, strict=False)
    pol.record(Fake(100, 5000))
    assert pol.next_call()[0] == int(1.5 * 5000) + 4096 and pol.invalid_frames == 1      # counted, regrown, no exception
    pol.record(Fake(pol.capacity, 10, flow=True, assumed_no_flow=True))
    pol.drain()
    assert pol.invalid_frames == 2 and not pol.no_flow         # a violated no-flow assumption invalidates the frame as well
    pol.disable()
    assert pol.next_call() == (0, False)
