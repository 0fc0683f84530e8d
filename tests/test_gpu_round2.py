"""GPU tests added in round 2 (run with `-m gpu` on an MI355X): host-surface fixes (fused-attribute cache,
render mode, shape checks).  The parity / full-size / multi-rank tests of this round live next to their subjects
(test_gpu_parity_full.py, test_gpu_dist.py)."""
import math

import numpy as np
import pytest
import torch

from tests import helpers as h

pytestmark = pytest.mark.gpu


def _model(cfg="cfg3", P=6000, fused=True, **kw):
    from ex4dgs_amd.scene import make_scene
    dev = torch.device("cuda:0")
    model, cam, bg = make_scene(cfg, P=P, device=dev, fused=fused, **kw)
    for p in model.parameters():
        p.requires_grad_(True)
    return model, cam.to(dev), bg.to(dev)


def test_fused_cache_survives_two_backwards_at_one_timestamp(hip_lib):
    """Gradient accumulation over two cameras that share a timestamp, no optimizer step in between (parameter versions
    unchanged): the second render must not reuse an autograd graph the first backward already freed."""
    from ex4dgs_amd.render import render
    model, cam, bg = _model()
    g = torch.Generator().manual_seed(0)
    gc = torch.randn(3, cam.image_height, cam.image_width, generator=g).cuda()
    grads = []
    for rep in range(2):
        out = render(cam, model, None, bg, timestamp=137, near=4.0, far=300.0)
        (out["render"] * gc).sum().backward()           # raised "backward through the graph a second time" before the fix
        grads.append(model._xyz_motion.grad.clone())
    tol = 1e-4 * float(grads[0].abs().max())                 # float atomics: the two passes agree to rounding, not bit-wise
    assert float((grads[1] - 2 * grads[0]).abs().max()) <= tol      # accumulated twice the same gradient
    # two renders BEFORE one backward share the cached evaluation and both reach the parameters
    model.zero_grad()
    a = render(cam, model, None, bg, timestamp=137, near=4.0, far=300.0)
    b = render(cam, model, None, bg, timestamp=137, near=4.0, far=300.0)
    ((a["render"] + b["render"]) * gc).sum().backward()
    assert float((model._xyz_motion.grad - grads[1]).abs().max()) <= tol


@pytest.mark.parametrize("fused", [False, True])
def test_render_mode_static_only_and_dynamic_only(hip_lib, fused):
    """mode=1 / mode=2 of gaussian_renderer.render() (static-only / dynamic-only getters, scene/c_gaussian_model.py:170-176)
    equal rasterizing the corresponding rows of the full attribute set."""
    from ex4dgs_amd.render import render
    from ex4dgs_amd.diff_gaussian_rasterization_df import GaussianRasterizer, GaussianRasterizationSettings
    model, cam, bg = _model(fused=fused)
    plain, _, _ = _model(fused=False)
    Ns = model.num_static
    H, W = cam.image_height, cam.image_width
    with torch.no_grad():
        full = [plain.get_xyz_at_t(7), plain.get_opacity_at_t(7), plain.get_features(), plain.get_scaling(), plain.get_rotation_at_t(7)]
        for mode, sl in ((1, slice(0, Ns)), (2, slice(Ns, None))):
            out = render(cam, model, None, bg, timestamp=7, near=4.0, far=300.0, mode=mode)
            s = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), 0.1, torch.zeros(H, W, 2, device="cuda"),
                                              bg, 1.0, cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, 4.0, 300.0, False)
            xyz, opa, shs, scl, rot = [x[sl].contiguous() for x in full]
            ref = GaussianRasterizer(s)(means3D=xyz, means2D=torch.zeros_like(xyz), dir3D=torch.zeros_like(xyz), opacities=opa, shs=shs,
                                        scales=scl, rotations=rot)
            assert out["radii"].shape[0] == xyz.shape[0]
            diff = (out["render"] - ref[0]).abs()
            if not fused:
                assert float(diff.max()) == 0.0
            else:       # the fused getters differ from the torch ones by float rounding of the attributes (threshold flips possible)
                assert float(diff.mean()) < 1e-5 and float(diff.max()) < 2e-2
    with pytest.raises(ValueError):
        render(cam, model, None, bg, timestamp=7, mode=3)
    from types import SimpleNamespace
    with pytest.raises(NotImplementedError):
        render(cam, model, SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False, debug=False), bg, timestamp=7)


def test_native_calls_reject_short_tensors(hip_lib):
    """Raw pointers carry no shape: every per-Gaussian input with fewer than P rows (and a wrong-sized subpixel_offset / matrix)
    is refused before any kernel sees it."""
    from ex4dgs_amd import _C
    ins, st = h.scene_inputs("cfg1")
    s = h.gpu_settings(st, "cuda")
    e = torch.Tensor([])
    d = {k: v.cuda() for k, v in ins.items()}

    def call(**over):
        a = dict(d); a.update(over)
        sub = over.get("subpixel_offset", s.subpixel_offset)
        view = over.get("viewmatrix", s.viewmatrix)
        return _C.rasterize_gaussians(s.bg, a["means3D"], a["dir3D"], e, a["opacities"], a["scales"], a["rotations"], 1.0, e, view,
                                      s.projmatrix, s.tanfovx, s.tanfovy, 0.1, sub, s.image_height, s.image_width, a["shs"], 3, s.campos,
                                      False, 4.0, 300.0, False)
    call()
    for k in ("opacities", "scales", "rotations", "shs", "dir3D"):
        with pytest.raises(RuntimeError, match="expected 256 rows"):
            call(**{k: d[k][:-1]})
    with pytest.raises(RuntimeError, match="subpixel_offset"):
        call(subpixel_offset=torch.zeros(10, 10, 2, device="cuda"))
    with pytest.raises(RuntimeError, match="viewmatrix"):
        call(viewmatrix=torch.zeros(3, 4, device="cuda"))
    with pytest.raises(RuntimeError):
        _C.mark_visible(d["means3D"], s.viewmatrix.double(), s.projmatrix, 4.0)
    with pytest.raises(RuntimeError):
        _C.mark_visible(d["means3D"], s.viewmatrix.cpu(), s.projmatrix, 4.0)


def test_sliced_keyframe_gradients_equal_the_dense_ones(hip_lib):
    """ex4d_attributes_backward_sliced returns exactly the 4 / 2 non-zero time slices of the dense keyframe gradients (and the slice
    hint says where they sit); every other gradient is identical."""
    from ex4dgs_amd import attributes as attr
    from ex4dgs_amd.scene import make_scene
    model, cam, bg = make_scene("cfg3", P=5000, device="cuda", fused=True)
    params = [getattr(model, n) for n in attr.PARAM_ORDER]
    g = torch.Generator().manual_seed(3)
    N = model.num_static + model.num_dynamic
    gin = [torch.randn(N, 3, generator=g).cuda(), torch.randn(N, 4, generator=g).cuda(), torch.randn(N, 1, generator=g).cuda(), torch.randn(N, 3, generator=g).cuda(), None]
    for t in (0, 7, 137, 299):
        scal = attr.time_scalars(t, model.num_static, model.num_dynamic, model._xyz_motion.shape[1], model.duration, model.interval, model.time_shift, model.var_pad)
        dense = attr.backward_raw(scal, params, gin, with_shs=False)
        sliced, hint = attr.backward_raw(scal, params, gin, with_shs=False, sliced=True)
        assert hint == (scal.k - 1, 4, scal.k, 2)
        for i, n in enumerate(attr.PARAM_ORDER):
            if dense[i] is None:
                assert sliced[i] is None
            elif n in ("_xyz_motion", "_rotation_motion"):
                lo, cnt = (hint[0], 4) if n == "_xyz_motion" else (hint[2], 2)
                assert torch.equal(sliced[i], dense[i][:, lo:lo + cnt])
                rest = dense[i].clone(); rest[:, lo:lo + cnt] = 0
                assert float(rest.abs().max()) == 0.0            # every other time slice of the dense gradient is zero
            else:
                assert torch.equal(sliced[i], dense[i]), n


def test_sliced_radam_is_bit_identical_to_the_dense_step(hip_lib):
    """ex4d_radam_step_sliced (keyframe tensor + windowed gradients, incl. two overlapping windows = two frames) against ex4d_radam_step
    on the dense gradient the windows add up to: parameters and both moments bit-identical over 9 steps (crossing rho_t > 5)."""
    from ex4dgs_amd.optim import radam_step_raw, radam_step_sliced_raw
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(11)
    for rows, K, Cc in ((1003, 35, 3), (777, 35, 4), (5, 7, 3)):
        p0 = torch.randn(rows, K, Cc, generator=g).cuda()
        A, B = p0.clone(), p0.clone()
        mA, vA, mB, vB = [torch.zeros_like(p0) for _ in range(4)]
        for step in range(1, 10):
            wins = []
            dense = torch.zeros_like(p0)
            for w in range(1 + step % 3):                       # 1..3 windows, possibly overlapping
                cnt = 4 if Cc == 3 else 2
                first = int(torch.randint(0, K - cnt + 1, (1,), generator=g))
                blk = torch.randn(rows, cnt, Cc, generator=g).cuda() * (0.0 if (step == 4 and w == 0) else 1.0)
                wins.append((first, cnt, blk))
                dense[:, first:first + cnt] += blk               # same order as the kernel: windows added in index order
            radam_step_raw([(A.data_ptr(), dense.data_ptr(), mA.data_ptr(), vA.data_ptr(), A.numel(), 1e-2, step)], (0.9, 0.999), 1e-8, dev)
            radam_step_sliced_raw([(B.data_ptr(), mB.data_ptr(), vB.data_ptr(), rows, K, Cc, 1e-2, step, [(f, c, b.data_ptr()) for f, c, b in wins])],
                                  (0.9, 0.999), 1e-8, dev)
            torch.cuda.synchronize()
            assert torch.equal(A, B) and torch.equal(mA, mB) and torch.equal(vA, vB), (rows, K, Cc, step)
    with pytest.raises(RuntimeError):
        radam_step_sliced_raw([(A.data_ptr(), mA.data_ptr(), vA.data_ptr(), 5, 7, 3, 1e-2, 1, [(6, 4, A.data_ptr())])], (0.9, 0.999), 1e-8, dev)   # window outside [0, K)


def test_trainer_with_sliced_optimizer_tracks_the_dense_one(hip_lib):
    """FrameTrainer with the replicated optimizer: sliced keyframe gradients (default) against dense ones on two copies of one model --
    equal to the rounding of the rasterizer's float atomics after several steps, and no dense keyframe gradient buffer exists."""
    from ex4dgs_amd.scene import make_scene
    from ex4dgs_amd.trainer import FrameTrainer
    ma, cam, bg = make_scene("cfg3", P=8000, device="cuda", fused=True)
    mb, _, _ = make_scene("cfg3", P=8000, device="cuda", fused=True)
    w = torch.rand(3, cam.image_height, cam.image_width, generator=torch.Generator().manual_seed(5)).cuda()
    up = lambda out: ([out["render"]], [w])
    lrs = {n: 1e-6 for n in ma.PARAM_NAMES}
    ta = FrameTrainer(ma, optimizer=True, lrs=lrs)
    tb = FrameTrainer(mb, optimizer=True, lrs=lrs, sliced=False)
    assert ta.sliced and not tb.sliced
    assert ta.pgrad[ta.names.index("_xyz_motion")].shape[1:] == (4, 3) and tb.pgrad[tb.names.index("_xyz_motion")].shape == mb._xyz_motion.shape
    p0 = {n: getattr(ma, n).clone() for n in ma.PARAM_NAMES}
    for t in (0, 137, 41, 299, 7):
        ta.step(cam, bg, t, up); tb.step(cam, bg, t, up)
    ta.flush(); tb.flush(); torch.cuda.synchronize()
    for n in ma.PARAM_NAMES:
        a, b = getattr(ma, n), getattr(mb, n)
        moved = float((a - p0[n]).abs().max())
        assert moved > 0 and torch.isfinite(a).all(), n
        assert float((a - b).abs().max()) <= 1e-3 * moved + 1e-12, (n, float((a - b).abs().max()), moved)


def test_tile_sort_without_materialised_tile_ids(hip_lib):
    """Default setting of the MSD tile sort ("binning_tile_ids" = 0): point_list, the tile ranges and every output are the same as
    with the ids written (what the parity tests run with), and the ids follow from the ranges.  Covers both sort paths: 1352x1014
    (13 tile-id bits: MSD on packed words) and 128x96 (6 bits: key/value LSD sort, which always writes the ids)."""
    from ex4dgs_amd import _C
    from ex4dgs_amd.scene import SceneConfig
    for cfg, P in (("cfg2", 30_000), (SceneConfig("small image", 3000, 128, 96, 110.0, seed=5), None)):
        ins, st = h.scene_inputs(cfg, P=P)
        assert _C.get_option("binning_tile_ids") == 1
        a = h.gpu_forward_raw(ins, st)
        try:
            _C.set_option("binning_tile_ids", 0)
            b = h.gpu_forward_raw(ins, st)
        finally:
            _C.set_option("binning_tile_ids", 1)
        assert a["num_rendered"] == b["num_rendered"] > 0
        for k in ("point_list", "ranges", "color", "depth", "acc", "flow", "idx", "n_contrib"):
            assert torch.equal(a[k], b[k]), k
        ranges = a["ranges"].long().cpu()
        T = ranges.shape[0]
        counts = ranges[:, 1] - ranges[:, 0]
        ids = torch.repeat_interleave(torch.arange(T), counts)
        assert torch.equal(ids, a["tile_ids"].long().cpu())
        occupied = counts > 0
        assert bool((ranges[occupied][1:, 0] == ranges[occupied][:-1, 1]).all()) and int(counts.sum()) == a["num_rendered"]


def test_compiled_host_path_matches_the_python_trainer(hip_lib):
    """include/ex4d_trainer.h (one C++ call per iteration, persistent workspace) against trainer.FrameTrainer + loss.l1_ssim_loss (Python,
    autograd) on two copies of one model: the same loss, the same parameter gradients, the same parameters after several RAdam steps --
    equal to the rounding of the rasterizer's float atomics, since both sequence the same kernels."""
    from ex4dgs_amd.loss import l1_ssim_loss
    from ex4dgs_amd.native_trainer import NativeTrainer
    from ex4dgs_amd.scene import make_scene
    from ex4dgs_amd.trainer import FrameTrainer
    ma, cam, bg = make_scene("cfg3", P=8000, device="cuda", fused=True)
    mb, _, _ = make_scene("cfg3", P=8000, device="cuda", fused=True)
    cam = cam.to("cuda"); bg = bg.cuda()
    gt = torch.rand(3, cam.image_height, cam.image_width, generator=torch.Generator().manual_seed(11)).cuda()
    lrs = {n: 1e-6 for n in ma.PARAM_NAMES}
    # (1) gradients only
    na = NativeTrainer(ma, cam, optimizer=False, lrs=lrs)
    fb = FrameTrainer(mb, optimizer=False)
    losses = []

    def up(out):
        loss = l1_ssim_loss(out["render"], gt, 0.2)[0]
        losses.append(loss.detach())
        return [loss], [None]
    na.step(cam, bg, 137, gt)
    out = fb.step(cam, bg, 137, up)
    fb.flush(); torch.cuda.synchronize()
    assert abs(float(na.output("loss")) - float(losses[0])) <= 1e-6
    assert float((na.output("render") - out["render"]).abs().max()) <= 1e-6 and torch.equal(na.output("radii"), out["radii"])
    assert na.num_rendered > 0
    ref = fb.grads()
    dense = {"_xyz_motion": 4, "_rotation_motion": 2}
    for n in ma.PARAM_NAMES:
        g, hint = na.grad(n)
        r = ref[n]
        if n in dense:          # FrameTrainer without an optimizer keeps dense keyframe gradients: compare the touched slices
            first = hint[0] if n == "_xyz_motion" else hint[2]
            assert float(r.abs().sum()) > 0 and float((r[:, first:first + dense[n]] - g).abs().max()) <= 1e-5 * max(1.0, float(r.abs().max()))
            z = r.clone(); z[:, first:first + dense[n]] = 0
            assert float(z.abs().max()) == 0.0
        else:
            assert g.shape == r.shape and float((g - r).abs().max()) <= 1e-5 * max(1.0, float(r.abs().max())), n
    na.close()
    # (2) with the optimizer, several timestamps.  The loss gradient is ~1e-6 per parameter: RAdam's first five steps (p -= lr * mhat) move
    # nothing representable, from the sixth on the step is ~lr whatever the gradient scale
    lrs = {n: 1e-4 for n in ma.PARAM_NAMES}
    p0 = {n: getattr(ma, n).clone() for n in ma.PARAM_NAMES}
    na = NativeTrainer(ma, cam, optimizer=True, lrs=lrs)
    fb = FrameTrainer(mb, optimizer=True, lrs=lrs)
    upg = lambda out: ([l1_ssim_loss(out["render"], gt, 0.2)[0]], [None])
    for t in (0, 137, 41, 299, 7, 138, 40, 139):
        na.step(cam, bg, t, gt); fb.step(cam, bg, t, upg)
    fb.flush(); torch.cuda.synchronize()
    assert na.bytes() > 0
    for n in ma.PARAM_NAMES:
        a, b = getattr(ma, n), getattr(mb, n)
        moved = float((a - p0[n]).abs().max())
        assert moved > 0 and torch.isfinite(a).all(), n
        ulp = 2.0 ** -23 * float(a.abs().max())          # the updates are a few ulp of the parameters: allow two of them
        assert float((a - b).abs().max()) <= 1e-3 * moved + 2 * ulp, (n, float((a - b).abs().max()), moved)
    with pytest.raises(RuntimeError, match="gt_image"):
        na.step(cam, bg, 0, gt[:, :10])
    na.close()


def test_compiled_host_path_static_only_model_and_growing_arenas(hip_lib):
    """NativeTrainer on a model without dynamic Gaussians (all eight motion tensors empty) and with views whose instance counts differ
    by more than the arena's growth step: the binning arena grows, results stay those of a fresh trainer."""
    from ex4dgs_amd.native_trainer import NativeTrainer
    from ex4dgs_amd.scene import make_scene
    model, cam, bg = make_scene("cfg2", P=6000, device="cuda", fused=True)
    assert model.num_dynamic == 0
    cam = cam.to("cuda"); bg = bg.cuda()
    gt = torch.rand(3, cam.image_height, cam.image_width, generator=torch.Generator().manual_seed(3)).cuda()
    nt = NativeTrainer(model, cam, optimizer=True, lrs={n: 1e-4 for n in model.PARAM_NAMES})
    b0 = nt.bytes()
    losses = []
    for t in range(8):
        nt.step(cam, bg, t, gt)
        losses.append(float(nt.output("loss")))
    assert all(np.isfinite(losses)) and nt.num_rendered > 0 and nt.bytes() > b0       # arenas were allocated by the first frame
    R0 = nt.num_rendered
    # every Gaussian three times as large: far more instances than the binning arena holds -> it grows inside step()
    with torch.no_grad():
        model._scaling += math.log(3.0)
    b1 = nt.bytes()
    nt.step(cam, bg, 0, gt)
    assert nt.num_rendered > 1.5 * R0 and nt.bytes() > b1 and np.isfinite(float(nt.output("loss")))
    render = nt.output("render")
    nt.close()
    fresh = NativeTrainer(model, cam, optimizer=False)
    # the step above already moved the parameters once more; compare against a fresh trainer on the CURRENT parameters
    fresh.step(cam, bg, 0, gt)
    r2 = fresh.output("render")
    fresh.close()
    assert torch.isfinite(render).all() and float((render - r2).abs().max()) < 5e-3      # one RAdam step of 1e-4 apart


def test_lean_geometry_path_and_optional_gradient_outputs(hip_lib):
    """Defaults of the library ("geom_debug_arrays" = 0: cov3D / tiles_touched not written, the backward recomputes the covariance;
    dL_dcolors / dL_dcov3D not written when the caller has no input to receive them) against the fully materialised run the parity
    tests use: identical forward, gradients equal to the rounding of the float atomics."""
    from ex4dgs_amd import _C
    ins, st = h.scene_inputs("cfg3", P=9000)
    a = h.gpu_forward_raw(ins, st)
    H, W = a["color"].shape[1:]
    grads = h.upstream_grads(a["acc"].cpu(), H, W, seed=5)
    ga = h.gpu_backward_raw(ins, a, grads)
    assert _C.get_option("geom_debug_arrays") == 1
    try:
        _C.set_option("geom_debug_arrays", 0)
        b = h.gpu_forward_raw(ins, st)
        gb = h.gpu_backward_raw(ins, b, grads)
    finally:
        _C.set_option("geom_debug_arrays", 1)
    for k in ("color", "depth", "acc", "flow", "idx", "radii", "point_list", "ranges", "n_contrib"):
        assert torch.equal(a[k], b[k]), k
    vis = a["radii"] > 0                                      # the records of culled Gaussians are never written
    assert torch.equal(a["records"][vis], b["records"][vis])
    for k in ga:
        if k == "acc16":
            continue
        x, y = ga[k], gb[k]
        assert float((x - y).abs().max()) <= 1e-5 * max(1.0, float(x.abs().max())), k
    # the C ABI accepts NULL for the two gradients nobody may need
    s = a["settings"]
    e = torch.Tensor([])
    d = lambda k: ins[k].cuda() if ins.get(k) is not None else e
    gc, gd, gf, gacc = [g.cuda() for g in grads]
    outs = _C.rasterize_gaussians_backward(
        s.bg, d("means3D"), a["radii"], d("colors_precomp"), d("scales"), d("rotations"), a["depth"], a["acc"], s.min_depth, s.max_depth,
        s.scale_modifier, d("cov3D_precomp"), s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy, s.kernel_size, s.subpixel_offset,
        gc, gd, gf, gacc, d("shs"), s.sh_degree, s.campos, a["geomBuffer"], a["num_rendered"], a["binningBuffer"], a["imgBuffer"], s.debug,
        need_colors=False, need_cov3D=False)
    assert outs[1].numel() == 0 and outs[4].numel() == 0
    names = ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations", "dL_ddir")
    for n, o in zip(names, outs):
        if n in ("dL_dcolors", "dL_dcov3D"):
            continue
        assert float((o - ga[n]).abs().max()) <= 1e-5 * max(1.0, float(ga[n].abs().max())), n


def test_radam_nan_to_num_flag_and_device_side_window_positions(hip_lib):
    """(1) Ex4dRadamTensor.nan_to_num: the gradient is read through torch.nan_to_num (train.py:244-247 does that to
    _opacity_duration_var.grad before optimizer.step()) -- an injected NaN / inf neither reaches the parameter nor the optimizer state, and
    the result equals torch.optim.RAdam stepped on the sanitised gradient.  (2) Ex4dRadamSlicedTensor.first_dev: window positions read
    from device memory (no host round trip) give the step of the host-side positions bit for bit."""
    from ex4dgs_amd.optim import radam_step_raw, radam_step_sliced_raw
    g0 = torch.Generator().manual_seed(4)
    p = torch.randn(5000, generator=g0).cuda()
    ref = torch.nn.Parameter(p.clone())
    opt = torch.optim.RAdam([ref], lr=1e-2)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for step in range(1, 9):
        g = torch.randn(5000, generator=g0).cuda()
        g[17] = float("nan")
        ref.grad = torch.nan_to_num(g)
        opt.step()
        radam_step_raw([(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), 1e-2, step, 1)], (0.9, 0.999), 1e-8, p.device)
    torch.cuda.synchronize()
    r = ref.detach()
    assert torch.isfinite(p).all() and torch.isfinite(m).all() and torch.isfinite(v).all()
    assert float(((p - r).abs() / r.abs().clamp_min(1.0)).max()) <= 2e-6
    # +-inf -> +-FLT_MAX (one step: exp_avg = 0.1 x FLT_MAX stays finite, with the sign of the gradient)
    pi, mi, vi = p.clone(), torch.zeros_like(p), torch.zeros_like(p)
    gi = torch.zeros_like(p); gi[99] = float("inf"); gi[100] = float("-inf")
    radam_step_raw([(pi.data_ptr(), gi.data_ptr(), mi.data_ptr(), vi.data_ptr(), pi.numel(), 1e-2, 1, 1)], (0.9, 0.999), 1e-8, p.device)
    torch.cuda.synchronize()
    assert torch.isfinite(mi).all() and float(mi[99]) > 1e37 and float(mi[100]) < -1e37
    # without the flag the NaN goes through (the caller asked for the plain step)
    p2, m2, v2 = p.clone(), torch.zeros_like(p), torch.zeros_like(p)
    radam_step_raw([(p2.data_ptr(), g.data_ptr(), m2.data_ptr(), v2.data_ptr(), p2.numel(), 1e-2, 1)], (0.9, 0.999), 1e-8, p.device)
    assert bool(torch.isnan(m2[17]))
    # (2) sliced step: positions on the host vs in device memory
    rows, K, Cc = 300, 35, 3
    q0 = torch.randn(rows, K, Cc, generator=g0).cuda()
    wins = [torch.randn(rows, 4, Cc, generator=g0).cuda() for _ in range(3)]
    firsts = [2, 17, 16]
    res = []
    for dev_side in (False, True):
        q, mq, vq = q0.clone(), torch.zeros_like(q0), torch.zeros_like(q0)
        fd = torch.tensor(firsts, dtype=torch.int32).cuda()
        for step in range(1, 8):
            w = [((0 if dev_side else f), 4, x.data_ptr()) for f, x in zip(firsts, wins)]
            item = (q.data_ptr(), mq.data_ptr(), vq.data_ptr(), rows, K, Cc, 1e-3, step, w) + ((fd.data_ptr(),) if dev_side else ())
            radam_step_sliced_raw([item], (0.9, 0.999), 1e-8, q.device)
        torch.cuda.synchronize()
        res.append((q, mq, vq))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    assert not torch.equal(res[0][0], q0)


def test_compiled_host_path_learning_rate_and_sh_degree_setters(hip_lib):
    """ex4d_trainer_set_lr / ex4d_trainer_set_sh_degree (the reference changes both during training: update_learning_rate,
    oneupSHdegree): a trainer whose rates are set to zero stops moving its parameters; lowering the SH degree changes the render to
    that of a trainer created at the lower degree."""
    from ex4dgs_amd.native_trainer import NativeTrainer
    from ex4dgs_amd.scene import make_scene
    model, cam, bg = make_scene("cfg3", P=6000, device="cuda", fused=True)
    cam = cam.to("cuda"); bg = bg.cuda()
    gt = torch.rand(3, cam.image_height, cam.image_width, generator=torch.Generator().manual_seed(3)).cuda()
    nt = NativeTrainer(model, cam, optimizer=True, lrs={n: 1e-3 for n in model.PARAM_NAMES})
    for t in range(7):
        nt.step(cam, bg, t, gt)
    before = {n: getattr(model, n).clone() for n in model.PARAM_NAMES}
    nt.set_lrs({n: 0.0 for n in model.PARAM_NAMES})
    nt.step(cam, bg, 7, gt)
    torch.cuda.synchronize()
    for n in model.PARAM_NAMES:
        assert torch.equal(getattr(model, n), before[n]), n
    with pytest.raises(RuntimeError):
        nt.set_lrs({"_xyz": -1.0})
    r3 = nt.output("render")
    nt.set_sh_degree(1)
    nt.step(cam, bg, 7, gt)
    r1 = nt.output("render")
    nt.close()
    assert float((r1 - r3).abs().max()) > 1e-3
    model.active_sh_degree = 1
    fresh = NativeTrainer(model, cam, optimizer=False)
    fresh.step(cam, bg, 7, gt)
    assert float((fresh.output("render") - r1).abs().max()) <= 1e-6
    fresh.close()
    with pytest.raises(RuntimeError):
        NativeTrainer(model, cam, optimizer=False).set_sh_degree(4)


def test_compacted_quadrant_lists_left_for_the_backward(hip_lib):
    """The forward compositing kernel leaves, per (tile, 8x8 quadrant), the compacted list the backward streams (include/ex4d_rasterizer.h:
    Ex4dBinningLayout.qlist / .qcount): entries are positions in the tile list, ascending (round 6: the position alone -- the Gaussian id is
    point_list[range start + position]), and every quadrant's list reaches its deepest contributor (max n_contrib over its pixels)."""
    ins, st = h.scene_inputs("cfg2", P=20000, dir_scale=0.0)
    g = h.gpu_forward_raw(ins, st)
    H, W = st["image_height"], st["image_width"]
    gx = (W + 15) // 16
    ranges = g["ranges"].cpu().numpy().astype(np.int64); pl = g["point_list"].cpu().numpy().astype(np.int64)
    ql = g["qlist"].cpu().numpy().astype(np.int64); qc = g["qcount"].cpu().numpy().astype(np.int64)
    ncon = g["n_contrib"].cpu().numpy().astype(np.int64)
    total = 0
    for t in range(ranges.shape[0]):
        r0, r1 = ranges[t]; n = r1 - r0
        ty, tx = divmod(t, gx)
        for q in range(4):
            qn = qc[t, q]
            assert 0 <= qn <= n
            ent = ql[4 * r0 + q * n: 4 * r0 + q * n + qn]
            total += qn
            if qn:
                k = ent
                assert np.all(np.diff(k) > 0) and k.min() >= 0 and k.max() < n and pl[r0 + k].min() >= 0, (t, q)
            sub = ncon[ty * 16 + (q >> 1) * 8: ty * 16 + (q >> 1) * 8 + 8, tx * 16 + (q & 1) * 8: tx * 16 + (q & 1) * 8 + 8]
            deepest = int(sub.max()) if sub.size else 0
            # the deepest contributor itself is a survivor of its quadrant's cull, so it is in the list
            assert deepest == 0 or (qn > 0 and (ent == deepest - 1).any()), (t, q, deepest)
    assert 0 < total < 4 * g["num_rendered"]


def test_sh_direction_sums_prepared_by_the_forward_and_second_backward(hip_lib):
    """Ex4dParams.prepare_backward: under autograd the forward leaves the SH direction sums in its geometry buffer and the backward does
    not read the SH tensors; a second backward through a retained graph reads the same (read-only) state.  Both give the gradients of
    the plain path (raw `_C` calls, the backward evaluating the SH derivative itself) up to the order of the float atomics."""
    from ex4dgs_amd.diff_gaussian_rasterization_df import rasterize_gaussians
    ins, st = h.scene_inputs("cfg3", P=9000, dir_scale=0.0)
    s = h.gpu_settings(st, "cuda")
    leaves = {k: ins[k].cuda().requires_grad_(True) for k in ("means3D", "dir3D", "shs", "opacities", "scales", "rotations")}
    means2D = torch.zeros_like(leaves["means3D"], requires_grad=True)
    e = torch.Tensor([])
    color, radii, depth, flow, acc, idx = rasterize_gaussians(leaves["means3D"], means2D, leaves["dir3D"], leaves["shs"], e, leaves["opacities"],
                                                            leaves["scales"], leaves["rotations"], e, s)
    H, W = st["image_height"], st["image_width"]
    grads = [x.cuda() for x in h.upstream_grads(acc.detach().cpu(), H, W, seed=4)]
    names = list(leaves)
    first = torch.autograd.grad([color, depth, flow, acc], [leaves[n] for n in names] + [means2D], grads, retain_graph=True)
    second = torch.autograd.grad([color, depth, flow, acc], [leaves[n] for n in names] + [means2D], grads)
    g = h.gpu_forward_raw(ins, st)
    raw = h.gpu_backward_raw(ins, g, grads)
    ref = dict(means3D=raw["dL_dmeans3D"], dir3D=raw["dL_ddir"], shs=raw["dL_dsh"], opacities=raw["dL_dopacity"], scales=raw["dL_dscales"],
               rotations=raw["dL_drotations"])
    for n, a, b in zip(names + ["means2D"], first, second):
        r = ref[n] if n != "means2D" else raw["dL_dmeans2D"]
        scale = max(1.0, float(r.abs().max()))
        assert float((a - r).abs().max()) <= 1e-5 * scale and float((b - r).abs().max()) <= 1e-5 * scale, n


def test_hand_scheduled_forward_walk_is_bit_identical_to_the_compiled_one(hip_lib):
    """The compositing forward's entry walk exists twice: hand-scheduled inline asm (default, frames without flow) and the compiler's
    loop (`composite_fwd_asm` = 0; also what frames with flow run).  Same operations on the same operands in the same order: every
    output and the per-pixel state must agree bit for bit -- on a shallow scene, a deep-overlap scene (saturating pixels: the
    rare path) and an image whose size is not a multiple of the tile."""
    from ex4dgs_amd import _C
    assert _C.get_option("composite_fwd_asm") == 1
    from ex4dgs_amd.scene import CONFIGS
    odd = CONFIGS["cfg3"]._replace(name="cfg3 at 333x217", width=333, height=217, focal=180.0)
    for cfg, P in (("cfg2", 20_000), ("cfg5", 30_000), (odd, 9_000)):
        ins, st = h.scene_inputs(cfg, P=P, dir_scale=0.0)
        a = h.gpu_forward_raw(ins, st)
        _C.set_option("composite_fwd_asm", 0)
        try:
            b = h.gpu_forward_raw(ins, st)
        finally:
            _C.set_option("composite_fwd_asm", 1)
        for k in ("color", "depth", "acc", "flow", "idx", "final_T", "n_contrib", "qcount"):
            assert torch.equal(a[k], b[k]), (cfg, k)
        assert float(a["final_T"].min()) < 1e-3 or cfg != "cfg5"          # the deep scene does saturate pixels
        cfg = cfg if isinstance(cfg, str) else cfg.name
        # the compacted lists agree on their valid prefixes
        T = a["qcount"].shape[0]
        ranges = a["ranges"].cpu().numpy().astype(np.int64)
        qa, qb, qc = a["qlist"].cpu().numpy(), b["qlist"].cpu().numpy(), a["qcount"].cpu().numpy()
        for t in range(0, T, max(1, T // 97)):
            r0, r1 = ranges[t]
            for q in range(4):
                s0 = 4 * r0 + q * (r1 - r0)
                assert np.array_equal(qa[s0: s0 + qc[t, q]], qb[s0: s0 + qc[t, q]]), (cfg, t, q)
