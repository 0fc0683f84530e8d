"""One training-iteration core per rank, built from the fused pieces of this package:

    fused attribute evaluation (attributes.forward_raw)            scene/c_gaussian_model.py:170-215,330-375 getters
 -> rasterizer forward + backward (GaussianRasterizer, SplitSH)    gaussian_renderer/__init__.py:19-124, train.py:139-153
 -> fused attribute backward into persistent gradient buffers      (autograd of the getters in the reference)
 -> sum of the 15 model-parameter gradients over the ranks         (the reference is single-process)
 -> optimizer step: replicated fused RAdam or the sharded one      scene/c_gaussian_model.py:430-449, train.py:250

One view (camera, timestamp) per rank per step: views shard round-robin (dist.shard_views), parameters are replicated.  The
attribute backward and the gradient exchange of frame i run on a side stream / the communicator's stream: the four feature gradients
(3/4 of the bytes) go on the wire before the attribute backward starts.  Without an optimizer the whole exchange hides behind the
rasterization of frame i+1; with one (the default of the multi-GPU bench) it has to finish before the optimizer step at the top of step
i+1 and is exposed except for the part beside the attribute backward.
No CPU fallback: everything here needs the HIP library and a ROCm device.
"""
import math

import torch
import torch.distributed as dist

from . import attributes as attr
from . import dist as xdist
from ._C import SplitSH
from .diff_gaussian_rasterization_df import GaussianRasterizationSettings, rasterize_gaussians

def reference_lrs(spatial_lr_scale=1.0):
    """The 15 optimizer groups of CGaussianModel.training_setup (scene/c_gaussian_model.py:430-447) with the default OptimizationParams
    (arguments/__init__.py:93-110); the two position groups carry the initial value of their exponential schedule times
    spatial_lr_scale.  Pinned by tests/golden/training_args.json (captured from the imported reference)."""
    return {"_xyz": 0.00016 * spatial_lr_scale, "_features_dc": 0.0025, "_features_rest": 0.0025 / 20.0, "_opacity": 0.05, "_scaling": 0.005,
            "_rotation": 0.00001, "_xyz_disp": 0.0001,
            "_xyz_motion": 0.00016 * spatial_lr_scale, "_features_dc_motion": 0.0025, "_features_rest_motion": 0.0025 / 20.0,
            "_scaling_motion": 0.005, "_opacity_motion": 0.05, "_opacity_duration_center": 0.001, "_opacity_duration_var": 0.0005,
            "_rotation_motion": 0.001}


# reference group name (c_gaussian_model.py:430-447) -> parameter attribute
REFERENCE_GROUP_NAMES = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling",
                         "rotation": "_rotation", "xyz_disp": "_xyz_disp", "motion_xyz": "_xyz_motion", "motion_f_dc": "_features_dc_motion",
                         "motion_f_rest": "_features_rest_motion", "motion_scaling": "_scaling_motion", "motion_opacity": "_opacity_motion",
                         "motion_opacity_center": "_opacity_duration_center", "motion_opacity_var": "_opacity_duration_var",
                         "motion_rotation": "_rotation_motion"}
DEFAULT_LRS = reference_lrs(1.0)
# parameters whose gradient the reference passes through nan_to_num before optimizer.step() (train.py:244-247)
NAN_TO_NUM = ("_opacity_duration_var",)


class FrameTrainer:
    """model: scene.DynamicGaussians on a ROCm device.  exchange: "none" | "allreduce" | "sharded" (reduce-scatter + sharded RAdam +
    all-gather; implies optimizer).  optimizer: False | True (replicated fused RAdam when exchange != "sharded")."""

    def __init__(self, model, exchange="none", optimizer=False, lrs=None, overlap=True, group=None, sliced=None, spatial_lr_scale=1.0,
                 force_collectives=False, async_forward=None, views_per_step=1):
        """lrs: overrides of the reference table reference_lrs(spatial_lr_scale).
        sliced (default: on whenever an optimizer runs, replicated or sharded): the keyframe gradients stay [Nd,4,3] / [Nd,2,4] slices from
        the attribute backward through the exchange into ex4d_radam_step_sliced -- no 196 MB zero fill, no dense read.  Replicated
        optimizer: all-gather of the ranks' windows (16 MB per rank sent instead of 196); sharded optimizer: the keyframe tensors are
        sharded by rows and one all-to-all hands every owner the rows of every rank's window (dist.SliceRowExchange: 14 MB sent per rank
        at 8 ranks).
        force_collectives: issue the exchange's collectives even in a process group of one rank (dist.py: the RCCL-native branches
        run and are checked on a one-GPU box).
        async_forward (default: off since round 6 -- with the synchronous forward's read-back off the caller's stream the two are equal at 1.0 M Gaussians
        and the synchronous one is 8 % faster at 100 k (bench.py --train-core [--sync-forward]); round 5: on for a single rank with exchange "none"): the rasterizer forward runs asynchronously (no instance-count read-back:
        include/ex4d_rasterizer.h Ex4dParams.instance_capacity, under an AsyncFrames policy of the trainer's own); the frame's status is looked at once, right before its gradients are applied, and a frame that overflowed its
        capacity is RE-RUN first -- the parameters are those of the synchronous path.  `replays` counts such re-runs.
        views_per_step = k > 1 (round 6): a rank renders k views per optimizer step; their gradients are added up locally (persistent
        accumulators), exchanged ONCE after the k-th view and applied once -- the batch of a step is N k views, the wire time per view
        1/k of the single-view step's (DESIGN.md section 6 has the projected efficiencies).  Keyframe gradients are dense in this mode (k
        views touch k windows; the sliced optimizer takes one window per rank), the forward is synchronous."""
        assert exchange in ("none", "allreduce", "sharded")
        self.k = int(views_per_step)
        assert self.k >= 1
        if self.k > 1:
            if sliced:
                raise ValueError("views_per_step > 1 accumulates dense keyframe gradients (one gradient window per view and rank): sliced=True does not apply")
            if async_forward:
                raise ValueError("views_per_step > 1 runs the synchronous forward (a re-run frame would be accumulated twice)")
            sliced, async_forward = False, False
        self._acc, self._nacc = None, 0
        self.model = model
        self.names = list(attr.PARAM_ORDER)
        self.params = [getattr(model, n) for n in self.names]
        self.device = self.params[0].device
        if self.device.type != "cuda":
            raise RuntimeError("FrameTrainer needs the model on a ROCm device (no CPU fallback)")
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        force = xdist._forced(force_collectives) and dist.is_initialized()
        self.mode = exchange if (self.world > 1 or exchange == "sharded" or force) else "none"
        self.overlap = overlap
        self.side = torch.cuda.Stream(device=self.device) if overlap else None
        self.feature_idx = [self.names.index(n) for n in attr.FEATURE_NAMES]
        # persistent gradient buffers of the 11 non-feature parameters (written once per step by the attribute backward);
        # the four feature gradients come out of the rasterizer's SplitSH path as fresh tensors every step
        self.pgrad = [None if i in self.feature_idx else torch.zeros_like(p) for i, p in enumerate(self.params)]
        lrs = dict(reference_lrs(spatial_lr_scale), **(lrs or {}))
        self.lrs = [lrs[n] for n in self.names]
        self.opt = None
        self.exchange = None
        # one gradient window per rank reaches ex4d_radam_step_sliced: more ranks than it takes windows -> dense keyframe gradients
        from .optim import MAX_WINDOWS
        gather_world = self.world if self.mode in ("allreduce", "sharded") else 1      # exchange "none": nothing is summed over ranks, keyframes included
        fits = gather_world <= MAX_WINDOWS
        has_opt = bool(optimizer) or self.mode == "sharded"
        self.sliced = (has_opt and model.num_dynamic > 0 and fits) if sliced is None else bool(sliced)
        if self.sliced and not has_opt:
            raise ValueError("sliced keyframe gradients need an optimizer in the step (plain gradient output is dense)")
        if self.sliced and not fits:
            raise ValueError(f"sliced keyframe gradients take one window per rank, at most {MAX_WINDOWS}")
        self.kf_idx = [self.names.index(n) for n in attr.SLICED_SHAPES] if self.sliced else []
        self.kf_gather = []
        if self.sliced:
            for i in self.kf_idx:
                shape = (self.params[i].shape[0],) + attr.SLICED_SHAPES[self.names[i]]
                self.pgrad[i] = torch.zeros(shape, dtype=torch.float32, device=self.device)
                if self.mode != "sharded":      # replicated optimizer: every rank needs every window (all-gather); sharded: row all-to-all inside ShardedRAdam
                    self.kf_gather.append(xdist.SliceGather(shape, self.device, group=group, local_only=(self.mode != "allreduce"), force=force))
        # two exchanges: the four feature gradients (3/4 of the bytes) leave the rasterizer backward and are on the wire while the
        # attribute backward still runs; the other parameters follow it
        self.feat_pos = [i for i in self.feature_idx]
        self.rest_pos = [i for i in range(len(self.params)) if i not in self.kf_idx and i not in self.feature_idx]
        self.exchange_feat = None
        if self.mode == "sharded":
            self.opt = xdist.ShardedRAdam(self.params, self.lrs, group=group, nan_to_num=[n in NAN_TO_NUM for n in self.names], force=force,
                                          sliced={i: attr.SLICED_SHAPES[self.names[i]][0] for i in self.kf_idx})
            self.exchange = self.opt.exchange
        else:
            if self.mode == "allreduce":
                self.exchange_feat = xdist.ParamGradExchange([self.params[i].shape for i in self.feat_pos], self.device, mode="allreduce", group=group, force=force)
                self.exchange = xdist.ParamGradExchange([self.params[i].shape for i in self.rest_pos], self.device, mode="allreduce", group=group, force=force)
            if optimizer:
                self.m = [torch.zeros_like(p) for p in self.params]
                self.v = [torch.zeros_like(p) for p in self.params]
                self.steps = 0
        self.optimizer = bool(optimizer) or self.mode == "sharded"
        # (round 5 made the non-blocking forward the default wherever the class can recover from an overflow by itself; round 6 measured the
        # synchronous forward -- its read-back off the caller's stream now -- equal or faster, and it needs no capacity policy: opt-in again)
        self.async_forward = False if async_forward is None else bool(async_forward)
        self.replays = 0
        self._frame = self._last_args = None
        if self.async_forward:
            if self.mode != "none":
                raise ValueError("async_forward re-runs an overflowing frame on its own: single rank, exchange 'none' only")
            from .diff_gaussian_rasterization_df import AsyncFrames
            self._policy = AsyncFrames().enable(headroom=1.25, strict=False)     # this trainer's own: other callers stay synchronous
        self._grads = None
        self._zero_sub = {}
        self.last = {}

    # ------------------------------------------------------------------------------------------
    def _settings(self, cam, bg, near, far):
        H, W = int(cam.image_height), int(cam.image_width)
        key = (H, W)
        if key not in self._zero_sub:
            self._zero_sub[key] = torch.zeros(H, W, 2, device=self.device)
        m = self.model
        return GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), kernel_size=m.kernel_size,
            subpixel_offset=self._zero_sub[key], bg=bg, scale_modifier=1.0, viewmatrix=cam.world_view_transform,
            projmatrix=cam.full_proj_transform, sh_degree=m.active_sh_degree, campos=cam.camera_center, prefiltered=False,
            min_depth=near, max_depth=far, debug=False)

    def finish_exchange(self):
        """Block the current stream on the pending gradient exchange (its results are needed by the optimizer / the caller)."""
        def wait_all():
            if self.exchange_feat is not None:
                self.exchange_feat.wait()
            if self.exchange is not None:
                self.exchange.wait()
            for gth in self.kf_gather:
                gth.wait()
            if self.mode == "sharded":
                for ex in self.opt.row_exchange.values():
                    ex.wait()
        if self.exchange is not None or self.kf_gather:
            if self.side is not None:
                with torch.cuda.stream(self.side):
                    wait_all()
                torch.cuda.current_stream(self.device).wait_stream(self.side)
            else:
                wait_all()

    def exchange_bytes_on_wire(self):
        """Payload bytes one rank contributes to one step's gradient exchange (all collectives of the step)."""
        n = 0
        for ex in (self.exchange_feat, self.exchange):
            if ex is not None:
                n += ex.bytes_on_wire()
        if self.world > 1:
            n += sum(g.bytes_on_wire() for g in self.kf_gather if not g.local_only)
            if self.mode == "sharded":
                n += sum(ex.bytes_on_wire() for ex in self.opt.row_exchange.values())
        return n

    def step(self, cam, bg, t, upstream, near=0.2, far=300.0):
        """upstream: callable(render dict) -> (list of outputs, list of their gradients), e.g. a loss evaluated with the fused
        L1+SSIM op, or fixed synthetic gradients.  Returns the render dict (tensors of this frame, detached)."""
        # the previous frame's exchange must be complete before this frame's optimizer-updated parameters are read (optimizer on) --
        # without an optimizer it only has to finish before its buffers are overwritten by this frame's attribute backward
        if self.optimizer and self._grads is not None:
            self._apply_optimizer()
        elif self.async_forward:
            self._settle_frame()
        self._last_args = (cam, bg, t, upstream, near, far)
        return self._run_frame(cam, bg, t, upstream, near, far)

    def _settle_frame(self):
        """async_forward: the pending frame's status; a frame whose tile lists were truncated is run again (with the capacity the policy
        has regrown) until it is whole -- before anybody consumes its gradients."""
        attempts = 0
        while True:
            fr, self._frame = self._frame, None
            if fr is None:
                return
            fr.wait()
            self._policy.poll()                     # (non-strict: counts the invalid frame, regrows the capacity, learns the flow flag)
            if fr.valid:
                return
            if attempts == 3:                       # the frame and three re-runs with regrown capacities were all truncated
                raise RuntimeError(f"a frame stayed invalid after {attempts} re-runs (last: {fr.num_rendered} instances for a capacity of {fr.capacity})")
            attempts += 1
            self.replays += 1
            self._run_frame(*self._last_args)       # leaves the re-run's pending status in self._frame: validated by the next turn of the loop

    def _run_frame(self, cam, bg, t, upstream, near, far):
        m = self.model
        main = torch.cuda.current_stream(self.device)
        scal = attr.time_scalars(t, m.num_static, m.num_dynamic, m._xyz_motion.shape[1] if m.num_dynamic else 0,
                                 m.duration, m.interval, m.time_shift, m.var_pad)
        with torch.no_grad():
            xyz, rot, opa, scl, _ = attr.forward_raw(scal, self.params, with_shs=False)
        leaves = [x.requires_grad_(True) for x in (xyz, rot, opa, scl)]
        feats = [self.params[i].detach().requires_grad_(True) for i in self.feature_idx]
        means2D = torch.empty_like(xyz, requires_grad=True)
        dir3D = torch.zeros_like(xyz, requires_grad=True)
        e = torch.Tensor([])
        st = self._settings(cam, bg, near, far)
        if self.async_forward:
            from .diff_gaussian_rasterization_df import use_policy
            with use_policy(self._policy):
                color, radii, depth, flow, acc, idx = rasterize_gaussians(leaves[0], means2D, dir3D, SplitSH(*feats), e, leaves[2], leaves[3], leaves[1], e, st)
            self._frame = self._policy.pending[-1] if self._policy.pending else None      # (None: the policy's synchronous seed frame)
        else:
            color, radii, depth, flow, acc, idx = rasterize_gaussians(leaves[0], means2D, dir3D, SplitSH(*feats), e, leaves[2], leaves[3], leaves[1], e, st)
        out = {"render": color, "depth": depth, "opticalflow": flow, "acc": acc, "radii": radii, "dominent_idxs": idx,
               "viewspace_points": means2D, "viewspace_l1points": dir3D, "visibility_filter": radii > 0}
        outs, gouts = upstream(out)
        torch.autograd.backward(outs, gouts)
        gin = [x.grad for x in leaves]                    # dL/d(means3D, rotations, opacities, scales)
        fgrads = [f.grad for f in feats]
        # ---- attribute backward + exchange: side stream, overlapping the next frame's rasterization
        if self.side is not None:
            self.side.wait_stream(main)
            for g in gin + fgrads:
                g.record_stream(self.side)
            ctx = torch.cuda.stream(self.side)
        else:
            ctx = torch.cuda.stream(main)
        with ctx:
            if self.exchange_feat is not None and self.k == 1:
                self.exchange_feat.wait()
                self.exchange_feat.launch(fgrads)            # on the wire before the attribute backward runs
            if self.exchange is not None:
                self.exchange.wait()                       # previous frame's collectives own the persistent buffers until here
            for gth in self.kf_gather:
                gth.wait()
            gout = attr.backward_raw(scal, self.params, (gin[0], gin[1], gin[2], gin[3], None), with_shs=False, out=self.pgrad, sliced=self.sliced)
            windows = None
            if self.sliced:
                gout, hint = gout
                self._hint = hint
                for gth, i, first in zip(self.kf_gather, self.kf_idx, (hint[0], hint[2])):
                    gth.launch(gout[i], first)               # all-gather of this rank's window (a plain copy for one rank)
                if self.mode == "sharded":                   # row all-to-all of the windows (inside the sharded optimizer)
                    windows = {i: (gout[i], first) for i, first in zip(self.kf_idx, (hint[0], hint[2]))}
            grads = [fgrads[self.feature_idx.index(i)] if i in self.feature_idx else gout[i] for i in range(len(self.params))]
            if self.k > 1:
                # views 1 .. k-1 of the group: added to the accumulators, nothing goes on the wire, no optimizer step follows
                if self._acc is None:
                    self._acc = [torch.zeros_like(p) for p in self.params]
                if self._nacc == 0:
                    if self.exchange_feat is not None:
                        self.exchange_feat.wait()            # (the previous group's collectives own the accumulators until here)
                    for a, g_ in zip(self._acc, grads):
                        a.copy_(g_)
                else:
                    torch._foreach_add_(self._acc, list(grads))
                self._nacc += 1
                if self._nacc < self.k:
                    self._grads = None
                    self.last = {"radii": radii}
                    return out
                self._nacc = 0
                grads = self._acc
                if self.exchange_feat is not None:
                    self.exchange_feat.launch([grads[i] for i in self.feat_pos])
            self._grads = grads
            if self.exchange is not None:
                if self.exchange_feat is not None:
                    self.exchange.launch([grads[i] for i in self.rest_pos])
                else:                                      # sharded: one reduce-scatter over the dense tensors + the window all-to-all
                    self.opt.launch_exchange(grads, windows)
        self.last = {"radii": radii}
        return out

    def _apply_optimizer(self):
        if self.async_forward:
            self._settle_frame()
        self.finish_exchange()
        if self.side is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.side)
        if self.mode == "sharded":
            self.opt.step()                                # (its exchange was launched by _run_frame: launch_exchange(grads, windows))
        else:
            from .optim import radam_step_raw, radam_step_sliced_raw
            self.steps += 1
            items = [(p.data_ptr(), g.data_ptr(), mm.data_ptr(), vv.data_ptr(), p.numel(), lr, self.steps, int(self.names[i] in NAN_TO_NUM))
                     for i, (p, g, mm, vv, lr) in enumerate(zip(self.params, self._grads, self.m, self.v, self.lrs)) if i not in self.kf_idx]
            radam_step_raw(items, (0.9, 0.999), 1e-8, self.device)
            if self.sliced:
                sl = []
                for gth, i in zip(self.kf_gather, self.kf_idx):
                    p = self.params[i]
                    count, Cc = attr.SLICED_SHAPES[self.names[i]]
                    sl.append((p.data_ptr(), self.m[i].data_ptr(), self.v[i].data_ptr(), p.shape[0], p.shape[1], Cc, self.lrs[i], self.steps,
                               gth.windows(count), gth.first_device_ptr()))
                radam_step_sliced_raw(sl, (0.9, 0.999), 1e-8, self.device)
            torch.autograd.graph.increment_version(self.params)
        self._grads = None

    def flush(self):
        """Finish whatever is pending (exchange, optimizer step of the last frame) on the current stream."""
        if self.optimizer and self._grads is not None:
            self._apply_optimizer()
        else:
            if self.async_forward:
                self._settle_frame()
            self.finish_exchange()
            if self.side is not None:
                torch.cuda.current_stream(self.device).wait_stream(self.side)

    def grads(self):
        """The 15 (summed) parameter gradients of the last frame, valid after flush() when no optimizer consumed them."""
        return dict(zip(self.names, self._grads)) if self._grads is not None else None
