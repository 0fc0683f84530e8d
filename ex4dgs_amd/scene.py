"""Host-side inputs of the rasterizer hot path: cameras, the static+dynamic Gaussian model's
per-timestamp getters, and the deterministic synthetic scene generator of SURVEY.md 8(d).

Reference anchors (all under /root/reference, never imported at run time):
  * camera conventions  scene/cameras.py:119-127, utils/graphics_utils.py:45-117
  * per-frame getters   scene/c_gaussian_model.py:170-215, :330-375, utils/interpolations.py:33-93
Pure PyTorch (plumbing for the kernels); works on CPU and on ROCm devices.
"""
import math
from typing import NamedTuple

import numpy as np
import torch


# ----------------------------------------------------------------------------------------------
# Cameras
# ----------------------------------------------------------------------------------------------
def world_to_view(R, t, translate=(0.0, 0.0, 0.0), scale=1.0):
    """utils/graphics_utils.py:45-56 (getWorld2View2): 4x4 float32 world->camera matrix."""
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = np.asarray(R).transpose()
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    C2W = np.linalg.inv(Rt)
    C2W[:3, 3] = (C2W[:3, 3] + np.asarray(translate)) * scale
    return np.linalg.inv(C2W).astype(np.float32)


def projection_matrix(znear, zfar, fovX, fovY, cx=0.0, cy=0.0, cv=False):
    """utils/graphics_utils.py:58-78 (centred) and :81-117 (off-centre 'CV' variant, cv=True)."""
    tanY, tanX = math.tan(fovY / 2), math.tan(fovX / 2)
    top, right = tanY * znear, tanX * znear
    bottom, left = -top, -right
    if cv:
        dx, dy = (2 * tanX * znear) * cx, (2 * tanY * znear) * cy
        left += dx; right += dx; top += dy; bottom += dy
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = (zfar + znear) / (zfar - znear) if cv else zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


class Camera(NamedTuple):
    image_height: int
    image_width: int
    FoVx: float
    FoVy: float
    world_view_transform: torch.Tensor   # (W2C)^T, scene/cameras.py:119
    full_proj_transform: torch.Tensor    # (W2C)^T P^T, :125
    camera_center: torch.Tensor          # :126
    timestamp: float = 0.0

    def to(self, device):
        return self._replace(world_view_transform=self.world_view_transform.to(device),
                             full_proj_transform=self.full_proj_transform.to(device),
                             camera_center=self.camera_center.to(device))


def make_camera(width, height, FoVx, FoVy, R=None, T=None, znear=0.01, zfar=100.0, cxr=0.0, cyr=0.0, timestamp=0.0):
    """Cameravideo.__init__ matrix block, scene/cameras.py:119-126 (off-centre projection iff cyr != 0)."""
    R = np.eye(3) if R is None else np.asarray(R, dtype=np.float64)
    T = np.zeros(3) if T is None else np.asarray(T, dtype=np.float64)
    wvt = torch.tensor(world_to_view(R, T), dtype=torch.float32).transpose(0, 1)
    proj = projection_matrix(znear, zfar, FoVx, FoVy, cxr, cyr, cv=(cyr != 0.0)).transpose(0, 1)
    full = wvt.unsqueeze(0).bmm(proj.unsqueeze(0)).squeeze(0)
    center = wvt.inverse()[3, :3]
    return Camera(int(height), int(width), float(FoVx), float(FoVy), wvt.contiguous(), full.contiguous(), center.contiguous(), timestamp)


def focal_camera(width, height, focal, **kw):
    """Pinhole camera from a focal length in pixels (tanfov = size / (2 focal))."""
    return make_camera(width, height, 2 * math.atan(width / (2 * focal)), 2 * math.atan(height / (2 * focal)), **kw)


# ----------------------------------------------------------------------------------------------
# Static + keyframe-interpolated dynamic Gaussians (per-frame getters of CGaussianModel)
# ----------------------------------------------------------------------------------------------
def _cube_interp(y0, y1, y2, y3, d):
    """utils/interpolations.py:81-93: Catmull-Rom Hermite; the basis is evaluated as Python doubles."""
    h00 = 2 * d ** 3 - 3 * d ** 2 + 1
    h10 = d ** 3 - 2 * d ** 2 + d
    h01 = -2 * d ** 3 + 3 * d ** 2
    h11 = d ** 3 - d ** 2
    m_k = (y2 - y0) / 2
    m_k1 = (y3 - y1) / 2
    return h00 * y1 + h10 * m_k + h01 * y2 + h11 * m_k1


def _quat_slerp(qa, qb, frac):
    """Normalised spherical interpolation of two keyframe quaternions with the reference's guards
    (utils/interpolations.py:33-52): cos(angle) clamped to +-(1 - 1e-4), angle / sin / weight-sum floored at 1e-4,
    fall-back to the first keyframe when the blend vanishes.  Weights are the two sine ratios, renormalised to sum 1."""
    floor = 1e-4
    qa = qa / qa.norm(dim=-1, keepdim=True)
    qb = qb / qb.norm(dim=-1, keepdim=True)
    angle = torch.acos((qa * qb).sum(-1, keepdim=True).clamp(floor - 1, 1 - floor)).clamp_min(floor)
    inv_sin = torch.sin(angle).clamp_min(floor)
    wa, wb = torch.sin((1 - frac) * angle) / inv_sin, torch.sin(frac * angle) / inv_sin
    total = (wa + wb).clamp_min(floor)
    blend = qa * (wa / total) + qb * (wb / total)
    blend = torch.where(blend.abs().sum(-1, keepdim=True) > floor, blend, qa)
    return blend / blend.norm(dim=-1, keepdim=True)


def _time_bigaussian(centers, log_widths, tau, var_min):
    """Temporal opacity window (utils/interpolations.py:55-61): 1 between the two centres, a Gaussian fall-off outside with the
    left / right width exp(log_width) + var_min / 2.36; distance measured to the nearer... (min of the two signed offsets)."""
    offset = (tau - centers).min(dim=1)[0]
    past_any = (tau > centers).any(dim=1)
    width = torch.where(past_any, log_widths[:, 1], log_widths[:, 0]).exp() + var_min / 2.36
    falloff = torch.exp(-1 * (offset.pow(2) / width.pow(2)))
    between = (centers[:, 0] - tau) * (centers[:, 1] - tau) < 0
    return torch.where(between, torch.ones_like(falloff), falloff)


class DynamicGaussians:
    """Parameter container + the five per-frame getters the rasterizer boundary consumes
    (scene/c_gaussian_model.py:170-215, :330-375; 'cube' xyz interpolation, 'slerp' rotation).

    Parameter names and shapes follow CGaussianModel: static `_xyz[Ns,3] _xyz_disp[Ns,3] _rotation[Ns,4]
    _opacity[Ns,1] _scaling[Ns,3] _features_dc[Ns,1,3] _features_rest[Ns,15,3]`; dynamic
    `_xyz_motion[Nd,K,3] _rotation_motion[Nd,K,4] _opacity_motion[Nd,1] _opacity_duration_center[Nd,2,1]
    _opacity_duration_var[Nd,2,1] _scaling_motion[Nd,3] _features_dc_motion[Nd,1,3] _features_rest_motion[Nd,15,3]`.
    """
    PARAM_NAMES = ("_xyz", "_xyz_disp", "_rotation", "_opacity", "_scaling", "_features_dc", "_features_rest",
                   "_xyz_motion", "_rotation_motion", "_opacity_motion", "_opacity_duration_center",
                   "_opacity_duration_var", "_scaling_motion", "_features_dc_motion", "_features_rest_motion")

    def __init__(self, params, duration=300, interval=10, time_pad=2, var_pad=3, kernel_size=0.1, sh_degree=3, fused=False, split_sh=True):
        for n in self.PARAM_NAMES:
            setattr(self, n, params[n])
        # fused=True: the five getters are served by ONE fused HIP evaluation per (timestamp, parameter version)
        # (ex4dgs_amd.attributes) instead of ~15 torch kernels + 3 torch.cat copies
        self.fused = fused
        self.split_sh = split_sh        # with fused: get_features() hands the four feature tensors to the rasterizer, no [N,16,3] copy
        self._fused_key = None
        self._fused_out = None
        self.duration = max(duration, 1)
        self.interval = interval
        self.time_pad = time_pad
        self.time_shift = time_pad + interval      # c_gaussian_model.py:76,119 ('cube')
        self.var_pad = var_pad
        self.kernel_size = kernel_size
        self.active_sh_degree = sh_degree
        self.max_sh_degree = 3

    @staticmethod
    def keyframe_count(duration=300, interval=10, time_pad=2):
        """c_gaussian_model.py:1254: ceil((duration + time_shift + 2 time_pad + 1)/interval) + 3 (35 for N3V)."""
        return math.ceil((duration + time_pad + interval + 2 * time_pad + 1) / interval) + 3

    def parameters(self):
        return [getattr(self, n) for n in self.PARAM_NAMES]

    def zero_grad(self, set_to_none=True):
        for p in self.parameters():
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    @property
    def num_static(self):
        return self._xyz.shape[0]

    @property
    def num_dynamic(self):
        return self._xyz_motion.shape[0]

    def _tk(self, t):
        t = t + self.time_shift
        return int(t // self.interval), (t % self.interval) / self.interval

    def evaluate_at_t(self, t):
        """(means3D, rotations, opacities, scales, shs) through the fused HIP op; cached per (t, parameter versions) so the
        reference's render(), which calls the five getters one after the other, triggers a single evaluation."""
        from .attributes import evaluate_attributes
        key = (t, torch.is_grad_enabled()) + tuple((p.data_ptr(), p._version, p.requires_grad) for p in self.parameters())
        self._last_t = t
        if key != self._fused_key:
            # the cached outputs carry an autograd graph whose saved tensors are freed by the first backward through it: that
            # backward drops the cache (on_backward), so a later render at the same timestamp re-evaluates instead of failing with
            # "backward through the graph a second time" (several cameras per timestamp, gradient accumulation)
            self._fused_out = evaluate_attributes({n: getattr(self, n) for n in self.PARAM_NAMES}, t, duration=self.duration,
                                                  interval=self.interval, time_shift=self.time_shift, var_pad=self.var_pad,
                                                  with_shs=not self.split_sh, on_backward=self._drop_fused_cache)
            self._fused_key = key
        return self._fused_out

    def _drop_fused_cache(self):
        self._fused_key = None
        self._fused_out = None

    def _rows(self, x, mode):
        """mode 0 = all rows, 1 = static rows only, 2 = dynamic rows only (the `mode` argument of the reference's getters,
        scene/c_gaussian_model.py:170-176; static rows come first)."""
        if mode == 0:
            return x
        if mode not in (1, 2):
            raise ValueError(f"mode must be 0 (all), 1 (static) or 2 (dynamic), got {mode}")
        return x[: self.num_static] if mode == 1 else x[self.num_static:]

    def get_xyz_at_t(self, t, mode=0, training=True):
        if self.fused:
            return self._rows(self.evaluate_at_t(t)[0], mode)
        static = self._xyz + self._xyz_disp * t / self.duration               # :180
        if self.num_dynamic == 0 or mode == 1:
            return static
        k, d = self._tk(t)
        y = self._xyz_motion
        dyn = _cube_interp(y[:, k - 1, :3], y[:, k, :3], y[:, k + 1, :3], y[:, k + 2, :3], d)   # :118
        return dyn if mode == 2 else torch.cat([static, dyn], dim=0).contiguous()

    def get_rotation_at_t(self, t, mode=0):
        if self.fused:
            return self._rows(self.evaluate_at_t(t)[1], mode)
        if self.num_dynamic == 0 or mode == 1:
            return self._rotation                                             # :198 (raw, un-normalised)
        k, d = self._tk(t)
        y = self._rotation_motion
        dyn = _quat_slerp(y[:, k, :], y[:, k + 1, :], d)
        return dyn if mode == 2 else torch.cat([self._rotation, dyn], dim=0).contiguous()

    def get_opacity_at_t(self, t, mode=0, training=False):
        if self.fused:
            return self._rows(self.evaluate_at_t(t)[2], mode)
        static = torch.sigmoid(self._opacity)
        if self.num_dynamic == 0 or mode == 1:
            return static
        tau = (t + self.time_shift) / self.interval                           # :364
        o = _time_bigaussian(self._opacity_duration_center, self._opacity_duration_var, tau,
                             var_min=self.var_pad / self.interval) * torch.sigmoid(self._opacity_motion)
        return o if mode == 2 else torch.cat([static, o], dim=0).contiguous()

    def _fused_time_independent(self, index):
        # scales / features do not depend on t: reuse the evaluation of the timestamp the other getters asked for
        return self.evaluate_at_t(getattr(self, "_last_t", 0))[index]

    def get_scaling(self, mode=0):
        if self.fused:
            return self._rows(self._fused_time_independent(3), mode)
        if self.num_dynamic == 0 or mode == 1:
            return torch.exp(self._scaling)
        if mode == 2:
            return torch.exp(self._scaling_motion)
        return torch.exp(torch.cat([self._scaling, self._scaling_motion], dim=0))   # :335

    def get_features(self, mode=0):
        if self.fused and self.split_sh:
            # zero-copy: the rasterizer reads the four tensors where they are (diff_gaussian_rasterization_df.SplitSH)
            from ._C import SplitSH
            parts = [self._features_dc, self._features_rest, self._features_dc_motion, self._features_rest_motion]
            if mode == 1:
                parts[2:] = [parts[2][:0], parts[3][:0]]          # static rows only: an empty dynamic half
            elif mode == 2:
                parts[:2] = [parts[0][:0], parts[1][:0]]
            return SplitSH(*parts)
        if self.fused:
            return self._rows(self._fused_time_independent(4), mode)
        s = torch.cat((self._features_dc, self._features_rest), dim=1)
        if self.num_dynamic == 0 or mode == 1:
            return s
        d = torch.cat((self._features_dc_motion, self._features_rest_motion), dim=1)
        return d if mode == 2 else torch.cat((s, d), dim=0).contiguous()


# ----------------------------------------------------------------------------------------------
# Synthetic scene generator; deterministic in (config, seed), generated on CPU
# ----------------------------------------------------------------------------------------------
class SceneConfig(NamedTuple):
    name: str
    P: int
    width: int
    height: int
    focal: float
    dyn_frac: float = 0.0
    min_depth: float = 4.0
    max_depth: float = 300.0
    z_lo: float = 4.5
    z_hi: float = 80.0
    sigma_px_med: float = 2.5
    sigma_px_logstd: float = 0.7
    cxr: float = 0.0
    cyr: float = 0.0
    seed: int = 0
    clusters: int = 0          # > 0: positions from this many anisotropic clusters instead of uniform in the frustum (round 6, config 3c)
    plane_frac: float = 0.0    # ... and this fraction of the Gaussians on two slanted planes (large surfaces)


CONFIGS = {
    # BASELINE.json configs[0..4]
    "cfg1": SceneConfig("cfg1: 256 static, 256x256", 256, 256, 256, 140.0, z_lo=4.5, z_hi=30.0, sigma_px_med=6.0, seed=1),
    "cfg2": SceneConfig("cfg2: 100k static, 1352x1014", 100_000, 1352, 1014, 730.0, seed=2),
    "cfg3": SceneConfig("cfg3: 1.0M static+dynamic (K=35), 1352x1014", 1_000_000, 1352, 1014, 730.0, dyn_frac=0.2, seed=3),
    # round 6: config 3 with the statistics of a real capture instead of a uniform cloud -- ~200 anisotropic clusters
    # (tile lists ten times the mean in their centres), 10 % of the Gaussians on two slanted planes, heavy-tailed scales (log-std 1.2: a few
    # rects of hundreds of tiles), popular Gaussians in the clusters (row-atomic contention in the compositing backward)
    "cfg3c": SceneConfig("cfg3c: 1.0M clustered static+dynamic (K=35), 1352x1014", 1_000_000, 1352, 1014, 730.0, dyn_frac=0.2, seed=33,
                         sigma_px_logstd=1.2, clusters=200, plane_frac=0.1),
    "cfg4": SceneConfig("cfg4: 2.0M static+dynamic x 300 frames, 1352x1014", 2_000_000, 1352, 1014, 730.0, dyn_frac=0.2, seed=4),
    "cfg5": SceneConfig("cfg5: 1.0M deep-overlap, 2048x1088 off-centre", 1_000_000, 2048, 1088, 1100.0, min_depth=0.01,
                        z_lo=0.5, z_hi=40.0, sigma_px_med=5.5, sigma_px_logstd=0.8, cxr=0.02, cyr=-0.01, seed=5),
}


def make_scene(cfg, P=None, device="cpu", duration=300, fused=False, split_sh=True):
    """Returns (DynamicGaussians, Camera, bg[3]).  Distributions: SURVEY.md 8(d)."""
    if isinstance(cfg, str):
        cfg = CONFIGS[cfg]
    P = cfg.P if P is None else P
    g = torch.Generator().manual_seed(cfg.seed)
    N = lambda *s: torch.randn(*s, generator=g)
    U = lambda *s: torch.rand(*s, generator=g)
    Nd = int(round(P * cfg.dyn_frac))
    Ns = P - Nd
    tanx, tany = cfg.width / (2 * cfg.focal), cfg.height / (2 * cfg.focal)

    z = torch.exp(math.log(cfg.z_lo) + U(P) * (math.log(cfg.z_hi) - math.log(cfg.z_lo)))
    x = z * tanx * (U(P) * 2.3 - 1.15)
    y = z * tany * (U(P) * 2.3 - 1.15)
    xyz = torch.stack([x, y, z], -1)
    if cfg.clusters > 0:
        # cluster centres uniform in the frustum (log-uniform depth like the cloud above), every cluster an anisotropic Gaussian blob whose
        # axes are 2-15 % of its depth; the cluster of a Gaussian by a heavy-tailed popularity (a few clusters hold most of the points)
        C = cfg.clusters
        cz = torch.exp(math.log(cfg.z_lo * 1.3) + U(C) * (math.log(cfg.z_hi * 0.6) - math.log(cfg.z_lo * 1.3)))
        cc = torch.stack([cz * tanx * (U(C) * 1.8 - 0.9), cz * tany * (U(C) * 1.8 - 0.9), cz], -1)
        axes = cz.unsqueeze(-1) * (0.02 + 0.13 * U(C, 3))
        rotc = torch.linalg.qr(N(C, 3, 3))[0]
        pop = torch.exp(1.2 * N(C))
        which = torch.multinomial(pop / pop.sum(), P, replacement=True, generator=g)
        local = N(P, 3) * axes[which]
        xyz = cc[which] + torch.einsum("pij,pj->pi", rotc[which], local)
        n_plane = int(round(P * cfg.plane_frac))
        if n_plane > 0:
            # two slanted planes through the scene (a floor and a wall), points uniform on them
            half = n_plane // 2
            for k, (nrm, d0) in enumerate((((0.0, 0.8, -0.2), 0.9), ((0.7, 0.0, -0.35), 1.1))):
                m = half if k == 0 else n_plane - half
                zz = torch.exp(math.log(cfg.z_lo * 1.2) + U(m) * (math.log(cfg.z_hi * 0.7) - math.log(cfg.z_lo * 1.2)))
                if k == 0:      # floor: y grows with depth
                    xx = zz * tanx * (U(m) * 2.0 - 1.0); yy = zz * tany * d0 - 0.05 * zz * nrm[2]
                else:           # wall: x fixed fraction of the frustum, y free
                    yy = zz * tany * (U(m) * 2.0 - 1.0); xx = zz * tanx * 0.6 * d0 - 0.1 * zz
                sl = slice(P - n_plane + (0 if k == 0 else half), P - n_plane + (half if k == 0 else n_plane))
                xyz[sl] = torch.stack([xx, yy, zz], -1)
        xyz[:, 2] = xyz[:, 2].clamp(min=cfg.z_lo * 0.9)
        z = xyz[:, 2]
    sigma_px = torch.exp(math.log(cfg.sigma_px_med) + cfg.sigma_px_logstd * N(P))
    scale = (z * sigma_px / cfg.focal).unsqueeze(-1) * torch.exp(0.35 * N(P, 3))
    q = torch.nn.functional.normalize(N(P, 4), dim=-1) * (1 + 0.05 * N(P, 1))
    opacity_logit = 2.0 * N(P, 1)
    f_dc = N(P, 1, 3)
    f_rest = 0.15 * N(P, 15, 3)

    K = DynamicGaussians.keyframe_count(duration)
    params = dict(
        _xyz=xyz[:Ns], _xyz_disp=(0.002 * z[:Ns]).unsqueeze(-1) * N(Ns, 3), _rotation=q[:Ns], _opacity=opacity_logit[:Ns],
        _scaling=torch.log(scale[:Ns]), _features_dc=f_dc[:Ns], _features_rest=f_rest[:Ns])
    zd = z[Ns:]
    walk = torch.cumsum((0.01 * zd).view(Nd, 1, 1) * N(Nd, K, 3), dim=1)
    rot_walk = torch.nn.functional.normalize(q[Ns:].unsqueeze(1) + torch.cumsum(0.05 * N(Nd, K, 4), dim=1), dim=-1)
    centers = torch.sort(2 + U(Nd, 2, 1) * (K - 5), dim=1)[0]
    params.update(
        _xyz_motion=xyz[Ns:].unsqueeze(1) + walk, _rotation_motion=rot_walk, _opacity_motion=opacity_logit[Ns:],
        _opacity_duration_center=centers, _opacity_duration_var=N(Nd, 2, 1), _scaling_motion=torch.log(scale[Ns:]),
        _features_dc_motion=f_dc[Ns:], _features_rest_motion=f_rest[Ns:])
    params = {k: v.to(device=device, dtype=torch.float32).contiguous() for k, v in params.items()}
    model = DynamicGaussians(params, duration=duration, fused=fused, split_sh=split_sh)
    cam = focal_camera(cfg.width, cfg.height, cfg.focal, znear=0.01, zfar=100.0, cxr=cfg.cxr, cyr=cfg.cyr).to(device)
    bg = U(3).to(device)
    return model, cam, bg


def upstream_grads(acc, H, W, seed=0, grad_acc_zero=True, device="cpu"):
    """Synthetic dL/d(outputs) of SURVEY.md 8(d): grad_color ~ N(0,1), grad_depth ~ 0.1 N(0,1),
    grad_flow = stack(acc, |N|, U) (mirrors the train.py:149-152 hook), grad_acc = 0 or N(0,1)."""
    g = torch.Generator().manual_seed(1000 + seed)
    gc = torch.randn(3, H, W, generator=g).to(device)
    gd = (0.1 * torch.randn(1, H, W, generator=g)).to(device)
    gf = torch.stack([acc.reshape(H, W).detach().to(device), torch.randn(H, W, generator=g).abs().to(device),
                      torch.rand(H, W, generator=g).to(device)], 0)
    ga = torch.zeros(1, H, W, device=device) if grad_acc_zero else torch.randn(1, H, W, generator=g).to(device)
    return gc, gd, gf, ga
