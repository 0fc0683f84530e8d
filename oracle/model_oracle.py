"""CPU oracle (numpy, float32 op-for-op) of the per-frame attribute evaluation of the reference's CGaussianModel
and of its backward -- TEST INFRASTRUCTURE ONLY (tests/, smoke, bench cpu legs).

Restates scene/c_gaussian_model.py:170-215 (get_xyz_at_t / get_rotation_at_t), :330-375 (get_scaling,
get_features, get_opacity_at_t) and utils/interpolations.py:33-61, :81-93 (quat_slerp_interp_uniiterval,
time_bigaussian, cube_interpolate) of /root/reference.  Pinned by tests/golden/model_getters.npz, which holds
outputs AND autograd gradients of the imported reference model for t in {0, 7, 137, 290, 299}.
The backward is the hand-derived adjoint of exactly these expressions (what torch autograd computes in the
reference); clamp gradients pass where the input is inside the closed interval, min() sends the gradient to
the first minimum.
"""
import math

import numpy as np

f32 = np.float32


def _time_index(t, time_shift, interval):
    tp = t + time_shift                          # Python numbers, c_gaussian_model.py:184-186
    return int(tp // interval), (tp % interval) / interval


def hermite_weights(delta):
    """Weights of (y_{k-1}, y_k, y_{k+1}, y_{k+2}) for p = h00 y1 + h10 (y2-y0)/2 + h01 y2 + h11 (y3-y1)/2; the basis is
    evaluated in Python doubles and enters the tensor arithmetic as float32 scalars (interpolations.py:81-93)."""
    d = delta
    h00 = 2 * d ** 3 - 3 * d ** 2 + 1
    h10 = d ** 3 - 2 * d ** 2 + d
    h01 = -2 * d ** 3 + 3 * d ** 2
    h11 = d ** 3 - d ** 2
    return f32(h00), f32(h10), f32(h01), f32(h11)


def forward(p, t, duration=300, interval=10, time_shift=12, var_pad=3):
    """p: dict of float32 arrays named like CGaussianModel's parameters.  Returns dict(means3D, rotations, opacities,
    scales, shs) with static rows first (c_gaussian_model.py:193,215,374,335,351-353)."""
    Ns, Nd = p["_xyz"].shape[0], p["_xyz_motion"].shape[0]
    out = {}
    tt = f32(t)
    static_xyz = p["_xyz"] + (p["_xyz_disp"] * tt) / f32(max(duration, 1))               # :180
    static_op = f32(1) / (f32(1) + np.exp(-p["_opacity"]))                                # sigmoid
    if Nd == 0:
        out["means3D"] = static_xyz
        out["rotations"] = p["_rotation"].copy()
        out["opacities"] = static_op
        out["scales"] = np.exp(p["_scaling"])
        out["shs"] = np.concatenate([p["_features_dc"], p["_features_rest"]], 1)
        return out
    k, delta = _time_index(t, time_shift, interval)
    h00, h10, h01, h11 = hermite_weights(delta)
    y = p["_xyz_motion"]
    y0, y1, y2, y3 = y[:, k - 1], y[:, k], y[:, k + 1], y[:, k + 2]
    m_k = (y2 - y0) / f32(2)
    m_k1 = (y3 - y1) / f32(2)
    dyn_xyz = h00 * y1 + h10 * m_k + h01 * y2 + h11 * m_k1
    out["means3D"] = np.concatenate([static_xyz, dyn_xyz], 0)

    q = p["_rotation_motion"]
    out["rotations"] = np.concatenate([p["_rotation"], slerp(q[:, k], q[:, k + 1], f32(delta))["out"]], 0)

    tau = f32((t + time_shift) / interval)                                                # :364
    big = bigaussian(p["_opacity_duration_center"], p["_opacity_duration_var"], tau, f32(var_pad / interval))["out"]
    dyn_op = big * (f32(1) / (f32(1) + np.exp(-p["_opacity_motion"])))
    out["opacities"] = np.concatenate([static_op, dyn_op], 0)
    out["scales"] = np.exp(np.concatenate([p["_scaling"], p["_scaling_motion"]], 0))
    out["shs"] = np.concatenate([np.concatenate([p["_features_dc"], p["_features_rest"]], 1),
                                 np.concatenate([p["_features_dc_motion"], p["_features_rest_motion"]], 1)], 0)
    return out


def slerp(q1, q2, t):
    """interpolations.py:33-52; returns every intermediate the backward needs."""
    n1 = np.sqrt((q1 * q1).sum(-1, keepdims=True)); n2 = np.sqrt((q2 * q2).sum(-1, keepdims=True))
    v1, v2 = q1 / n1, q2 / n2
    raw = (v1 * v2).sum(-1, keepdims=True)
    lo, hi = f32(-1 + 1e-4), f32(1 - 1e-4)
    d = np.clip(raw, lo, hi)
    ac = np.arccos(d)
    omega = np.maximum(ac, f32(1e-4))
    sn = np.sin(omega)
    s = np.maximum(sn, f32(1e-4))
    p0 = np.sin((f32(1) - t) * omega) / s
    p1 = np.sin(t * omega) / s
    ps_raw = p0 + p1
    psum = np.maximum(ps_raw, f32(1e-4))
    p0n, p1n = p0 / psum, p1 / psum
    r_mix = v1 * p0n + v2 * p1n
    fallback = ~(np.abs(r_mix).sum(-1, keepdims=True) > f32(1e-4))
    r = np.where(fallback, v1, r_mix)
    nr = np.sqrt((r * r).sum(-1, keepdims=True))
    return dict(out=r / nr, n1=n1, n2=n2, v1=v1, v2=v2, raw=raw, d=d, ac=ac, omega=omega, sn=sn, s=s, p0=p0, p1=p1,
                ps_raw=ps_raw, psum=psum, p0n=p0n, p1n=p1n, fallback=fallback, r=r, nr=nr, lo=lo, hi=hi)


def slerp_backward(c, t, G):
    out = c["r"] / c["nr"]
    g_r = (G - out * (out * G).sum(-1, keepdims=True)) / c["nr"]
    fb = c["fallback"]
    g_v1 = np.where(fb, g_r, g_r * c["p0n"])
    g_v2 = np.where(fb, 0, g_r * c["p1n"]).astype(f32)
    g_p0n = np.where(fb, 0, (g_r * c["v1"]).sum(-1, keepdims=True)).astype(f32)
    g_p1n = np.where(fb, 0, (g_r * c["v2"]).sum(-1, keepdims=True)).astype(f32)
    g_p0 = g_p0n / c["psum"]
    g_p1 = g_p1n / c["psum"]
    g_psum = -(g_p0n * c["p0"] + g_p1n * c["p1"]) / (c["psum"] * c["psum"])
    pass_ps = c["ps_raw"] >= f32(1e-4)
    g_p0 = g_p0 + np.where(pass_ps, g_psum, 0)
    g_p1 = g_p1 + np.where(pass_ps, g_psum, 0)
    om, s = c["omega"], c["s"]
    g_om = g_p0 * (f32(1) - t) * np.cos((f32(1) - t) * om) / s + g_p1 * t * np.cos(t * om) / s
    g_s = -(g_p0 * c["p0"] + g_p1 * c["p1"]) / s
    g_om = g_om + np.where(c["sn"] >= f32(1e-4), g_s * np.cos(om), 0)
    g_d = np.where(c["ac"] >= f32(1e-4), -g_om / np.sqrt(f32(1) - c["d"] * c["d"]), 0)
    g_raw = np.where((c["raw"] >= c["lo"]) & (c["raw"] <= c["hi"]), g_d, 0)
    g_v1 = g_v1 + g_raw * c["v2"]
    g_v2 = g_v2 + g_raw * c["v1"]
    g_q1 = (g_v1 - c["v1"] * (c["v1"] * g_v1).sum(-1, keepdims=True)) / c["n1"]
    g_q2 = (g_v2 - c["v2"] * (c["v2"] * g_v2).sum(-1, keepdims=True)) / c["n2"]
    return g_q1.astype(f32), g_q2.astype(f32)


def bigaussian(mean, var, tau, var_min):
    """interpolations.py:55-61 with mean/var [Nd,2,1] and scalar tau; returns out [Nd,1] and intermediates."""
    diff = tau - mean                                   # [Nd,2,1]
    m = diff.min(axis=1)                                # [Nd,1]
    arg = diff.argmin(axis=1)                           # first minimum
    after = (tau > mean).any(axis=1)                    # [Nd,1]
    v = np.where(after, var[:, 1], var[:, 0])
    D = np.exp(v) + var_min / f32(2.36)
    u = (m * m) / (D * D)
    o = np.exp(f32(-1) * u)
    inside = (mean[:, 0] - tau) * (mean[:, 1] - tau) < 0
    return dict(out=np.where(inside, f32(1), o).astype(f32), m=m, arg=arg, after=after, v=v, D=D, u=u, o=o, inside=inside)


def backward(p, t, grads, duration=300, interval=10, time_shift=12, var_pad=3):
    """grads: dict(means3D, rotations, opacities, scales, shs) of float32 arrays.  Returns gradients for every parameter."""
    Ns, Nd = p["_xyz"].shape[0], p["_xyz_motion"].shape[0]
    g = {k: np.zeros_like(v) for k, v in p.items()}
    gm, gr, go, gs, gf = (grads[k] for k in ("means3D", "rotations", "opacities", "scales", "shs"))
    g["_xyz"] = gm[:Ns].copy()
    g["_xyz_disp"] = (gm[:Ns] / f32(max(duration, 1))) * f32(t)
    g["_rotation"] = gr[:Ns].copy()
    so = f32(1) / (f32(1) + np.exp(-p["_opacity"]))
    g["_opacity"] = go[:Ns] * (so * (f32(1) - so))
    g["_scaling"] = gs[:Ns] * np.exp(p["_scaling"])
    g["_features_dc"] = gf[:Ns, :1].copy()
    g["_features_rest"] = gf[:Ns, 1:].copy()
    if Nd == 0:
        return g
    k, delta = _time_index(t, time_shift, interval)
    h00, h10, h01, h11 = hermite_weights(delta)
    G = gm[Ns:]
    g["_xyz_motion"][:, k - 1] = -(h10 * G) / f32(2)
    g["_xyz_motion"][:, k] = h00 * G - (h11 * G) / f32(2)
    g["_xyz_motion"][:, k + 1] = (h10 * G) / f32(2) + h01 * G
    g["_xyz_motion"][:, k + 2] = (h11 * G) / f32(2)
    q = p["_rotation_motion"]
    c = slerp(q[:, k], q[:, k + 1], f32(delta))
    g["_rotation_motion"][:, k], g["_rotation_motion"][:, k + 1] = slerp_backward(c, f32(delta), gr[Ns:])
    tau = f32((t + time_shift) / interval)
    b = bigaussian(p["_opacity_duration_center"], p["_opacity_duration_var"], tau, f32(var_pad / interval))
    sg = f32(1) / (f32(1) + np.exp(-p["_opacity_motion"]))
    Go = go[Ns:]
    g["_opacity_motion"] = Go * b["out"] * (sg * (f32(1) - sg))
    g_big = np.where(b["inside"], 0, Go * sg).astype(f32)
    g_u = -g_big * b["o"]
    g_m = g_u * f32(2) * b["m"] / (b["D"] * b["D"])
    g_D = -f32(2) * b["u"] / b["D"] * g_u
    g_v = g_D * np.exp(b["v"])
    rows = np.arange(Nd)
    g["_opacity_duration_center"][rows, b["arg"][:, 0], 0] = -g_m[:, 0]
    g["_opacity_duration_var"][rows, b["after"][:, 0].astype(np.int64), 0] = g_v[:, 0]
    g["_scaling_motion"] = gs[Ns:] * np.exp(p["_scaling_motion"])
    g["_features_dc_motion"] = gf[Ns:, :1].copy()
    g["_features_rest_motion"] = gf[Ns:, 1:].copy()
    return g
