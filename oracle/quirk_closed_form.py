"""Closed form of the reference's compositing backward INCLUDING its non-derivative terms (TEST INFRASTRUCTURE ONLY).

Written from the operation-level specification (i.e. CR/backward.cu:505-535 pre-scaling, :571-680 per-pair
terms; CR/ = submodules/diff_gaussian_rasterization_df/cuda_rasterizer/) as dense float64 array algebra over the pixels x Gaussians
tensors of oracle/oracle_torch.py -- NOT from oracle/ex4d_oracle.c, whose sequential replay (running T divided back, accum_rec /
last_alpha / last_color recurrences, dL_dacc compounded in place) it cross-checks.  Covers what torch autograd cannot
(tests/test_cpu_oracle_and_host.py): the depth term and its `dep > min_depth` gate (:603-613, :621), dL_dacc compounding into the
opacity gradient (:650, :679), the alpha-clamp pass-through (:588, :662), the flow channel (:640-647) and the pre-scaling by acc.

Per pixel p, list entries j in front-to-back order, U = contributes in the backward (position < n_contrib, power <= 0, alpha >= 1/255):
    T_j      transmittance in front of j = prod_{k<j, U} (1 - alpha_k)              (the reference divides T_final back: same value)
    wgt_j    = alpha_j T_j
    S_j      = sum_{k>j, U} c_k wgt_k                 accum_rec_j = S_j / (T_j (1 - alpha_j))        (blend of everything behind j)
    dA_j     = T_j [ gate_j (final_depth - depth_j) gd T_j + sum_ch (c_j - accum_rec_j) g_pix ] - T_final / (1 - alpha_j) (bg . g_pix)
    ga_j     = ga prod_{k>=j, U} T_k                                                 (dL_dacc *= T at every contributor, back to front)
with gd = grad_depth / acc, gf = grad_flow / acc, ga = grad_acc where acc > 0, else grad_depth, 0, 0.
"""
import numpy as np


def accumulators(dense, *, final_T, acc, final_depth, bg, grad_color, grad_depth, grad_flow, grad_acc, W, H, min_depth, n_visible_total):
    """dense: oracle_torch.rasterize(..., return_dense=True)['dense'].  Returns [P,13] float64 in the index convention of
    ex4d_oracle_render_bwd (0..2 dL_dmean2D, 3..5 dL_dconic.(x,y,w), 6 dL_dopacity, 7..9 dL_dcolor, 10..12 dL_ddir)."""
    f = lambda t: t.detach().numpy().astype(np.float64)
    U = dense["use"].numpy()
    order = dense["order"].numpy()
    al, G, dx, dy = (np.where(U, f(dense[k]), 0.0) for k in ("alpha", "G", "dx", "dy"))
    cn, w, rgb, dep = f(dense["conic"]), f(dense["w"]), f(dense["rgb"]), f(dense["depth"])
    HW, n = U.shape
    fT = np.asarray(final_T, np.float64).reshape(HW, 1)
    ac = np.asarray(acc, np.float64).reshape(HW, 1)
    fd = np.asarray(final_depth, np.float64).reshape(HW, 1)
    gp = np.asarray(grad_color, np.float64).reshape(3, HW).T                      # [HW,3]
    pos = ac > 0
    gd = np.where(pos, np.asarray(grad_depth, np.float64).reshape(HW, 1) / np.where(pos, ac, 1.0), np.asarray(grad_depth, np.float64).reshape(HW, 1))
    gf = np.where(pos, np.asarray(grad_flow, np.float64).reshape(3, HW).T / np.where(pos, ac, 1.0), 0.0)
    ga = np.where(pos, np.asarray(grad_acc, np.float64).reshape(HW, 1), 0.0)

    one_m = 1.0 - al                                                              # 1 where not contributing
    T = np.concatenate([np.ones((HW, 1)), np.cumprod(one_m, 1)[:, :-1]], 1)       # in front of j
    wgt = al * T
    cw = wgt[:, :, None] * rgb[None]                                              # [HW,n,3]
    S = np.flip(np.cumsum(np.flip(cw, 1), 1), 1) - cw                             # strictly behind j
    accum_rec = S / (T * one_m)[:, :, None]
    gate = ((dep[None] > min_depth) & (wgt > 0.0)).astype(np.float64)
    colour = ((rgb[None] - accum_rec) * gp[:, None, :]).sum(2)
    bgdot = (gp * np.asarray(bg, np.float64)[None]).sum(1, keepdims=True)
    dA = T * (gate * (fd - dep[None]) * gd * T + colour) - fT / one_m * bgdot
    # dL_dacc compounding: product of T over the contributors at and behind j
    Tu = np.where(U, T, 1.0)
    ga_j = ga * np.flip(np.cumprod(np.flip(Tu, 1), 1), 1)
    dG = w[None] * dA
    gdx, gdy = G * dx, G * dy
    A_, B_, C_ = cn[:, 0][None], cn[:, 1][None], cn[:, 2][None]
    terms = np.zeros((HW, n, 13))
    terms[:, :, 0] = dG * (-gdx * A_ - gdy * B_) * (0.5 * W)
    terms[:, :, 1] = dG * (-gdy * C_ - gdx * B_) * (0.5 * H)
    terms[:, :, 2] = gate * gd * wgt
    terms[:, :, 3] = -0.5 * gdx * dx * dG
    terms[:, :, 4] = -0.5 * gdx * dy * dG
    terms[:, :, 5] = -0.5 * gdy * dy * dG
    terms[:, :, 6] = G * dA + G * ga_j
    terms[:, :, 7:10] = wgt[:, :, None] * gp[:, None, :]
    terms[:, :, 10:13] = wgt[:, :, None] * gf[:, None, :]
    terms *= U[:, :, None]
    out = np.zeros((n_visible_total, 13))
    out[order] = terms.sum(0)
    return out
