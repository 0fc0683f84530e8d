"""Developer prototype (numpy, float32) of the MFMA compositing backward's arithmetic -- NOT part of the product or the
tests.  It replays, per 8x8 pixel quadrant, exactly what ex4d_composite.hip's composite_bwd_mfma_kernel computes:

  * batches of 16 list entries (descending from the quadrant's deepest contributor), lane = (entry n, pixel slot);
  * the per-pixel recurrences of CR/backward.cu:571-680 as 16-lane scans with a carry per pixel:
        T_i   = T_carry * prod_{j<=i} 1/(1-alpha_j)                      (inclusive prefix product, Hillis-Steele 1,2,4,8)
        E_i   = E_carry + sum_{j<i} alpha_j T_j (c_j . dL_dpixel)          (exclusive prefix sum) -- accum_rec in closed form:
                (c_i - accum_rec_i) . dL_dpixel * T_i = (c_i . dL_dpixel) T_i - E_i / (1 - alpha_i)
        gacc_i = gacc_carry * prod_{j<=i} T_j                              (dL_dacc compounding)
  * the 13 per-Gaussian sums as three contractions over the 64 pixels (the f32 MFMAs of the kernel):
        D1 = [gdepth, gp0..2, gflow0..2] . dcc      D2 = [1, x, y, xx, xy, yy] . sG      D3 = [1] . s6
    with x, y relative to the quadrant origin, converted to Gaussian-centred moments per batch.

and compares the result with the oracle's double-precision sums under the tolerance model of tests/helpers.py.

    python tools/dev/proto_bwd_scan.py [cfg] [P]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import helpers as h          # noqa: E402
from oracle import oracle               # noqa: E402

f32 = np.float32


def scan_mul(x):      # inclusive prefix product along axis 0 (16 entries), Hillis-Steele order, float32
    x = x.copy()
    for s in (1, 2, 4, 8):
        y = x.copy()
        y[s:] = x[s:] * x[:-s]
        x = y
    return x


def scan_add(x):
    x = x.copy()
    for s in (1, 2, 4, 8):
        y = x.copy()
        y[s:] = x[s:] + x[:-s]
        x = y
    return x


def run(cfg="cfg1", P=None, grad_acc_zero=False):
    ins, st = h.scene_inputs(cfg, P=P)
    o = h.oracle_forward(ins, st)
    H, W = o["H"], o["W"]
    grads = [g.numpy() for g in h.upstream_grads(torch.from_numpy(o["acc"]), H, W, seed=3, grad_acc_zero=grad_acc_zero)]
    ob = oracle.backward(o, *grads)
    gc, gd, gf, ga = grads
    Pn = o["P"]
    acc = np.zeros((Pn, 13), f32)
    gx = (W + 15) // 16
    mean2D, co, depths = o["means2D"], o["conic_opacity"], o["depths"]
    colors = o["rgb"]
    bg = o["_inputs"]["bg"]
    min_depth = f32(st["min_depth"])
    log2e = f32(1.4426950408889634)
    for tile in range(o["ranges"].shape[0]):
        r0, r1 = (int(v) for v in o["ranges"][tile])
        if r1 <= r0:
            continue
        for quad in range(4):
            px0 = (tile % gx) * 16 + (quad & 1) * 8
            py0 = (tile // gx) * 16 + (quad >> 1) * 8
            ys, xs = np.meshgrid(np.arange(py0, py0 + 8), np.arange(px0, px0 + 8), indexing="ij")
            ys, xs = ys.reshape(-1), xs.reshape(-1)
            inside = (xs < W) & (ys < H)
            yc, xc = np.minimum(ys, H - 1), np.minimum(xs, W - 1)
            lastc = np.where(inside, o["n_contrib"][yc, xc], 0).astype(np.int64)
            deepest = int(lastc.max())
            if deepest == 0:
                continue
            T_final = np.where(inside, o["final_T"][yc, xc], 0).astype(f32)
            accw = np.where(inside, o["acc"][0, yc, xc], 0).astype(f32)
            fd = np.where(inside, o["depth"][0, yc, xc], 0).astype(f32)
            pos = accw > 0
            safe = np.where(pos, accw, f32(1))
            gdepth = np.where(inside, gd[0, yc, xc], 0).astype(f32)
            gdepth = np.where(pos, gdepth / safe, gdepth).astype(f32)
            gp = np.where(inside[None], gc[:, yc, xc], 0).astype(f32)
            gflow = np.where((inside & pos)[None], gf[:, yc, xc] / safe[None], 0).astype(f32)
            gacc0 = np.where(inside & pos, ga[0, yc, xc], 0).astype(f32)
            bgT = (-T_final * (bg[0] * gp[0] + bg[1] * gp[1] + bg[2] * gp[2])).astype(f32)
            fx = xs.astype(f32); fy = ys.astype(f32)
            ox, oy = f32(px0), f32(py0)
            xr, yr = fx - ox, fy - oy
            A1 = np.stack([gdepth, gp[0], gp[1], gp[2], gflow[0], gflow[1], gflow[2]], 0)                # [7,64]
            A2 = np.stack([np.ones(64, f32), xr, yr, xr * xr, xr * yr, yr * yr], 0).astype(f32)           # [6,64]
            Tc = T_final.copy(); Ec = np.zeros(64, f32); gaccc = gacc0.copy()
            k_desc = np.arange(deepest - 1, -1, -1)
            for b0 in range(0, len(k_desc), 16):
                ks = k_desc[b0:b0 + 16]
                ids = o["point_list"][r0 + ks].astype(np.int64)
                n = len(ks)
                gxm, gym = mean2D[ids, 0], mean2D[ids, 1]
                ap = (co[ids, 0] * f32(-0.5) * log2e).astype(f32); bp = (co[ids, 1] * -log2e).astype(f32); cp = (co[ids, 2] * f32(-0.5) * log2e).astype(f32)
                w = co[ids, 3]
                dx = (gxm[:, None] - fx[None]).astype(f32); dy = (gym[:, None] - fy[None]).astype(f32)
                power2 = (dx * (ap[:, None] * dx + bp[:, None] * dy) + (cp[:, None] * dy) * dy).astype(f32)
                G = np.exp2(power2).astype(f32)
                alpha = np.minimum(f32(0.99), w[:, None] * G).astype(f32)
                ok = (ks[:, None] < lastc[None]) & (power2 <= 0) & ~(alpha < f32(1.0 / 255.0))
                alpha_m = np.where(ok, alpha, f32(0)); G_m = np.where(ok, G, f32(0))
                inv = (f32(1) / (f32(1) - alpha_m)).astype(f32)
                Tn = (scan_mul(np.concatenate([inv[:1] * Tc[None], inv[1:]], 0))).astype(f32)               # carry seeded into entry 0
                dcc = (alpha_m * Tn).astype(f32)
                cgp = (colors[ids, 0][:, None] * gp[0][None] + colors[ids, 1][:, None] * gp[1][None] + colors[ids, 2][:, None] * gp[2][None]).astype(f32)
                e = (dcc * cgp).astype(f32)
                Einc = scan_add(np.concatenate([e[:1] + Ec[None], e[1:]], 0)).astype(f32)
                Eexc = (Einc - e).astype(f32)
                flag = (depths[ids] > min_depth).astype(f32)
                u = ((fd[None] - depths[ids][:, None]) * (gdepth[None] * flag[:, None]) * Tn).astype(f32)
                col = (cgp * Tn - Eexc * inv).astype(f32)
                dLa = (u * Tn + col).astype(f32)
                dLa = (dLa + bgT[None] * inv).astype(f32)
                gaccn = scan_mul(np.concatenate([np.where(ok[:1], Tn[:1], f32(1)) * gaccc[None], np.where(ok[1:], Tn[1:], f32(1))], 0)).astype(f32)
                sG = ((w[:, None] * G_m) * dLa).astype(f32)
                s6 = (G_m * (dLa + gaccn)).astype(f32)
                Tc, Ec, gaccc = Tn[-1].copy(), Einc[-1].copy(), gaccn[-1].copy()
                D1 = (dcc @ A1.T).astype(f32)      # [n,7]
                D2 = (sG @ A2.T).astype(f32)       # [n,6]: M0 Mx My Mxx Mxy Myy
                D3 = s6.sum(1).astype(f32)
                dx0 = (gxm - ox).astype(f32); dy0 = (gym - oy).astype(f32)
                M0, Mx, My, Mxx, Mxy, Myy = (D2[:, i] for i in range(6))
                out = np.zeros((n, 13), f32)
                out[:, 0] = dx0 * M0 - Mx
                out[:, 1] = dy0 * M0 - My
                out[:, 2] = D1[:, 0] * flag
                out[:, 3] = (dx0 * dx0) * M0 - (f32(2) * dx0) * Mx + Mxx
                out[:, 4] = (dx0 * dy0) * M0 - dx0 * My - dy0 * Mx + Mxy
                out[:, 5] = (dy0 * dy0) * M0 - (f32(2) * dy0) * My + Myy
                out[:, 6] = D3
                out[:, 7:13] = D1[:, 1:7]
                np.add.at(acc, ids, out)
    # to reference units
    A, B, Cc = co[:, 0].astype(np.float64), co[:, 1].astype(np.float64), co[:, 2].astype(np.float64)
    a64 = acc.astype(np.float64)
    ref = a64.copy()
    ref[:, 0] = -(A * a64[:, 0] + B * a64[:, 1]) * (0.5 * W)
    ref[:, 1] = -(Cc * a64[:, 1] + B * a64[:, 0]) * (0.5 * H)
    ref[:, 3:6] = -0.5 * a64[:, 3:6]
    tol = 1e-5 + 64 * 2.0 ** -24 * ob["abs13"] + 3e-6 * np.abs(ob["sum13"])
    err = np.abs(ref - ob["sum13"])
    ratio = err / tol
    print(f"{cfg} P={Pn} R={o['num_rendered']}: worst err/tol per accumulator:", np.round(ratio.max(0), 3))
    print("   max abs err per accumulator:", np.array2string(err.max(0), precision=2))
    print("   worst overall", ratio.max(), "PASS" if ratio.max() <= 1 else "FAIL")
    return ratio.max()


if __name__ == "__main__":
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
    P = int(sys.argv[2]) if len(sys.argv) > 2 else None
    run(cfg, P)
    run(cfg, P, grad_acc_zero=True)
