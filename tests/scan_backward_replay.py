"""Numpy (float32) replay of the arithmetic of the scan compositing backward, ex4d_composite.hip: composite_bwd_scan_kernel -- used by
tests/test_cpu_oracle_and_host.py as a CPU-side check of the reformulations against the oracle's double-precision sums (the -m gpu
tests check the kernel itself).  Per 8x8 pixel quadrant it does what the kernel does:

  * batches of 16 list entries (descending from the quadrant's deepest contributor), lane = (entry n, pixel slot);
  * the per-pixel recurrences of CR/backward.cu:571-680 as 16-lane scans with a carry per pixel:
        T_i    = T_carry * prod_{j<=i} 1/(1-alpha_j)                      (inclusive prefix product, Hillis-Steele 1,2,4,8)
        Q_i    = Q_carry - sum_{j<=i} alpha_j T_j (c_j . dL_dpixel)        (Q = bgT - E: the background term folded into the carry)
        dL_dalpha (colour + background) = inv_i ((c_i . dL_dpixel) T_i + Q_i)
        gacc_i = gacc_carry * prod_{j<=i} T_j                              (dL_dacc compounding)
  * the exponent as q2 = -power log2(e) = fma(dx, fma(dx, -a', -b' dy), -(c' dy) dy) (the forward's contraction with every operand
    negated), pixel-row terms hoisted; the reference's two skips (power > 0, alpha < 1/255) as ONE unsigned compare of bit patterns
    bits(q2) <= bits(log2(255 w)) (round 3);
  * the position moments accumulated per STEP PARITY and per pixel-slot lane (dx is constant over the even and over the odd steps of a
    batch), over s6 = G dL_dalpha, the factor w of sG = w s6 applied ONCE per batch and lane (round 3):
        S_e/o = sum s6,  Y_e/o = sum s6 dy,  V = sum s6 dy^2   ->   sum sG dx = w (dxe S_e + dxo S_o), etc.;
    the conic mix of dL_dmean2D per lane, then the four pixel-slot lanes of a Gaussian added.

    python tests/scan_backward_replay.py [cfg] [P]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
            Tc = T_final.copy(); Qc = bgT.copy(); gaccc = gacc0.copy()
            k_desc = np.arange(deepest - 1, -1, -1)
            for b0 in range(0, len(k_desc), 16):
                ks = k_desc[b0:b0 + 16]
                ids = o["point_list"][r0 + ks].astype(np.int64)
                n = len(ks)
                gxm, gym = mean2D[ids, 0], mean2D[ids, 1]
                ap = (co[ids, 0] * f32(-0.5) * log2e).astype(f32); bp = (co[ids, 1] * -log2e).astype(f32); cp = (co[ids, 2] * f32(-0.5) * log2e).astype(f32)
                w = co[ids, 3]
                dx = (gxm[:, None] - fx[None]).astype(f32); dy = (gym[:, None] - fy[None]).astype(f32)
                fma = lambda a_, b_, c_: (a_.astype(np.float64) * b_.astype(np.float64) + c_.astype(np.float64)).astype(f32)   # one rounding
                bdy = (bp[:, None] * dy).astype(f32); cdydy = ((cp[:, None] * dy).astype(f32) * dy).astype(f32)
                q2 = fma(dx, fma(dx, np.broadcast_to(-ap[:, None], dx.shape), -bdy), -cdydy)          # = -power2, bit for bit
                G = np.exp2(-q2).astype(f32)
                alpha = np.minimum(f32(0.99), w[:, None] * G).astype(f32)
                with np.errstate(divide="ignore", invalid="ignore"):
                    tauq = np.log2((f32(255.0) * w).astype(f32)).astype(f32)
                # one unsigned compare: a negative q2 (power > 0) has its sign bit set and fails; q2 > tauq <=> w 2^-q2 < 1/255.  Gaussians with
                # w < 1/255 (tauq < 0) never reach the kernel: the forward's cull drops them before staging (ex4d_preprocess.hip: tau = -inf)
                ok = ((ks[:, None] < lastc[None]) & (w[:, None] >= f32(1.0 / 255.0))
                      & (np.ascontiguousarray(q2).view(np.uint32) <= np.ascontiguousarray(tauq).view(np.uint32)[:, None]))
                alpha_m = np.where(ok, alpha, f32(0)); G_m = np.where(ok, G, f32(0))
                inv = (f32(1) / (f32(1) - alpha_m)).astype(f32)
                Tn = (Tc[None] * scan_mul(inv)).astype(f32)                     # row-uniform carry times the identity-seeded scan
                dcc = (alpha_m * Tn).astype(f32)
                cgp = (colors[ids, 0][:, None] * gp[0][None] + colors[ids, 1][:, None] * gp[1][None] + colors[ids, 2][:, None] * gp[2][None]).astype(f32)
                e = (dcc * cgp).astype(f32)
                Q = (Qc[None] - scan_add(e)).astype(f32)                        # bgT - E (inclusive)
                flag = (depths[ids] > min_depth).astype(f32)
                depflag = (depths[ids] * flag).astype(f32)
                dLa = ((cgp * Tn + Q) * inv).astype(f32)
                gdT = (gdepth[None] * Tn).astype(f32)
                dLa = (dLa + ((fd[None] * flag[:, None] - depflag[:, None]) * gdT) * Tn).astype(f32)
                gaccn = (gaccc[None] * scan_mul(np.where(ok, Tn, f32(1)))).astype(f32)
                s6 = (G_m * dLa).astype(f32)
                s6g = (G_m * gaccn).astype(f32)
                Tc, Qc, gaccc = Tn[-1].copy(), Q[-1].copy(), gaccn[-1].copy()
                # pixel p = 4 s + g: step parity = (p >> 2) & 1; dx is one value on the even steps and one on the odd steps (per pixel slot g)
                par = ((np.arange(64) >> 2) & 1).astype(bool)
                out = np.zeros((n, 13), f32)
                for g_ in range(4):                                              # one pixel-slot lane of every Gaussian
                    slot = (np.arange(64) & 3) == g_
                    SY = {}
                    for odd in (False, True):
                        m = slot & (par == odd)
                        SY[odd] = (s6[:, m].sum(1, dtype=f32), (s6[:, m] * dy[:, m]).sum(1, dtype=f32), dx[:, m][:, 0])     # S, Y, the 8 pixels' dx
                    (Se, Ye, dxe), (So, Yo, dxo) = SY[False], SY[True]
                    V = (s6[:, slot] * (dy[:, slot] * dy[:, slot]).astype(f32)).sum(1, dtype=f32)
                    m0 = ((dxe * Se + dxo * So).astype(f32) * w).astype(f32)
                    m1 = ((Ye + Yo).astype(f32) * w).astype(f32)
                    out[:, 3] += (((dxe * dxe) * Se + (dxo * dxo) * So).astype(f32) * w).astype(f32)
                    out[:, 4] += ((dxe * Ye + dxo * Yo).astype(f32) * w).astype(f32)
                    out[:, 5] += (w * V).astype(f32)
                    out[:, 0] += ((f32(2) * ap) * m0 + bp * m1).astype(f32)
                    out[:, 1] += ((f32(2) * cp) * m1 + bp * m0).astype(f32)
                out[:, 2] = (alpha_m * gdT).sum(1, dtype=f32) * flag
                out[:, 6] = (s6 + s6g).sum(1, dtype=f32)
                out[:, 7:10] = (dcc @ gp.T).astype(f32)
                out[:, 10:13] = (dcc @ gflow.T).astype(f32)
                np.add.at(acc, ids, out)
    # accumulator layout 0 -> reference units (factors the per-Gaussian backward kernel applies: ln2 W/2, ln2 H/2, -1/2)
    a64 = acc.astype(np.float64)
    ref = a64.copy()
    ln2 = 0.6931471805599453
    ref[:, 0] = a64[:, 0] * (ln2 * 0.5 * W)
    ref[:, 1] = a64[:, 1] * (ln2 * 0.5 * H)
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
