import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests import helpers as h
from oracle import oracle
from ex4dgs_amd import _C
_C.load(); _C.set_option("binning_tile_ids", 1); _C.set_option("geom_debug_arrays", 1)
for cfg, P, ds in (("cfg1", None, 0.1),):
    ins, st = h.scene_inputs(cfg, P=P, dir_scale=ds)
    o = h.oracle_forward(ins, st)
    g = h.gpu_forward_raw(ins, st)
    H, W = st["image_height"], st["image_width"]
    gx = (W + 15) // 16
    ranges = g["ranges"].cpu().numpy().astype(np.int64); pl = g["point_list"].cpu().numpy().astype(np.int64)
    ql = g["qlist"].cpu().numpy().astype(np.int64); qc = g["qcount"].cpu().numpy().astype(np.int64)
    ncon = g["n_contrib"].cpu().numpy().astype(np.int64)
    bad = 0; tot = 0; miss = 0
    for t in range(ranges.shape[0]):
        r0, r1 = ranges[t]; n = r1 - r0
        ty, tx = divmod(t, gx)
        for q in range(4):
            qn = qc[t, q]
            ent = ql[4 * r0 + q * n: 4 * r0 + q * n + qn]
            tot += qn
            if qn:
                k = ent[:, 1]
                if not (np.all(np.diff(k) > 0) and k.max() < n and np.array_equal(ent[:, 0], pl[r0 + k])): bad += 1
            # every contributing position of the quadrant's pixels must be in the list up to the deepest contributor
            y0, x0 = ty * 16 + (q >> 1) * 8, tx * 16 + (q & 1) * 8
            sub = ncon[y0:y0 + 8, x0:x0 + 8]
            deepest = int(sub.max()) if sub.size else 0
            if deepest > 0 and (qn == 0 or ent[:, 1].max() < deepest - 1): miss += 1
    print(cfg, P, ds, "list entries", tot, "bad quadrants", bad, "quadrants whose list ends before the deepest contributor", miss, flush=True)
    grads = list(h.upstream_grads(torch.from_numpy(o["acc"]), H, W, seed=3))
    solid = torch.from_numpy(o["fragile"] > 1e-4)
    grads = [x * solid[None] for x in grads]
    st2 = dict(o); st2.update(depth=h.to_np(g["depth"]), acc=h.to_np(g["acc"]), final_T=np.ascontiguousarray(h.to_np(g["final_T"])), n_contrib=np.ascontiguousarray(h.to_np(g["n_contrib"]).astype(np.uint32)))
    ob = oracle.backward(st2, *grads)
    for rep in range(4):
        _C.set_option("debug_old_alpha_test", rep // 2)
        gb = h.gpu_backward_raw(ins, g, grads)
        acc = h.acc16_in_reference_units(gb["acc16"], W, H)[:, :13].astype(np.float64)
        err = np.abs(acc - ob["sum13"]); scale = np.maximum(1.0, np.abs(ob["sum13"]).max(0))
        tol = 1e-5 + 64 * 2.0 ** -24 * ob["abs13"] + 3e-6 * np.abs(ob["sum13"])
        iw = np.unravel_index((err / tol).argmax(), err.shape)
        print("    worst err/tol", float((err / tol).max()), "at", iw, "gpu", acc[iw], "oracle", ob["sum13"][iw], "abs13", ob["abs13"][iw], "row gpu", np.round(acc[iw[0]], 4), "row oracle", np.round(ob["sum13"][iw[0]], 4))
        print("  run", rep, "old test" if rep // 2 else "new test", "max err / column scale:", np.round(err.max(0) / scale, 6), "rows wrong:", int((err.max(1) > 1e-3 * np.maximum(1, np.abs(ob["sum13"]).max(1))).sum()), "of", int((o["radii"] > 0).sum()))
