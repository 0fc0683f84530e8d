"""Developer script (GPU box): prints the achieved parity numbers of the compositing-backward variants without asserting,
to calibrate / document the bounds written in tests/helpers.py.   python tools/dev/calib_parity.py [variants...]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import helpers as h        # noqa: E402
from oracle import oracle             # noqa: E402
from ex4dgs_amd import _C, build      # noqa: E402

build.build()
_C.load()
variants = [int(v) for v in sys.argv[1:]] or [4, 8]          # (the round-1 / round-2 variants 0 and 2 were removed in round 3)
cases = [("cfg1", None, 0), ("cfg2", 20000, 0), ("cfg3", 12000, 137), ("cfg5", 6000, 0), ("cfg2", None, 0)]
for cfg, P, t in cases:
    ins, st = h.scene_inputs(cfg, P=P, t=t)
    t0 = time.time()
    o = h.oracle_forward(ins, st)
    H, W = st["image_height"], st["image_width"]
    grads = list(h.upstream_grads(torch.from_numpy(o["acc"]), H, W, seed=3, grad_acc_zero=False))
    solid = torch.from_numpy(o["fragile"] > 1e-4)
    grads = [x * solid[None] for x in grads]
    ob_e2e = oracle.backward(o, *grads)
    print(f"== {cfg} P={o['P']} R={o['num_rendered']} fragile={int((~solid).sum())} px  (oracle {time.time() - t0:.1f} s)", flush=True)
    for v in variants:
        _C.set_option("composite_bwd_variant", v)
        g = h.gpu_forward_raw(ins, st)
        try:
            h.compare_forward(o, g)
        except AssertionError as e:
            print("   forward mismatch:", str(e)[:200])
        st2 = dict(o)
        st2.update(depth=h.to_np(g["depth"]), acc=h.to_np(g["acc"]), final_T=np.ascontiguousarray(h.to_np(g["final_T"])),
                   n_contrib=np.ascontiguousarray(h.to_np(g["n_contrib"]).astype(np.uint32)))
        ob = oracle.backward(st2, *grads)
        gb = h.gpu_backward_raw(ins, g, grads)
        for name, ref in (("gpu-state", ob), ("end-to-end", ob_e2e)):
            acc = h.acc16_in_reference_units(gb["acc16"], W, H, conic=o["conic_opacity"])[:, :13].astype(np.float64)
            tol = 1e-5 + 64 * 2.0 ** -24 * ref["abs13"] + 3e-6 * np.abs(ref["sum13"])
            err = np.abs(acc - ref["sum13"])
            ge = h.gradient_errors(ref, gb, o["P"])
            worst = max(r["rel_to_tensor_max"] for r in ge.values())
            print(f"   variant {v} {name:10s}: acc worst err/tol {float((err / tol).max()):.3f}; grads worst max-abs/tensor-max {worst:.2e}; "
                  + " ".join(f"{k[3:]}:{r['max_abs']:.1e}/{r['ref_max']:.1e}" for k, r in ge.items()), flush=True)
