#!/usr/bin/env python
"""CPU calibration of the noise-floor comparison (tests/helpers.py: compare_with_noise) without a GPU: a stand-in for "another float32
implementation of the same backward" is the oracle source rebuilt with FMA contraction (what nvcc does to the reference), run on the
plain oracle's forward state; it is compared with the exact sums in units of the reference's replayed atomics-order spread."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import helpers as h
from oracle import oracle
sys.path.insert(0, os.path.join(ROOT, "tools"))
from fma_sensitivity import load_fma

cfg, P = sys.argv[1], (int(sys.argv[2]) if len(sys.argv) > 2 else None)
ins, st = h.scene_inputs(cfg, P=P, t=137 if cfg in ("cfg3", "cfg4") else 0)
plain = oracle.lib()
o = h.oracle_forward(ins, st)
H, W = st["image_height"], st["image_width"]
grads = list(h.upstream_grads(torch.from_numpy(o["acc"]), H, W, seed=3))
solid = torch.from_numpy(o["fragile"] > 1e-4)
grads = [x * solid[None] for x in grads]
t0 = time.time(); ob = oracle.backward(o, *grads); t1 = time.time()
noise = oracle.backward_noise(o, *grads, orders=8); t2 = time.time()
print(f"P={o['P']} R={o['num_rendered']} backward {t1 - t0:.1f}s, 8 replays {t2 - t1:.1f}s")
fma = load_fma()
oracle._LIB = fma
obf = oracle.backward(o, *grads, want_sums=False)
oracle._LIB = plain
gb = {k: obf[k] for k in h.GRAD_NAMES}
ref = h._stage(o, ob["sum13"])
ref["dL_dmeans2D"], ref["dL_dcolors"] = ob["sum13"][:, 0:3], ob["sum13"][:, 7:10]
ref["dL_dopacity"], ref["dL_ddir"] = ob["sum13"][:, 6:7], ob["sum13"][:, 10:13]
dev = h.noise_floor(o, noise, ob["sum13"])
eps = 2.0 ** -24
floor13 = h.NOISE_FLOOR_EPS * eps * ob["abs13"]
fl = h.propagated_tolerance(o, floor13)
rep = {}
try:
    h.compare_with_noise(rep, ref, gb, dev, o["P"], floor_acc=floor13, floor_derived={k: fl[k] for k in h.DERIVED}, assert_rows=False)
except AssertionError as e:
    print("ASSERT:", e)
for k, r in rep["noise_floor"].items():
    print(f"{k:14s} err/noise {r['err_over_ref_noise']:.2f}  row max {r['row_ratio_max']:.2f} p999 {r['row_ratio_p999']:.2f} noise-only p999 {r['row_ratio_noise_only_p999']:.2f} rows>c {r['rows_above_c']}/{r['rows']}")
# the plain oracle's own row-major float32 result is one more draw of the same distribution
rep2 = {}
gb2 = {k: ob[k] for k in h.GRAD_NAMES}
h.compare_with_noise(rep2, ref, gb2, dev, o["P"], floor_acc=floor13, floor_derived={k: fl[k] for k in h.DERIVED}, assert_rows=False)
for k, r in rep2["noise_floor"].items():
    print(f"[row-major] {k:14s} err/noise {r['err_over_ref_noise']:.2f}  row max {r['row_ratio_max']:.2f} p999 {r['row_ratio_p999']:.2f} rows>c {r['rows_above_c']}/{r['rows']}")
