import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import helpers as h
from oracle import oracle
ins, st = h.scene_inputs("cfg3", P=12000, t=0)
o = h.oracle_forward(ins, st)
g = h.gpu_forward_raw(ins, st)
H, W = st["image_height"], st["image_width"]
grads = list(h.upstream_grads(torch.from_numpy(o["acc"]), H, W, seed=3, grad_acc_zero=False))
solid = torch.from_numpy(o["fragile"] > 1e-4)
grads = [x * solid[None] for x in grads]
ob = oracle.backward(o, *grads)
G = 3693
for rep in range(2):
    gb = h.gpu_backward_raw(ins, g, grads)
    acc = h.acc16_in_reference_units(gb["acc16"], W, H)
    print("gpu   ", acc[G, :13])
print("oracle", ob["sum13"][G])
print("diff  ", acc[G, :13] - ob["sum13"][G])
# pixels touched: compare forward state around the Gaussian
x, y = o["means2D"][G]; xi, yi = int(x), int(y)
sl = (slice(max(0, yi - 4), yi + 5), slice(max(0, xi - 4), xi + 5))
print("final_T diff", np.abs(o["final_T"][sl] - h.to_np(g["final_T"])[sl]).max(), "n_contrib eq", (o["n_contrib"][sl].astype(np.int64) == h.to_np(g["n_contrib"])[sl]).all())
print("final_T", o["final_T"][sl].min(), o["final_T"][sl].max(), "fragile min", o["fragile"][sl].min())
err = np.abs(acc[:, :13] - ob["sum13"]); eps = 2.0**-24
tol = 1e-5 + 64 * eps * ob["abs13"] + 3e-6 * np.abs(ob["sum13"])
r = err / tol
idx = np.argsort(r.max(1))[::-1][:8]
for i in idx: print(i, r[i].max(), r[i].argmax(), "w", o["conic_opacity"][i, 3], "radius", o["radii"][i])
