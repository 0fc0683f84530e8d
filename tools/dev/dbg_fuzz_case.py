"""Dev: re-run ONE case of tools/dev/fuzz_parity.py (same random draws) and print the worst accumulator rows.
Usage: python tools/dev/dbg_fuzz_case.py <case> <seed> <dir_scale> [repeats]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests import test_gpu_parity as T
from tests import helpers as h
from ex4dgs_amd.scene import SceneConfig
from ex4dgs_amd import _C
_C.load()
case, seed, dir_scale = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
rng = np.random.default_rng(seed)
for i in range(case + 1):
    W = int(rng.integers(17, 700)); H = int(rng.integers(17, 500))
    P = int(rng.integers(1, 6000))
    cfg = SceneConfig(f"fuzz{i}", P, W, H, float(rng.uniform(0.4, 1.5) * W), dyn_frac=float(rng.choice([0.0, 0.3])), seed=int(rng.integers(1 << 30)),
                      sigma_px_med=float(rng.uniform(0.3, 25.0)), sigma_px_logstd=float(rng.uniform(0.2, 1.2)),
                      cxr=float(rng.choice([0.0, 0.15])), cyr=float(rng.choice([0.0, -0.1])), z_lo=4.5, z_hi=float(rng.uniform(10, 120)))
    deg = int(rng.integers(0, 4)); t = int(rng.integers(0, 300))
    kw = dict(sh_degree=deg, t=t, grad_acc_zero=bool(rng.integers(0, 2)), seed=int(rng.integers(1 << 20)), dir_scale=dir_scale)
    if rng.random() < 0.3:
        kw["kernel_size"] = float(rng.choice([0.0, 0.05, 0.3]))
    if rng.random() < 0.3:
        kw["scale_modifier"] = float(rng.uniform(0.5, 1.5))
    sub = None
    if rng.random() < 0.3:
        sub = torch.tensor(rng.uniform(-0.5, 0.5, (H, W, 2)).astype(np.float32))
print("case", case, cfg, kw, "subpixel" if sub is not None else "")
from oracle import oracle
ins, st = h.scene_inputs(cfg, t=kw["t"], sh_degree=kw["sh_degree"], dir_scale=dir_scale)
for k in ("kernel_size", "scale_modifier"):
    if k in kw: st[k] = kw[k]
o = h.oracle_forward(ins, st, subpixel_offset=sub)
H_, W_ = st["image_height"], st["image_width"]
grads = list(h.upstream_grads(torch.from_numpy(o["acc"]), H_, W_, seed=kw["seed"], grad_acc_zero=kw["grad_acc_zero"]))
solid = torch.from_numpy(o["fragile"] > h.FRAG_EPS)
grads = [x * solid[None] for x in grads]
for asm in (1, 0):
    _C.set_option("composite_fwd_asm", asm)
    g = h.gpu_forward_raw(ins, st, subpixel_offset=sub)
    ob_state = dict(o)
    ob_state.update(depth=h.to_np(g["depth"]), acc=h.to_np(g["acc"]), final_T=np.ascontiguousarray(h.to_np(g["final_T"])),
                    n_contrib=np.ascontiguousarray(h.to_np(g["n_contrib"]).astype(np.uint32)))
    ob = oracle.backward(ob_state, *grads)
    dstate = [np.abs(o[k].astype(np.float64) - h.to_np(g[k]).astype(np.float64)).astype(np.float32).reshape(H_, W_) for k in ("depth", "acc", "final_T")]
    ob2 = oracle.backward(o, *grads, state_delta=dstate)
    print("asm", asm, "n_contrib equal to oracle on solid:", bool((torch.from_numpy(o["n_contrib"].astype(np.int64))[solid] == g["n_contrib"].cpu()[solid]).all()),
          "max |final_T diff|", float(np.abs(h.to_np(g["final_T"]) - o["final_T"]).max()))
    for r in range(reps):
        gb = h.gpu_backward_raw(ins, g, grads)
        acc = h.acc16_in_reference_units(gb["acc16"], W_, H_, conic=o["conic_opacity"])[:, :13].astype(np.float64)
        for name, obx, atol, keps in (("gpu-state", ob, 1e-5, 64.0), ("end-to-end", ob2, 1e-5, 64.0)):
            tol = atol + keps * 2.0 ** -24 * obx["abs13"] + 3e-6 * np.abs(obx["sum13"]) + (2.0 * obx["state13"] if obx.get("state13") is not None else 0.0)
            err = np.abs(acc - obx["sum13"])
            ratio = err / tol
            i, j = np.unravel_index(ratio.argmax(), ratio.shape)
            print(f"  rep {r} {name}: worst ratio {ratio.max():.3f} at {(int(i), int(j))}: gpu {acc[i, j]:.9g} ref {obx['sum13'][i, j]:.9g} abs13 {obx['abs13'][i, j]:.6g} tol {tol[i, j]:.3g}")
        if r == 0:
            ratio = np.abs(acc - ob["sum13"]) / (1e-5 + 64.0 * 2.0 ** -24 * ob["abs13"] + 3e-6 * np.abs(ob["sum13"]))
            i = int(np.unravel_index(ratio.argmax(), ratio.shape)[0])
            px, py = o["means2D"][i]
            x0, y0 = max(0, int(px) - 2), max(0, int(py) - 2)
            print("   fragile margins around the mean:", o["fragile"][y0:y0 + 5, x0:x0 + 5].min(), "acc there", o["acc"][0, y0:y0 + 5, x0:x0 + 5].min(), o["acc"][0, y0:y0 + 5, x0:x0 + 5].max(),
                  "n_contrib gpu/oracle equal there:", bool((h.to_np(g["n_contrib"])[y0:y0 + 5, x0:x0 + 5] == o["n_contrib"][y0:y0 + 5, x0:x0 + 5]).all()))
            print("   row", i, "gpu", acc[i, :7], "\n   ref", ob["sum13"][i, :7], "\n   abs", ob["abs13"][i, :7], "\n   conic/opacity", o["conic_opacity"][i], "mean2D", o["means2D"][i], "radius", o["radii"][i], "tiles", o["tiles_touched"][i])
