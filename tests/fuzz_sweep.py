"""Randomised parity sweep of the rasterizer forward + backward against the oracle (small scenes, many shapes): image sizes
17..700 x 17..500, 1..6000 Gaussians, footprints 0.3..25 px, SH degrees 0-3, static / dynamic, off-centre projection, kernel
sizes, scale modifiers, sub-pixel offsets.  Shared by tests/test_gpu_parity.py (the -m gpu suite) and the command-line front end
tools/dev/fuzz_parity.py."""
import numpy as np
import torch


def run(n=30, seed=0, dir_scale=0.1, verbose=True):
    """Runs `n` random cases (generator seeded with `seed`); dir_scale 0 = frames without flow (the hand-scheduled forward walk).
    Returns the list of failure messages (empty = all green)."""
    from tests import test_gpu_parity as T
    from ex4dgs_amd.scene import SceneConfig
    from ex4dgs_amd import _C
    _C.load()
    rng = np.random.default_rng(seed)
    fails = []
    for i in range(n):
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
        try:
            # random scenes stack many faint Gaussians per pixel: more pixels sit within 1e-4 of an alpha threshold than in the bench scenes
            o, g, ob, gb, rep = T._fwd_bwd(cfg, subpixel=sub, max_fragile_frac=1e-2, **kw)
            if verbose:
                e2e = rep.get("e2e", {})
                print(f"case {i}: {W}x{H} P={P} deg={deg} t={t} R={o['num_rendered']} worst={rep.get('worst', 0):.2e} acc={rep.get('acc16_worst_ratio', 0):.3f} "
                      f"e2e={e2e.get('acc16_worst_ratio', 0):.3f} (state rows {e2e.get('extra13_rows_above_atol', 0)}) idx_mismatch={rep.get('idx_mismatch_pixels', 0)} ok", flush=True)
        except Exception as e:
            fails.append(f"case {i}: {W}x{H} P={P} deg={deg} t={t} kw={kw} FAILED: {type(e).__name__}: {str(e)[:300]}")
            if verbose:
                print(fails[-1], flush=True)
    return fails
