#!/usr/bin/env python
"""How much do the INTEGER decisions of the path depend on FMA contraction?   (CPU only; writes profiles/archive/r02_fma_sensitivity.json)

The parity oracle (oracle/ex4d_oracle.c) is compiled with -ffp-contract=off and the HIP preprocess kernel likewise, so "bit-exact
integers" means bit-exact against a NO-FMA evaluation of the reference's expressions.  The reference itself is built by nvcc with
its default -fmad=true (DGR/setup.py:21-29 passes no -fmad=false), i.e. WITH contraction, and cannot be run here.  This script
rebuilds the same oracle source with -ffp-contract=fast -mfma (gcc contracts every a*b+c it can see; nvcc's choices differ in
detail, so this is an error bar, not a replay) and counts, over the randomised parity corpus and the BASELINE generators, how many
cull / radii / tiles_touched / sort decisions move.
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import helpers as h            # noqa: E402
from oracle import oracle                 # noqa: E402
from ex4dgs_amd.scene import SceneConfig  # noqa: E402


def load_fma():
    so = os.path.join(ROOT, "oracle", "libex4d_oracle_fma.so")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "libex4d_oracle_fma.so"])
    lib = C.CDLL(so)
    lib.ex4d_oracle_preprocess.restype = C.c_int64
    lib.ex4d_oracle_getHigherMsb.restype = C.c_uint32
    return lib


def run(lib, ins, st, sub=None):
    oracle._LIB = lib
    return h.oracle_forward(ins, st, subpixel_offset=sub, want_fragile=False)


def corpus(n_fuzz, seed):
    rng = np.random.default_rng(seed)
    for i in range(n_fuzz):       # the generator of tools/dev/fuzz_parity.py
        W = int(rng.integers(17, 700)); H = int(rng.integers(17, 500))
        P = int(rng.integers(1, 6000))
        cfg = SceneConfig(f"fuzz{i}", P, W, H, float(rng.uniform(0.4, 1.5) * W), dyn_frac=float(rng.choice([0.0, 0.3])), seed=int(rng.integers(1 << 30)),
                          sigma_px_med=float(rng.uniform(0.3, 25.0)), sigma_px_logstd=float(rng.uniform(0.2, 1.2)),
                          cxr=float(rng.choice([0.0, 0.15])), cyr=float(rng.choice([0.0, -0.1])), z_lo=4.5, z_hi=float(rng.uniform(10, 120)))
        yield cfg.name, cfg, None, int(rng.integers(0, 300)), int(rng.integers(0, 4))
    yield "cfg2 (100k, full size)", "cfg2", None, 0, 3
    yield "cfg3 generator at 250k", "cfg3", 250_000, 137, 3
    yield "cfg5 generator at 60k", "cfg5", 60_000, 0, 3


def main():
    n_fuzz = int(sys.argv[1]) if len(sys.argv) > 1 else 250
    plain = oracle.lib()
    fma = load_fma()
    tot = dict(cases=0, gaussians=0, visible=0, cull_flips=0, radii_differ=0, tiles_touched_differ=0, num_rendered_differ_cases=0,
               depth_bits_differ=0, mean2D_bits_differ=0, conic_opacity_bits_differ=0, rgb_bits_differ=0, instances=0,
               point_list_positions_differ=0, cases_with_identical_point_list=0, n_contrib_pixels_differ=0, pixels=0, max_abs_color_diff=0.0)
    per_case = []
    for name, cfg, P, t, deg in corpus(n_fuzz, 0):
        ins, st = h.scene_inputs(cfg, P=P, t=t, sh_degree=deg)
        a = run(plain, ins, st)
        b = run(fma, ins, st)
        va, vb = a["radii"] > 0, b["radii"] > 0
        both = va & vb
        bits = lambda x: np.ascontiguousarray(x).view(np.uint32)
        row_diff = lambda k: int((bits(a[k])[both].reshape(int(both.sum()), -1) != bits(b[k])[both].reshape(int(both.sum()), -1)).any(1).sum())
        c = dict(name=name, P=int(a["P"]), visible=int(va.sum()), cull_flips=int((va != vb).sum()), radii_differ=int((a["radii"] != b["radii"]).sum()),
                 tiles_touched_differ=int((a["tiles_touched"] != b["tiles_touched"]).sum()), R=(int(a["num_rendered"]), int(b["num_rendered"])),
                 depth_bits_differ=row_diff("depths"), mean2D_bits_differ=row_diff("means2D"), conic_opacity_bits_differ=row_diff("conic_opacity"),
                 rgb_bits_differ=row_diff("rgb"))
        if a["num_rendered"] == b["num_rendered"]:
            c["point_list_positions_differ"] = int((a["point_list"] != b["point_list"]).sum())
        else:
            c["point_list_positions_differ"] = None
        c["n_contrib_pixels_differ"] = int((a["n_contrib"] != b["n_contrib"]).sum())
        c["max_abs_color_diff"] = float(np.abs(a["color"] - b["color"]).max())
        per_case.append(c)
        tot["cases"] += 1; tot["gaussians"] += c["P"]; tot["visible"] += c["visible"]; tot["instances"] += c["R"][0]; tot["pixels"] += a["H"] * a["W"]
        for k in ("cull_flips", "radii_differ", "tiles_touched_differ", "depth_bits_differ", "mean2D_bits_differ", "conic_opacity_bits_differ",
                  "rgb_bits_differ", "n_contrib_pixels_differ"):
            tot[k] += c[k]
        tot["num_rendered_differ_cases"] += int(c["R"][0] != c["R"][1])
        if c["point_list_positions_differ"] is not None:
            tot["point_list_positions_differ"] += c["point_list_positions_differ"]
            tot["cases_with_identical_point_list"] += int(c["point_list_positions_differ"] == 0)
        tot["max_abs_color_diff"] = max(tot["max_abs_color_diff"], c["max_abs_color_diff"])
        print(name, {k: v for k, v in c.items() if k != "name"}, flush=True)
    oracle._LIB = plain
    out = dict(what="oracle/ex4d_oracle.c built with -ffp-contract=off (the parity oracle) vs -ffp-contract=fast -mfma (gcc's contraction; proxy "
                    "for nvcc -fmad=true of the reference build): counts of decisions / bit patterns that differ over the parity corpus",
               corpus=f"{n_fuzz} random scenes of tools/dev/fuzz_parity.py (seed 0) + cfg2 at 100k + cfg3 generator at 250k + cfg5 generator at 60k",
               totals=tot, large_cases=[c for c in per_case if not c["name"].startswith("fuzz")])
    with open(os.path.join(ROOT, "profiles", "archive", "r02_fma_sensitivity.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(tot, indent=1))


if __name__ == "__main__":
    main()
