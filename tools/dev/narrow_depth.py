"""Depth-sort stage time on scenes whose visible depths occupy a narrow band of [min_depth, max_depth] (round 5: the MSD depth sort cuts
its top digit from the key range the frame occupies; cut from the static range such scenes fell into a handful of oversize buckets).
    python tools/dev/narrow_depth.py            -> one line per (scene, depth sort) with the stage times of the forward"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import helpers as h                      # noqa: E402
from tests.test_gpu_round4 import _raw_forward      # noqa: E402
from ex4dgs_amd import _C, build                    # noqa: E402


def squeezed(P, z_lo, z_hi, seed=5):
    ins, st = h.scene_inputs("cfg3", P=P, t=137)
    g = torch.Generator().manual_seed(seed)
    m = ins["means3D"]
    z = z_lo * (z_hi / z_lo) ** torch.rand(P, generator=g)
    scale = (z / m[:, 2]).unsqueeze(1)
    ins["means3D"] = (m * scale).contiguous()
    ins["means3D"][:, 2] = z
    ins["scales"] = (ins["scales"] * scale).contiguous()
    return {k: v.cuda() for k, v in ins.items()}, st


def main():
    build.build(); _C.load()
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    for name, z_lo, z_hi in (("bench range 4.5-80", 4.5, 80.0), ("8-12 m", 8.0, 12.0), ("5-5.5 m", 5.0, 5.5), ("wall: 30 % at 10 m +- 1 cm", 4.5, 80.0), ("wall: 90 % at 6 m +- 2 mm", 4.5, 80.0),
                             ("one depth 7 m", 7.0, 7.0)):
        ins, st = squeezed(P, z_lo, z_hi)
        if name.startswith("wall"):
            frac, zw, th = (0.3, 10.0, 0.01) if "30 %" in name else (0.9, 6.0, 0.002)
            g = torch.Generator().manual_seed(3)
            m = ins["means3D"].cpu()
            on = torch.rand(P, generator=g) < frac
            z = torch.where(on, zw + th * (2 * torch.rand(P, generator=g) - 1), m[:, 2])
            sc = (z / m[:, 2]).unsqueeze(1)
            m = m * sc; m[:, 2] = z
            ins["means3D"] = m.contiguous().cuda()
            ins["scales"] = (ins["scales"].cpu() * sc).contiguous().cuda()
        for mode in (2, 0):
            _C.set_option("depth_sort_msd", mode)
            for _ in range(3):
                _raw_forward(ins, st)
            torch.cuda.synchronize()
            _C.profile_enable(True)
            agg, n = {}, 8
            for _ in range(n):
                f = _raw_forward(ins, st)[1]
                torch.cuda.synchronize()
                for k, ms in _C.profile_read(0):
                    agg[k] = agg.get(k, 0.0) + ms / n
            _C.profile_enable(False)
            R = int(f[0])
            print(f"{name:28s} P={P} R={R:9d} depth sort {'MSD (adaptive digit)' if mode == 2 else 'LSD 3 passes      '}: depth_sort {agg['depth_sort'] * 1e3:7.1f} us  scan_tiles {agg['scan_tiles'] * 1e3:6.1f} us  "
                  f"forward {sum(agg.values()) * 1e3:7.1f} us", flush=True)
        # the default (auto): mean forward time over a run of frames from a cold start -- the slow frames it pays (one per doubling hold) included
        line = f"{name:28s} forward, mean of {RUN} frames:"
        for mode, label in ((3, "auto"), (0, "LSD"), (2, "MSD forced")):
            _C.set_option("depth_sort_msd", mode)        # (resets the auto mode's hold)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(RUN if not (mode == 2 and name.startswith(("wall", "one"))) else RUN // 10):
                _raw_forward(ins, st)
            e1.record(); torch.cuda.synchronize()
            n = RUN if not (mode == 2 and name.startswith(("wall", "one"))) else RUN // 10
            line += f"  {label} {e0.elapsed_time(e1) / n * 1e3:7.1f} us"
            if mode == 3:
                line += f" ({_C.get_option('depth_sort_trips')} reports)"
        print(line, flush=True)
    _C.set_option("depth_sort_msd", 3)


RUN = 1000
if __name__ == "__main__":
    main()
