"""Static issue-cost estimate of a kernel's straight-line stretches from the built object's disassembly (CPU only).

    python tools/isa_cost.py ex4dgs_amd/csrc/ex4d_composite.o composite_bwd_scan_kernelILi1ELb0 [--top 12] [--min 30]

For every stretch of instructions between labels / branches of the kernels whose symbol contains the given substring: the instruction mix
by class and the issue cycles it costs one wave, priced with the per-instruction costs MEASURED on MI355X (wave64, 4 waves per SIMD;
DESIGN.md section 4 "Why compositing is VALU-bound", tools/dev/micro/pk_rate.hip, mfma_dpp_probe.hip).  The compositing kernels run at
0.88-0.94 VALU-busy, so for them the sum over the hot stretches IS the kernel time to first order; the tool exists to price a rewrite of
such a loop before a GPU is spent on it (round 4: the two-pixels-per-lane packing of the backward step).
s_nop / s_waitcnt / scalar / LDS / memory instructions are listed but priced at 0: with several waves per SIMD they overlap the VALU
stream of the other waves."""
import argparse
import collections
import os
import re
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ex4dgs_amd import isa_check          # noqa: E402

# cycles per instruction per SIMD (measured; VOP2 v_cndmask with vcc is the one outlier worth knowing about)
COST = (("dpp", 5.0), ("trans", 9.0), ("pk", 5.7), ("cndmask_vcc", 23.6), ("cmp_min_cndmask", 5.0), ("permlane_swap", 8.8), ("mfma", 25.0),
        ("fma", 3.5), ("valu", 3.2), ("other", 0.0))


def classify(text):
    c = isa_check.decode(text)
    mn = re.sub(r"_(e32|e64|dpp|sdwa)$", "", c["mnemonic"])
    if not c["valu"]:
        return "other"
    if c["dpp"]:
        return "dpp"
    if c["trans"]:
        return "trans"
    if mn.startswith("v_pk_"):
        return "pk"
    if mn.startswith("v_mfma"):
        return "mfma"
    if mn.startswith(("v_permlane16_swap", "v_permlane32_swap")):
        return "permlane_swap"
    if mn == "v_cndmask_b32" and not c["mnemonic"].endswith("_e64") and "vcc" in c["mask"]:
        return "cndmask_vcc"
    if mn.startswith(("v_cmp", "v_min", "v_max", "v_cndmask")):
        return "cmp_min_cndmask"
    if mn.startswith(("v_fma", "v_fmac", "v_mad")):
        return "fma"
    return "valu"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("obj")
    ap.add_argument("kernel", help="substring of the (mangled) kernel symbol")
    ap.add_argument("--top", type=int, default=12)
    ap.add_argument("--min", type=int, default=30, help="shortest stretch (instructions) worth listing")
    a = ap.parse_args()
    cost = dict(COST)
    d = tempfile.mkdtemp(prefix="ex4d_isa_")
    co = isa_check.device_code(a.obj, d)
    rows = []
    for sym, run in isa_check.disassembly(co):
        if a.kernel not in sym:
            continue
        stretch, start = [], 0
        for i, t in enumerate(run + ["s_endpgm"]):
            if t.split(None, 1)[0].startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
                if len(stretch) >= a.min:
                    mix = collections.Counter(classify(x) for x in stretch)
                    rows.append((sum(cost[k] * n for k, n in mix.items()), sym, len(stretch), mix, stretch[0]))
                stretch = []
            else:
                stretch.append(t)
    rows.sort(key=lambda r: -r[0])
    for cycles, sym, n, mix, first in rows[:a.top]:
        valu = sum(v for k, v in mix.items() if k != "other")
        print(f"{cycles:7.0f} cycles  {n:4d} instructions ({valu} VALU)  " + "  ".join(f"{k} {mix[k]}" for k, _ in COST if mix.get(k)) + f"   | starts: {first[:60]}")
    print(f"{len(rows)} stretches of >= {a.min} instructions in kernels matching '{a.kernel}'")


if __name__ == "__main__":
    main()
