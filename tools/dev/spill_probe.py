"""Developer probe (GPU box) for VERDICT r03 weak #8: a build of preprocess_bwd_kernel forced to 128 VGPRs (__launch_bounds__(256, 4))
spills three registers and returned wrong gradients for ~0.06 % of the Gaussians.  This script builds that variant of the CURRENT sources
into /tmp, runs the same forward + backward through the default library and through the variant (two processes: one library per
process) on BASELINE config 2 at 100 k Gaussians, and prints WHICH Gaussians differ, in which tensors, and what they have in common.

    python tools/dev/spill_probe.py            # driver: builds, runs both, compares
    python tools/dev/spill_probe.py run <out>  # worker: forward + backward with the library EX4D_HIP_LIB names, results -> <out>
    python tools/dev/spill_probe.py build_fix <dir>  # CPU: the 128-register build as is and with a 32-bit visibility predicate (the source fix)
    python tools/dev/spill_probe.py variants <dir>   # the assembly-patched libraries of tools/dev/spill_asm_variants.py, one line each

Result (round 4): not the spills -- a 64-bit shift whose shift amount the register allocator had put into the wave's
last register (v127) reads VGPR0 instead on gfx950 in waves that share their SIMD; stand-alone: tools/dev/micro/topreg_probe.hip."""
import os, re, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "ex4dgs_amd", "csrc")
SIZES = tuple(int(x) for x in os.environ.get("EX4D_SPILL_PROBE_SIZES", "100000,65536,40000,131072").split(","))


def build_variant(dst, patch, info=None):
    """All objects of the in-tree build except ex4d_preprocess.o, which is compiled from a patched copy of the source.
    info (a dict, optional) receives {kernel name: (VGPRs, spilled VGPRs)} of the preprocess_bwd instantiations and "obj"."""
    from ex4dgs_amd import build
    build.build()
    os.makedirs(dst, exist_ok=True)
    src = open(os.path.join(CSRC, "ex4d_preprocess.hip")).read()
    src2 = patch(src)
    assert src2 != src
    tmp = os.path.join(dst, "ex4d_preprocess.hip")
    open(tmp, "w").write(src2.replace('#include "ex4d_internal.h"', f'#include "{CSRC}/ex4d_internal.h"'))
    obj = os.path.join(dst, "ex4d_preprocess.o")
    cmd = [build._hipcc()] + build.COMMON + build.SOURCES["ex4d_preprocess.hip"] + ["-Rpass-analysis=kernel-resource-usage", "-c", tmp, "-o", obj]
    r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    for m in re.finditer(r"Function Name: (\S+).*?VGPRs: (\d+).*?VGPRs Spill: (\d+)", r.stderr, re.S):
        if "preprocess_bwd" in m.group(1):
            print("variant", m.group(1)[:60], "VGPRs", m.group(2), "spilled", m.group(3), flush=True)
            if info is not None: info[m.group(1)] = (int(m.group(2)), int(m.group(3)))
    if info is not None: info["obj"] = obj
    objs = [obj] + [os.path.join(CSRC, f.replace(".hip", ".o")) for f in build.SOURCES if f != "ex4d_preprocess.hip"]
    lib = os.path.join(dst, "libex4d_hip_variant.so")
    subprocess.check_call([build._hipcc(), "-shared", "-fPIC", f"--offload-arch={build.ARCH}", "-o", lib] + objs)
    return lib


def worker(out):
    import numpy as np, torch
    from tests import helpers as h
    from ex4dgs_amd import _C
    _C.load()
    res = {}
    for P in SIZES:                              # 100 000 = 390 blocks + 160 threads (a 32-row wave, an empty wave); 65 536 = 256 blocks: one per CU
        ins, st = h.scene_inputs("cfg2", P=P)
        g = h.gpu_forward_raw(ins, st)
        grads = [x.cuda() for x in h.upstream_grads(g["acc"].cpu(), st["image_height"], st["image_width"], seed=3)]
        d = {k: v.cuda() for k, v in ins.items()}
        for rep in range(2):
            b = h.gpu_backward_raw(d, g, grads)
            torch.cuda.synchronize()
            for k in ("dL_dmeans3D", "dL_dsh", "dL_dscales", "dL_drotations", "dL_dmeans2D", "dL_dopacity", "acc16"):
                res[f"{P}/{rep}/{k}"] = b[k].detach().cpu().numpy().copy()
        res[f"{P}/radii"] = g["radii"].cpu().numpy()
    np.savez(out, **res)


def main():
    import numpy as np
    dst = "/tmp/ex4d_spill_probe"
    shutil.rmtree(dst, ignore_errors=True)
    lib = build_variant(dst, lambda s: s.replace(*BOUNDS))
    outs = {}
    for name, env in (("default", {}), ("variant", {"EX4D_HIP_LIB": lib})):
        out = os.path.join(dst, name + ".npz")
        e = dict(os.environ, **env); e.pop("EX4D_HIP_LIB", None) if not env else None
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "run", out], env=e)
        outs[name] = np.load(out)
    a, v = outs["default"], outs["variant"]
    for P in SIZES:
        radii = a[f"{P}/radii"]
        # the per-Gaussian stage is deterministic given its accumulators: compare on rows whose accumulator rows are bit-equal
        same_acc = (a[f"{P}/0/acc16"].view(np.uint32) == v[f"{P}/0/acc16"].view(np.uint32)).all(1)
        print(f"P={P}: accumulator rows bit-equal between the two processes: {int(same_acc.sum())} of {P} (float atomics: the others differ in the last bits)")
        for k in ("dL_dmeans3D", "dL_dsh", "dL_dscales", "dL_drotations", "dL_dmeans2D", "dL_dopacity"):
            x, y = a[f"{P}/0/{k}"].reshape(P, -1), v[f"{P}/0/{k}"].reshape(P, -1)
            scale = np.maximum(np.abs(x).max(1), 1e-30)
            rel = np.abs(x - y).max(1) / scale
            bad = np.nonzero((rel > 1e-3) & (radii > 0))[0]
            rep_noise = np.abs(a[f"{P}/0/{k}"] - a[f"{P}/1/{k}"]).reshape(P, -1).max(1) / scale
            print(f"  {k}: rows differing by > 1e-3 relative: {len(bad)} (default-vs-default rerun: {int(((rep_noise > 1e-3) & (radii > 0)).sum())})")
            if len(bad):
                print("    first rows", bad[:24].tolist())
                print("    row % 64 histogram:", np.bincount(bad % 64, minlength=64).tolist())
                print("    (row // 64) % 4 (wave in block) histogram:", np.bincount((bad // 64) % 4, minlength=4).tolist(), " blocks:", sorted(set((bad // 256).tolist()))[:20])
                i = int(bad[0])
                print(f"    row {i}: default {x[i][:8]} variant {y[i][:8]} bit-equal accumulators: {bool(same_acc[i])}")


BOUNDS = ("template <bool DSUMS>\n__global__ __launch_bounds__(256) void preprocess_bwd_kernel(",
          "template <bool DSUMS>\n__global__ __launch_bounds__(256, 4) void preprocess_bwd_kernel(")
NEED32 = "mask_bit(need, g)"                       # what the sources hold since round 4
NEED64 = "(((need >> g) & 1ull) != 0ull)"          # what they held before: a 64-bit shift by a per-lane amount


def build_fix(outdir):
    """CPU: the forced-128-register build of the sources as they are (lib_fixed32.so: the visibility predicate of the SH chunk loads
    comes from the two 32-bit halves of the mask, so no 64-bit shift by a per-lane amount is left for the allocator to place in v127) and
    with the predicate they had before round 4 (lib_base.so: `(need >> g) & 1`); says what build.shift_amount_in_last_vgpr finds in
    each.  Run them with `variants <outdir>` on the GPU: base has the wrong rows, fixed32 none -- with the same three spills."""
    from ex4dgs_amd import build
    os.makedirs(outdir, exist_ok=True)
    for name, patch in (("base", lambda s: s.replace(*BOUNDS).replace(NEED32, NEED64)), ("fixed32", lambda s: s.replace(*BOUNDS))):
        dst = f"/tmp/ex4d_spill_fix_{name}"
        shutil.rmtree(dst, ignore_errors=True)
        lib = build_variant(dst, patch)
        print(name, "-> 64-bit shifts by the last register:", build.shift_amount_in_last_vgpr(os.path.join(dst, "ex4d_preprocess.o")), flush=True)
        shutil.copy(lib, os.path.join(outdir, f"lib_{name}.so"))


def variants(libdir):
    """the assembly-patched libraries of tools/dev/spill_asm_variants.py against the default build, P = 100 000"""
    import glob
    import numpy as np
    os.environ["EX4D_SPILL_PROBE_SIZES"] = "100000"
    dst = "/tmp/ex4d_spill_probe_variants"
    shutil.rmtree(dst, ignore_errors=True); os.makedirs(dst)
    libs = sorted(glob.glob(os.path.join(libdir, "lib_*.so")), key=lambda p: (os.path.basename(p) != "lib_base.so", p))
    runs = [("default", None)] + [(os.path.basename(p)[4:-3], p) for p in libs]
    ref = None
    for name, lib in runs:
        out = os.path.join(dst, name + ".npz")
        e = dict(os.environ)
        e.pop("EX4D_HIP_LIB", None)
        if lib: e["EX4D_HIP_LIB"] = lib
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "run", out], env=e, stderr=subprocess.PIPE, text=True)
        if r.returncode != 0:
            print(f"{name}: FAILED to run\n{r.stderr[-800:]}", flush=True); continue
        d = np.load(out)
        if ref is None: ref = d; continue
        P = 100000
        x, y = ref[f"{P}/0/dL_dmeans3D"].reshape(P, -1), d[f"{P}/0/dL_dmeans3D"].reshape(P, -1)
        radii = ref[f"{P}/radii"]
        scale = np.maximum(np.abs(x).max(1), 1e-30)
        rel = np.abs(x - y).max(1) / scale
        bad = np.nonzero(((rel > 1e-3) | ~np.isfinite(y).all(1)) & (radii > 0))[0]
        nan_rows = np.nonzero(~np.isfinite(y).all(1) & (radii > 0))[0]
        lanes = np.bincount(bad % 64, minlength=64)
        nz = np.nonzero(lanes)[0]
        print(f"{name}: rows off by > 1e-3: {len(bad)} (non-finite: {len(nan_rows)}); lanes {nz.min() if len(nz) else '-'}..{nz.max() if len(nz) else '-'}; "
              f"blocks from {int(bad.min()) // 256 if len(bad) else '-'}; other tensors off: "
              + ", ".join(f"{k} {int((np.abs(ref[f'{P}/0/{k}'].reshape(P, -1) - d[f'{P}/0/{k}'].reshape(P, -1)).max(1) / np.maximum(np.abs(ref[f'{P}/0/{k}'].reshape(P, -1)).max(1), 1e-30) > 1e-3)[radii > 0].sum())}"
                          for k in ("dL_dsh", "dL_dscales", "dL_drotations")), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "build_fix":
        build_fix(sys.argv[2])
    elif len(sys.argv) > 2 and sys.argv[1] == "variants":
        variants(sys.argv[2])
    elif len(sys.argv) > 2 and sys.argv[1] == "run":
        worker(sys.argv[2])
    else:
        main()
