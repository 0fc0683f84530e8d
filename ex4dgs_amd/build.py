"""Builds libex4d_hip.so (the C-ABI library of include/ex4d_rasterizer.h) for gfx950 with hipcc.

In-tree build: objects and the .so land next to the sources (ex4dgs_amd/csrc/), so the built library
travels with a repo snapshot.  hipcc cross-compiles without a GPU present.
"""
import os
import shutil
import subprocess
import sys

from . import isa_check          # machine-code checks of the built objects

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libex4d_hip.so")
ARCH = "gfx950"

# per-file flags: the per-Gaussian preprocess must not fuse multiply-adds (bit-exact integer decisions); its SLP-vectorised form
# (v_pk_*_f32: the same IEEE results per element) needs 12-20 more VGPRs, which costs the backward kernel a wave per SIMD;
# the compositing kernels are VALU-issue bound and v_pk_*_f32 is slower than two scalar ops there (measured:
# -fno-slp-vectorize = -5 % kernel time), so the SLP vectoriser is off for that file
SOURCES = {
    "ex4d_preprocess.hip": ["-ffp-contract=off", "-fno-slp-vectorize"],
    "ex4d_binning.hip": [],
    "ex4d_rowsort.hip": [],
    "ex4d_composite.hip": ["-ffp-contract=fast", "-munsafe-fp-atomics", "-fno-slp-vectorize"],
    "ex4d_api.hip": [],
    "ex4d_attributes.hip": ["-ffp-contract=off"],
    "ex4d_loss.hip": [],
    "ex4d_optim.hip": ["-ffp-contract=off"],
    "ex4d_knn.hip": ["-ffp-contract=off"],
    "ex4d_trainer.hip": [],          # host code only: the compiled host path of one training iteration (include/ex4d_trainer.h)
}
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libex4d_hip.so cannot be built")


def _without_remarks(text):
    """The compiler's other diagnostics (warnings, errors) without the resource-usage remarks and their source snippets."""
    out, skip = [], 0
    for line in text.splitlines():
        if "-Rpass-analysis=kernel-resource-usage" in line:
            skip = 2
            continue
        if skip and (line.strip().startswith("|") or line.split("|")[0].strip().isdigit()):
            skip -= 1
            continue
        skip = 0
        if "remark" in line and "generated" in line:      # "12 remarks generated."
            continue
        out.append(line + "\n")
    return "".join(out)


def _no_vgpr_spills(src, remarks, obj):
    """No kernel of this library may spill vector registers: a performance rule (every kernel here is laid out for its register
    budget; a spill means a change blew it).  The wrong gradients of the 128-register build of preprocess_bwd_kernel that once
    motivated this check were not caused by its three spills: see shift_amount_in_last_vgpr below.  The compiler's resource remarks
    are checked at build time; the object is removed on a violation."""
    name, bad = None, []
    for line in remarks.splitlines():
        if "Function Name:" in line:
            name = line.split("Function Name:")[1].split("[")[0].strip()
        elif "VGPRs Spill:" in line:
            n = int(line.split("VGPRs Spill:")[1].split("[")[0])
            if n:
                bad.append((name, n))
    if bad:
        if os.path.exists(obj):
            os.remove(obj)
        raise RuntimeError(f"{src}: vector-register spills in {bad}: restructure the kernel or relax its __launch_bounds__")


shift_amount_in_last_vgpr = isa_check.shift_amount_in_last_vgpr


def _no_shift_amount_in_last_vgpr(src, obj):
    bad = shift_amount_in_last_vgpr(obj)
    if bad:
        os.remove(obj)
        raise RuntimeError(f"{src}: 64-bit shifts with their shift amount in the wave's last vector register (wrong results on gfx950 for waves "
                           f"allocated at the top of the register file): {bad}; change the kernel's __launch_bounds__ / register pressure")


def _report_wait_state_violations(src, obj):
    """Missing software wait states around the inline assembly (isa_check.wait_state_violations): reported, not fatal -- the rules are
    restated from the compiler's hazard recognizer, and tests/test_cpu_oracle_and_host.py holds the objects to zero findings."""
    bad = isa_check.wait_state_violations(obj)
    for kernel, rule, cons, prod in bad[:20]:
        sys.stderr.write(f"{src}: wait states: {rule}: `{prod}` -> `{cons}` in {kernel[:60]}\n")
    return bad


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=()):
    """Compile every HIP source for gfx950 and link libex4d_hip.so.  Returns the library path."""
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, "ex4d_internal.h"), os.path.join(HERE, "..", "include", "ex4d_rasterizer.h"),
               os.path.join(HERE, "..", "include", "ex4d_attributes.h"), os.path.join(HERE, "..", "include", "ex4d_loss.h"), os.path.join(HERE, "..", "include", "ex4d_optim.h"), os.path.join(HERE, "..", "include", "ex4d_knn.h"), os.path.join(HERE, "..", "include", "ex4d_trainer.h"),
               os.path.abspath(__file__)]
    objs = []
    for src, flags in SOURCES.items():
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [hipcc] + COMMON + flags + list(extra_flags) + os.environ.get("EX4D_EXTRA_HIPCC_FLAGS", "").split() + \
                  ["-Rpass-analysis=kernel-resource-usage", "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
            remarks = r.stderr or ""
            sys.stderr.write(_without_remarks(remarks))
            if r.returncode:
                raise subprocess.CalledProcessError(r.returncode, cmd)
            _no_vgpr_spills(src, remarks, o)
            _no_shift_amount_in_last_vgpr(src, o)
            _report_wait_state_violations(src, o)
    if force or _stale(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
