"""Build-time checks on the gfx950 machine code of the library's objects (CPU only: llvm-objdump on the embedded code object).

Two kinds of rule the compiler does not enforce for us:

* `shift_amount_in_last_vgpr` -- a gfx950 operand fault found in round 4 (DESIGN.md section 4 "Round 4",
  tools/dev/micro/topreg_probe.hip): a 64-bit shift whose 32-bit shift amount sits in the LAST vector register of the wave's
  allocation shifts by VGPR0 instead in waves that share their SIMD.  The register allocator hands that register out like any other.
* `wait_state_violations` -- the software-managed wait states of gfx940-class hardware that matter to this library's INLINE ASSEMBLY
  (the compiler pads its own instructions, but it neither looks inside an asm string nor re-checks the boundary when its scheduling
  around an asm block changes with a toolchain): a DPP instruction (or v_permlane16/32_swap) reading a VGPR a VALU instruction wrote fewer than
  2 wait states earlier; a VALU instruction reading the result of a transcendental instruction in the very next slot; a VALU instruction reading,
  as a CONSTANT (not as the lane mask of a select / carry), an SGPR / VCC that a VALU instruction wrote fewer than 2 wait states earlier;
  v_readlane / v_writelane with a lane select, or v_div_fmas with a VCC, written by VALU < 4 states earlier.  (Rules as LLVM's GCNHazardRecognizer states them for gfx940; the second and
  third were hit on hardware while the hand-scheduled forward walk was written.)  One wait state = one issued instruction of the wave;
  `s_nop N` counts N + 1.  The check is per straight-line run of instructions (labels and branches end a run: conservative in
  the sense of not inventing violations, so a hazard that spans a branch is NOT seen).
"""
import os
import re
import shutil
import subprocess
import tempfile

ARCH = "gfx950"


class IsaCheckError(RuntimeError):
    """The machine-code checks could not be carried out (tools missing, metadata unreadable): the build fails CLOSED on it."""


def _llvm_dir():
    """Directory of llvm-objcopy / llvm-objdump / llvm-readelf / clang-offload-bundler: next to the ROCm installation hipcc belongs
    to ($ROCM_PATH, the resolved hipcc, /opt/rocm), never a hard-coded path alone."""
    cands = []
    if os.environ.get("ROCM_PATH"):
        cands.append(os.path.join(os.environ["ROCM_PATH"], "lib", "llvm", "bin"))
    hipcc = shutil.which("hipcc")
    if hipcc:
        root = os.path.dirname(os.path.dirname(os.path.realpath(hipcc)))
        cands += [os.path.join(root, "lib", "llvm", "bin"), os.path.join(root, "llvm", "bin")]
    cands.append("/opt/rocm/lib/llvm/bin")
    need = ("llvm-objcopy", "llvm-objdump", "llvm-readelf", "clang-offload-bundler")
    for c in cands:
        if all(os.path.exists(os.path.join(c, t)) for t in need):
            return c
    raise IsaCheckError(f"the LLVM tools {need} were not found in any of {cands}: set ROCM_PATH; the machine-code checks of the build cannot run")


class _Llvm:
    """`f"{LLVM}/tool"` resolves the directory on first use (import of this module must not need the tools)."""
    _dir = None

    def __format__(self, spec):
        if _Llvm._dir is None:
            _Llvm._dir = _llvm_dir()
        return _Llvm._dir


LLVM = _Llvm()
SHIFT64 = ("v_lshlrev_b64", "v_lshrrev_b64", "v_ashrrev_i64")
TRANS = ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rcp_iflag_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32",
         "v_exp_f16", "v_log_f16", "v_rcp_f16", "v_rsq_f16", "v_sqrt_f16", "v_sin_f16", "v_cos_f16", "v_exp_legacy_f32", "v_log_legacy_f32")
_DPP_CTRL = ("quad_perm:", "row_shl:", "row_shr:", "row_ror:", "wave_shl:", "wave_shr:", "wave_rol:", "wave_ror:", "row_mirror",
             "row_half_mirror", "row_bcast:", "row_newbcast:")


def device_code(obj, workdir):
    """The gfx950 code object embedded in a host object; None for host-only objects."""
    fat, co = os.path.join(workdir, "fat.bin"), os.path.join(workdir, "dev.co")
    sections = subprocess.run([f"{LLVM}/llvm-readelf", "-S", obj], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if sections.returncode != 0:
        raise IsaCheckError(f"llvm-readelf cannot read {obj}: {sections.stderr.strip()[:200]}")
    if ".hip_fatbin" not in sections.stdout:
        return None                              # host-only object (no device code at all)
    r = subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj], stderr=subprocess.PIPE, text=True)
    if r.returncode != 0 or not os.path.exists(fat) or os.path.getsize(fat) == 0:
        raise IsaCheckError(f"{obj} has a .hip_fatbin section that llvm-objcopy could not extract: {r.stderr.strip()[:200]}")
    subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--type=o", f"--targets=hipv4-amdgcn-amd-amdhsa--{ARCH}", f"--input={fat}",
                           f"--output={co}", "--unbundle"])
    return co


def kernel_registers(co):
    """{kernel symbol: (vgpr_count, agpr_count)} from the code object's metadata note."""
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], stdout=subprocess.PIPE, text=True, check=True).stdout
    regs = {}
    for m in re.finditer(r"\.agpr_count:\s+(\d+).*?\.name:\s+(\S+).*?\.vgpr_count:\s+(\d+)", notes, re.S):
        regs[m.group(2)] = (int(m.group(3)), int(m.group(1)))
    return regs


def disassembly(co):
    """[(kernel symbol, [instruction text, ...])], one entry per straight-line stretch between labels (`--symbolize-operands` marks every
    branch target with a label <L..>)."""
    dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", "--symbolize-operands", co], stdout=subprocess.PIPE, text=True,
                         check=True).stdout
    out, sym, run = [], None, []
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line.strip())
        if m:
            if sym is not None and run:
                out.append((sym, run))
            run = []
            if not re.fullmatch(r"L\d+", m.group(1)):
                sym = m.group(1)
            continue
        t = line.split("//")[0].strip()
        if not t or sym is None or t.startswith("Disassembly") or "file format" in t:
            continue
        run.append(t)
    if sym is not None and run:
        out.append((sym, run))
    return out


def shift_amount_in_last_vgpr(obj, tmpdir=None):
    """[(kernel, instruction)] for every 64-bit shift of the object's gfx950 code whose shift amount is the last register of the wave's
    allocation.  Registers are handed out in blocks of 8, so only kernels whose register count is a multiple of 8 (and that use no
    accumulation registers) can name their allocation's last register."""
    d = tmpdir or tempfile.mkdtemp(prefix="ex4d_isa_")
    try:
        co = device_code(obj, d)
        if co is None:
            return []
        regs = kernel_registers(co)
        found = []
        runs = disassembly(co)
        # every kernel of the code object (one `<name>.kd` descriptor symbol each) must have register metadata AND disassembled code:
        # a metadata / symbol layout this parser does not understand must not turn the check into a silent pass
        symtab = subprocess.run([f"{LLVM}/llvm-readelf", "-s", "-W", co], stdout=subprocess.PIPE, text=True, check=True).stdout
        descriptors = sorted({m.group(1) for m in re.finditer(r"\s(\S+)\.kd\s*$", symtab, re.M)})
        seen = {k for k, _ in runs}
        missing = [k for k in descriptors if k not in regs or k not in seen]
        if missing:
            raise IsaCheckError(f"{obj}: kernels without register metadata or disassembly: {missing[:3]} (tool output layout changed?): "
                                f"the shift-operand check cannot be made")
        for kernel, run in runs:
            if kernel not in regs:
                continue
            vgprs, agprs = regs[kernel]
            if agprs or vgprs % 8:
                continue
            for t in run:
                if t.startswith(SHIFT64):
                    ops = [x.strip() for x in t.split(None, 1)[1].split(",")]
                    if ops[1] == f"v{vgprs - 1}":
                        found.append((kernel, t))
        return found
    finally:
        if not tmpdir:
            shutil.rmtree(d, ignore_errors=True)


# ------------------------------------------------------------------------------------------------ wait states
def _regs(tok):
    """Register names an operand token covers: 'v5' -> {'v5'}, 'v[4:7]' -> v4..v7, 's[4:5]', 'vcc', 'exec', '-v3', '|v3|', 'v3 op_sel...'."""
    tok = tok.strip().split()[0] if tok.strip() else ""
    tok = tok.strip("-|")
    m = re.fullmatch(r"([vsa])\[(\d+):(\d+)\]", tok)
    if m:
        return {f"{m.group(1)}{i}" for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    if re.fullmatch(r"[vsa]\d+", tok):
        return {tok}
    if tok in ("vcc", "vcc_lo", "vcc_hi"):
        return {"vcc"}
    if tok in ("exec", "exec_lo", "exec_hi"):
        return {"exec"}
    return set()


def decode(text):
    """A coarse decode of one disassembled instruction: dict(mnemonic, valu, trans, dpp, dsts, srcs (data operands), src0, mask (the
    SGPR / VCC operand a select or carry instruction reads as its lane mask), lane_sel (v_readlane / v_writelane), states)."""
    parts = text.split(None, 1)
    mn = parts[0]
    rest = parts[1] if len(parts) > 1 else ""
    ops = [o.strip() for o in rest.split(",")] if rest else []
    dpp = mn.endswith("_dpp") or any(c in rest for c in _DPP_CTRL)          # (modifiers ride on the last operand token)
    valu = mn.startswith("v_") and mn != "v_nop"
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", mn)
    info = dict(mnemonic=mn, valu=valu, trans=base in TRANS, dpp=dpp, dsts=set(), srcs=set(), src0=set(), mask=set(), lane_sel=set(), states=1)
    if mn == "s_nop":
        info["states"] = int(ops[0], 0) + 1
        return info
    if not valu:
        # scalar instructions matter only as writers that END a hazard (the register no longer holds what the VALU wrote)
        if mn.startswith("s_") and ops and not mn.startswith(("s_cmp", "s_bitcmp", "s_waitcnt", "s_cbranch", "s_branch", "s_setprio", "s_sleep",
                                                                "s_barrier", "s_endpgm", "s_sendmsg", "s_setreg", "s_store", "s_dcache", "s_icache")):
            info["dsts"] = _regs(ops[0])
            if "saveexec" in mn:
                info["dsts"] |= {"exec"}
        return info
    n_dst = 1
    if base.startswith("v_cmpx"):
        info["dsts"] = {"exec"}
        n_dst = 0 if len(ops) == 2 else 1
    elif base.startswith("v_cmp"):
        if len(ops) == 2:                       # e32: implicit vcc
            info["dsts"] = {"vcc"}
            n_dst = 0
    elif base in ("v_mad_u64_u32", "v_mad_i64_i32", "v_div_scale_f32", "v_div_scale_f64", "v_add_co_u32", "v_sub_co_u32", "v_subrev_co_u32",
                  "v_addc_co_u32", "v_subb_co_u32", "v_subbrev_co_u32"):
        n_dst = 2
    for o in ops[:n_dst]:
        info["dsts"] |= _regs(o)
    src_ops = ops[n_dst:]
    if base in ("v_cndmask_b32", "v_addc_co_u32", "v_subb_co_u32", "v_subbrev_co_u32"):
        if len(src_ops) == 3:                   # explicit mask / carry-in: the last source
            info["mask"] = _regs(src_ops[-1])
            src_ops = src_ops[:-1]
        else:
            info["mask"] = {"vcc"}
    if base in ("v_div_fmas_f32", "v_div_fmas_f64"):
        info["mask"] = {"vcc"}
    if base in ("v_readlane_b32", "v_writelane_b32") and src_ops:
        info["lane_sel"] = _regs(src_ops[-1])
        src_ops = src_ops[:-1]
    for i, o in enumerate(src_ops):
        r = _regs(o)
        info["srcs"] |= r
        if i == 0:
            info["src0"] = r
    return info


def run_violations(run):
    """Violations inside one straight-line run: [(index, rule, consumer text, producer text)]."""
    dec = [decode(t) for t in run]
    out = []
    for i, c in enumerate(dec):
        if not c["valu"]:
            continue
        div_fmas = c["mnemonic"].startswith("v_div_fmas")
        swap = c["mnemonic"].startswith(("v_permlane16_swap", "v_permlane32_swap"))         # (both operands are read and written)
        states, j, rewritten = 0, i - 1, set()
        while j >= 0 and states < 4:
            p = dec[j]
            if p["valu"]:
                vg = {r for r in p["dsts"] if r[0] == "v" and r != "vcc"} - rewritten
                sg = {r for r in p["dsts"] if r[0] == "s" or r in ("vcc", "exec")} - rewritten
                if c["dpp"] and states < 2 and (vg & c["src0"]):
                    out.append((i, "VALU write -> DPP read needs 2 wait states", run[i], run[j]))
                if swap and states < 2 and (vg & (c["srcs"] | c["dsts"])):
                    out.append((i, "VALU write -> v_permlane16/32_swap operand needs 2 wait states", run[i], run[j]))
                if p["trans"] and not c["trans"] and states < 1 and (vg & c["srcs"]):
                    out.append((i, "transcendental result -> VALU read needs 1 wait state", run[i], run[j]))
                if states < 2 and (sg & {r for r in c["srcs"] if r[0] == "s" or r == "vcc"}):
                    out.append((i, "VALU write of SGPR/VCC -> VALU read as a constant needs 2 wait states", run[i], run[j]))
                if states < 4 and (sg & c["lane_sel"]):
                    out.append((i, "VALU write of SGPR -> v_readlane/v_writelane lane select needs 4 wait states", run[i], run[j]))
                if div_fmas and states < 4 and "vcc" in sg:
                    out.append((i, "VALU write of VCC -> v_div_fmas needs 4 wait states", run[i], run[j]))
            rewritten |= p["dsts"]
            states += p["states"]
            j -= 1
    return out


def wait_state_violations(obj, tmpdir=None):
    """[(kernel, rule, consumer, producer)] over every kernel of the object's gfx950 code."""
    d = tmpdir or tempfile.mkdtemp(prefix="ex4d_isa_")
    try:
        co = device_code(obj, d)
        if co is None:
            return []
        found = []
        for kernel, insts in disassembly(co):
            run = []
            for t in insts + ["s_endpgm"]:
                mn = t.split(None, 1)[0]
                if mn.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc", "s_barrier")):
                    found += [(kernel, rule, cons, prod) for _, rule, cons, prod in run_violations(run)]
                    run = []
                else:
                    run.append(t)
        return found
    finally:
        if not tmpdir:
            shutil.rmtree(d, ignore_errors=True)


def coverage(obj):
    """How much the wait-state check had to look at: {"dpp": DPP instructions, "trans": transcendental instructions, "valu_sgpr_writes":
    VALU instructions writing an SGPR / VCC / EXEC, "instructions": all} over the object's gfx950 code (a check that finds nothing
    because it parsed nothing would otherwise look the same as a clean object)."""
    d = tempfile.mkdtemp(prefix="ex4d_isa_")
    try:
        co = device_code(obj, d)
        n = dict(dpp=0, trans=0, valu_sgpr_writes=0, instructions=0)
        if co is None:
            return n
        for _, run in disassembly(co):
            for t in run:
                c = decode(t)
                n["instructions"] += 1
                n["dpp"] += c["valu"] and c["dpp"]
                n["trans"] += c["trans"]
                n["valu_sgpr_writes"] += c["valu"] and any(r[0] == "s" or r in ("vcc", "exec") for r in c["dsts"])
        return n
    finally:
        shutil.rmtree(d, ignore_errors=True)
