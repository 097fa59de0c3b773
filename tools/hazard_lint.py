#!/usr/bin/env python3
"""Static wait-state (data hazard) lint over the gfx950 assembly the library is built from.

hipcc pads the hazards of ITS OWN instructions (GCNHazardRecognizer) but treats an `asm` statement as one opaque
instruction: nothing inside `;;#ASMSTART` .. `;;#ASMEND` is padded, and -- what cost `k_hodina_m` a round -- the registers an
asm statement WRITES are not checked against MFMAs that are still in flight.  This tool re-does the recogniser's job on the
final ISA with the inline-asm instructions included: for every instruction it walks back over all control-flow paths, counts
wait states (one per instruction, N + 1 for `s_nop N`) and reports every producer / consumer pair that is closer than the
table below requires.

Input: the device assembly of the library (`make -C vipsy_amd/csrc` keeps it: vipsy_amd/_lib/*gfx950.s), or any `hipcc -S
--cuda-device-only` output.  A pair whose producer or consumer sits inside an asm region is an ERROR (exit status 1); a pair
between two compiler-emitted instructions is reported as CALIBRATION (the table stricter than the compiler -- or a compiler
bug) and does not fail the run.

Table (gfx950; the compiler's own padding of probe kernels, tools/hazard_lint.py --selftest builds them, agrees with it):
  R1  MFMA writes D          -> any read or write of it that is not an MFMA            passes + 4 (XDL), passes + 2 (fp32 MFMA)
  R2  MFMA reads SrcC        -> vector write of it (write-after-read)                  passes - 1 (8 -> 7, 16 -> 13, 4 -> 3)
  R3  MFMA writes D          -> MFMA reads it as SrcA / SrcB                           as R1;  as SrcC: 0 when the ranges are
                                                                                       equal, else passes + 2 (XDL) / passes
  R4  vector write           -> MFMA reads it (SrcA / SrcB / SrcC)                     2
  R5  transcendental write   -> non-transcendental vector read                         1
  R6  write of a 16-bit half (v_fma_mixhi, op_sel dst, SDWA dst_sel) -> vector read    1
  R7  vector write           -> v_readlane / v_readfirstlane read                      1
  R8  vector write           -> DPP read 2;  vector write of EXEC -> DPP               5
  R9  vector write           -> v_permlane*_swap read                                  2
  R10 vector write of an SGPR / VCC -> vector read 2; -> lane select of v_readlane / v_writelane 4; -> vector-memory read 5;
      VCC -> v_div_fmas 4
  R11 scalar write of M0     -> LDS-DMA (`global_load_lds_*`, `buffer_load ... lds`), `ds_*_addtid`, GWS        1
  R13 store of more than 8 bytes -> vector write of its data registers                 2
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile
from collections import defaultdict

MAX_LOOKBACK = 21

# ---------------------------------------------------------------------------------------------------------------------
# parsing
# ---------------------------------------------------------------------------------------------------------------------
REG_RE = re.compile(r"\b([vas])\[(\d+):(\d+)\]|\b([vas])(\d+)\b|\b(vcc_lo|vcc_hi|vcc|exec_lo|exec_hi|exec|m0|scc)\b")
TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")
TWO_DST = ("v_div_scale_", "v_add_co_", "v_sub_co_", "v_subrev_co_", "v_addc_co_", "v_subb_co_", "v_subbrev_co_",
           "v_mad_u64_u32", "v_mad_i64_i32")
# instructions whose destination is also a source (accumulating forms, lane writes, the two-way swaps)
DST_ALSO_READ = ("v_fmac_", "v_mac_", "v_pk_fmac_", "v_dot2c_", "v_dot4c_", "v_dot8c_", "v_writelane_", "v_movreld_")
DPP_RE = re.compile(r"\b(quad_perm|row_shl|row_shr|row_ror|wave_shl|wave_shr|wave_rol|wave_ror|row_mirror|row_half_mirror|"
                    r"row_bcast|row_newbcast|dpp8)\b")


def regs_of(text):
    """set of (file, index) named in an operand string; vcc / exec / m0 as ('vcc', 0), ('vcc', 1), ('exec', 0/1), ('m0', 0)."""
    out = set()
    for m in REG_RE.finditer(text):
        if m.group(1):
            f, lo, hi = m.group(1), int(m.group(2)), int(m.group(3))
            for i in range(lo, hi + 1):
                out.add((f, i))
        elif m.group(4):
            out.add((m.group(4), int(m.group(5))))
        else:
            name = m.group(6)
            if name.startswith("vcc"):
                idx = {"vcc": (0, 1), "vcc_lo": (0,), "vcc_hi": (1,)}[name]
                out.update(("vcc", i) for i in idx)
            elif name.startswith("exec"):
                idx = {"exec": (0, 1), "exec_lo": (0,), "exec_hi": (1,)}[name]
                out.update(("exec", i) for i in idx)
            elif name == "m0":
                out.add(("m0", 0))
    return out


def split_operands(s):
    """top-level comma split (brackets of register ranges and op_sel lists kept together)"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "[(":
            depth += 1
        elif ch in "])":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


class Inst:
    __slots__ = ("line", "text", "mn", "ops", "in_asm", "defs", "uses", "kind", "passes", "xdl", "srcc", "srcab", "is_dpp",
                 "is_trans", "hi_write", "lane_sel", "waits", "store_data", "is_valu", "is_vmem", "is_ds", "is_salu",
                 "branch_target", "ends_block", "lds_dma")

    def __init__(self, line, text, in_asm):
        self.line, self.text, self.in_asm = line, text, in_asm
        body = text.split(";")[0].strip()
        parts = body.split(None, 1)
        self.mn = parts[0]
        opstr = parts[1] if len(parts) > 1 else ""
        self.ops = split_operands(opstr)
        mn = self.mn
        self.is_valu = mn.startswith("v_")
        self.is_salu = mn.startswith("s_") and not mn.startswith(("s_load", "s_buffer_load", "s_store", "s_waitcnt", "s_nop",
                                                                    "s_barrier", "s_branch", "s_cbranch", "s_endpgm",
                                                                    "s_sleep", "s_setprio", "s_sethalt", "s_dcache",
                                                                    "s_icache", "s_sendmsg", "s_trap", "s_memtime",
                                                                    "s_memrealtime", "s_atomic", "s_scratch"))
        self.is_vmem = mn.startswith(("global_", "buffer_", "flat_", "scratch_", "tbuffer_"))
        self.is_ds = mn.startswith("ds_")
        self.waits = 1
        if mn == "s_nop":
            self.waits = int(self.ops[0], 0) + 1
        self.branch_target = None
        self.ends_block = False
        if mn == "s_branch" or mn.startswith("s_cbranch"):
            self.branch_target = self.ops[-1] if self.ops else None
            self.ends_block = mn == "s_branch"
        if mn in ("s_endpgm", "s_setpc_b64", "s_swappc_b64"):
            self.ends_block = True
        self.kind = None
        self.passes = 0
        self.xdl = False
        self.srcc = set()
        self.srcab = set()
        self.is_dpp = bool(DPP_RE.search(opstr)) and self.is_valu
        self.is_trans = mn.startswith(TRANS)
        self.hi_write = False
        self.lane_sel = set()
        self.store_data = set()
        self.lds_dma = False
        self.defs, self.uses = set(), set()
        self._operands(opstr)

    def _operands(self, opstr):
        mn, ops = self.mn, self.ops
        # operand strings that carry registers (modifier lists such as op_sel:[..] hold none)
        reg_ops = [o for o in ops if REG_RE.search(o.split(":")[0] if re.match(r"^[a-z_0-9]+:", o) else o)]
        if self.is_valu:
            if mn.startswith("v_mfma") or mn.startswith("v_smfma"):
                self.kind = "mfma"
                d, a, b = reg_ops[0], reg_ops[1], reg_ops[2]
                self.defs = regs_of(d)
                self.srcab = regs_of(a) | regs_of(b)
                c = ops[3] if len(ops) > 3 else ""
                self.srcc = regs_of(c) if REG_RE.search(c) else set()
                self.uses = self.srcab | self.srcc
                m = re.match(r"v_mfma_[a-z0-9]+_(\d+)x(\d+)x(\d+)_?([a-z0-9_]+)", mn)
                M, _, K, ty = int(m.group(1)), int(m.group(2)), int(m.group(3)), m.group(4)
                if ty.startswith("f32") or ty == "xf32":
                    self.xdl = False                                     # "SGEMM": fp32 in, the vector rate
                    self.passes = {32: 16, 16: 8, 4: 2}[M]
                else:
                    self.xdl = True
                    self.passes = {32: 8, 16: 4, 4: 2}[M]
                    if "f8f6f4" in mn:
                        self.passes *= 2
                return
            if mn.startswith("v_cmpx"):
                self.defs = {("exec", 0), ("exec", 1)}
                self.uses = set().union(*[regs_of(o) for o in reg_ops]) if reg_ops else set()
                return
            if mn.startswith("v_cmp") and mn.endswith("_e32"):
                self.defs = {("vcc", 0), ("vcc", 1)}
                srcs = reg_ops[1:] if reg_ops and reg_ops[0].startswith("vcc") else reg_ops
                self.uses = set().union(*[regs_of(o) for o in srcs]) if srcs else set()
                return
            ndst = 2 if (mn.startswith(TWO_DST) or mn.startswith("v_permlane") and "swap" in mn or mn.startswith("v_swap_")) else 1
            if mn in ("v_nop",) or not reg_ops:
                return
            dst_ops = reg_ops[:ndst]
            src_ops = reg_ops[ndst:]
            for o in dst_ops:
                self.defs |= regs_of(o)
            for o in src_ops:
                self.uses |= regs_of(o)
            if mn.startswith(DST_ALSO_READ) or (mn.startswith("v_permlane") and "swap" in mn) or mn.startswith("v_swap_"):
                for o in dst_ops:
                    self.uses |= regs_of(o)
            if mn.startswith(("v_cndmask_b32_e32", "v_addc_co_u32_e32", "v_subb_co_u32_e32", "v_subbrev_co_u32_e32")) or \
                    mn.startswith("v_div_fmas"):
                self.uses |= {("vcc", 0), ("vcc", 1)}
            if mn.startswith(("v_readlane_", "v_writelane_")) and len(reg_ops) >= 3:
                self.lane_sel = {r for r in regs_of(reg_ops[2]) if r[0] in ("s", "vcc", "m0")}
            # 16-bit half writes that keep the other half: v_*_mixhi, op_sel with the dst bit, SDWA dst_sel
            if "mixhi" in mn or re.search(r"dst_sel:(WORD_1|BYTE_[0-3]|WORD_0)", opstr):
                self.hi_write = True
            m = re.search(r"op_sel:\[([01,]+)\]", opstr)
            if m and "mix" not in mn:
                bits = m.group(1).split(",")
                nsrc = len(src_ops)
                if len(bits) > nsrc and bits[-1] == "1":
                    self.hi_write = True
            return
        if self.is_vmem:
            is_store = "_store" in mn
            is_atomic = "_atomic" in mn
            self.lds_dma = "_lds_" in mn or mn.startswith("global_load_lds") or re.search(r"\blds\b", opstr) is not None
            if is_store:
                for o in reg_ops:
                    self.uses |= regs_of(o)
                if mn.startswith("buffer_") or mn.startswith("tbuffer_"):
                    data = regs_of(reg_ops[0]) if reg_ops else set()
                else:                                   # global / flat / scratch: addr, data, saddr
                    data = regs_of(reg_ops[1]) if len(reg_ops) > 1 else set()
                if len([r for r in data if r[0] in "va"]) > 2:
                    self.store_data = data
            elif self.lds_dma:
                for o in reg_ops:
                    self.uses |= regs_of(o)
                self.uses.add(("m0", 0))
            else:
                returns = (not is_atomic) or re.search(r"\b(glc|sc0)\b", opstr) is not None
                if returns and reg_ops:
                    self.defs = regs_of(reg_ops[0])
                    rest = reg_ops[1:]
                else:
                    rest = reg_ops
                for o in rest:
                    self.uses |= regs_of(o)
            return
        if self.is_ds:
            if "addtid" in mn or mn.startswith("ds_gws"):
                self.uses.add(("m0", 0))
            has_dst = mn.startswith(("ds_read", "ds_bpermute", "ds_permute", "ds_swizzle", "ds_consume", "ds_append")) or \
                "_rtn" in mn
            if has_dst and reg_ops:
                self.defs = regs_of(reg_ops[0])
                rest = reg_ops[1:]
            else:
                rest = reg_ops
            for o in rest:
                self.uses |= regs_of(o)
            return
        if mn.startswith(("s_load", "s_buffer_load")):
            if reg_ops:
                self.defs = regs_of(reg_ops[0])
                for o in reg_ops[1:]:
                    self.uses |= regs_of(o)
            return
        if self.is_salu:
            if mn.startswith(("s_cmp", "s_bitcmp", "s_setreg", "s_cbranch")):
                for o in reg_ops:
                    self.uses |= regs_of(o)
                return
            if reg_ops:
                self.defs = regs_of(reg_ops[0])
                for o in reg_ops[1:]:
                    self.uses |= regs_of(o)
            return
        # everything else (s_waitcnt, s_nop, s_barrier ...): no registers that matter
        if mn.startswith("s_cbranch_vcc"):
            self.uses |= {("vcc", 0), ("vcc", 1)}


class Func:
    def __init__(self, name):
        self.name = name
        self.insts = []            # Inst
        self.labels = {}           # label -> index of the first instruction behind it
        self.preds = None

    def build_cfg(self):
        n = len(self.insts)
        preds = [[] for _ in range(n)]
        for i, ins in enumerate(self.insts):
            if i + 1 < n and not ins.ends_block:
                preds[i + 1].append(i)
            if ins.branch_target is not None and ins.branch_target in self.labels:
                t = self.labels[ins.branch_target]
                if t < n:
                    preds[t].append(i)
        self.preds = preds


def parse(path):
    funcs = []
    cur = None
    in_asm = False
    pending_labels = []
    func_names = set()
    with open(path) as f:
        lines = f.readlines()
    for ln in lines:
        m = re.match(r"\s*\.type\s+([^,\s]+),@function", ln)
        if m:
            func_names.add(m.group(1))
    for no, ln in enumerate(lines, 1):
        s = ln.rstrip("\n")
        st = s.strip()
        if not st:
            continue
        m = re.match(r"^([A-Za-z_.$][\w.$]*):", s)
        if m:
            lab = m.group(1)
            if lab in func_names:
                cur = Func(lab)
                funcs.append(cur)
                in_asm = False
                pending_labels = []
            elif cur is not None:
                pending_labels.append(lab)
            continue
        if cur is None:
            continue
        if st.startswith(".Lfunc_end"):
            cur = None
            continue
        if st.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if st.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if st.startswith((";", ".", "//")):
            if re.match(r"^\.Lfunc_end", st):
                cur = None
            continue
        if not s.startswith(("\t", " ")):
            continue
        ins = Inst(no, st, in_asm)
        for lab in pending_labels:
            cur.labels[lab] = len(cur.insts)
        pending_labels = []
        cur.insts.append(ins)
    for fn in funcs:
        fn.build_cfg()
    return funcs


# ---------------------------------------------------------------------------------------------------------------------
# the table
# ---------------------------------------------------------------------------------------------------------------------
def vec(rs):
    return {r for r in rs if r[0] in ("v", "a")}


def sgpr_like(rs):
    return {r for r in rs if r[0] in ("s", "vcc")}


def required(p, c):
    """list of (rule, needed wait states, registers) for producer p in front of consumer c"""
    out = []
    pv, cv_def, cv_use = vec(p.defs), vec(c.defs), vec(c.uses)
    if p.kind == "mfma":
        # XDL: passes + 4 (2-pass: 5).  The fp32-input forms: what the compiler leaves in front of a reader of
        # v_mfma_f32_32x32x2_f32 in this library is 11 states (tools/hazard_lint.py --calibration prints the minima)
        raw_waw = p.passes + 4 if p.xdl else {16: 11, 8: 7, 2: 4}.get(p.passes, p.passes + 2)
        if p.passes == 2 and p.xdl:
            raw_waw = 5
        if c.kind == "mfma":
            ab = pv & c.srcab
            if ab:
                out.append(("R3ab", raw_waw, ab))
            cc = pv & c.srcc
            if cc:
                if p.defs == c.srcc and p.xdl == c.xdl:
                    need = 0
                elif p.xdl:
                    need = p.passes if not c.xdl and p.defs == c.srcc else p.passes + (1 if p.passes == 2 else 2)
                else:
                    need = p.passes
                if need:
                    out.append(("R3c", need, cc))
        else:
            hit = pv & (cv_use | cv_def)
            if hit and (c.is_valu or c.is_vmem or c.is_ds):
                out.append(("R1", raw_waw, hit))
        war = p.srcc & cv_def
        if war and c.is_valu and c.kind != "mfma":
            out.append(("R2", {2: 1, 4: 3, 8: 7, 16: 13}.get(p.passes, 13), war))
        return out
    if p.is_valu:
        if c.kind == "mfma":
            hit = pv & c.uses
            if hit:
                out.append(("R4", 2, hit))
        if c.is_valu and c.kind != "mfma":
            if p.is_trans and not c.is_trans:
                hit = pv & cv_use
                if hit:
                    out.append(("R5", 1, hit))
            if p.hi_write:
                hit = pv & cv_use
                if hit:
                    out.append(("R6", 1, hit))
            if c.mn.startswith(("v_readlane", "v_readfirstlane")):
                hit = pv & cv_use
                if hit:
                    out.append(("R7", 1, hit))
            if c.is_dpp:
                hit = pv & cv_use
                if hit:
                    out.append(("R8", 2, hit))
                if ("exec", 0) in p.defs or ("exec", 1) in p.defs:
                    out.append(("R8x", 5, {("exec", 0)}))
            if c.mn.startswith("v_permlane") and "swap" in c.mn:
                hit = pv & cv_use
                if hit:
                    out.append(("R9", 2, hit))
        ps = sgpr_like(p.defs)
        if ps:
            if c.is_valu:
                hit = ps & c.lane_sel
                if hit:
                    out.append(("R10l", 4, hit))
                hit = ps & sgpr_like(c.uses) - c.lane_sel
                if hit:
                    out.append(("R10", 2, hit))
                if c.mn.startswith("v_div_fmas") and (("vcc", 0) in ps or ("vcc", 1) in ps):
                    out.append(("R10d", 4, {("vcc", 0)}))
            if c.is_vmem:
                hit = ps & sgpr_like(c.uses)
                if hit:
                    out.append(("R10m", 5, hit))
        return out
    if p.is_salu:
        if ("m0", 0) in p.defs and ("m0", 0) in c.uses and (c.lds_dma or c.is_ds):
            out.append(("R11", 1, {("m0", 0)}))
        return out
    if p.is_vmem and p.store_data:
        hit = vec(p.store_data) & cv_def
        if hit and c.is_valu:
            out.append(("R13", 2, hit))
    return out


def physical_distance(p, path):
    """wait states between MFMA `p` and the consumer when the instructions between them (`path`, in program order) issue
    as early as the hardware lets them: one state per instruction (N + 1 for s_nop N), and an MFMA not before the matrix
    pipe has finished the passes of the MFMA in front of it (one pipe a SIMD: 32 cycles per 32x32x16 back to back whether or
    not the accumulators differ, MI355X_MICROARCH.md cycle constants).  The compiler counts such an MFMA as ONE state; the
    kernels that place asm reads "four MFMAs behind their producer" (k_mvn_bwd_hb2.hip) rely on the pipe instead."""
    t = 0                     # issue time of the instruction in hand, in wait states after p's issue
    free = p.passes           # when the matrix pipe has finished what was issued so far
    nxt = 1                   # earliest issue of the next instruction
    for ins in path:
        t = nxt
        if ins.kind == "mfma":
            t = max(t, free)
            free = t + ins.passes
        nxt = t + ins.waits
    return nxt - 1


def lint(funcs, only=None, strict=False, everything=False):
    """findings: (function, rule, needed, distance, producer, consumer, registers).  `strict`: count every instruction as one
    wait state, as the compiler's recogniser does; `everything`: also pairs of two compiler-emitted instructions."""
    findings = []
    for fn in funcs:
        if only and not any(o in fn.name for o in only):
            continue
        insts, preds = fn.insts, fn.preds
        n = len(insts)
        if not everything:
            if not any(i.in_asm for i in insts):
                continue
            # consumers worth a walk: asm instructions, and whatever issues within the lookback behind one
            succs = [[] for _ in range(n)]
            for i in range(n):
                for q in preds[i]:
                    succs[q].append(i)
            cand = [False] * n
            for i, ins in enumerate(insts):
                if not ins.in_asm:
                    continue
                cand[i] = True
                seen = {}
                stack = [(j, ins.waits - 1) for j in succs[i]]
                while stack:
                    j, w = stack.pop()
                    if w >= MAX_LOOKBACK or (j in seen and seen[j] <= w):
                        continue
                    seen[j] = w
                    cand[j] = True
                    for k in succs[j]:
                        stack.append((k, w + insts[j].waits))
        for ci, c in enumerate(insts):
            if not (c.defs or c.uses):
                continue
            if not everything and not cand[ci]:
                continue
            # walk back over every path: (index, wait states between that instruction and c, MFMAs passed, path)
            seen = {}
            stack = [(pi, 0, 0, ()) for pi in preds[ci]]
            while stack:
                pi, w, nm, path = stack.pop()
                if w >= MAX_LOOKBACK:
                    continue
                if (pi, nm) in seen and seen[(pi, nm)] <= w:
                    continue
                seen[(pi, nm)] = w
                p = insts[pi]
                if (p.defs or p.srcc or p.store_data) and (everything or p.in_asm or c.in_asm):
                    for rule, need, regs in required(p, c):
                        dist = w
                        if p.kind == "mfma" and nm and not strict:
                            dist = physical_distance(p, path)
                        if dist < need:
                            findings.append((fn.name, rule, need, dist, p, c, regs))
                npath = (p,) + path
                nw = w + p.waits
                nnm = nm + (1 if p.kind == "mfma" else 0)
                for q in preds[pi]:
                    stack.append((q, nw, nnm, npath))
    # one finding per (consumer line, producer line, rule): the shortest distance
    best = {}
    for f in findings:
        key = (f[0], f[1], f[4].line, f[5].line)
        if key not in best or f[3] < best[key][3]:
            best[key] = f
    return sorted(best.values(), key=lambda f: (f[0], f[5].line))


def fmt_regs(regs):
    by = defaultdict(list)
    for f, i in sorted(regs):
        by[f].append(i)
    return " ".join("%s{%s}" % (f, ",".join(map(str, idx[:6])) + ("..." if len(idx) > 6 else "")) for f, idx in by.items())


SELFTEST_SRC = r"""
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// the shape of the k_hodina_m fault: an asm statement's scratch output lands in the dead upper registers of an MFMA result
__global__ void bad_waw(float* o, const bf16x8* in) {
    bf16x8 a = in[threadIdx.x], b = in[threadIdx.x + 64];
    f32x16 c; for (int i = 0; i < 16; ++i) c[i] = 0.f;
    float s = o[threadIdx.x];
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    unsigned x = __builtin_bit_cast(unsigned, s), t;
    asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "=&v"(t));
    o[threadIdx.x] = c[0] + c[1] + __builtin_bit_cast(float, x) + __builtin_bit_cast(float, t);
}
__global__ void bad_raw(float* o, const bf16x8* in) {            // an MFMA result read inside asm right behind it
    bf16x8 a = in[threadIdx.x], b = in[threadIdx.x + 64];
    f32x16 c; for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    float r;
    asm volatile("v_add_f32 %0, %1, %1" : "=v"(r) : "v"(c[3]));
    o[threadIdx.x] = r;
}
__global__ void good(float* o, const bf16x8* in) {
    bf16x8 a = in[threadIdx.x], b = in[threadIdx.x + 64];
    f32x16 c; for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    float e = __builtin_amdgcn_exp2f(c[3]) + 1.0f;
    int d = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, e), 0xB1, 0xF, 0xF, true);
    o[threadIdx.x] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(d, 63)) + c[0];
}
"""


def selftest(hipcc):
    with tempfile.TemporaryDirectory() as td:
        src, out = os.path.join(td, "t.hip"), os.path.join(td, "t.s")
        with open(src, "w") as f:
            f.write(SELFTEST_SRC)
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "--cuda-device-only", "-S", "-o", out, src],
                              stderr=subprocess.DEVNULL)
        res = lint(parse(out), everything=True)
    by = defaultdict(list)
    for f in res:
        by[f[0]].append(f)
    ok = True
    for name, want in (("bad_raw", "R1"), ("good", None)):
        fn = [k for k in by if name in k]
        got = sorted({f[1] for k in fn for f in by[k] if f[4].in_asm or f[5].in_asm})
        if want is None:
            if any(by[k] for k in fn):
                print("selftest: compiler-padded kernel flagged:", [(f[1], f[4].text, f[5].text) for k in fn for f in by[k]])
                ok = False
        elif want not in got:
            print("selftest: %s not flagged with %s (got %s)" % (name, want, got))
            ok = False
    print("selftest", "ok" if ok else "FAILED", "(bad_waw findings: %s)" %
          sorted({f[1] for k in by if "bad_waw" in k for f in by[k]}))
    return ok


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("asm", nargs="?", help="device assembly (.s)")
    ap.add_argument("--only", action="append", help="only functions whose name contains this")
    ap.add_argument("--selftest", action="store_true")
    ap.add_argument("--hipcc", default="/opt/rocm/bin/hipcc")
    ap.add_argument("--calibration", action="store_true", help="also list the compiler-only pairs")
    ap.add_argument("--strict", action="store_true", help="count an MFMA between producer and consumer as ONE wait state (the "
                    "compiler's rule) instead of the time it holds the matrix pipe")
    ap.add_argument("--quiet", action="store_true")
    a = ap.parse_args()
    if a.selftest:
        sys.exit(0 if selftest(a.hipcc) else 1)
    funcs = parse(a.asm)
    res = lint(funcs, a.only, strict=a.strict, everything=a.calibration)
    errs = [f for f in res if f[4].in_asm or f[5].in_asm]
    cal = [f for f in res if not (f[4].in_asm or f[5].in_asm)]
    n_asm = sum(1 for fn in funcs for i in fn.insts if i.in_asm)
    if not a.quiet:
        for f in errs + (cal if a.calibration else []):
            name, rule, need, have, p, c, regs = f
            tag = "ERROR" if (p.in_asm or c.in_asm) else "calibration"
            print("%s %s in %s: needs %d wait states, has %d, registers %s" % (tag, rule, name, need, have, fmt_regs(regs)))
            print("    %7d%s  %s" % (p.line, " asm" if p.in_asm else "    ", p.text))
            print("    %7d%s  %s" % (c.line, " asm" if c.in_asm else "    ", c.text))
    by_rule = defaultdict(int)
    for f in cal:
        by_rule[f[1]] += 1
    print("hazard_lint: %d functions, %d instructions (%d inside asm statements): %d asm hazards, %d compiler-only pairs %s" %
          (len(funcs), sum(len(fn.insts) for fn in funcs), n_asm, len(errs), len(cal), dict(by_rule) if cal else ""))
    sys.exit(1 if errs else 0)


if __name__ == "__main__":
    main()
