#!/usr/bin/env python3
"""Lint of the compiled kernels for the hazard behind round 3's lost carried scale:

    a SCALAR load (s_load_*, served by the scalar cache) of memory the same kernel also writes with VECTOR stores, where
    (A) the load can execute AFTER such a store -- the scalar cache is not coherent with vector stores -- or
    (B) the store can issue while the load has not been waited for (no ``s_waitcnt lgkmcnt(0)`` on some path between them):
        the two memory paths are not ordered by the hardware, so the store can overtake the load.

The compiler picks a scalar load for any uniform address it believes unclobbered *before* the load; nothing stops a later
store of the same wave from reaching memory first.  ``k_colscale_after_truncate`` did exactly that (tests/golden/
k_colscale_r03.s is the ISA it compiled to, kept as the known-bad input).

    python arnoldi-py_amd/csrc/check_scalar_hazards.py            # compiles csrc/aks_kernels.hip to ISA and checks it
    python arnoldi-py_amd/csrc/check_scalar_hazards.py FILE.s     # checks an assembly listing
    ... -v                                                        # every kernel's coverage line, not only the notable ones

How (round 5: whole control-flow graph, every address followed):

  * per kernel the instructions form a control-flow graph (labels, s_branch, s_cbranch_*); a forward may-analysis runs to
    its fixed point over it -- loops included -- and gives for every register the set of KERNEL ARGUMENTS its value may
    derive from.  Kernel-argument loads seed it (the kernarg pointer is s[0:1]); EVERY scalar or vector ALU instruction
    hands the union of its sources' sets to its destination (s_mul / s_lshl / v_mad_u64_u32 / ... -formed offsets
    included: no opcode list to fall behind); values loaded from memory derive from nothing.  (A vector instruction is
    taken to overwrite its destination in the lanes that later use it -- the EXEC mask is not modelled; v_writelane, which
    SGPR spills go through, adds to its destination's set.)
  * a memory instruction's address is *followed* when its set holds a POINTER argument -- ``global_buffer`` arguments and
    the words of by-value structs, from the code object's metadata (listings without metadata: every argument counts).
  * a scalar data load L and a vector store / atomic S are RELATED when their pointer sets intersect, or when either could
    not be followed (then it may point anywhere).  A related pair passes if both go through the very same, loop-invariant
    SGPR base pair with disjoint constant byte ranges (fields of one struct, e.g. the control block); otherwise
    rule (A): S must not reach L in the graph, and rule (B): every path from L to S must cross ``s_waitcnt lgkmcnt(0)``.
  * per kernel the followed fractions are reported, and a kernel that scalar-loads and stores through one argument (has a
    related pair) with ANY unfollowed scalar load or vector store is itself a finding: below 100 % the verdict on that
    kernel would rest on a guess.

Exit 1 and the candidates' names if there are any; exit 0 = none.
"""
import os
import re
import subprocess
import sys
import tempfile
from collections import defaultdict

HERE = os.path.dirname(os.path.abspath(__file__))
KARG = "karg"                       # root of the kernarg segment pointer itself
ANYARG = "anyarg"                   # an argument read at a non-constant offset: may be any of them
REG1 = re.compile(r"^([sva])(\d+)$")
REGN = re.compile(r"^([sva])\[(\d+):(\d+)\]$")
LOAD_BYTES = {"dword": 4, "dwordx2": 8, "dwordx3": 12, "dwordx4": 16, "dwordx8": 32, "dwordx16": 64}
STORE_BYTES = {"byte": 1, "short": 2, "dword": 4, "dwordx2": 8, "dwordx3": 12, "dwordx4": 16}
TWO_DESTS = ("v_add_co_u32", "v_sub_co_u32", "v_subrev_co_u32", "v_addc_co_u32", "v_subb_co_u32", "v_subbrev_co_u32",
             "v_mad_u64_u32", "v_mad_i64_i32", "v_div_scale_f64", "v_div_scale_f32", "v_add_co_ci_u32", "v_sub_co_ci_u32")
EMPTY = frozenset()


def regs(op):
    """Register names ('s4', 'v12', 'a3', ...) of one operand; [] for immediates, vcc, exec, off, labels, modifiers."""
    op = op.strip().split()[0] if op.strip() else ""
    op = op.lstrip("-|").rstrip("|")
    m = REG1.match(op)
    if m:
        return [m.group(1) + m.group(2)]
    m = REGN.match(op)
    if m:
        return [m.group(1) + str(i) for i in range(int(m.group(2)), int(m.group(3)) + 1)]
    return []


def imm(a):
    a = a.strip()
    return int(a, 0) if re.match(r"^(0x[0-9a-fA-F]+|\d+)$", a) else None


def store_bytes(op):
    tail = op.split("_")
    if "atomic" in tail:
        return 8 if tail[-1] in ("x2", "f64", "u64", "i64", "b64") else 4
    return STORE_BYTES.get(tail[-1])


class Insn:
    __slots__ = ("no", "text", "op", "args", "kind", "block")

    def __init__(self, no, text):
        self.no, self.text = no, text
        parts = text.split(None, 1)
        self.op = parts[0]
        self.args = [a.strip() for a in parts[1].split(",")] if len(parts) > 1 else []
        op = self.op
        if op.startswith("s_load_") or op.startswith("s_buffer_load"):
            self.kind = "sload"
        elif op.startswith(("global_store", "flat_store", "buffer_store", "scratch_store")):
            self.kind = "vstore"
        elif op.startswith(("global_atomic", "flat_atomic", "buffer_atomic")):
            self.kind = "vatomic"
        elif op.startswith(("global_load", "flat_load", "buffer_load", "scratch_load", "ds_read", "ds_load", "ds_bpermute",
                            "ds_permute", "ds_swizzle", "ds_consume", "ds_append")):
            self.kind = "vload"
        elif op.startswith("ds_"):
            self.kind = "lds"                      # LDS writes / atomics: not global memory
        elif op == "s_waitcnt":
            self.kind = "wait"
        elif op in ("s_branch",) or op.startswith("s_cbranch"):
            self.kind = "branch"
        elif op in ("s_endpgm", "s_endpgm_saved"):
            self.kind = "end"
        else:
            self.kind = "alu"

    def waits_for_scalar_loads(self):
        return "lgkmcnt(0)" in self.text or re.search(r"s_waitcnt\s+(0x0+|0)\s*$", self.text) is not None


def build_cfg(lines):
    """(instructions, successor lists) of one kernel body [(line no, raw text)]."""
    insns, labels = [], {}
    for no, raw in lines:
        text = raw.split(";")[0].strip()
        if not text:
            continue
        m = re.match(r"^([.\w$]+):$", text)
        if m:
            labels[m.group(1)] = len(insns)
            continue
        if text.startswith("."):
            continue
        insns.append(Insn(no, text))
    succ = [[] for _ in insns]
    for i, ins in enumerate(insns):
        if ins.kind == "end":
            continue
        if ins.kind == "branch":
            tgt = labels.get(ins.args[0]) if ins.args else None
            if tgt is not None and tgt < len(insns):
                succ[i].append(tgt)
            if ins.op == "s_branch":
                continue
        if i + 1 < len(insns):
            succ[i].append(i + 1)
    return insns, succ


def basic_blocks(insns, succ):
    leaders = {0}
    for i, ins in enumerate(insns):
        if ins.kind in ("branch", "end") and i + 1 < len(insns):
            leaders.add(i + 1)
        if ins.kind == "branch":
            for t in succ[i]:
                leaders.add(t)
    order = sorted(leaders)
    blocks, block_of = [], [0] * len(insns)
    for b, start in enumerate(order):
        end = order[b + 1] if b + 1 < len(order) else len(insns)
        blocks.append((start, end))
        for i in range(start, end):
            block_of[i] = b
            insns[i].block = b
    bsucc = [sorted({block_of[t] for t in succ[end - 1]}) if end > start else [] for start, end in blocks]
    return blocks, block_of, bsucc


def karg_base(base, roots, defs):
    """(is this SGPR pair the kernarg pointer or a copy of it plus something, the constant it was moved by or None)."""
    if len(base) != 2 or not all(roots.get(r, EMPTY) == frozenset([KARG]) for r in base):
        return False, None
    lo, hi = defs.get(("k", base[0])), defs.get(("k", base[1]))
    return True, (lo if isinstance(lo, int) and hi == "hi" else None)


def transfer(ins, i, roots, defs):
    """Apply one instruction to (roots, defs) in place.  roots: register -> frozenset of argument roots its value may
    derive from; defs: register -> index of the defining instruction (-1: kernel entry, None: differs between paths)."""
    k, a = ins.kind, ins.args
    if k in ("vstore", "wait", "branch", "end", "lds") or not a:
        return
    if k == "vatomic" and "sc0" not in ins.text.split(a[-1])[-1] and len(a) < 4:
        return                                              # no returned value
    dst = regs(a[0])
    if k == "sload":
        base = regs(a[1]) if len(a) > 1 else []
        off = imm(a[2]) if len(a) > 2 else None
        karg, delta = karg_base(base, roots, defs)
        for n, d in enumerate(dst):
            if karg and off is not None and delta is not None:
                roots[d] = frozenset([(delta + off + 4 * n) & ~7])   # a kernel argument: the 8-byte word it belongs to
            elif karg:
                roots[d] = frozenset([ANYARG])              # an argument at an offset that is not a constant
            else:
                roots[d] = EMPTY                            # data from memory derives from no argument
            defs[d] = i
            defs.pop(("k", d), None)
        return
    if k in ("vload", "vatomic"):
        # a VECTOR load from the kernarg segment (a by-value struct of pointers indexed per lane: k_oneshot_post's
        # P.box[peer]) yields some argument; any other load yields data that derives from no argument
        src = EMPTY
        for operand in a[1:]:
            for r in regs(operand):
                src |= roots.get(r, EMPTY)
        loaded = frozenset([ANYARG]) if (k == "vload" and KARG in src and ins.op.startswith(("global_load", "flat_load"))) else EMPTY
        for d in dst:
            roots[d] = loaded
            defs[d] = i
            defs.pop(("k", d), None)
        return
    # the kernarg pointer plus a constant (hidden arguments are read through such a copy): remember the constant
    kd = {}
    if ins.op in ("s_mov_b32", "s_mov_b64") and len(a) == 2:
        srcs = regs(a[1])
        if len(srcs) == len(dst):
            kd = {d: defs.get(("k", s_)) for d, s_ in zip(dst, srcs)}
    elif ins.op == "s_add_u32" and len(a) == 3 and len(dst) == 1:
        r0, c = regs(a[1]), imm(a[2])
        if len(r0) == 1 and c is not None and isinstance(defs.get(("k", r0[0])), int):
            kd = {dst[0]: defs[("k", r0[0])] + c}
    elif ins.op == "s_addc_u32" and len(a) == 3 and len(dst) == 1:
        r0 = regs(a[1])
        if len(r0) == 1 and imm(a[2]) == 0 and defs.get(("k", r0[0])) == "hi":
            kd = {dst[0]: "hi"}
    # ALU: destination(s) <- union of the sources' roots
    n_dst = 1
    if ins.op.startswith(TWO_DESTS) and len(a) > 1:
        n_dst = 2
    if ins.op.startswith(("s_cmp", "s_bitcmp", "v_cmpx", "s_nop", "s_barrier", "s_sleep", "s_setprio", "s_setreg", "s_sendmsg",
                          "s_waitcnt_", "s_inst_prefetch", "s_dcache", "s_icache", "buffer_wbl2", "buffer_inv", "s_trap",
                          "s_version", "s_code_end", "s_wait_", "s_clause", "s_delay", "s_set_gpr", "s_ttrace")):
        return
    src = EMPTY
    for operand in a[n_dst:]:
        for r in regs(operand):
            src |= roots.get(r, EMPTY)
    boolean = ins.op.startswith(("v_cmp", "s_and_saveexec", "s_or_saveexec", "s_andn2_saveexec", "s_xor_saveexec"))
    if ins.op.startswith("v_writelane"):                    # one lane of the register changes (SGPR spills live in such lanes)
        for d in dst:
            src |= roots.get(d, EMPTY)
    for d in dst:
        roots[d] = EMPTY if boolean else src
        defs[d] = i
        if kd.get(d) is not None:
            defs[("k", d)] = kd[d]
        else:
            defs.pop(("k", d), None)
    if n_dst == 2:
        for d in regs(a[1]):                                # carry-out / scale flag: not an address
            roots[d] = EMPTY
            defs[d] = i


def analyse(insns, succ):
    """Fixed point of the may-derive-from analysis; returns per-instruction (roots, defs) IN states of the instructions
    that matter (scalar loads, vector stores / atomics)."""
    blocks, block_of, bsucc = basic_blocks(insns, succ)
    entry_roots = {"s0": frozenset([KARG]), "s1": frozenset([KARG])}
    entry_defs = {"s0": -1, "s1": -1, ("k", "s0"): 0, ("k", "s1"): "hi"}
    in_roots = [None] * len(blocks)
    in_defs = [None] * len(blocks)
    in_roots[0], in_defs[0] = dict(entry_roots), dict(entry_defs)
    work, queued = [0], {0}
    while work:
        b = work.pop()
        queued.discard(b)
        roots, defs = dict(in_roots[b]), dict(in_defs[b])
        start, end = blocks[b]
        for i in range(start, end):
            transfer(insns[i], i, roots, defs)
        for t in bsucc[b]:
            changed = False
            if in_roots[t] is None:
                in_roots[t], in_defs[t] = dict(roots), dict(defs)
                changed = True
            else:
                tr, td = in_roots[t], in_defs[t]
                for r, v in roots.items():
                    old = tr.get(r)
                    if old is None:
                        tr[r] = v
                        changed = changed or bool(v)
                    elif not v <= old:
                        tr[r] = old | v
                        changed = True
                for r in set(td) | set(defs):
                    if td.get(r, "absent") != defs.get(r, "absent") and td.get(r, "absent") is not None:
                        td[r] = None                        # different definitions meet here
                        changed = True
            if changed and t not in queued:
                queued.add(t)
                work.append(t)
    states = {}
    for b, (start, end) in enumerate(blocks):
        if in_roots[b] is None:
            continue                                        # unreachable
        roots, defs = dict(in_roots[b]), dict(in_defs[b])
        for i in range(start, end):
            if insns[i].kind in ("sload", "vstore", "vatomic"):
                states[i] = (dict(roots), dict(defs))
            transfer(insns[i], i, roots, defs)
    return blocks, block_of, bsucc, states


def in_cycle(i, insns, block_of, bsucc, blocks):
    """Whether instruction i can execute more than once (its block lies on a cycle)."""
    b0 = block_of[i]
    seen, stack = set(), list(bsucc[b0])
    while stack:
        b = stack.pop()
        if b == b0:
            return True
        if b in seen:
            continue
        seen.add(b)
        stack.extend(bsucc[b])
    return False


def check_kernel(name, lines, ptr_roots=None):
    """Findings [(kernel, line no, text)] of one kernel body; coverage goes to check_kernel.report[name]."""
    insns, succ = build_cfg(lines)
    if not insns:
        return []
    blocks, block_of, bsucc, states = analyse(insns, succ)

    def pointers(rootset):
        if ptr_roots is None:
            return frozenset(r for r in rootset if r != KARG)
        if ANYARG in rootset:
            return frozenset(ptr_roots) | frozenset([ANYARG])
        return frozenset(r for r in rootset if r in ptr_roots)

    loads, stores = [], []
    for i, ins in enumerate(insns):
        if i not in states:
            continue
        roots, defs = states[i]
        a = ins.args
        if ins.kind == "sload":
            base = regs(a[1]) if len(a) > 1 else []
            if karg_base(base, roots, defs)[0]:
                continue                                    # a kernel-argument load
            addr = EMPTY
            for operand in a[1:]:
                for r in regs(operand):
                    addr |= roots.get(r, EMPTY)
            off = imm(a[2]) if len(a) > 2 else None
            m = re.search(r"offset:(0x[0-9a-fA-F]+|\d+)", ins.text)
            if off is None and m and not (len(a) > 2 and regs(a[2])):
                off = int(m.group(1), 0)
            size = LOAD_BYTES.get(ins.op.split("_")[-1])
            loads.append({"i": i, "ptr": pointers(addr), "base": tuple((r, defs.get(r)) for r in base),
                          "range": (off, off + size) if off is not None and size else None})
        else:
            ret = ins.kind == "vatomic" and len(a) >= 4
            addr_ops = a[1:2] + a[3:] if ret else a[0:1] + a[2:]                 # skip the data operand (and the returned value)
            if ins.op.startswith("buffer_"):
                addr_ops = a[1:]
            addr = EMPTY
            saddr = None
            for operand in addr_ops:
                rs = regs(operand)
                for r in rs:
                    addr |= roots.get(r, EMPTY)
                if len(rs) == 2 and rs[0][0] == "s":
                    saddr = tuple((r, defs.get(r)) for r in rs)
            vaddr = regs(addr_ops[0]) if addr_ops else []
            vzero = False
            if saddr is not None and len(vaddr) == 1:
                d = defs.get(vaddr[0])
                vzero = d is not None and d >= 0 and re.match(r"^v_mov_b32(_e32)?\s+v\d+,\s*(0|0x0)$", insns[d].text) is not None
            m = re.search(r"offset:(-?\d+)", ins.text)
            stores.append({"i": i, "ptr": pointers(addr), "saddr": saddr, "off": int(m.group(1)) if m else 0,
                           "size": store_bytes(ins.op), "vzero": vzero})

    first_wait = []
    for start, end in blocks:
        first_wait.append(next((i for i in range(start, end) if insns[i].kind == "wait" and insns[i].waits_for_scalar_loads()), None))

    def reaches(src, dst):
        """dst executes after src on some path (>= 1 step)."""
        bs, bd = block_of[src], block_of[dst]
        if bs == bd and src < dst:
            return True
        seen, stack = set(), list(bsucc[bs])
        while stack:
            b = stack.pop()
            if b == bd:
                return True
            if b in seen:
                continue
            seen.add(b)
            stack.extend(bsucc[b])
        return False

    def reaches_unwaited(load, store):
        """store can issue after load with no s_waitcnt lgkmcnt(0) in between on some path."""
        bl = block_of[load]
        start, end = blocks[bl]
        for i in range(load + 1, end):
            if i == store:
                return True
            if insns[i].kind == "wait" and insns[i].waits_for_scalar_loads():
                return False
        seen, stack = set(), list(bsucc[bl])
        while stack:
            b = stack.pop()
            if b in seen:
                continue
            seen.add(b)
            w = first_wait[b]
            if block_of[store] == b and (w is None or store < w):
                return True
            if w is None:
                stack.extend(bsucc[b])
        return False

    findings, related = [], 0
    loop_cache = {}

    def invariant(key):
        """An SGPR base pair defined once, outside every loop (or at kernel entry)."""
        for _, d in key:
            if d is None:
                return False
            if d >= 0:
                if d not in loop_cache:
                    loop_cache[d] = in_cycle(d, insns, block_of, bsucc, blocks)
                if loop_cache[d]:
                    return False
        return True

    for ld in loads:
        for st in stores:
            lp, sp = ld["ptr"], st["ptr"]
            if lp and sp and not (lp & sp):
                continue                                    # two different arguments
            related += 1
            if (st["saddr"] is not None and st["saddr"] == ld["base"] and ld["range"] and st["size"] and st["vzero"]
                    and invariant(ld["base"])):
                lo, hi = ld["range"]
                if st["off"] + st["size"] <= lo or st["off"] >= hi:
                    continue                                # same base, disjoint bytes of one struct
            li, si = insns[ld["i"]], insns[st["i"]]
            which = (f"argument +{min(lp & sp):#x}" if lp and sp else "memory that cannot be told apart from it")
            if reaches(st["i"], ld["i"]):
                findings.append((name, li.no, f"scalar load of {which} can execute AFTER the vector store of line {si.no} "
                                              f"({si.text}): the scalar cache is not coherent with vector stores | {li.text}"))
            elif reaches_unwaited(ld["i"], st["i"]):
                findings.append((name, si.no, f"vector store to {which} while the scalar load of line {li.no} ({li.text}) "
                                              f"has not been waited for: the store can overtake it | {si.text}"))
    lf = sum(1 for x in loads if x["ptr"])
    sf = sum(1 for x in stores if x["ptr"])
    rep = {"scalar_loads": len(loads), "scalar_loads_followed": lf, "stores": len(stores), "stores_followed": sf,
           "related_pairs": related, "line": insns[0].no}
    check_kernel.report[name] = rep
    if related and (lf < len(loads) or sf < len(stores)):
        findings.append((name, insns[0].no, f"scalar-loads and stores through one argument but only {lf} of {len(loads)} scalar data "
                                             f"loads and {sf} of {len(stores)} vector stores could be followed to a kernel argument"))
    return findings


check_kernel.report = {}


def parse_metadata(text):
    """kernel symbol -> set of kernarg byte offsets that can hold a pointer (None if the listing has no metadata)."""
    m = re.search(r"\.amdgpu_metadata(.*?)\.end_amdgpu_metadata", text, re.S)
    if not m:
        return None
    out, args, cur = {}, [], None
    for line in m.group(1).split("\n"):
        s = line.strip()
        if s.startswith("- .") and line.startswith("  - "):       # next kernel
            args, cur = [], None
        if s.startswith("- ."):
            s = s[2:]
            if line.startswith("      - "):
                cur = {}
                args.append(cur)
        if cur is not None and line.startswith("        ") or line.startswith("      - "):
            mm = re.match(r"^\.(offset|size|value_kind):\s*(\S+)", s)
            if mm and cur is not None:
                cur[mm.group(1)] = mm.group(2)
        mm = re.match(r"^\.symbol:\s*(\S+)\.kd", s)
        if mm:
            ptrs = set()
            for a in args:
                if "offset" not in a:
                    continue
                off, size, kind = int(a["offset"]), int(a.get("size", 0)), a.get("value_kind", "")
                if kind == "global_buffer":
                    ptrs.add(off)
                elif kind == "by_value" and size > 8:                # a struct passed by value: any word may be a pointer
                    ptrs.update(range(off & ~7, off + size, 8))
            out[mm.group(1)] = ptrs
    return out


def check_listing(text):
    """Kernels = the text between a function label and its ``.Lfunc_end`` label (a kernel has several s_endpgm)."""
    meta = parse_metadata(text)
    findings, kernels = [], 0
    name, body = None, []
    for no, line in enumerate(text.split("\n"), 1):
        if name is None:
            m = re.match(r"^(_Z\w+|[A-Za-z_]\w*):\s*(;.*)?$", line)
            if m:
                name, body = m.group(1), []
            continue
        if re.match(r"^\.Lfunc_end\d+:", line):
            findings += check_kernel(name, body, meta.get(name) if meta is not None else None)
            kernels += 1
            name, body = None, []
        elif line.startswith("\t.end_amdgpu_metadata") or line.startswith("\t.amdgpu_metadata"):
            break
        else:
            body.append((no, line))
    if name is not None and body:                     # (a listing cut short: the known-bad fixture)
        findings += check_kernel(name, body, None)
        kernels += 1
    return findings, kernels


def compile_listing():
    src = os.path.join(HERE, "aks_kernels.hip")
    inc = os.path.join(HERE, "..", "..", "include")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "aks.s")
        cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + inc, "-S",
               "--cuda-device-only", "-o", out, src]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(out):
            raise SystemExit("hipcc -S failed:\n" + r.stderr[-2000:])
        return open(out).read()


def short(name):
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    return dem.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def main(argv):
    verbose = "-v" in argv
    argv = [a for a in argv if a != "-v"]
    text = open(argv[1]).read() if len(argv) > 1 else compile_listing()
    check_kernel.report.clear()
    findings, kernels = check_listing(text)
    tot = defaultdict(int)
    print("kernel: scalar data loads followed / all, vector stores followed / all, related (load, store) pairs")
    for name, r in check_kernel.report.items():
        for k, v in r.items():
            tot[k] += v
        notable = r["related_pairs"] or r["scalar_loads_followed"] < r["scalar_loads"] or r["stores_followed"] < r["stores"]
        if verbose or notable:
            print(f"  {short(name)}: loads {r['scalar_loads_followed']}/{r['scalar_loads']}, stores {r['stores_followed']}/{r['stores']}, "
                  f"related pairs {r['related_pairs']}")
    for name, no, what in findings:
        print(f"{short(name)}: line {no}: {what}")
    print(f"{kernels} kernels checked, {len(findings)} candidate hazard(s); followed to a kernel argument: "
          f"{tot['scalar_loads_followed']} of {tot['scalar_loads']} scalar data loads, "
          f"{tot['stores_followed']} of {tot['stores']} vector stores / atomics; {tot['related_pairs']} related (load, store) pair(s)"
          + (" (the unexplained ones are listed above)" if findings else
             ", each one explained (disjoint bytes of one struct, or the load is complete before the store can issue and never follows it)"))
    return 1 if findings else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
