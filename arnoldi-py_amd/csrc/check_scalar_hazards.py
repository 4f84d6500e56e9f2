#!/usr/bin/env python3
"""Lint of the compiled kernels for the hazard behind round 3's lost carried scale:

    a SCALAR load (s_load_*, served by the scalar cache) of memory the same kernel also writes with VECTOR stores,
    with no ``s_waitcnt lgkmcnt(0)`` between the load and the store (the store can overtake the load: the two paths
    are not ordered by the hardware), or issued after such a store (the scalar cache is not coherent with vector stores).

The compiler picks a scalar load for any uniform address it believes unclobbered *before* the load; nothing stops a
later store of the same wave from reaching memory first.  ``k_colscale_after_truncate`` did exactly that.

    python arnoldi-py_amd/csrc/check_scalar_hazards.py            # compiles csrc/aks_kernels.hip to ISA and checks it
    python arnoldi-py_amd/csrc/check_scalar_hazards.py FILE.s     # checks an assembly listing

How: per kernel, a linear walk over the instructions (branches ignored) that tracks which SGPRs / VGPRs hold addresses
derived from which kernel-argument pointer (kernarg loads, s_mov, s_add_u32 / s_addc_u32, v_mov, v_add_co / v_addc,
v_lshl_add_u64); a scalar data load from argument P while a vector store through P is possible later -- before the
load's wait -- or earlier, is reported.

Not every address can be followed to an argument (pointers read from memory, addresses built across branches), so the
"store overtakes a pending load" half is also checked without roots: EVERY vector store issued while a scalar data load
has not been waited for must be explained -- different arguments; or the very same, unmodified SGPR base pair with
disjoint byte ranges (fields of one struct); anything else is reported as unresolved.  The summary line says how many
such stores there are at all (a handful: the compiler normally waits for its scalar loads long before it stores).

Heuristic by design: it names candidates for a human to read, and exits 1 if there are any.  Exit 0 = none.
"""
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
SREG = re.compile(r"^s(\d+)$")
SRANGE = re.compile(r"^s\[(\d+):(\d+)\]$")
VREG = re.compile(r"^v(\d+)$")
VRANGE = re.compile(r"^v\[(\d+):(\d+)\]$")


def regs(op):
    """('s' | 'v', [indices]) of a register operand, else (None, [])."""
    op = op.strip()
    for kind, single, rng in (("s", SREG, SRANGE), ("v", VREG, VRANGE)):
        m = single.match(op)
        if m:
            return kind, [int(m.group(1))]
        m = rng.match(op)
        if m:
            return kind, list(range(int(m.group(1)), int(m.group(2)) + 1))
    return None, []


LOAD_BYTES = {"dword": 4, "dwordx2": 8, "dwordx3": 12, "dwordx4": 16, "dwordx8": 32, "dwordx16": 64}
STORE_BYTES = {"byte": 1, "short": 2, "dword": 4, "dwordx2": 8, "dwordx3": 12, "dwordx4": 16}


def store_bytes(op):
    """Bytes a vector store / atomic touches, from its mnemonic (None if unknown)."""
    tail = op.split("_")
    if "atomic" in tail:
        return 8 if tail[-1] in ("x2", "f64", "u64", "i64", "b64") else 4
    return STORE_BYTES.get(tail[-1])


def check_kernel(name, lines):
    findings = []
    sroot, vroot = {}, {}           # register index -> kernarg byte offset of the pointer it derives from
    sver = {}                       # SGPR index -> how often it has been written (is a base pair still the same value?)
    vzero = {}                      # VGPR index -> True while it holds the literal 0 (v_mov_b32 vN, 0)
    karg = None                     # first SGPR of the kernarg segment pointer
    pending = []                    # scalar data loads not yet waited for: dicts (root, line no, text, base, byte range)
    stored = {}                     # root -> line no of the first vector store through it
    stats = check_kernel.stats

    def root_of(kind, idx):
        return (sroot if kind == "s" else vroot).get(idx)

    def base_key(idx):
        return tuple((i, sver.get(i, 0)) for i in idx)

    def imm(a):
        return int(a, 0) if re.match(r"^(0x[0-9a-fA-F]+|\d+)$", a) else None

    for no, raw in lines:
        text = raw.split(";")[0].strip()
        if not text or text.startswith(".") or text.endswith(":"):
            continue
        parts = text.split(None, 1)
        op = parts[0]
        args = [a.strip() for a in parts[1].split(",")] if len(parts) > 1 else []
        if op.startswith("s_load_") or op.startswith("s_buffer_load"):
            dk, dst = regs(args[0])
            bk, base = regs(args[1])
            off_imm = imm(args[2]) if len(args) > 2 else 0
            off = off_imm or 0
            if karg is None and bk == "s":
                karg = base[0]
            if bk == "s" and base and base[0] == karg:
                for i, d in enumerate(dst):                   # kernel arguments: dword i of the load
                    sroot[d] = (off + 4 * i) & ~7
            else:
                r = root_of("s", base[0]) if base else None
                key = base_key(base)                          # (before the destination is written: it may be the base)
                for d in dst:
                    sroot.pop(d, None)
                stats["scalar_data_loads"] += 1
                if r is not None:
                    stats["scalar_data_loads_followed"] += 1
                    if r in stored:
                        findings.append((name, no, f"scalar load of argument +{r:#x} AFTER a vector store through it (line {stored[r]}): "
                                                   f"the scalar cache is not coherent with vector stores | {text}"))
                size = LOAD_BYTES.get(op.split("_")[-1])
                rng = (off_imm, off_imm + size) if off_imm is not None and size else None
                pending.append({"root": r, "no": no, "text": text, "base": key, "range": rng})
            for d in dst:
                sver[d] = sver.get(d, 0) + 1
            continue
        if op == "s_waitcnt":
            if "lgkmcnt(0)" in text or re.search(r"s_waitcnt\s+(0x0+|0)\b", text):
                pending.clear()
            continue
        if op in ("s_mov_b64", "s_mov_b32"):
            dk, dst = regs(args[0])
            sk, src = regs(args[1])
            for i, d in enumerate(dst):
                r = root_of("s", src[i]) if sk == "s" and i < len(src) else None
                sver[d] = sver.get(d, 0) + 1
                if r is None:
                    sroot.pop(d, None)
                else:
                    sroot[d] = r
            continue
        if op in ("s_add_u32", "s_addc_u32", "s_add_i32", "s_sub_u32", "s_subb_u32"):
            dk, dst = regs(args[0])
            r = None
            for a in args[1:]:
                k, idx = regs(a)
                if k == "s" and idx and root_of("s", idx[0]) is not None:
                    r = root_of("s", idx[0])
                    break
            for d in dst:
                sver[d] = sver.get(d, 0) + 1
                if r is None:
                    sroot.pop(d, None)
                else:
                    sroot[d] = r
            continue
        if op.startswith("s_") and args:                       # any other scalar ALU result is not an address we follow
            dk, dst = regs(args[0])                            # (s_cmp_* only read their first operand: counted as a write,
            if dk == "s":                                      #  which errs towards "cannot be told apart")
                for d in dst:
                    sroot.pop(d, None)
                    sver[d] = sver.get(d, 0) + 1
            continue
        if op.startswith(("v_mov_b32", "v_mov_b64", "v_add_co_u32", "v_addc_co_u32", "v_add_u32", "v_lshl_add_u64", "v_add_co_ci_u32")):
            dk, dst = regs(args[0])
            r = None
            for a in args[1:]:
                k, idx = regs(a)
                if idx and root_of(k, idx[0]) is not None:
                    r = root_of(k, idx[0])
                    break
            zero = op.startswith("v_mov_b32") and len(args) == 2 and args[1] in ("0", "0x0")
            for d in dst:
                if dk == "v":
                    vzero[d] = zero
                    if r is None:
                        vroot.pop(d, None)
                    else:
                        vroot[d] = r
            continue
        if op.startswith(("global_store", "flat_store", "global_atomic", "flat_atomic", "buffer_store", "buffer_atomic")):
            r = None
            for a in args:                                     # the address operand: an SGPR pair (saddr) or a VGPR pair
                k, idx = regs(a)
                if len(idx) == 2 and root_of(k, idx[0]) is not None:
                    r = root_of(k, idx[0])
                    break
            stats["vector_stores"] += 1
            if r is not None:
                stats["vector_stores_followed"] += 1
                stored.setdefault(r, no)
            if pending:
                stats["stores_with_a_load_pending"] += 1
                saddr, soff = None, 0                          # the store's SGPR base pair and immediate offset, if it has one
                for a in args[1:]:
                    k, idx = regs(a.split()[0]) if a else (None, [])
                    if k == "s" and len(idx) == 2:
                        saddr = base_key(idx)
                m = re.search(r"offset:(-?\d+)", text)
                soff = int(m.group(1)) if m else 0
                size = store_bytes(op)
                for pl in pending:
                    if r is not None and pl["root"] is not None and r != pl["root"]:
                        continue                               # two different kernel arguments
                    if saddr is not None and saddr == pl["base"] and pl["range"] and size:
                        lo, hi = pl["range"]                   # saddr form: address = SGPR pair + VGPR offset + immediate; a
                        vk, vidx = regs(args[0])               # uniform struct field has the VGPR offset set to literal 0
                        if (soff + size <= lo or soff >= hi) and vk == "v" and vzero.get(vidx[0]) is True:
                            continue                           # same base, disjoint bytes of one struct
                    why = (f"through argument +{r:#x}" if r is not None and r == pl["root"] else "that cannot be told apart from it")
                    findings.append((name, no, f"vector store {why} while the scalar load of line {pl['no']} ({pl['text']}) "
                                               f"has not been waited for: the store can overtake it | {text}"))
            continue
        if op.startswith(("global_load", "flat_load", "buffer_load", "ds_read", "ds_load", "scratch_load")) and args:
            dk, dst = regs(args[0])
            if dk == "v":
                for d in dst:
                    vroot.pop(d, None)
                    vzero.pop(d, None)
            continue
        if op.startswith("v_") and args:
            dk, dst = regs(args[0])
            for d in dst:
                if dk == "v":
                    vroot.pop(d, None)
                    vzero.pop(d, None)
                elif dk == "s":                                # v_readfirstlane / v_cmp into an SGPR pair
                    sroot.pop(d, None)
                    sver[d] = sver.get(d, 0) + 1
    return findings


check_kernel.stats = {"scalar_data_loads": 0, "scalar_data_loads_followed": 0, "vector_stores": 0, "vector_stores_followed": 0,
                      "stores_with_a_load_pending": 0}


def check_listing(text):
    """Kernels = the text between a function label and its ``.Lfunc_end`` label (a kernel has several s_endpgm)."""
    findings, kernels = [], 0
    name, body = None, []
    for no, line in enumerate(text.split("\n"), 1):
        if name is None:
            m = re.match(r"^(_Z\w+|[A-Za-z_]\w*):\s*(;.*)?$", line)
            if m:
                name, body = m.group(1), []
            continue
        if re.match(r"^\.Lfunc_end\d+:", line):
            findings += check_kernel(name, body)
            kernels += 1
            name, body = None, []
        else:
            body.append((no, line))
    if name is not None and body:                     # (a listing cut short: the known-bad fixture)
        findings += check_kernel(name, body)
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


def main(argv):
    text = open(argv[1]).read() if len(argv) > 1 else compile_listing()
    findings, kernels = check_listing(text)
    for name, no, what in findings:
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        dem = dem.replace("(anonymous namespace)::", "").replace("void ", "")
        print(f"{dem.split('(')[0]}: line {no}: {what}")
    st = check_kernel.stats
    print(f"{kernels} kernels checked, {len(findings)} candidate hazard(s); followed to a kernel argument: "
          f"{st['scalar_data_loads_followed']} of {st['scalar_data_loads']} scalar data loads, "
          f"{st['vector_stores_followed']} of {st['vector_stores']} vector stores / atomics; "
          f"{st['stores_with_a_load_pending']} store(s) issued with a scalar data load outstanding"
          + (" (the unexplained ones are listed above)" if findings else ", each one explained (other argument, or disjoint bytes of one struct)"))
    return 1 if findings else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
