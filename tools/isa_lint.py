#!/usr/bin/env python3
"""Static check of the hand-scheduled kernels' assembly.

Loads written as inline asm (ds_read_*, buffer_load_dwordx4 into VGPRs) are invisible to the compiler's
wait bookkeeping: it treats their destination registers as valid from the instruction on, although the
data only arrives after the hand-placed s_waitcnt.  The sources therefore pass every such value through a
fence right behind its wait, which prints `; srhip-fence <registers>` into the assembly.  This script
compiles the given .hip files to gfx950 assembly and verifies, on every control-flow path, that between an
asm load and the fence of its destination registers NO instruction reads or writes those registers (a
register move there would copy stale data; a re-use would be overwritten when the load lands).  It also checks
that no VALU instruction overwrites the data registers of a 96/128-bit store directly behind it, and that no
compiler-generated instruction touches the destination of an inline-asm MFMA within the MFMA's latency.

usage: isa_lint.py [--keep DIR] file.hip [file.hip ...]      exit status 1 when a violation is found
"""
import os
import re
import subprocess
import sys
import tempfile

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-std=c++17", "-S", "--cuda-device-only"]   # the product build's code generation flags (csrc/Makefile)
LOAD_RE = re.compile(r"^(ds_read_b128|ds_read_b64_tr_b16|buffer_load_dwordx4|buffer_load_dword)\s+(v\[\d+:\d+\]|v\d+)\s*,(.*)$")
REG_RE = re.compile(r"\bv(?:(\d+)\b|\[(\d+):(\d+)\])")
FENCE = "; srhip-fence"


def regs_of(text):
    out = set()
    for m in REG_RE.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


AREG_RE = re.compile(r"\ba(?:(\d+)\b|\[(\d+):(\d+)\])")
MFMA_WAIT_STATES = 18   # a 32x32 MFMA's result is readable by other instruction types this many wait states after issue


def aregs_of(text):
    out = set()
    for m in AREG_RE.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def split_functions(lines):
    """Yield (name, [instruction lines]) for every kernel."""
    name, body = None, []
    for ln in lines:
        m = re.match(r"^(_Z\w+):", ln)
        if m and name is None:
            name, body = m.group(1), []
            continue
        if name is not None:
            body.append(ln.rstrip("\n"))
            if ln.strip().startswith("s_endpgm"):
                yield name, body
                name = None


def lint_function(name, body):
    # instruction list with asm-block flags; labels -> index of the next instruction
    insts, labels, in_asm = [], {}, False
    for ln in body:
        t = ln.strip()
        if not t:
            continue
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = re.match(r"^(\.LBB\w+):", t)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        if t.startswith(FENCE):
            insts.append(("fence", t[len(FENCE):], in_asm))
            continue
        if t.startswith(";") or t.startswith("."):
            continue
        code = t.split(";")[0].strip()
        if code:
            insts.append(("inst", code, in_asm))

    def successors(i):
        kind, code, _ = insts[i]
        if kind == "inst":
            op = code.split()[0]
            if op == "s_endpgm":
                return []
            if op == "s_branch":
                return [labels[code.split()[1]]]
            if op.startswith("s_cbranch"):
                return [labels[code.split()[1]], i + 1]
        return [i + 1] if i + 1 < len(insts) else []

    problems = []
    n_loads = 0
    for i, (kind, code, in_asm) in enumerate(insts):
        if kind != "inst" or not in_asm:
            continue
        m = LOAD_RE.match(code)
        if not m or code.rstrip().endswith(" lds"):
            continue
        n_loads += 1
        dst = frozenset(regs_of(m.group(2)))
        seen = set()
        stack = [(s, dst) for s in successors(i)]
        while stack:
            j, pend = stack.pop()
            if (j, pend) in seen:
                continue
            seen.add((j, pend))
            k2, c2, asm2 = insts[j]
            if k2 == "fence":
                pend = pend - regs_of(c2)
                if not pend:
                    continue
            else:
                touched = regs_of(c2) & pend
                if touched:
                    m2 = LOAD_RE.match(c2) if asm2 else None
                    if m2 and not c2.rstrip().endswith(" lds") and regs_of(m2.group(2)) & pend and not (regs_of(m2.group(3)) & pend):
                        continue  # re-issued into the same registers: the earlier value was dead on this path
                    problems.append((i, code, j, c2, sorted(touched)))
                    continue
            for s in successors(j):
                stack.append((s, pend))
    # Second check: a store of more than 64 bits reads its data registers after it has issued; the instruction right
    # behind it must not overwrite them (the compiler's hazard table has exempted one store form before).
    store_re = re.compile(r"^(buffer|global|flat|scratch)_store_dwordx[34]\s+(.*)$")
    for i, (kind, code, _) in enumerate(insts):
        if kind != "inst":
            continue
        m = store_re.match(code)
        if not m:
            continue
        ops = [o.strip() for o in m.group(2).split(",")]
        data = regs_of(ops[0] if m.group(1) == "buffer" else ops[1] if len(ops) > 1 else "")
        waited = 0
        for j in range(i + 1, min(i + 4, len(insts))):
            k2, c2, _ = insts[j]
            if k2 != "inst":
                continue
            op = c2.split()[0]
            if op == "s_nop":
                waited += int(c2.split()[1]) + 1
                continue
            if op.startswith(("s_", "buffer_", "global_", "ds_")) or waited >= 2:
                break
            dst = regs_of(c2.split(",")[0])
            if op.startswith("v_") and dst & data:
                problems.append((i, code, j, c2, sorted(dst & data)))
            break
    # Third check: the MFMAs are inline asm, so the compiler's hazard recognizer does not know that their result registers
    # are written many cycles after issue.  A compiler-generated instruction (a v_accvgpr_read / _write it adds when it
    # splits an accumulator's live range under register pressure) that touches an asm MFMA's destination within the
    # MFMA's latency reads a stale value or is overwritten.  Dependent asm MFMAs (the accumulation chain) interlock in
    # hardware and are not flagged; the sources' own reads sit behind enough independent work or an s_nop.
    mfma_re = re.compile(r"^v_mfma_\w+\s+(a\[\d+:\d+\])")
    for i, (kind, code, in_asm) in enumerate(insts):
        if kind != "inst" or not in_asm:
            continue
        m = mfma_re.match(code)
        if not m:
            continue
        dst = aregs_of(m.group(1))
        waited = 0
        for j in range(i + 1, len(insts)):
            k2, c2, asm2 = insts[j]
            if k2 != "inst":
                continue
            op = c2.split()[0]
            if op in ("s_branch", "s_endpgm", "s_barrier") or op.startswith("s_cbranch") or waited >= MFMA_WAIT_STATES:
                break
            if op == "s_nop":
                waited += int(c2.split()[1]) + 1
                continue
            if op.startswith("v_mfma"):
                waited += 8
                continue
            if not asm2 and aregs_of(c2) & dst:
                problems.append((i, code, j, c2, sorted(aregs_of(c2) & dst)))
                break
            waited += 1
    return n_loads, problems


def lint_file(path, keep=None):
    out_dir = keep or tempfile.mkdtemp(prefix="isa_lint_")
    os.makedirs(out_dir, exist_ok=True)
    asm = os.path.join(out_dir, os.path.basename(path) + ".s")
    inc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include")
    subprocess.run([HIPCC] + FLAGS + ["-I", inc, path, "-o", asm], check=True, stderr=subprocess.DEVNULL)
    total, bad = 0, 0
    with open(asm) as f:
        lines = f.readlines()
    for name, body in split_functions(lines):
        n_loads, problems = lint_function(name, body)
        total += n_loads
        reported = set()
        for (i, code, j, c2, regs) in problems:
            key = (code, c2)
            if key in reported:
                continue
            reported.add(key)
            bad += 1
            print(f"{os.path.basename(path)}: {name[:90]}\n    load  #{i}: {code}\n    touch #{j}: {c2}   ({regs}) too early")
    print(f"{os.path.basename(path)}: {total} asm loads checked, {bad} violation(s)")
    return bad


def main(argv):
    keep = None
    if argv and argv[0] == "--keep":
        keep, argv = argv[1], argv[2:]
    bad = 0
    for p in argv:
        bad += lint_file(p, keep)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
