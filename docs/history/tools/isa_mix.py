#!/usr/bin/env python3
"""Instruction mix of a kernel's loops from hipcc's -save-temps assembly (gfx950).

  python tools/isa_mix.py file.s <kernel-name-substring> [--top N]

For every backward branch (loop) in the kernel: its span in instructions and the count per class (MFMA, VALU, TRANS, SALU,
DS read / write, VMEM, waits, branches), innermost loops first. Used to budget the VALU issue slots next to the MFMAs
(attention softmax, GEMM k-loop)."""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "MFMA"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
        return "TRANS"
    if op.startswith("v_"):
        return "VALU"
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "DS_RD"
    if op.startswith("ds_"):
        return "DS_WR/OTHER"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    if op.startswith("s_waitcnt"):
        return "WAIT"
    if op.startswith("s_barrier"):
        return "BARRIER"
    if op.startswith(("s_cbranch", "s_branch")):
        return "BRANCH"
    if op.startswith("s_nop"):
        return "NOP"
    if op.startswith("s_"):
        return "SALU"
    return "OTHER"


def main():
    path, name = sys.argv[1], sys.argv[2]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 6
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and name in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end + 1]
    insts, labels = [], {}
    for l in body:
        s = l.split(";")[0].strip()
        if not s or s.startswith("."):
            m = re.match(r"^(\.LBB\w+):", s)
            if m:
                labels[m.group(1)] = len(insts)
            continue
        m = re.match(r"^(\.LBB\w+):", s)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        if re.match(r"^_Z\w*:", s):
            continue
        insts.append(s)
    print(f"{name}: {len(insts)} instructions")
    tot = collections.Counter(classify(i.split()[0]) for i in insts)
    print("  whole kernel:", dict(tot))
    loops = []
    for i, ins in enumerate(insts):
        op = ins.split()[0]
        if op.startswith(("s_cbranch", "s_branch")):
            tgt = ins.split()[-1]
            if tgt in labels and labels[tgt] <= i:
                loops.append((labels[tgt], i))
    loops.sort(key=lambda ab: ab[1] - ab[0])
    for a, b in loops[:top]:
        c = collections.Counter(classify(i.split()[0]) for i in insts[a:b + 1])
        ops = collections.Counter(i.split()[0] for i in insts[a:b + 1] if classify(i.split()[0]) in ("VALU", "TRANS"))
        print(f"  loop [{a}, {b}] {b - a + 1} instr: {dict(c)}")
        print("     top VALU:", ops.most_common(14))


if __name__ == "__main__":
    main()
