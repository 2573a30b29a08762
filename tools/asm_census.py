#!/usr/bin/env python3
"""Basic-block instruction census of one kernel in `make asm` output (blurrily_amd/csrc/find_kernels.gfx950.s).

    python tools/asm_census.py [kernel-name-substring] [--blocks]

Per block: instruction counts by class -- V VALU (v_readlane/v_writelane/v_readfirstlane included: they take a
VALU issue slot), S SALU, L LDS, G global/scratch memory, B barrier, W s_waitcnt, J branch -- and what marks it
(LDS atomics, barriers, 128-bit LDS reads/writes, DPP).  Used for profiles/r03_step_budget.md."""
import re
import sys


def classify(op):
    if op.startswith("v_"):
        return "V"
    if op.startswith("ds_"):
        return "L"
    if op.startswith("s_barrier"):
        return "B"
    if op.startswith("s_waitcnt"):
        return "W"
    if op.startswith(("s_cbranch", "s_branch")):
        return "J"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_", "s_load", "s_buffer")):
        return "G"
    if op.startswith("s_"):
        return "S"
    return "?"


def blocks_of(text, name_part):
    m = re.search(r"^(\S*%s\S*):\s" % re.escape(name_part), text, flags=re.M)
    a = m.start()
    b = text.index(".Lfunc_end", a)
    cur = ["<entry>", []]
    out = []
    for ln in text[a:b].split("\n"):
        t = ln.strip()
        lab = re.match(r"^(\.LBB\d+_\d+):", t)
        if lab:
            out.append(cur)
            cur = [lab.group(1), []]
            continue
        if not t or t.startswith((";", ".")) or re.match(r"^[A-Za-z_$][\w.$]*:", t):
            continue
        cur[1].append(re.sub(r"\s*;.*", "", t))
    out.append(cur)
    return m.group(1), out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    part = args[0] if args else "find_kernelIhLi1024ELb0ELb1E"
    path = args[1] if len(args) > 1 else "blurrily_amd/csrc/find_kernels.gfx950.s"
    name, blocks = blocks_of(open(path).read(), part)
    print(name)
    total = {}
    for i, (lab, ins) in enumerate(blocks):
        c = {}
        for x in ins:
            k = classify(x.split()[0])
            c[k] = c.get(k, 0) + 1
            total[k] = total.get(k, 0) + 1
        marks = []
        n_add = sum("ds_add_u32" in x for x in ins)
        if n_add:
            marks.append(f"ADDx{n_add}")
        for tag, pat in (("BARRIER", "s_barrier"), ("RD128", "ds_read_b128"), ("WR128", "ds_write_b128"),
                         ("DPP", "dpp"), ("LANE", "v_readlane|v_writelane"), ("GLOAD", "global_load")):
            n = sum(bool(re.search(pat, x)) for x in ins)
            if n:
                marks.append(f"{tag}x{n}" if n > 1 else tag)
        tgt = [x.split()[-1] for x in ins if x.startswith(("s_cbranch", "s_branch"))]
        if "--blocks" in sys.argv:
            print(f"{i:4d} {lab:12s} n={len(ins):4d} " + " ".join(f"{k}{v}" for k, v in sorted(c.items())) +
                  "  " + " ".join(marks) + ("  -> " + ",".join(tgt) if tgt else ""))
    print("total", total)


if __name__ == "__main__":
    main()
