#!/usr/bin/env python3
"""Scratch loads / stores of a kernel in `make asm` output with the loop depth of their block.  python tools/asm_scratch_depth.py [kernel substring]"""
import re, sys
part = sys.argv[1] if len(sys.argv) > 1 else "find_kernelIhLi1024ELb0ELb1E"
txt = open("blurrily_amd/csrc/find_kernels.gfx950.s").read()
m = re.search(r"^(\S*%s\S*):\s" % re.escape(part), txt, flags=re.M)
start = m.start(); end = txt.index(".Lfunc_end", start)
blk, depth = "entry", ""
for l in txt[start:end].split("\n"):
    t = l.strip()
    lab = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?", t)
    if lab:
        blk, depth = lab.group(1), (lab.group(2) or ""); continue
    if t.startswith(";") and "Loop" in t:
        depth += " " + t; continue
    if t.startswith("scratch_"):
        d = re.findall(r"Depth=(\d)", depth)
        print(blk, "depth", d[-1] if d else "-", t[:80])
