#!/usr/bin/env python3
"""Where a kernel of `make asm` output touches scratch (VGPR spills) and moves SGPR spills: per basic block, beside the
marks that place the block (barriers, LDS atomics, global loads).  python tools/asm_spills.py [kernel substring] [asm]"""
import re
import sys

part = sys.argv[1] if len(sys.argv) > 1 else "find_kernelIhLi1024ELb0ELb1E"
path = sys.argv[2] if len(sys.argv) > 2 else "blurrily_amd/csrc/find_kernels.gfx950.s"
txt = open(path).read()
m = re.search(r"^(\S*%s\S*):\s" % re.escape(part), txt, flags=re.M)
start = m.start()
end = txt.index(".Lfunc_end", start)
blk, order, info = "entry", ["entry"], {"entry": dict(n=0, scr=[], bar=0, dsadd=0, gld=0, rl=0, wl=0, call=0)}
for l in txt[start:end].split("\n"):
    t = l.strip()
    lab = re.match(r"^(\.LBB\d+_\d+):", t)
    if lab:
        blk = lab.group(1); order.append(blk); info[blk] = dict(n=0, scr=[], bar=0, dsadd=0, gld=0, rl=0, wl=0, call=0)
        continue
    if not t or t.startswith((";", ".")):
        continue
    d = info[blk]; d["n"] += 1
    op = t.split()[0]
    if op.startswith("scratch_"): d["scr"].append(op.replace("scratch_", "") + " " + t.split("offset:")[-1].split()[0] if "offset:" in t else op.replace("scratch_", ""))
    d["bar"] += op == "s_barrier"; d["dsadd"] += op.startswith("ds_add"); d["gld"] += op.startswith("global_load")
    d["rl"] += op == "v_readlane_b32"; d["wl"] += op == "v_writelane_b32"; d["call"] += op.startswith("s_swappc")
tot = dict(scr=0, rl=0, wl=0)
for b in order:
    d = info[b]
    tot["scr"] += len(d["scr"]); tot["rl"] += d["rl"]; tot["wl"] += d["wl"]
    if d["scr"] or d["bar"] or d["dsadd"] >= 8 or d["call"]:
        print(f"{b:12s} n {d['n']:3d} bar {d['bar']} dsadd {d['dsadd']:2d} gload {d['gld']} call {d['call']} readlane {d['rl']:2d} writelane {d['wl']:2d}  scratch {d['scr']}")
print(m.group(1), "totals:", tot)
