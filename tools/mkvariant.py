#!/usr/bin/env python3
"""usage: tools/mkvariant.py <name> <edits.py> -- a copy of blurrily_amd/csrc under csrc_x<name> with the edits applied, built into
blurrily_amd/libx_<name>.so (experiments; git-ignored).  edits.py defines EDITS = [(file, old, new), ...] (exact text, must
occur once) and optionally EXTRA = "-D..."."""
import os, shutil, subprocess, sys, glob
name, edits = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = os.path.join(root, "blurrily_amd", "csrc_x" + name)
shutil.rmtree(d, ignore_errors=True); os.makedirs(d)
for f in glob.glob(os.path.join(root, "blurrily_amd", "csrc", "*")):
    if os.path.isfile(f) and not f.endswith(".s"): shutil.copy(f, d)
shutil.copytree(os.path.join(root, "blurrily_amd", "csrc", "kernels"), os.path.join(d, "kernels"))   # (EDITS name them as kernels/x.inc)
ns = {}; exec(open(edits).read(), ns)
for f, old, new in ns["EDITS"]:
    p = os.path.join(d, f); s = open(p).read()
    assert s.count(old) == 1, (f, old[:60], s.count(old))
    open(p, "w").write(s.replace(old, new))
r = subprocess.run(["make", "-s", "-j4", "-C", d, f"OUT=../libx_{name}.so", "EXTRA=" + ns.get("EXTRA", "")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
err = [l for l in r.stdout.split("\n") if "error" in l]
print("\n".join(err[:10]) if err else f"built libx_{name}.so")
