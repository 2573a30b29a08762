#!/bin/bash
# usage: bash tools/spills_report.sh > profiles/rNN_spills.txt  -- registers, scratch and spills of the hot kernels from the compiler
# (make resources) and, per kernel, the basic blocks that touch scratch (tools/asm_spills.py over make asm's output).  No GPU needed.
cd "$(dirname "$0")/.."
echo "# register spills of the hot kernels at commit $(git rev-parse --short HEAD) (+ working tree)"
echo "# (make -C blurrily_amd/csrc resources: the compiler's own figures; tools/asm_spills.py: where the scratch accesses sit --"
echo "#  basic blocks with a barrier, >= 8 LDS atomics or a scratch access; 'scratch [...]' lists the accesses of the block)"
echo
make -s -C blurrily_amd/csrc resources 2>&1 | python3 -c "
import sys,re
cur=None;d={};order=[]
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur=m.group(1); d[cur]={}; order.append(cur)
    for k,pat in (('VGPRs',r' VGPRs: (\d+)'),('scratch',r'ScratchSize \[bytes/lane\]: (\d+)'),('occ',r'Occupancy \[waves/SIMD\]: (\d+)'),('sgpr_sp',r'SGPRs Spill: (\d+)'),('vgpr_sp',r'VGPRs Spill: (\d+)')):
        m=re.search(pat,l)
        if m and cur: d[cur][k]=m.group(1)
for k in order:
    if any(x in k for x in ('wsweep_kernel','find_small_kernel','find_one_kernel','find_kernelIh')) and 'counted' not in k:
        v=d[k]; print(f\"{k}  VGPRs {v.get('VGPRs')}  scratch {v.get('scratch')} B/lane  occupancy {v.get('occ')}  SGPR spills {v.get('sgpr_sp')}  VGPR spills {v.get('vgpr_sp')}\")
"
make -s -C blurrily_amd/csrc asm 2>&1 | grep -i error
for k in find_kernelIhLi1024ELb0ELb1E find_kernelIhLi1024ELb1ELb1E find_small_kernel find_one_kernel wsweep_kernel; do
  echo; echo "## $k: blocks with scratch accesses"
  python3 tools/asm_spills.py $k | grep -E "scratch \['|totals"
done
