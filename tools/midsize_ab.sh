# usage (GPU box): bash tools/midsize_ab.sh <tag>  -- mid-size batches at Geonames scale: the plain sweep against the one that leaves slices out (tools/leave_ab.py)
tag=$1; mkdir -p gpurun_out/$tag
for n in 1024 4096 12000; do for sw in 1 3; do AB_N=$n AB_SWEEP=$sw timeout 300 python tools/leave_ab.py 2>&1 | grep "kernel ms" | sed "s/^/n=$n /"; done; done | tee gpurun_out/$tag/ab.txt
