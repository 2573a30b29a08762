# on the GPU box: phase clocks of the needle-major sweep, plain against slices left out (usage: bash tools/run_phase.sh <tag> [needles])
tag=${1:-phase}; nq=${2:-100000}
mkdir -p gpurun_out/$tag
WS_AUTOTUNE=0 WSWEEP=0 NM_CMIN=0 python tools/phase_profile.py 1.0 $nq > gpurun_out/$tag/plain.txt 2>&1
WS_AUTOTUNE=0 WSWEEP=0 NM_MIN_WINDOWS=0 python tools/phase_profile.py 1.0 $nq > gpurun_out/$tag/leave.txt 2>&1
tail -22 gpurun_out/$tag/plain.txt; tail -22 gpurun_out/$tag/leave.txt
