# the whole GPU suite on the box, log under gpurun_out/<tag>/ (usage: bash tools/run_gpu_suite.sh <tag> [pytest args])
tag=${1:-suite}; shift
mkdir -p gpurun_out/$tag
timeout 1500 python -m pytest tests -m gpu -q "$@" > gpurun_out/$tag/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$tag/pytest.log
tail -15 gpurun_out/$tag/pytest.log
