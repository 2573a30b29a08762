# on the GPU box: bash tools/run_profiles.sh <tag> <commit>
export PROFILE_COMMIT=$2
bash tools/collect_profiles.sh $1 > gpurun_out/$1.collect.log 2>&1
tail -5 gpurun_out/$1.collect.log
head -c 1500 gpurun_out/$1/bench.json; echo
grep -v "^\[bench\] \(geonames\|words\|skewed\)" gpurun_out/$1/bench.log | tail -5
