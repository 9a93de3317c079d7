#!/bin/bash
set -x
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/prof_train
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train -o run -- python bench.py --mode train --steps 3 --warmup 1 --no-roofline --no-cpu-baseline > gpurun_out/bench_train_prof.json 2> gpurun_out/bench_train_prof.err; echo "rocprof train rc=$?"
DB=$(find gpurun_out/prof_train -name '*.db' | head -1); python scripts/prof_summary.py $DB > gpurun_out/kernel_stats_train.txt 2>&1; head -40 gpurun_out/kernel_stats_train.txt
find gpurun_out -name '*.db' -size +30M -delete
