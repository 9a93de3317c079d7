#!/bin/bash
# GPU: training-path tests + train bench + rocprof kernel trace of the training step.
set -x
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout=300 -rf -x -k "train or period or decimate or strided" > gpurun_out/pytest_gpu_train.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_train.log
tail -25 gpurun_out/pytest_gpu_train.log
timeout 600 python bench.py --mode train --steps 6 --warmup 2 > gpurun_out/bench_train.json 2> gpurun_out/bench_train.err; echo "bench rc=$?"
cat gpurun_out/bench_train.json; tail -5 gpurun_out/bench_train.err
rm -rf gpurun_out/prof_train
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train -o run -- python bench.py --mode train --steps 3 --warmup 1 --no-roofline > gpurun_out/bench_train_prof.json 2> gpurun_out/bench_train_prof.err; echo "rocprof train rc=$?"
DB=$(find gpurun_out/prof_train -name '*.db' | head -1); python scripts/prof_summary.py $DB > gpurun_out/kernel_stats_train.txt 2>&1; head -60 gpurun_out/kernel_stats_train.txt
find gpurun_out -name '*.db' -size +30M -delete
