#!/bin/bash
# One gpurun call: GPU parity tests, bench line, rocprofv3 kernel-trace stats.  Outputs under gpurun_out/.
set -x
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout=240 -rf > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -30 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 30 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"
cat gpurun_out/bench.json; tail -5 gpurun_out/bench.err
rm -rf gpurun_out/prof_stats
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats -o run -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-graph > gpurun_out/bench_prof.json 2> gpurun_out/bench_prof.err; echo "rocprof rc=$?"
DB=$(find gpurun_out/prof_stats -name '*.db' | head -1); python scripts/prof_summary.py $DB > gpurun_out/kernel_stats.txt 2>&1; head -40 gpurun_out/kernel_stats.txt
find gpurun_out/prof_stats -name '*.db' -size +40M -delete
