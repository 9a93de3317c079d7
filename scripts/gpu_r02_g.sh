#!/bin/bash
set -x
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python bench.py --mode train --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/bench_train_g.json 2> gpurun_out/bench_train_g.err; python -c "
import json; d=json.load(open('gpurun_out/bench_train_g.json')); print('train', round(d['ms_per_step'],2), d['value']); print({k:v for k,v in list(d['families'].items())[:6]}, d['families']['_kernel_ms_total'])"; tail -3 gpurun_out/bench_train_g.err
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_train_loop_gpu.py tests/test_train_ops_gpu.py tests/test_data_parallel_gpu.py tests/test_diffusion.py tests/test_data_utils.py -m gpu -q --timeout=600 -rf > gpurun_out/pytest_gpu_g.log 2>&1; tail -8 gpurun_out/pytest_gpu_g.log
SVC_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --mode train --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/bench_train_dp2.json 2> gpurun_out/bench_train_dp2.err; cat gpurun_out/bench_train_dp2.json; tail -5 gpurun_out/bench_train_dp2.err
rm -rf gpurun_out/prof_train
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train -o run -- python bench.py --mode train --steps 3 --warmup 1 --no-roofline --no-cpu-baseline > gpurun_out/bench_train_prof.json 2> gpurun_out/bench_train_prof.err; echo "rocprof train rc=$?"
DB=$(find gpurun_out/prof_train -name '*.db' | head -1); python scripts/prof_summary.py $DB > gpurun_out/kernel_stats_train.txt 2>&1; head -30 gpurun_out/kernel_stats_train.txt
find gpurun_out -name '*.db' -size +30M -delete
