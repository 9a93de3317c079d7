#!/bin/bash
set -x
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() { SVC_D_STREAMS=$1 timeout 600 python bench.py --mode train --steps 8 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/bench_train_l.json 2> gpurun_out/bench_train_l.err; python -c "
import json; d=json.load(open('gpurun_out/bench_train_l.json')); print('D streams $1: train', round(d['ms_per_step'],2), d['value'], d['losses'])"; tail -2 gpurun_out/bench_train_l.err; }
run 0; run 1; run 0; run 1
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_train_loop_gpu.py tests/test_data_parallel_gpu.py -m gpu -q --timeout=600 -rf > gpurun_out/pytest_gpu_l.log 2>&1; tail -5 gpurun_out/pytest_gpu_l.log
