#!/bin/bash
set -x
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() { SVC_MRF_STREAMS=$1 timeout 300 python bench.py --mode infer --steps 30 --warmup 5 --no-cpu-baseline $2 > gpurun_out/bench_i.json 2> gpurun_out/bench_i.err; python -c "
import json; d=json.load(open('gpurun_out/bench_i.json')); print('streams $1 $2', round(d['ms_per_step'],3), {k:v for k,v in d['roofline']['families'].items() if k in ('conv1d_mfma','resblock_pair')})"; tail -2 gpurun_out/bench_i.err; }
run 0; run 1; run 0; run 1; run 1 --no-graph; run 0 --no-graph
timeout 900 python -m pytest tests/test_infer_gpu.py tests/test_nsf_hifigan.py -m gpu -q --timeout=600 -x -rf > gpurun_out/pytest_gpu_i.log 2>&1; tail -4 gpurun_out/pytest_gpu_i.log
