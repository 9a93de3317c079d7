#!/bin/bash
set -x
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() { SVC_MRF_STREAMS=$1 timeout 300 python bench.py --mode infer --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench_k.json 2> gpurun_out/bench_k.err; python -c "
import json; d=json.load(open('gpurun_out/bench_k.json')); print('streams $1', round(d['ms_per_step'],3), d['roofline']['frac'])"; tail -2 gpurun_out/bench_k.err; }
run 1; run 0; run 1
timeout 1200 python -m pytest tests/test_infer_gpu.py tests/test_nsf_hifigan.py tests/test_svc_gpu.py tests/test_boundary_gpu.py tests/test_ops_gpu.py -m gpu -q --timeout=600 -x -rf > gpurun_out/pytest_gpu_k.log 2>&1; tail -4 gpurun_out/pytest_gpu_k.log
