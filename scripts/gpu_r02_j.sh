#!/bin/bash
set -x
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python scripts/bench_attention.py > gpurun_out/bench_attention.txt 2>&1; cat gpurun_out/bench_attention.txt
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_hubert.py -m gpu -q --timeout=600 -x -rf > gpurun_out/pytest_gpu_j.log 2>&1; tail -4 gpurun_out/pytest_gpu_j.log
run() { SVC_CONV_CFG=$1 timeout 300 python bench.py --mode infer --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench_j.json 2> gpurun_out/bench_j.err; python -c "
import json; d=json.load(open('gpurun_out/bench_j.json')); print('cfg $1', round(d['ms_per_step'],3), {k:v for k,v in d['roofline']['families'].items() if k in ('conv1d_mfma','resblock_pair','attention')})"; tail -2 gpurun_out/bench_j.err; }
run 0; run 100000000; run 0; run 100000000
