#!/bin/bash
set -x
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_conv1d_gpu.py -m gpu -q --timeout=300 -x -rf -k resblock_pair > gpurun_out/pytest_gpu_f.log 2>&1; tail -12 gpurun_out/pytest_gpu_f.log
timeout 300 python scripts/bench_pair.py > gpurun_out/bench_pair.txt 2>&1; cat gpurun_out/bench_pair.txt
run() { SVC_MRF_FUSE_PAIR=$1 timeout 300 python bench.py --mode infer --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench_f.json 2> gpurun_out/bench_f.err; python -c "
import json; d=json.load(open('gpurun_out/bench_f.json')); print('fuse $1', round(d['ms_per_step'],3), {k:v for k,v in d['roofline']['families'].items() if k in ('conv1d_mfma','resblock_pair')})"; tail -2 gpurun_out/bench_f.err; }
run 0; run 1; run 0; run 1
timeout 900 python -m pytest tests/test_infer_gpu.py tests/test_nsf_hifigan.py -m gpu -q --timeout=600 -x -rf > gpurun_out/pytest_gpu_f2.log 2>&1; tail -4 gpurun_out/pytest_gpu_f2.log
