#!/bin/bash
set -x
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python scripts/bench_conv.py 10000000 0 > gpurun_out/bench_conv_d.txt 2>&1; grep -v "Cin= 192\|Cin= 768" gpurun_out/bench_conv_d.txt
run() { SVC_CONV_CFG=$1 SVC_MRF_POSTACT=$2 timeout 300 python bench.py --mode infer --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench_d.json 2> gpurun_out/bench_d.err; python -c "
import json; d=json.load(open('gpurun_out/bench_d.json')); print('cfg $1 postact $2', round(d['ms_per_step'],3), d['roofline']['families']['conv1d_mfma'])"; }
run 10000000 1; run 0 1; run 10000000 1; run 0 1
timeout 900 python -m pytest tests/test_conv1d_gpu.py tests/test_ops_gpu.py tests/test_infer_gpu.py -m gpu -q --timeout=600 -x -rf > gpurun_out/pytest_gpu_d.log 2>&1; tail -4 gpurun_out/pytest_gpu_d.log
