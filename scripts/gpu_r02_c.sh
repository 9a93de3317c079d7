#!/bin/bash
set -x
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python scripts/bench_small_conv.py > gpurun_out/bench_small_conv.txt 2>&1; cat gpurun_out/bench_small_conv.txt
for cfg in 1000000 0 1000000 0; do
SVC_CONV_CFG=$cfg timeout 300 python bench.py --mode infer --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench_c_$cfg.json 2> gpurun_out/bench_c.err; python -c "
import json; d=json.load(open('gpurun_out/bench_c_$cfg.json')); print('cfg $cfg', d['ms_per_step'], d['roofline']['families']['conv1d_mfma'], d['roofline']['families']['attention'])"
done
timeout 600 python -m pytest tests/test_conv1d_gpu.py tests/test_ops_gpu.py -m gpu -q --timeout=600 -x -rf > gpurun_out/pytest_gpu_c.log 2>&1; tail -4 gpurun_out/pytest_gpu_c.log
