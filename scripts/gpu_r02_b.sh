#!/bin/bash
set -x
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python scripts/diag_attention_bwd.py > gpurun_out/diag_attention.txt 2>&1; cat gpurun_out/diag_attention.txt | tail -20
timeout 900 python -m pytest tests/test_boundary_gpu.py tests/test_train_gpu.py -m gpu -q --timeout=600 -rf > gpurun_out/pytest_gpu_b.log 2>&1; tail -15 gpurun_out/pytest_gpu_b.log
