#!/bin/bash
set -x
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 300 python scripts/diag_discp_flip.py > gpurun_out/t_discp_flip_diag.txt 2>&1; echo "diag rc=$?"; tail -12 gpurun_out/t_discp_flip_diag.txt | cut -c1-400
for i in 1 2; do timeout 600 python -m pytest tests/test_train_ops_gpu.py -m gpu -q -k "discriminator_p or lrelu_tail" 2>&1 | tail -3 | cut -c1-300; done
