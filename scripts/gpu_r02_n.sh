#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python scripts/diag_direct_conv.py > gpurun_out/diag_direct.txt 2>&1; grep -E "BAD|bad:" gpurun_out/diag_direct.txt | head -40; tail -3 gpurun_out/diag_direct.txt
