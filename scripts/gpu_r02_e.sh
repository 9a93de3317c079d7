#!/bin/bash
set -x
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python scripts/bench_conv.py 10000000 10004000 10002000 10001000 10008000 > gpurun_out/bench_conv_e.txt 2>&1; grep -v "Cin= 192\|Cin= 768" gpurun_out/bench_conv_e.txt
