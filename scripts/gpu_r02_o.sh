#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -f gpurun_out/determinism7.txt
for i in 1 2 3 4; do SVC_BENCH_TRACE=1 SVC_D_STREAMS=0 timeout 300 python bench.py --mode train --steps 8 --warmup 2 --no-cpu-baseline --no-roofline 2>&1 | grep -E "TRACE|ms_per_step" | cut -c1-250 | tee -a gpurun_out/determinism7.txt; done
for s in at2 at2; do N=10 SYNC=$s GRAPH=1 SVC_D_STREAMS=0 timeout 300 python scripts/diag_train_determinism.py 2>/dev/null | cut -c1-250 | tee -a gpurun_out/determinism7.txt; done
