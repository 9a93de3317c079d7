#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() { env $1 timeout 600 python bench.py --mode train --steps 8 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/bench_train_m.json 2> gpurun_out/bench_train_m.err; python -c "
import json; d=json.load(open('gpurun_out/bench_train_m.json')); print('$1', round(d['ms_per_step'],2), {k:v for k,v in d['losses'].items() if k in ('loss_disc','loss_fm','loss_mel','loss_kl')})" | tee -a gpurun_out/determinism2.txt; tail -1 gpurun_out/bench_train_m.err | grep -v amdgpu; }
rm -f gpurun_out/determinism2.txt
B="SVC_D_STREAMS=0 SVC_TIME_ALIGN=0"
run "$B SVC_CONV_CFG=1000000"; run "$B SVC_CONV_CFG=1000000"; run "$B SVC_CONV_CFG=1000000"
run "$B"; run "$B"; run "$B SVC_CONV_CFG=10000000"; run "$B SVC_CONV_CFG=10000000"
