#!/bin/bash
# Round-2 call q: DiscriminatorP gradient diagnostic (padded vs unpadded vs fp64), RCCL dry run with forced collectives, training kernel trace.
set -x
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python scripts/diag_discp_padded.py > gpurun_out/q_discp_diag.txt 2>&1; echo "diag rc=$?"; cat gpurun_out/q_discp_diag.txt | tail -30
SVC_DP_FORCE=1 timeout 600 python bench.py --mode train --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/q_train_rccl_world1.json 2> gpurun_out/q_train_rccl_world1.err; echo "rccl dry run rc=$?"
python -c "
import json;d=json.loads(open('gpurun_out/q_train_rccl_world1.json').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['losses'],d['allreduce'])"
tail -3 gpurun_out/q_train_rccl_world1.err
SVC_DP_FORCE=1 SVC_DP_CAPTURE_COLLECTIVES=1 timeout 600 python bench.py --mode train --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/q_train_rccl_captured.json 2> gpurun_out/q_train_rccl_captured.err; echo "rccl captured rc=$?"
python -c "
import json;d=json.loads(open('gpurun_out/q_train_rccl_captured.json').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['losses'],d['allreduce'])"
tail -3 gpurun_out/q_train_rccl_captured.err
rm -rf gpurun_out/prof_train_q
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train_q -o run -- python bench.py --mode train --steps 3 --warmup 1 --no-roofline --no-cpu-baseline > gpurun_out/q_train_prof.json 2> gpurun_out/q_train_prof.err; echo "rocprof train rc=$?"
DB=$(find gpurun_out/prof_train_q -name '*.db' | head -1); python scripts/prof_summary.py $DB > gpurun_out/q_kernel_stats_train.txt 2>&1; head -40 gpurun_out/q_kernel_stats_train.txt
find gpurun_out -name '*.db' -size +30M -delete
