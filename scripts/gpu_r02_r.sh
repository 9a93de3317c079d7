#!/bin/bash
# Round-2 call r: full GPU suite on the weight-plan build, training A/B (SVC_WEIGHT_PLAN), RCCL dry runs under the thread_local
# capture mode, aten-op census of one training iteration.
set -x
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 -rf --maxfail=30 > gpurun_out/r_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r_pytest_gpu.log
tail -15 gpurun_out/r_pytest_gpu.log | cut -c1-300
for wp in 0 1 0 1; do
  SVC_WEIGHT_PLAN=$wp timeout 600 python bench.py --mode train --steps 8 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/r_train_wp$wp.json 2> gpurun_out/r_train_wp$wp.err; echo "wp=$wp rc=$?"
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r_train_wp$wp.json").read().splitlines() if l.startswith("{")][-1])
print("WEIGHT_PLAN=$wp ms_per_step", d["ms_per_step"], {k: round(v,4) for k,v in d["losses"].items()})
PY
done
SVC_DP_FORCE=1 timeout 600 python bench.py --mode train --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r_train_rccl_world1.json 2> gpurun_out/r_train_rccl_world1.err; echo "rccl two-graph rc=$?"
grep '^{' gpurun_out/r_train_rccl_world1.json | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['ms_per_step'],d['losses'],d['allreduce'])"
SVC_DP_FORCE=1 SVC_DP_CAPTURE_COLLECTIVES=1 timeout 600 python bench.py --mode train --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r_train_rccl_captured.json 2> gpurun_out/r_train_rccl_captured.err; echo "rccl captured rc=$?"
grep '^{' gpurun_out/r_train_rccl_captured.json | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['ms_per_step'],d['losses'],d['allreduce'])"
grep -v "^frame\|^$" gpurun_out/r_train_rccl_captured.err | tail -6 | cut -c1-300
timeout 600 python scripts/train_op_census.py > gpurun_out/r_train_aten_op_census.txt 2> gpurun_out/r_census.err; echo "census rc=$?"; head -40 gpurun_out/r_train_aten_op_census.txt
