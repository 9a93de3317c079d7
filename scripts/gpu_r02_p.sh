#!/bin/bash
# Round-2 call p: parity of the transformer flow + DiscriminatorP padded rows, training A/B (SVC_DISCP_PAD), RCCL dry run at N=1.
set -x
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_train_ops_gpu.py tests/test_train_gpu.py tests/test_data_parallel_gpu.py "tests/test_infer_gpu.py::test_infer_matches_reference_golden" -m gpu -q --timeout=600 -rf --maxfail=30 > gpurun_out/p_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/p_pytest.log
tail -25 gpurun_out/p_pytest.log
for pad in 0 1 0 1; do
  SVC_DISCP_PAD=$pad timeout 600 python bench.py --mode train --steps 8 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/p_train_pad$pad.json 2> gpurun_out/p_train_pad$pad.err; echo "pad=$pad rc=$?"
  python - <<PY
import json
d=json.loads(open("gpurun_out/p_train_pad$pad.json").read().strip().splitlines()[-1])
print("PAD=$pad ms_per_step", d["ms_per_step"], {k: round(v,4) for k,v in d["losses"].items()})
PY
done
SVC_DP_FORCE=1 timeout 600 python bench.py --mode train --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/p_train_rccl_world1.json 2> gpurun_out/p_train_rccl_world1.err; echo "rccl dry run rc=$?"
cat gpurun_out/p_train_rccl_world1.json; tail -5 gpurun_out/p_train_rccl_world1.err
SVC_DP_FORCE=1 SVC_DP_CAPTURE_COLLECTIVES=1 timeout 600 python bench.py --mode train --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/p_train_rccl_captured.json 2> gpurun_out/p_train_rccl_captured.err; echo "rccl captured rc=$?"
cat gpurun_out/p_train_rccl_captured.json; tail -5 gpurun_out/p_train_rccl_captured.err
