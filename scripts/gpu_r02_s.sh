#!/bin/bash
# Round-2 call s: unguarded wgrad MFMA loops — training-op parity, golden training tests, default bench line, training kernel trace.
set -x
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_train_ops_gpu.py tests/test_train_gpu.py tests/test_train_loop_gpu.py tests/test_data_parallel_gpu.py tests/test_diffusion.py -m gpu -q --timeout=600 -rf > gpurun_out/s_pytest_training.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s_pytest_training.log
tail -6 gpurun_out/s_pytest_training.log | cut -c1-300
timeout 900 python bench.py --steps 30 --warmup 5 > gpurun_out/s_bench.json 2> gpurun_out/s_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/s_bench.json").read().splitlines() if l.startswith("{")][-1])
t=d["train"]; print("infer ms", d["ms_per_step"], "train ms", t["ms_per_step"], t["losses"]); print(t["families"])
PY
rm -rf gpurun_out/prof_train
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train -o run -- python bench.py --mode train --steps 3 --warmup 1 --no-roofline --no-cpu-baseline > gpurun_out/s_bench_train_prof.json 2> gpurun_out/s_bench_train_prof.err; echo "rocprof train rc=$?"
DB=$(find gpurun_out/prof_train -name '*.db' | head -1); python scripts/prof_summary.py $DB > gpurun_out/s_kernel_stats_train.txt 2>&1; head -34 gpurun_out/s_kernel_stats_train.txt
find gpurun_out -name '*.db' -size +30M -delete
