#!/bin/bash
# Round-2 closing call w: default bench line and training kernel trace on the final build.
set -x
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python bench.py --steps 30 --warmup 5 > gpurun_out/w_bench.json 2> gpurun_out/w_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/w_bench.json").read().splitlines() if l.startswith("{")][-1])
t=d["train"]; print("infer ms", d["ms_per_step"], d["value"], "host_io", d["host_io"]["ms_per_step"], "train ms", t["ms_per_step"], t["losses"]); print(t["families"]); print(d["roofline"]["frac"], d["cpu_baseline"]["value"], t["cpu_baseline"])
PY
rm -rf gpurun_out/prof_train
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train -o run -- python bench.py --mode train --steps 3 --warmup 1 --no-roofline --no-cpu-baseline > gpurun_out/w_bench_train_prof.json 2> gpurun_out/w_bench_train_prof.err; echo "rocprof train rc=$?"
DB=$(find gpurun_out/prof_train -name '*.db' | head -1); python scripts/prof_summary.py $DB > gpurun_out/w_kernel_stats_train.txt 2>&1; head -20 gpurun_out/w_kernel_stats_train.txt
find gpurun_out -name '*.db' -size +30M -delete
