#!/bin/bash
# Round-2 final GPU call: full GPU parity suite, default bench line (infer + train), kernel-trace stats (infer serialised + train),
# PMC HBM-traffic passes (separate runs, no other trace domains).  Outputs under gpurun_out/.
set -x
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 -rf --maxfail=20 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -12 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 30 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"
cat gpurun_out/bench.json; tail -3 gpurun_out/bench.err
rm -rf gpurun_out/prof_stats gpurun_out/prof_train gpurun_out/pmc_fetch gpurun_out/pmc_write
SVC_MRF_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats -o run -- python bench.py --mode infer --steps 10 --warmup 3 --no-cpu-baseline --no-graph > gpurun_out/bench_prof.json 2> gpurun_out/bench_prof.err; echo "rocprof rc=$?"
DB=$(find gpurun_out/prof_stats -name '*.db' | head -1); python scripts/prof_summary.py $DB > gpurun_out/kernel_stats.txt 2>&1; head -45 gpurun_out/kernel_stats.txt
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train -o run -- python bench.py --mode train --steps 3 --warmup 1 --no-roofline --no-cpu-baseline > gpurun_out/bench_train_prof.json 2> gpurun_out/bench_train_prof.err; echo "rocprof train rc=$?"
DB=$(find gpurun_out/prof_train -name '*.db' | head -1); python scripts/prof_summary.py $DB > gpurun_out/kernel_stats_train.txt 2>&1; head -30 gpurun_out/kernel_stats_train.txt
SVC_MRF_STREAMS=0 timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o run -- python bench.py --mode infer --steps 3 --warmup 1 --no-cpu-baseline --no-graph --no-roofline > gpurun_out/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
SVC_MRF_STREAMS=0 timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o run -- python bench.py --mode infer --steps 3 --warmup 1 --no-cpu-baseline --no-graph --no-roofline > gpurun_out/pmc_write.log 2>&1; echo "pmc write rc=$?"
python scripts/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write 4 gpurun_out/pmc_conv1d_mfma.json > gpurun_out/pmc_summary.txt 2>&1; cat gpurun_out/pmc_summary.txt
find gpurun_out -name '*.db' -size +30M -delete
find gpurun_out -name '*counter_collection.csv' -size +20M -delete
du -sh gpurun_out
