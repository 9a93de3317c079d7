#!/bin/bash
# Round-2 closing GPU call: the tests touched since call r, default bench line (infer + train + host_io), the inference path under an
# initialised process group, kernel-trace stats (infer serialised + train), PMC HBM-traffic passes.  Outputs under gpurun_out/.
set -x
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_train_ops_gpu.py tests/test_train_gpu.py tests/test_train_loop_gpu.py tests/test_data_parallel_gpu.py tests/test_boundary_gpu.py -m gpu -q --timeout=600 -rf > gpurun_out/f2_pytest_subset.log 2>&1; echo "pytest rc=$?" >> gpurun_out/f2_pytest_subset.log
tail -4 gpurun_out/f2_pytest_subset.log | cut -c1-300
timeout 900 python bench.py --steps 30 --warmup 5 > gpurun_out/f2_bench.json 2> gpurun_out/f2_bench.err; echo "bench rc=$?"
cat gpurun_out/f2_bench.json; tail -3 gpurun_out/f2_bench.err
SVC_DP_FORCE=1 timeout 600 python bench.py --mode infer --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/f2_bench_infer_pg.json 2> gpurun_out/f2_bench_infer_pg.err; echo "infer under process group rc=$?"
grep '^{' gpurun_out/f2_bench_infer_pg.json | cut -c1-400
rm -rf gpurun_out/prof_stats gpurun_out/prof_train gpurun_out/pmc_fetch gpurun_out/pmc_write
SVC_MRF_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats -o run -- python bench.py --mode infer --steps 10 --warmup 3 --no-cpu-baseline --no-graph > gpurun_out/f2_bench_prof.json 2> gpurun_out/f2_bench_prof.err; echo "rocprof rc=$?"
DB=$(find gpurun_out/prof_stats -name '*.db' | head -1); python scripts/prof_summary.py $DB > gpurun_out/f2_kernel_stats.txt 2>&1; head -12 gpurun_out/f2_kernel_stats.txt
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train -o run -- python bench.py --mode train --steps 3 --warmup 1 --no-roofline --no-cpu-baseline > gpurun_out/f2_bench_train_prof.json 2> gpurun_out/f2_bench_train_prof.err; echo "rocprof train rc=$?"
DB=$(find gpurun_out/prof_train -name '*.db' | head -1); python scripts/prof_summary.py $DB > gpurun_out/f2_kernel_stats_train.txt 2>&1; head -30 gpurun_out/f2_kernel_stats_train.txt
SVC_MRF_STREAMS=0 timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o run -- python bench.py --mode infer --steps 3 --warmup 1 --no-cpu-baseline --no-graph --no-roofline --no-host-io > gpurun_out/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
SVC_MRF_STREAMS=0 timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o run -- python bench.py --mode infer --steps 3 --warmup 1 --no-cpu-baseline --no-graph --no-roofline --no-host-io > gpurun_out/pmc_write.log 2>&1; echo "pmc write rc=$?"
python scripts/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write 4 gpurun_out/f2_pmc_conv1d_mfma.json > gpurun_out/f2_pmc_summary.txt 2>&1; cat gpurun_out/f2_pmc_summary.txt
find gpurun_out -name '*.db' -size +30M -delete
find gpurun_out -name '*counter_collection.csv' -size +20M -delete
du -sh gpurun_out
