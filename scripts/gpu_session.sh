#!/bin/bash
# One gpurun call = a list of named steps (arguments), each writing under gpurun_out/<tag>_*.  Replaces the one-shot
# scripts of rounds 1-2.  usage: gpurun -- 'bash scripts/gpu_session.sh TAG step [step ...]'
#   convtests   pytest of tests/test_conv1d_gpu.py          fulltests  whole -m gpu suite
#   convbench   bench_conv.py strip=0 / strip=1 in the pair's two forms
#   pairbench   bench_pair.py (fused ResBlock pair vs two launches)   gemmbench  bench_train_kernels.py gemm
#   inferab     bench.py --mode infer under SVC_CONV_STRIP x SVC_MRF_STREAMS
#   bench       the driver's default bench.py line          prof       rocprofv3 kernel-trace stats of the infer step (serialised)
#   profsplit   the same for the split pipeline (bench.py --mode infer --split)      profhalf  ... for the half mode (--half)
#   trainprof   rocprofv3 kernel-trace stats of the training step (trainprof_bf16: the bf16 mode)  pmc  FETCH_SIZE / WRITE_SIZE passes of the infer step
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
TAG=$1; shift
O=gpurun_out/$TAG
for step in "$@"; do
case $step in
convtests) timeout 600 python -m pytest tests/test_conv1d_gpu.py -m gpu -q --timeout=300 -rf > ${O}_convtests.log 2>&1; tail -15 ${O}_convtests.log ;;
fulltests) timeout 1200 python -m pytest tests -m gpu -q --timeout=300 -rf > ${O}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> ${O}_pytest_gpu.log; tail -15 ${O}_pytest_gpu.log ;;
convbench) for m in conv1 conv2; do BENCH_CONV_MODE=$m timeout 300 python scripts/bench_conv.py strip=0 0 strip=1 0 > ${O}_convbench_$m.txt 2>&1; cat ${O}_convbench_$m.txt; done ;;
pairbench) timeout 300 python scripts/bench_pair.py > ${O}_pairbench.txt 2>&1; tail -8 ${O}_pairbench.txt ;;
gemmbench) timeout 300 python scripts/bench_train_kernels.py gemm > ${O}_gemmbench.txt 2>&1; cat ${O}_gemmbench.txt ;;
inferab) for st in 0 1; do for ms in 1 0; do SVC_CONV_STRIP=$st SVC_MRF_STREAMS=$ms timeout 300 python bench.py --mode infer --steps 30 --warmup 5 --no-cpu-baseline > ${O}_infer_strip${st}_streams${ms}.json 2> ${O}_infer_strip${st}_streams${ms}.err; cat ${O}_infer_strip${st}_streams${ms}.json; done; done ;;
bench) timeout 900 python bench.py > ${O}_bench.json 2> ${O}_bench.err; echo "bench rc=$?"; cat ${O}_bench.json; tail -3 ${O}_bench.err ;;
prof) rm -rf gpurun_out/prof_stats; SVC_MRF_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats -o run -- python bench.py --mode infer --steps 10 --warmup 3 --no-cpu-baseline --no-graph --no-roofline --no-extras --no-host-io --no-steady > ${O}_prof_bench.json 2> ${O}_prof_bench.err
      DB=$(find gpurun_out/prof_stats -name '*.db' | head -1); python scripts/prof_summary.py $DB > ${O}_infer_T862_kernel_stats_serialised.txt 2>&1; head -40 ${O}_infer_T862_kernel_stats_serialised.txt ;;
profsplit) rm -rf gpurun_out/prof_split; SVC_MRF_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_split -o run -- python bench.py --mode infer --split --steps 10 --warmup 3 --no-cpu-baseline --no-graph --no-roofline --no-extras --no-host-io --no-steady --no-pmc > ${O}_profsplit_bench.json 2> ${O}_profsplit_bench.err
      DB=$(find gpurun_out/prof_split -name '*.db' | head -1); python scripts/prof_summary.py $DB > ${O}_infer_split_T862_kernel_stats_serialised.txt 2>&1; head -30 ${O}_infer_split_T862_kernel_stats_serialised.txt; rm -rf gpurun_out/prof_split ;;
profhalf) rm -rf gpurun_out/prof_half; SVC_MRF_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_half -o run -- python bench.py --mode infer --half --steps 10 --warmup 3 --no-cpu-baseline --no-graph --no-roofline --no-extras --no-host-io --no-steady --no-pmc > ${O}_profhalf_bench.json 2> ${O}_profhalf_bench.err
      DB=$(find gpurun_out/prof_half -name '*.db' | head -1); python scripts/prof_summary.py $DB > ${O}_infer_half_T862_kernel_stats_serialised.txt 2>&1; head -30 ${O}_infer_half_T862_kernel_stats_serialised.txt; rm -rf gpurun_out/prof_half ;;
trainprof_bf16) rm -rf gpurun_out/prof_train16; timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train16 -o run -- python bench.py --mode train --bf16 --steps 3 --warmup 1 --no-roofline --no-cpu-baseline --no-extras > ${O}_trainprof_bf16_bench.json 2> ${O}_trainprof_bf16_bench.err
      DB=$(find gpurun_out/prof_train16 -name '*.db' | head -1); python scripts/prof_summary.py $DB > ${O}_train_bf16_B16_kernel_stats.txt 2>&1; head -40 ${O}_train_bf16_B16_kernel_stats.txt ;;
trainprof) rm -rf gpurun_out/prof_train; timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train -o run -- python bench.py --mode train --steps 3 --warmup 1 --no-roofline --no-cpu-baseline --no-extras > ${O}_trainprof_bench.json 2> ${O}_trainprof_bench.err
      DB=$(find gpurun_out/prof_train -name '*.db' | head -1); python scripts/prof_summary.py $DB > ${O}_train_B16_kernel_stats.txt 2>&1; head -60 ${O}_train_B16_kernel_stats.txt ;;
pmc) rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
     SVC_MRF_STREAMS=0 timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o run -- python bench.py --mode infer --steps 3 --warmup 1 --no-cpu-baseline --no-graph --no-roofline --no-extras --no-host-io --no-steady > ${O}_pmc_fetch.log 2>&1
     SVC_MRF_STREAMS=0 timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o run -- python bench.py --mode infer --steps 3 --warmup 1 --no-cpu-baseline --no-graph --no-roofline --no-extras --no-host-io --no-steady > ${O}_pmc_write.log 2>&1
     python scripts/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write 4 ${O}_pmc_conv.json > ${O}_pmc_summary.txt 2>&1; cat ${O}_pmc_summary.txt ;;
*) echo "custom step: $step"; eval "$step" ;;
esac
done
# raw profiler output stays on the box: only the summaries travel (gpurun merges at most 64 MiB back — a session with four
# profiling steps once lost everything to that limit)
rm -rf gpurun_out/prof_stats gpurun_out/prof_train gpurun_out/prof_train16 gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_k gpurun_out/pmc_wgrad
find gpurun_out -name '*.db' -size +5M -delete
find gpurun_out -name '*counter_collection.csv' -size +5M -delete
du -sh gpurun_out
