#!/bin/bash
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/pmc_conv
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > gpurun_out/pmc_conv/sq_counters.txt
for cfg in 0 10000; do
export SVC_CONV_CFG=$cfg
echo "=== SVC_CONV_CFG=$cfg (0: LDS-DMA double-buffered kernel, 10000: register-staged kernel)"
rm -rf gpurun_out/pmc_conv/SQ_* gpurun_out/pmc_conv/GRBM_*
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_conv/$tag -o run -- python scripts/conv_one.py 128 128 55168 11 5 4 > gpurun_out/pmc_conv/$tag.log 2>&1; echo "rc=$?"
done
python - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob('gpurun_out/pmc_conv/*/*counter_collection.csv')):
    agg=collections.defaultdict(lambda:[0.0,0])
    for r in csv.DictReader(open(f)):
        if 'conv1d_mfma' in r['Kernel_Name']:
            a=agg[r['Counter_Name']]; a[0]+=float(r['Counter_Value']); a[1]+=1
    for k,(v,n) in agg.items(): print(f"{k:36s} {v/n:16.1f} per launch ({n} launches)")
PY
done
