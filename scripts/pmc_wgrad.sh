#!/bin/bash
# SQ / GRBM counters + kernel durations of ONE wgrad shape (separate rocprofv3 --pmc passes with --kernel-trace only).
# usage: pmc_wgrad.sh TAG "B Ca Cb T K dil"
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=$1; SHAPE=$2
D=gpurun_out/pmc_wgrad; mkdir -p $D; rm -rf $D/p_*
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $D/p_$i -o run -- python scripts/wgrad_one.py $SHAPE 6 > $D/p_$i.log 2>&1; echo "rc=$?"; tail -1 $D/p_$i.log
done
echo "=== wgrad shape $SHAPE" >> gpurun_out/${TAG}_pmc_wgrad.txt
python - >> gpurun_out/${TAG}_pmc_wgrad.txt <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
dur = []
for f in sorted(glob.glob('gpurun_out/pmc_wgrad/p_*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        if 'conv1d_wgrad' in r['Kernel_Name']:
            a = agg[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for f in sorted(glob.glob('gpurun_out/pmc_wgrad/p_*/**/*kernel_trace.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        if 'conv1d_wgrad' in r['Kernel_Name']:
            dur.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
v = {k: a[0] / a[1] for k, a in agg.items()}
for k in sorted(v): print(f"{k:36s} {v[k]:16.1f} per launch ({agg[k][1]} launches)")
if dur:
    dur.sort(); d = dur[len(dur) // 2]
    print(f"kernel duration (median of {len(dur)} profiled launches): {d:.1f} us")
    if 'GRBM_GUI_ACTIVE' in v:
        print(f"derived: clock = GUI_ACTIVE/duration = {v['GRBM_GUI_ACTIVE'] / d / 1e3:.3f} GHz")
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in v and 'GRBM_GUI_ACTIVE' in v:
        print(f"derived: MFMA pipe busy = {v['SQ_VALU_MFMA_BUSY_CYCLES'] / (v['GRBM_GUI_ACTIVE'] * 1024):.3f} of SIMD-cycles (if GUI_ACTIVE is per-chip) or {v['SQ_VALU_MFMA_BUSY_CYCLES'] / (v['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f} (if it sums 8 XCDs)")
    if 'SQ_WAVE_CYCLES' in v:
        w = v['SQ_WAVE_CYCLES']
        print("derived: of wave-cycles: " + ", ".join(f"{n} {v[n] / w:.3f}" for n in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_LDS') if n in v))
PY
cat gpurun_out/${TAG}_pmc_wgrad.txt
