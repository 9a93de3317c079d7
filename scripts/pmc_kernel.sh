#!/bin/bash
# SQ / GRBM counters + durations of the launches of ONE kernel (name substring) in a command; separate rocprofv3 --pmc passes with
# --kernel-trace only.  (Eight counters per pass at most: a ninth in the first set — SQ_INSTS_VALU_MFMA_MOPS_F16, round 5 — made both
# passes of that set sit until their 300 s timeout and cost ten GPU-minutes for nothing.)  usage: pmc_kernel.sh TAG KERNEL_SUBSTRING "command ..."   -> gpurun_out/TAG_pmc_KERNEL.txt
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=$1; KNAME=$2; CMD=$3
D=gpurun_out/pmc_k; mkdir -p $D; rm -rf $D/p_*
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAVE32_INSTS"; do
  i=$((i+1))
  timeout 90 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $D/p_$i -o run -- $CMD > $D/p_$i.log 2>&1; echo "rc=$?"; tail -1 $D/p_$i.log
done
SAFE=$(echo "$KNAME" | tr -c 'A-Za-z0-9_\n' '_')     # (template arguments in the substring: not a file name)
OUT="gpurun_out/${TAG}_pmc_${SAFE}.txt"
echo "=== kernel *$KNAME* in: $CMD" >> "$OUT"
KNAME="$KNAME" python - >> "$OUT" <<'PY'
import csv, glob, collections, os
kn = os.environ["KNAME"]
agg = collections.defaultdict(lambda: [0.0, 0])
dur = []
for f in sorted(glob.glob('gpurun_out/pmc_k/p_*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        if kn in r['Kernel_Name']:
            a = agg[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for f in sorted(glob.glob('gpurun_out/pmc_k/p_*/**/*kernel_trace.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        if kn in r['Kernel_Name']:
            dur.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
v = {k: a[0] / a[1] for k, a in agg.items()}
for k in sorted(v): print(f"{k:36s} {v[k]:16.1f} per launch ({agg[k][1]} launches)")
if dur:
    dur.sort(); d = dur[len(dur) // 2]
    print(f"kernel duration (median of {len(dur)} profiled launches): {d:.1f} us")
    if 'GRBM_GUI_ACTIVE' in v:
        print(f"derived: GUI_ACTIVE/duration = {v['GRBM_GUI_ACTIVE'] / d / 1e3:.3f} GHz (x8 if it sums the XCDs)")
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in v and 'GRBM_GUI_ACTIVE' in v:
        print(f"derived: MFMA pipe busy = {v['SQ_VALU_MFMA_BUSY_CYCLES'] / (v['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f} of SIMD-cycles (GUI_ACTIVE summed over 8 XCDs)")
    if 'SQ_WAVE_CYCLES' in v:
        w = v['SQ_WAVE_CYCLES']
        print("derived: of wave-cycles: " + ", ".join(f"{n} {v[n] / w:.3f}" for n in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_INST_CYCLES_VMEM', 'SQ_ACTIVE_INST_VMEM') if n in v))
PY
cat $OUT
