#!/bin/bash
# Round-2 call u: A/B of the wgrad time-split target for the one-workgroup-per-CU instantiations (SVC_WGRAD_TARGET).
set -x
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
for tg in 128 256 512 128; do
  SVC_WGRAD_SMALL_TARGET=$tg timeout 300 python bench.py --mode train --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/u_train_tg$tg.json 2> gpurun_out/u_train_tg$tg.err; echo "tg=$tg rc=$?"
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/u_train_tg$tg.json").read().splitlines() if l.startswith("{")][-1])
print("WGRAD_SMALL_TARGET=$tg ms_per_step", round(d["ms_per_step"],2), "wgrad", d["families"]["conv1d_wgrad"], {k: round(v,4) for k,v in list(d["losses"].items())[:3]})
PY
done
