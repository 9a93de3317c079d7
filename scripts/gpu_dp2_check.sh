#!/bin/bash
# 2 ranks on ONE GPU over gloo: functional check of the data-parallel training path (hooks, arena buckets, broadcast)
# with the real HIP modules; compares against the 1-rank run on the union batch is not possible here (weak scaling), so
# it checks: both ranks finish, losses finite, all-reduce statistics reported, parameters identical across ranks.
set -x
cd $GRAFT_REPO_ROOT
export SVC_DIST_BACKEND=gloo
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --mode train --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/dp2.json 2> gpurun_out/dp2.err; echo "rc=$?"
cat gpurun_out/dp2.json | tail -2; tail -5 gpurun_out/dp2.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 scripts/dp2_consistency.py 2>&1 | tail -6
