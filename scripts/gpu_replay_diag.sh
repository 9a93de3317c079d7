#!/bin/bash
# Replay-corruption diagnostics: per-iteration losses of the bench's training graph, 10 iterations, under
# variants of host synchronisation (SYNC=none | atK: one synchronize after iteration K | 1: after every iteration) and of what the
# captured iteration contains.  usage: gpu_replay_diag.sh TAG "ENV=VAL ..." ["ENV=VAL ..." ...]   (each variant runs 4 times)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=$1; shift
O=gpurun_out/${TAG}_replay_diag.txt; : > $O
for v in "$@"; do
  echo "## $v" >> $O
  for i in 1 2 3 4; do env $v N=10 python scripts/diag_train_determinism.py 2>/dev/null | grep "^SYNC" >> $O; done
done
cat $O
