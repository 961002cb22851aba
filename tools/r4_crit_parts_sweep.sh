#!/bin/bash
# which of the critical path's kernels gain from the raised wave priority at which size: ZKG_CRIT_PRIO_PARTS (1 matrix-vector + pointwise, 2 transforms,
# 4 the H job) with the size threshold off (ZKG_CRIT_PRIO_MIN_LOG=0), 30 sparse-witness proofs each, one box
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export ZKG_CRIT_PRIO_MIN_LOG=0
for k in ${KS:-8 20 37}; do
  for parts in 0 2 4 6 7 0 2; do
    ZKG_CRIT_PRIO_PARTS=$parts REPS=30 timeout -k 10 200 python3 tools/zklaim_prove_profile.py $k 2>/dev/null | tail -1 | sed "s/^/parts $parts /"
  done
done
