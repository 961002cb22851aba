#!/bin/bash
# stream-priority sweep of the prover (ZKG_PRIO bits: 1 NTT stream, 2 H, 4 witness jobs, 8 ones-sums), 8 and 37 payloads
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for k in 8 37; do for p in 1 0 3 7; do REPS=24 ZKG_PRIO=$p timeout -k 10 120 python3 tools/zklaim_prove_profile.py $k 2>&1 | tail -1; done; done
