#!/bin/bash
# window size of the witness tables (ZKG_TABLE_C_W) against the prove time, 8 and 37 payloads
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for k in 8 37; do for c in default 10 12 13 14 16; do if [ $c = default ]; then unset ZKG_TABLE_C_W; else export ZKG_TABLE_C_W=$c; fi; echo -n "c_w=$c "; REPS=24 timeout -k 10 120 python3 tools/zklaim_prove_profile.py $k 2>&1 | tail -1; done; done
