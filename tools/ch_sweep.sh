#!/bin/bash
# Window size of the H query's table (ZKG_TABLE_C_H) against proof latency at small payload counts: run through gpurun from the repo root.
for k in 1 2 4; do
  for c in 10 11 12 13 14 15 16; do
    echo -n "c_h=$c "; ZKG_TABLE_C_H=$c REPS=30 timeout -k 10 120 python3 tools/zklaim_prove_profile.py $k | tail -1 || exit 1
  done
done
