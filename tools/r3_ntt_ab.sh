#!/bin/bash
# NTT on the 29-bit representation: parity tests, then lone transforms and the prove legs in the variants below
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3_ntt
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_ntt.py tests/test_gpu_step_domain.py -x -q > $OUT/pytest.log 2>&1; rc=$?
tail -3 $OUT/pytest.log
[ $rc -eq 0 ] || exit 1
for v in pad0 ntt32; do
  unset ZKG_NTT_32 ZKG_NTT29_PAD ZKG_NTT_TILE_LOG
  case $v in pad1) export ZKG_NTT29_PAD=1;; tile10) export ZKG_NTT_TILE_LOG=10;; ntt32) export ZKG_NTT_32=1;; esac
  echo "== $v"
  for lg in 18 20 22; do timeout -k 10 100 python3 tools/ntt_profile.py $lg 50 2>&1 | tail -1; done
done
