cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for w in 0 3 2 0 3 2; do
  if [ $w = 0 ]; then unset ZKG_ACC29_WAVES; else export ZKG_ACC29_WAVES=$w; fi
  for k in 8 37; do REPS=30 timeout -k 10 200 python3 tools/zklaim_prove_profile.py $k 2>/dev/null | tail -1 | sed "s/^/waves $w /"; done
done
