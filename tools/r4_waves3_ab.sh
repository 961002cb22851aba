cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for w in 2 3 2 3; do
  ZKG_ACC29_WAVES=$w timeout -k 10 300 python3 bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/w3_$w.json 2>/dev/null || exit 1
  python3 -c "
import json; j=json.load(open('gpurun_out/w3_$w.json')); r=j['scalars_resident']; print('waves $w: headline median', j['ms_per_step_stats']['median'], 'accum in step', j['roofline']['kernel_ms'], '| resident median', r['ms_per_step']['median'], 'accum alone', r['accumulation_kernel']['kernel_ms'])"
done
