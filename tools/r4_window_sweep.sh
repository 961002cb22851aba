cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in 16 15 14 16 15 17; do
  ZKG_MSM_C=$c timeout -k 10 300 python3 bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/csweep_$c.json 2>/dev/null || { echo "c=$c failed"; continue; }
  python3 -c "
import json; j=json.load(open('gpurun_out/csweep_$c.json')); r=j['scalars_resident']; print('c=$c: headline median', j['ms_per_step_stats']['median'], 'accum in step', j['roofline']['kernel_ms'], '| resident median', r['ms_per_step']['median'], 'accum alone', r['accumulation_kernel']['kernel_ms'], 'same', r['same_result'])"
done
