export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r5refresh; mkdir -p $OUT
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f)  frac %.3f  traffic %s (%s)  host-event %.1f us' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], r['frac'], r['traffic'], (r['traffic_source'] or 'floor'), (d['host_ms']-d['event_ms'])*1e3))
except Exception as ex: print('$1 FAILED', ex)"; }
for w in keycorridor unlock unlockpickup blockedunlockpickup multiroom babyai_goto bosslevel; do
  timeout 300 python bench.py --workload $w --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | tee $OUT/bench_$w.json | line "$w"
done | tee $OUT/bench_lines_generator_families_final_tree.txt
