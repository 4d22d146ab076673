#!/bin/bash
# Round 6, call 7: the spare's agent record prefetched with its grid; who bounds MultiRoom now -- kernel trace of the de-phased run (refill on packed lanes / on
# cooperative wavefronts) beside the bench lines; the other big-grid families
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f) episodes in the timed region %d (share %.5f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], c.get('episodes_finished_in_timed_region_rank0', -1), c.get('autoreset_share_timed', -1)))
except Exception as ex: print('$1 FAILED', ex)"; }
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_roll.py tests/test_gpu_fused_full.py tests/test_gpu_fused.py -q -m gpu -n 4 2>&1 | tail -3 | tee $OUT/pytest_gpu_call7.log
for w in multiroom babyai_goto bosslevel keycorridor; do
  python bench.py --workload $w --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "$w steps 1024 (de-phased)"
  python bench.py --workload $w --steps 1024 --warmup 128 --no-cpu-baseline --dephase 0 2>/dev/null | line "$w steps 1024 --dephase 0"
done | tee $OUT/bench_lines_generators_call7.txt
cd /tmp
for cfg in "MG_X=0" "MG_LANE_BURST=0"; do
  env $cfg timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_multiroom_$cfg -o mr -- python $ROOT/bench.py --workload multiroom --steps 1024 --warmup 128 --no-cpu-baseline > $OUT/kt_multiroom_$cfg.log 2>&1
  grep "^{" $OUT/kt_multiroom_$cfg.log | line "multiroom under the tracer $cfg"
  head -8 $(find $OUT/kt_multiroom_$cfg -name '*kernel_stats.csv' | head -1) | cut -c1-200 | tee $OUT/kernel_stats_multiroom_call7_$cfg.txt
  python - $(find $OUT/kt_multiroom_$cfg -name '*kernel_trace.csv' | head -1) <<'PY' | tee -a $OUT/kernel_stats_multiroom_call7_$cfg.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t_end = int(rows[-1]["End_Timestamp"])
# the last 40 ms of the run: busy time per kernel family and stream occupancy
win = [r for r in rows if int(r["Start_Timestamp"]) > t_end - 12_000_000]
fam = {}
for r in win:
    n = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void mg::", "")
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    f = fam.setdefault(n, [0, 0]); f[0] += 1; f[1] += d
span = int(win[-1]["End_Timestamp"]) - int(win[0]["Start_Timestamp"])
print("last 12 ms of the trace: span %.2f ms" % (span / 1e6))
for n, (c, d) in sorted(fam.items(), key=lambda x: -x[1][1]): print("  %-28s calls %4d  busy %.2f ms  avg %.1f us" % (n, c, d / 1e6, d / c / 1e3))
PY
  rm -rf $OUT/kt_multiroom_$cfg
done
