#!/bin/bash
# Round 6, call 24 (after gen_goto_lane): BabyAI-GoToRedBall x 32 768 -- who is busy: kernel trace of the timed region + instruction counters per kernel
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
cd /tmp
CMD="python $ROOT/bench.py --workload gotoredball --steps 2048 --warmup 256 --no-cpu-baseline"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_gtrb -o g -- $CMD > $OUT/kt_gtrb.log 2>&1
head -8 $(find $OUT/kt_gtrb -name '*kernel_stats.csv' | head -1) | cut -c1-220 | tee $OUT/kernel_stats_gotoredball_call24.txt
python - $(find $OUT/kt_gtrb -name '*kernel_trace.csv' | head -1) <<'PY' | tee -a $OUT/kernel_stats_gotoredball_call24.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t_end = int(rows[-1]["End_Timestamp"])
win = [r for r in rows if int(r["Start_Timestamp"]) > t_end - 5_000_000]
fam = {}
for r in win:
    n = r["Kernel_Name"].split("(")[0].replace("void mg::", "")[:60]
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    f = fam.setdefault(n, [0, 0]); f[0] += 1; f[1] += d
span = int(win[-1]["End_Timestamp"]) - int(win[0]["Start_Timestamp"])
print("last 5 ms of the trace: span %.2f ms" % (span / 1e6))
for n, (c, d) in sorted(fam.items(), key=lambda x: -x[1][1]): print("  %-62s calls %4d  busy %.2f ms  avg %.1f us" % (n, c, d / 1e6, d / c / 1e3))
PY
rm -rf $OUT/kt_gtrb
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/sq_gtrb -o x -- $CMD > $OUT/sq_gtrb.log 2>&1
python $ROOT/profiles/summarize_pmc.py $(find $OUT/sq_gtrb -name '*counter_collection.csv' | head -1) | cut -c1-220 | tee $OUT/sq_counters_gotoredball_all_kernels_call24.txt
rm -rf $OUT/sq_gtrb
