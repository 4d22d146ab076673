#!/bin/bash
# round 3, final evidence of the final build (quad encode, batched prologue, 172 ids): full GPU suite, smoke(), bench lines + rocprofv3 kernel stats + PMC passes with meta,
# SQ counters of the headline, slow-family lines, one launch per step.
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r3z; mkdir -p $OUT
echo "== full GPU suite"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee $OUT/pytest_gpu_full_suite.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $OUT/smoke.log
echo "== evidence"
bash profiles/collect_r3.sh r3z 2>&1 | tee $OUT/collect.log
line() { python - $1 $2 <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-30s %-20s n=%-7d %7.3f G steps/s %6.2f us/step frac %.3f spl %d" % (sys.argv[2], d["config"]["env_id"][:20], d["config"]["envs_per_gpu"], d["value"]/1e9, d["ms_per_step"]*1e3, d["roofline"]["frac"], d["config"]["steps_per_launch"]))
except Exception as ex:
    print(sys.argv[2], "FAILED", ex)
PY
}
echo "== other lines" | tee $OUT/bench_lines_other.txt
for w in bosslevel dynobs16x16; do timeout 300 python bench.py --workload $w --steps 256 --warmup 64 --no-cpu-baseline > $OUT/bench_$w.json 2> $OUT/b.err; line $OUT/bench_$w.json $w; done 2>&1 | tee -a $OUT/bench_lines_other.txt
timeout 300 python bench.py --workload bosslevel --envs-per-gpu 32768 --steps 256 --warmup 64 --no-cpu-baseline > $OUT/bench_bosslevel_32768.json 2> $OUT/b.err; line $OUT/bench_bosslevel_32768.json bosslevel_32768 | tee -a $OUT/bench_lines_other.txt
for w in empty8x8 doorkey8x8 gotoredball lavacrossing_full; do timeout 200 python bench.py --workload $w --fused 0 --steps 512 --warmup 64 --no-cpu-baseline > $OUT/bench_${w}_unfused.json 2> $OUT/b.err; line $OUT/bench_${w}_unfused.json ${w}_unfused; done 2>&1 | tee -a $OUT/bench_lines_other.txt
for w in empty8x8_rgb doorkey8x8_rgb_partial; do timeout 200 python bench.py --workload $w --steps 64 --warmup 16 --no-cpu-baseline > $OUT/bench_$w.json 2> $OUT/b.err; line $OUT/bench_$w.json $w; done 2>&1 | tee -a $OUT/bench_lines_other.txt
echo "== SQ counters of the default headline run"
cd /tmp
CMD="python $ROOT/bench.py --workload empty8x8 --steps 256 --warmup 64 --no-cpu-baseline"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/sq$i -o x -- $CMD > $OUT/sq$i.log 2>&1
  python $ROOT/profiles/summarize_pmc.py $(find $OUT/sq$i -name '*counter_collection.csv' | head -1) | grep k_roll
  rm -rf $OUT/sq$i
done 2>&1 | tee $OUT/sq_counters_empty8x8.txt
