#!/bin/bash
# round 3, GPU call 1: crash hunt as the box's first GPU process, the new parity tests of the benchmarked configuration,
# occupancy experiments on the r2 kernel, the sanitizer loop.
export TMPDIR=/tmp
OUT=gpurun_out/r3a; mkdir -p $OUT
echo "== first process of the box (plain build)" | tee $OUT/first_process.log
timeout 180 python profiles/first_process.py >> $OUT/first_process.log 2>&1; echo "rc=$?" | tee -a $OUT/first_process.log
tail -3 $OUT/first_process.log
echo "== pytest: benchmarked configuration at full size + ADVICE regressions"
timeout 600 python -m pytest tests/test_gpu_fused_full.py -x -q 2>&1 | tail -15 | tee $OUT/pytest_fused_full.log
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "BossLevel or GoToSeq-v0 or OpenDoorsOrderN4-v0 or MoveTwoAcrossS8N9" 2>&1 | tail -5 | tee $OUT/pytest_sentence.log
echo "== occupancy experiment: the r2 fused kernel at 1 / 2 / 4 waves per SIMD worth of Empty-8x8 envs"
for n in 65536 131072 262144; do
  timeout 120 python bench.py --workload empty8x8 --envs-per-gpu $n --steps 1024 --warmup 128 --no-cpu-baseline > $OUT/bench_empty_n$n.json 2> $OUT/bench_empty_n$n.err
  python - $OUT/bench_empty_n$n.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["config"]["envs_per_gpu"], "%.3f G steps/s"%(d["value"]/1e9), "%.2f us/step"%(d["ms_per_step"]*1e3), "event %.2f ms host %.2f ms"%(d["event_ms"], d["host_ms"]), "frac %.3f"%d["roofline"]["frac"])
PY
done 2>&1 | tee $OUT/occupancy.txt
timeout 60 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver.json 2> $OUT/bench_driver.err; cut -c1-400 $OUT/bench_driver.json
echo "== one launch per step: kernel duration vs launch period (rocprofv3 kernel trace)"
( cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_unfused -o unfused -- python $OLDPWD/bench.py --fused 0 --steps 512 --warmup 64 --no-cpu-baseline > $OLDPWD/$OUT/prof_unfused.log 2>&1 )
cp $(find $OUT/prof_unfused -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_empty8x8_unfused.csv 2>/dev/null; head -4 $OUT/kernel_stats_empty8x8_unfused.csv | cut -c1-200
tail -1 $OUT/prof_unfused.log | cut -c1-300
rm -rf $OUT/prof_unfused
echo "== sanitizer loop"
HSA_XNACK=1 rocminfo 2>/dev/null | grep -i -m3 xnack | tee $OUT/xnack.txt
RT=$(python profiles/asan_build.py --runtime)
if [ -f minigrid_amd/libminigrid_hip_asan.so ]; then
  t_end=$((SECONDS + 240)); i=0; bad=0
  while [ $SECONDS -lt $t_end ] && [ $i -lt 50 ]; do
    i=$((i+1))
    HSA_XNACK=1 LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 MINIGRID_AMD_LIB=$PWD/minigrid_amd/libminigrid_hip_asan.so \
      timeout 120 python profiles/first_process.py 4096 > $OUT/asan_run_$i.log 2>&1
    rc=$?; echo "asan run $i rc=$rc $(tail -1 $OUT/asan_run_$i.log | cut -c1-80)"
    if [ $rc -ne 0 ]; then bad=$((bad+1)); else rm -f $OUT/asan_run_$i.log; fi
    [ $bad -ge 3 ] && break
  done 2>&1 | tee $OUT/asan_loop.txt
fi
echo "== serialized-kernel loop (AMD_SERIALIZE_KERNEL=3), plain build"
for i in 1 2 3 4 5 6; do AMD_SERIALIZE_KERNEL=3 timeout 120 python profiles/first_process.py > $OUT/serial_$i.log 2>&1; echo "serialized run $i rc=$? $(tail -1 $OUT/serial_$i.log | cut -c1-60)"; done 2>&1 | tee $OUT/serial_loop.txt
