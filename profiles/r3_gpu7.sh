#!/bin/bash
# round 3, GPU call 7: evidence for profiles/r3 (bench lines, kernel stats, PMC passes with meta), slow-family traces, reset latency.
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r3g; mkdir -p $OUT
echo "== new tests"
timeout 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_roll.py -q -k "pickl or wrapping or primitives or split" 2>&1 | tail -5 | tee $OUT/pytest_new.log
echo "== evidence"
bash profiles/collect_r3.sh r3g 2>&1 | tee $OUT/collect.log
cd /tmp
for w in bosslevel dynobs16x16; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$w -o $w -- python $ROOT/bench.py --workload $w --steps 200 --warmup 40 --no-cpu-baseline > $OUT/prof_$w.log 2>&1
  cp $(find $OUT/prof_$w -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_$w.csv; rm -rf $OUT/prof_$w
  head -6 $OUT/kernel_stats_$w.csv | cut -c1-170
  tail -1 $OUT/prof_$w.log | cut -c1-200 > $OUT/bench_$w.json
done
echo "== SQ counters of the default headline run (NW = 4)"
CMD="python $ROOT/bench.py --workload empty8x8 --steps 256 --warmup 64 --no-cpu-baseline"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/sq$i -o x -- $CMD > $OUT/sq$i.log 2>&1
  python $ROOT/profiles/summarize_pmc.py $(find $OUT/sq$i -name '*counter_collection.csv' | head -1) | grep k_roll
  rm -rf $OUT/sq$i
done 2>&1 | tee $OUT/sq_counters_empty8x8.txt
cd $ROOT
echo "== one launch per step"
for w in empty8x8 doorkey8x8 gotoredball lavacrossing_full; do timeout 200 python bench.py --workload $w --fused 0 --steps 512 --warmup 64 --no-cpu-baseline > $OUT/bench_${w}_unfused.json 2> $OUT/bench_${w}_unfused.err; echo "rc=$? $(cut -c1-160 $OUT/bench_${w}_unfused.json)"; done
echo "== reset(seed) latency"
python - <<'PY' 2>&1 | tee $OUT/reset_latency.txt
import time, sys
sys.path.insert(0, ".")
import torch
import minigrid_amd as mg
for env_id, n in (("MiniGrid-DoorKey-8x8-v0", 262144), ("BabyAI-GoToRedBall-v0", 32768)):
    env = mg.make_vec(env_id, n, output="torch")
    env.reset(seed=0); env.sync()
    for k in range(3):
        t0 = time.perf_counter(); obs, _ = env.reset(seed=k + 1); torch.cuda.current_stream().synchronize(); first = obs["image"][0, 3, 6].cpu(); t1 = time.perf_counter()
        env.sync(); t2 = time.perf_counter()
        obs, *_ = env.step(torch.zeros(n, dtype=torch.uint8, device="cuda")); torch.cuda.current_stream().synchronize(); t3 = time.perf_counter()
        print(f"{env_id} x {n} reset(seed): first observation on the host after {1e3*(t1-t0):.2f} ms, ring redrawn after {1e3*(t2-t0):.2f} ms, next step {1e3*(t3-t2):.3f} ms")
    env.close()
PY
