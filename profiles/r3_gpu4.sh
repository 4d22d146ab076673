#!/bin/bash
# round 3, GPU call 4: k_roll7 with aligned LDS accesses only -- parity subset, sweeps, attribution, SQ counters, sanitizer retry.
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r3d; mkdir -p $OUT
echo "== parity subset"
timeout 600 python -m pytest tests/test_gpu_roll.py tests/test_gpu_fused_full.py tests/test_gpu_multi.py -x -q 2>&1 | tail -6 | tee $OUT/pytest_subset.log
line() { python - $1 $2 <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-26s %-20s n=%-7d %7.3f G steps/s %6.2f us/step" % (sys.argv[2], d["config"]["env_id"][:20], d["config"]["envs_per_gpu"], d["value"]/1e9, d["ms_per_step"]*1e3))
except Exception as ex:
    print(sys.argv[2], "FAILED", ex)
PY
}
echo "== sweeps" | tee $OUT/sweeps.txt
for nw in 1 2 4; do for n in 65536 262144; do
  MG_ROLL_NW=$nw timeout 100 python bench.py --workload empty8x8 --envs-per-gpu $n --steps 1024 --warmup 128 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; line $OUT/b.json empty_nw$nw
done; done 2>&1 | tee -a $OUT/sweeps.txt
for nw in 1 2 4; do for w in doorkey8x8 gotoredball; do
  MG_ROLL_NW=$nw timeout 100 python bench.py --workload $w --steps 1024 --warmup 128 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; line $OUT/b.json ${w}_nw$nw
done; done 2>&1 | tee -a $OUT/sweeps.txt
for a in "--fused 0 --steps 512 --warmup 64" "--steps 20 --warmup 5" "--workload doorkey8x8 --fused 0 --steps 512 --warmup 64"; do
  timeout 100 python bench.py $a --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; line $OUT/b.json "$(echo $a | tr ' ' '_' | cut -c1-26)"
done 2>&1 | tee -a $OUT/sweeps.txt
echo "== MG_EXP attribution, NW=1" | tee $OUT/exp.txt
for n in 65536 262144; do for x in 0 2 6 30; do
  MG_ROLL_NW=1 MG_EXP=$x timeout 100 python bench.py --workload empty8x8 --envs-per-gpu $n --steps 512 --warmup 64 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; line $OUT/b.json nw1_exp$x
done; done 2>&1 | tee -a $OUT/exp.txt
for x in 0 1 2 6; do MG_ROLL_NW=1 MG_EXP=$x timeout 100 python bench.py --workload doorkey8x8 --steps 512 --warmup 64 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; line $OUT/b.json doorkey_nw1_exp$x; done 2>&1 | tee -a $OUT/exp.txt
echo "== SQ counters, NW=1, 65536 envs"
cd /tmp
CMD="python $ROOT/bench.py --workload empty8x8 --steps 256 --warmup 64 --no-cpu-baseline"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  MG_ROLL_NW=1 timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/sq$i -o x -- $CMD > $OUT/sq$i.log 2>&1
  python $ROOT/profiles/summarize_pmc.py $(find $OUT/sq$i -name '*counter_collection.csv' | head -1) | grep k_roll
  rm -rf $OUT/sq$i
done 2>&1 | tee $OUT/sq_counters_empty8x8_nw1.txt
echo "== sanitizer, torch-free process with the ROCm runtime"
cd $ROOT
RT=$(python profiles/asan_build.py --runtime)
HSA_XNACK=1 LD_PRELOAD=$RT LD_LIBRARY_PATH=/opt/rocm/lib ASAN_OPTIONS=detect_leaks=0 MINIGRID_AMD_NO_TORCH=1 MINIGRID_AMD_LIB=$PWD/minigrid_amd/libminigrid_hip_asan.so timeout 200 python profiles/first_process.py 4096 > $OUT/asan_C.log 2>&1; echo "asan C rc=$? $(tail -1 $OUT/asan_C.log | cut -c1-100)"; head -12 $OUT/asan_C.log | cut -c1-200
