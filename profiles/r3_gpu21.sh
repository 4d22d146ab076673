#!/bin/bash
# round 3: quad encode, plain vs software-pipelined loop (libminigrid_hip_pipe.so, -DMG_QUAD_PIPELINE=1); SQ counters of the quad build
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r3u; mkdir -p $OUT
line() { python - $1 "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-44s n=%-7d %7.3f G steps/s %6.2f us/step frac %.3f" % (sys.argv[2], d["config"]["envs_per_gpu"], d["value"]/1e9, d["ms_per_step"]*1e3, d["roofline"]["frac"]))
except Exception as ex:
    print(sys.argv[2], "FAILED", ex)
PY
}
echo "== quad encode: plain loop (default library) vs pipelined" | tee $OUT/encode_pipe_ab.txt
for rep in 1 2; do for lib in plain pipe; do
  if [ $lib = pipe ]; then export MINIGRID_AMD_LIB=$ROOT/minigrid_amd/libminigrid_hip_pipe.so; else unset MINIGRID_AMD_LIB; fi
  for w in empty8x8 doorkey8x8 gotoredball; do
    timeout 200 python bench.py --workload $w --steps 1024 --warmup 128 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; line $OUT/b.json "$lib $w"
  done
  timeout 100 python bench.py --workload empty8x8 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; line $OUT/b.json "$lib empty8x8 driver-sized"
done; done 2>&1 | tee -a $OUT/encode_pipe_ab.txt
unset MINIGRID_AMD_LIB
MINIGRID_AMD_LIB=$ROOT/minigrid_amd/libminigrid_hip_pipe.so timeout 300 python -m pytest tests/test_gpu_roll.py -m gpu -q -x 2>&1 | tail -2
echo "== SQ counters, quad build, headline"
cd /tmp
CMD="python $ROOT/bench.py --workload empty8x8 --steps 256 --warmup 64 --no-cpu-baseline"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/sq$i -o x -- $CMD > $OUT/sq$i.log 2>&1
  python $ROOT/profiles/summarize_pmc.py $(find $OUT/sq$i -name '*counter_collection.csv' | head -1) | grep k_roll
  rm -rf $OUT/sq$i
done 2>&1 | tee $OUT/sq_counters_empty8x8_quads.txt
