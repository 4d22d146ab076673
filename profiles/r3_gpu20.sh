#!/bin/bash
# round 3: the quad encode (12 bytes per lane from one code dword) against the chunk encode (libminigrid_hip_chunk.so, -DMG_ENCODE_QUADS=0)
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r3t; mkdir -p $OUT
echo "== parity (quad encode)"
timeout 900 python -m pytest tests/test_gpu_roll.py tests/test_gpu_fused.py -m gpu -q -x 2>&1 | tail -3 | tee $OUT/pytest_quads.log
timeout 900 python -m pytest tests/test_gpu_fused_full.py tests/test_gpu_parity.py -m gpu -q -x -k "Empty-8x8 or LavaCrossingS9N1 or ragged or DoorKey-8x8" 2>&1 | tail -3 | tee -a $OUT/pytest_quads.log
line() { python - $1 "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-44s n=%-7d %7.3f G steps/s %6.2f us/step frac %.3f" % (sys.argv[2], d["config"]["envs_per_gpu"], d["value"]/1e9, d["ms_per_step"]*1e3, d["roofline"]["frac"]))
except Exception as ex:
    print(sys.argv[2], "FAILED", ex)
PY
}
echo "== quad encode (default library) vs chunk encode" | tee $OUT/encode_ab.txt
for rep in 1 2; do for lib in quad chunk; do
  if [ $lib = chunk ]; then export MINIGRID_AMD_LIB=$ROOT/minigrid_amd/libminigrid_hip_chunk.so; else unset MINIGRID_AMD_LIB; fi
  for w in empty8x8 doorkey8x8 lavacrossing_full gotoredball; do
    timeout 200 python bench.py --workload $w --steps 1024 --warmup 128 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; line $OUT/b.json "$lib $w"
  done
  timeout 100 python bench.py --workload empty8x8 --fused 0 --steps 512 --warmup 64 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; line $OUT/b.json "$lib empty8x8 unfused"
  timeout 100 python bench.py --workload empty8x8 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; line $OUT/b.json "$lib empty8x8 driver-sized"
done; done 2>&1 | tee -a $OUT/encode_ab.txt
