#!/bin/bash
# round 3: BabyAI-SynthS5R2-v0 on the device; anatomy of the one-step launch (Env.step) and the rotating stepping wave
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r3r; mkdir -p $OUT
echo "== SynthS5R2"
timeout 900 python -m pytest tests/test_gpu_synths5r2.py tests/test_gpu_parity.py -m gpu -q -k "synths5r2 or SynthS5R2" 2>&1 | tail -25 | tee $OUT/pytest_synths5r2.log
line() { python - $1 "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-44s n=%-7d %7.3f G steps/s %6.2f us/step" % (sys.argv[2], d["config"]["envs_per_gpu"], d["value"]/1e9, d["ms_per_step"]*1e3))
except Exception as ex:
    print(sys.argv[2], "FAILED", ex)
PY
}
echo "== one launch per step: which wave steps (MG_ROLL_SHARE: 16 = wave 0, k = (wg >> (k-1)) & 3, 0 = no shared encode)" | tee $OUT/unfused_stepping_wave.txt
for w in empty8x8 gotoredball; do for sh in 16 1 2 3 4 6 9 0; do
  MG_ROLL_SHARE=$sh timeout 100 python bench.py --workload $w --fused 0 --steps 512 --warmup 64 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; line $OUT/b.json "$w unfused MG_ROLL_SHARE=$sh"
done; done 2>&1 | tee -a $OUT/unfused_stepping_wave.txt
MG_ROLL_SHARE=9 timeout 100 python bench.py --workload doorkey8x8 --envs-per-gpu 131072 --fused 0 --steps 512 --warmup 64 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; line $OUT/b.json "doorkey 131072 unfused MG_ROLL_SHARE=9" | tee -a $OUT/unfused_stepping_wave.txt
MG_ROLL_SHARE=16 timeout 100 python bench.py --workload doorkey8x8 --envs-per-gpu 131072 --fused 0 --steps 512 --warmup 64 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; line $OUT/b.json "doorkey 131072 unfused MG_ROLL_SHARE=16" | tee -a $OUT/unfused_stepping_wave.txt
echo "== anatomy of the one-step launch (MG_EXP: 16 no transition, 4 no codes, 2 no encode + obs stores, 8 no scalar outputs)" | tee $OUT/unfused_anatomy.txt
for sh in 16 0; do for ex in 0 2 6 22 30; do
  MG_ROLL_SHARE=$sh MG_EXP=$ex timeout 100 python bench.py --workload empty8x8 --fused 0 --steps 512 --warmup 64 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; line $OUT/b.json "empty8x8 unfused share=$sh MG_EXP=$ex"
done; done 2>&1 | tee -a $OUT/unfused_anatomy.txt
echo "== parity of the one-step path with a rotating stepping wave"
for sh in 1 9; do MG_ROLL_SHARE=$sh timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "ragged or 4096_envs" 2>&1 | tail -2; done | tee $OUT/pytest_share_rotation.log
