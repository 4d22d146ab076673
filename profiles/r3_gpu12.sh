#!/bin/bash
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r3l; mkdir -p $OUT
line() { python - $1 $2 <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-30s %-20s n=%-7d %7.3f G steps/s %6.2f us/step frac %.3f" % (sys.argv[2], d["config"]["env_id"][:20], d["config"]["envs_per_gpu"], d["value"]/1e9, d["ms_per_step"]*1e3, d["roofline"]["frac"]))
except Exception as ex:
    print(sys.argv[2], "FAILED", ex)
PY
}
echo "== parity: FullyObs through k_roll7, RGB with any view / tile size"
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_fused_full.py tests/test_gpu_parity.py -x -q -k "True or full or Lava or wrapping or pickl or rgb" 2>&1 | tail -6 | tee $OUT/pytest_quick.log
echo "== FullyObs bench" | tee $OUT/full_bench.txt
for nw in 1 2 3 4; do MG_ROLL_NW=$nw timeout 100 python bench.py --workload lavacrossing_full --steps 1024 --warmup 128 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; line $OUT/b.json lava_full_roll_nw$nw; done 2>&1 | tee -a $OUT/full_bench.txt
for x in 2 16; do MG_ROLL_NW=2 MG_EXP=$x timeout 100 python bench.py --workload lavacrossing_full --steps 512 --warmup 64 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; line $OUT/b.json lava_full_nw2_exp$x; done 2>&1 | tee -a $OUT/full_bench.txt
timeout 100 python bench.py --workload lavacrossing_full --fused 0 --steps 256 --warmup 64 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; line $OUT/b.json lava_full_unfused | tee -a $OUT/full_bench.txt
for nw in 2 3; do MG_ROLL_NW=$nw timeout 100 python bench.py --workload doorkey8x8 --obs-mode full --steps 512 --warmup 64 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; line $OUT/b.json doorkey_full_nw$nw; done | tee -a $OUT/full_bench.txt
timeout 100 python bench.py --workload empty8x8 --obs-mode rgb_partial --steps 64 --warmup 16 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; line $OUT/b.json empty_rgb_partial | tee -a $OUT/full_bench.txt
