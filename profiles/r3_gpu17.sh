#!/bin/bash
# round 3: BabyAI-SynthS5R2-v0 on the device (exact test of RoomGrid.place_agent's endless loop), unfused launch anatomy
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r3q; mkdir -p $OUT
echo "== SynthS5R2"
timeout 900 python -m pytest tests/test_gpu_synths5r2.py tests/test_gpu_parity.py -m gpu -q -x -k "synths5r2 or SynthS5R2" 2>&1 | tail -15 | tee $OUT/pytest_synths5r2.log
echo "== sentence levels (FLAG_STUCK check in take_spare) + fused suites"
timeout 900 python -m pytest tests/test_gpu_roll.py tests/test_gpu_fused.py -m gpu -q -x 2>&1 | tail -4 | tee $OUT/pytest_roll_fused.log
echo "== headline sanity"
timeout 200 python bench.py --workload empty8x8 --steps 1024 --warmup 128 --no-cpu-baseline > $OUT/bench_empty8x8.json 2> $OUT/b.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3q/bench_empty8x8.json").read().strip().splitlines()[-1])
print("empty8x8 %.3f G %.2f us/step frac %.3f" % (d["value"]/1e9, d["ms_per_step"]*1e3, d["roofline"]["frac"]))
PY
echo "== one launch per step: kernel time vs step time (rocprofv3)"
cd /tmp
for w in empty8x8 doorkey8x8; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/unf_$w -o x -- python $ROOT/bench.py --workload $w --fused 0 --steps 512 --warmup 64 --no-cpu-baseline > $OUT/unf_$w.json 2> $OUT/unf_$w.err
  f=$(find $OUT/unf_$w -name '*kernel_stats.csv' | head -1); head -4 $f | tee $OUT/kernel_stats_${w}_unfused.csv; rm -rf $OUT/unf_$w
  tail -1 $OUT/unf_$w.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w unfused under rocprof: %.2f us/step' % (d['ms_per_step']*1e3))"
done
