#!/bin/bash
# Round 5: the device-side half of round 4's UBSan finding.  RG::door_xy / RG::mark (mg_gen.h) shift by a negative count on the
# exhausted-draw-budget path of add_door; host builds mask the count (MG_SHC), the device build still has the expression the GPU suite validated because
# the explicit mask changes the register allocation of eight generator translation units.  This run validates the masked form ON the device:
#   build here first:  python profiles/variant_build.py shcmask --units=$(python - <<'PY'
# from minigrid_amd import build as B; print(",".join(u for u in B.UNITS if u.startswith("mg_gen_")))
# PY
# ) -DMG_SHC_MASK_DEVICE=1
# then on the GPU box: the generator goldens and multi-episode oracle runs of the RoomGrid / BabyAI levels with the variant library, and their refill
# rates beside the product library's.  Green + same rates: delete the `&& !defined(MG_SHC_MASK_DEVICE)` in mg_gen.h.
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r5shcmask; mkdir -p $OUT
LIB=$ROOT/minigrid_amd/libminigrid_hip_shcmask.so
[ -f $LIB ] || { echo "build the variant first (see the header)"; exit 1; }
MINIGRID_AMD_LIB=$LIB timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_roll.py tests/test_gpu_fused.py tests/test_gpu_synths5r2.py -q -m gpu -n 4 2>&1 | tail -6 | tee $OUT/pytest_shcmask.log
for w in keycorridor babyai_goto bosslevel; do
  for lib in "" $LIB; do
    MINIGRID_AMD_LIB=$lib timeout 300 python bench.py --workload $w --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', '${lib:+masked}', '%.3f G env-steps/s' % (d['value']/1e9))"
  done
done | tee $OUT/bench_lines.txt
