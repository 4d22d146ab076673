#!/bin/bash
# Round 5, GPU call: how long the waves of k_roll7's log split wait for each other (attribution build: profiles/spin_counts.py)
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r5spin; mkdir -p $OUT
export MINIGRID_AMD_LIB=$ROOT/minigrid_amd/libminigrid_hip_spin.so
for rep in 1 2; do
python profiles/spin_counts.py MiniGrid-Empty-8x8-v0 65536 16
python profiles/spin_counts.py MiniGrid-DoorKey-8x8-v0 262144 8
python profiles/spin_counts.py MiniGrid-DoorKey-8x8-v0 65536 16
python profiles/spin_counts.py BabyAI-GoToRedBall-v0 32768 16
python profiles/spin_counts.py MiniGrid-Empty-8x8-v0 32768 16
done 2>&1 | grep -v amdgpu.ids | tee $OUT/spin_counts.txt
for x in 0 32 2 6; do MG_EXP=$x python bench.py --steps 1024 --warmup 256 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('attr MG_EXP=$x  %.3f us/step' % (d['ms_per_step']*1e3))"; done | tee -a $OUT/spin_counts.txt
