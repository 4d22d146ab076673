#!/bin/bash
# round 3: where the cycles of ONE DynamicObstacles-16x16 episode generation go (MG_DEBUG_TIMING stamps of wave 0)
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r3ac; mkdir -p $OUT
for id in MiniGrid-Dynamic-Obstacles-16x16-v0 MiniGrid-DoorKey-8x8-v0 BabyAI-GoToRedBall-v0; do timeout 120 python profiles/gen_stamps.py $id 64 2>&1 | tail -6; done | tee $OUT/gen_stamps.txt
