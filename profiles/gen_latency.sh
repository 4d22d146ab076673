#!/bin/bash
export TMPDIR=/tmp; ROOT=$PWD; mkdir -p $ROOT/gpurun_out/genlat; cd /tmp
for id in MiniGrid-DoorKey-8x8-v0 MiniGrid-LavaCrossingS9N1-v0 BabyAI-GoToRedBall-v0; do
  for n in 256 2048; do
    rm -rf /tmp/gl; timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gl -o x -- python $ROOT/profiles/gen_latency.py $id $n > /dev/null 2>&1
    echo "$id n=$n $(grep k_generate $(find /tmp/gl -name '*kernel_stats.csv') | cut -d, -f2-8)"
  done
done
