#!/bin/bash
# kernel-trace stats of one bench workload: bash profiles/kstats.sh <workload> [extra env assignments...]
export TMPDIR=/tmp; ROOT=$PWD; W=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp; rm -rf /tmp/ks
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o x -- python $ROOT/bench.py --workload $W --steps 500 --warmup 100 --no-cpu-baseline > /tmp/ks.log 2>&1
echo "== $W $@"; python $ROOT/profiles/trace_gaps.py $(find /tmp/ks -name '*kernel_trace.csv'); head -4 $(find /tmp/ks -name '*kernel_stats.csv') | cut -d, -f1-8 | cut -c1-150
