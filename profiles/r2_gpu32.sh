#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r2ag
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r2ag/t_all.log 2>&1; echo "all rc=$?" | tee -a gpurun_out/r2ag/summary.txt; tail -3 gpurun_out/r2ag/t_all.log | cut -c1-300
bash profiles/r2_final.sh > gpurun_out/r2ag_log.txt 2>&1; grep -E "G steps/s|driver" gpurun_out/r2ag_log.txt | head -20
