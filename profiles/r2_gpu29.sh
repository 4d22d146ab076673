#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r2ad
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r2ad/t_all.log 2>&1; echo "all rc=$?" | tee -a gpurun_out/r2ad/summary.txt; tail -3 gpurun_out/r2ad/t_all.log | cut -c1-300
bash profiles/r2_final.sh > gpurun_out/r2ad_log.txt 2>&1; tail -16 gpurun_out/r2ad_log.txt | cut -c1-200
