#!/bin/bash
export TMPDIR=/tmp
bash profiles/pmc_sq_r4.sh r4d empty8x8 split
bash profiles/pmc_sq_r4.sh r4d empty8x8 timesplit MG_ROLL_SPLIT=0
rocm-smi --showclocks 2>/dev/null | head -20
