#!/bin/bash
# VGPR / SGPR / scratch of every k_step variant (device-only compile, no GPU needed): bash profiles/regstats.sh
set -e; ROOT=$PWD; mkdir -p /tmp/mg_regstats
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -I$ROOT/include --cuda-device-only -S -o /tmp/mg_regstats/dev.s $ROOT/minigrid_amd/csrc/mg_api.hip
python - <<'PY'
import re
t = open('/tmp/mg_regstats/dev.s').read()
for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.wavefront_size', t, re.S):
    n, b = m.group(1), m.group(2)
    if not any(k in n for k in ('k_step', 'k_generate', 'k_move', 'k_render')): continue
    g = lambda k: re.search(k + r':\s+(\d+)', b).group(1)
    print(f"{n[:70]:70s} scratch {g('.private_segment_fixed_size'):>4s}  sgpr {g('.sgpr_count'):>3s}  vgpr {g('.vgpr_count'):>3s}  spills {g('.vgpr_spill_count')}")
PY
