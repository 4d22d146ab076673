#!/bin/bash
# k_render launch-geometry sweep + the plain write-bandwidth ceiling of this box: bash profiles/render_sweep.sh
export TMPDIR=/tmp; ROOT=$PWD
python - <<'PY'
import torch, time
x = torch.empty(805306368, dtype=torch.uint8, device="cuda")
y = torch.empty_like(x)
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
us = t(lambda: x.zero_()); print(f"ceiling: memset 805 MB        {us:8.1f} us  {805.306368e6/us/1e6:6.2f} TB/s written")
us = t(lambda: x.fill_(7)); print(f"ceiling: fill_ 805 MB         {us:8.1f} us  {805.306368e6/us/1e6:6.2f} TB/s written")
us = t(lambda: y.copy_(x)); print(f"ceiling: copy 805 MB -> 805 MB {us:8.1f} us  {805.306368e6/us/1e6:6.2f} TB/s written (+ same read)")
PY
run() { echo "== $*"; env "$@" timeout 100 python $ROOT/bench.py --workload ${W:-empty8x8_rgb} --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('   us/step %.1f  G env-steps/s %.3f  frac %.3f' % (j['roofline']['avg_launch_us'], j['value']/1e9, j['roofline']['frac']))"; }
run MG_RENDER_EPW=16
run MG_RENDER_EPW=16
for epw in 4 8 32 64; do run MG_RENDER_EPW=$epw; done
for b in 512 1024 4096 65536; do run MG_RENDER_BLOCKS=$b MG_RENDER_EPW=16; done
W=doorkey8x8_rgb_partial run MG_RENDER_EPW=16
W=doorkey8x8_rgb_partial run MG_RENDER_EPW=8
