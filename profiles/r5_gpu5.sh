#!/bin/bash
# Round 5, GPU call 5: (1) the device-side shift mask of RG::door_xy / RG::mark (variant build, generator TUs only) against the product library,
# (2) the evidence of the final step kernels (profiles/r5_final.sh: profiler passes bound to the build by hash + bench lines).
bash profiles/r5_shift_mask.sh
bash profiles/r5_final.sh skip-suite
