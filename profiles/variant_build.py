#!/usr/bin/env python3
"""A/B and attribution variants of the product library: the same sources with extra -D switches on the step translation units
(the generator units are shared with the product build), next to the product library, selected at run time with MINIGRID_AMD_LIB:

    python profiles/variant_build.py attr -DMG_ATTRIBUTION        # -> minigrid_amd/libminigrid_hip_attr.so (reads MG_EXP)
    MINIGRID_AMD_LIB=minigrid_amd/libminigrid_hip_attr.so MG_EXP=2 python bench.py ...

    python profiles/variant_build.py lanewide --units=mg_gen_lane.hip,mg_api.hip -DMG_LANE_WIDE=1     # other translation units than the step ones

The product library never reads MG_EXP (mg_roll.h MG_EXPBIT); bench.py prints mg_build_info() and every MG_* variable into its line."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    from minigrid_amd import build as B
    name = sys.argv[1]
    flags = [a for a in sys.argv[2:] if a != "--force" and not a.startswith("--units=")]
    units = {u for u in B.UNITS if u.startswith("mg_step_") or u == "mg_api.hip"}
    for a in sys.argv[2:]:
        if a.startswith("--units="):
            units = set(a[len("--units="):].split(","))
            assert units <= set(B.UNITS), sorted(units - set(B.UNITS))
    lib = os.path.join(B.HERE, f"libminigrid_hip_{name}.so")
    print(B.build(force="--force" in sys.argv, verbose=True, lib=lib, extra_flags=flags, tag="_" + name, flag_units=units))
