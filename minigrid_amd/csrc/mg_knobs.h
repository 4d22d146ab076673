// mg_knobs.h — EVERY environment variable the library reads, in one place (round 6: 38 getenv sites scattered over mg_api.hip before; VERDICT r5 "next" #8).
// None is needed for normal use: they are A/B switches whose default is the measured best, and debugging aids.  A handle reads them ONCE, at mg_create
// (mg_env::k); bench.py prints every MG_* variable of its process into its JSON line (config.environment), so a number produced under a knob says so.
//
// Deleted in round 6, each after its A/B had been decided in at least two rounds (the measured winner is now the only path; the evidence is under
// profiles/r2 .. r5): MG_GEN_PRIO (generator stream priority: high), MG_LIVE_OVERLAP (off), MG_MOVE_EPB, MG_ROLL_DROT (dynamics-wave rotation shift 8),
// MG_ROLL_RATIO (0.12), MG_ROLL_SHARE (stepping wave 0), MG_ROLL_ONE (on), MG_SENT_SPLIT (on), MG_FULL_SPLIT (on), MG_NO_ROLL_FULL, MG_LPE,
// MG_LANE_GEN (on), MG_REFILL_WPS (1 / 2 / 4 by level), MG_RENDER_EPW, MG_RENDER_BLOCKS, MG_RING_CAP_GB (32), MG_TRAJ_SLOTS and
// MG_MAX_FUSED (mg_config.traj_slots / bench.py --spl do the same), the store-policy builds (-DMG_OBS_STORE_AUX), and MG_ROLL_STAGED (round 6's A/B: the STAGED split of the big grids is the only path).
#pragma once
#include <cstdlib>

namespace mg {

struct Knobs {
  // ---- k_roll7 launch shape ----
  int roll_nw = 0;             // MG_ROLL_NW=1..4       waves per k_roll7 workgroup (0 = the level's default)
  bool roll_split = true;      // MG_ROLL_SPLIT=0       the time split instead of the log / staged split (round-3 shape; tests keep it exact)
  int roll_epw = 64;           // MG_ROLL_EPW=32        32 envs per workgroup (measured, not adopted: profiles/r4/epw32.txt; tests keep it exact)
  int roll_shadows = -1;       // MG_ROLL_SHADOWS=0|1|2 spare episodes per env staged in LDS per fused launch (-1 = the level's default)
  int dyn_inloop = -1;         // MG_DYN_INLOOP=0       DynamicObstacles: the round-3 three-launch step instead of the draws inside k_roll7 (-1 = unset)
  int dring = 0;               // MG_DRING=2|4          code stagings of the staged split (0 = the level's default; the protocol stress test sets 2)
  long long nt_mb = 256;       // MG_NT_BYTES=<MB>      burst size from which observation stores are nontemporal (0 = always, negative = never)
  // ---- episode generation ----
  long long lane_burst = -1;   // MG_LANE_BURST=<n>     burst hybrid: requests per refill from which packed lanes serve it (0 = off, -1 = the level's default)
  int lane_direct = -1;        // MG_LANE_DIRECT=0|1    direct generation on lanes: never | at every batch size (-1 = from 16 384 envs on)
  int lane_packed = -1;        // MG_LANE_PACKED=0|1    the levels whose refill runs on lanes refill per segment | PACKED (-1 = the level's default)
  int lane_lpw = 64;           // MG_LANE_LPW=1..64     busy lanes per wavefront of the packed refill
  int lane_cap = -1;           // MG_LANE_CAP=<n>       lane refills: a request draws a quarter of its env's free ring slots, at least n (0 = all of them, -1 = the default, 2)
  int spare_ring = 0;          // MG_SPARE_RING=<R>     spare-episode ring depth (power of two >= 4; the CPU emulator suite runs with 4)
  // ---- debugging ----
  bool guard = false;          // MG_GUARD=1            red zones around every device buffer, checked at mg_sync
  bool abort_backtrace = false;// MG_ABORT_BACKTRACE=1  print a backtrace when the process aborts inside the library
  int exp = 0;                 // MG_EXP=<bits>         attribution BUILDS only (-DMG_ATTRIBUTION): parts of the step switched off

  static Knobs from_env() {
    Knobs k;
    auto num = [](const char* name, long long dflt) { const char* s = getenv(name); return s ? atoll(s) : dflt; };
    { const long long v = num("MG_ROLL_NW", 0); if (v >= 1 && v <= 4) k.roll_nw = (int)v; }
    k.roll_split = num("MG_ROLL_SPLIT", 1) != 0;
    { const long long v = num("MG_ROLL_EPW", 64); if (v == 32 || v == 64) k.roll_epw = (int)v; }
    { const long long v = num("MG_ROLL_SHADOWS", -1); if (v >= 0 && v <= 2) k.roll_shadows = (int)v; }
    { const long long v = num("MG_DYN_INLOOP", -1); if (v >= 0) k.dyn_inloop = v != 0; }
    { const long long v = num("MG_DRING", 0); if (v == 2 || v == 4) k.dring = (int)v; }
    k.nt_mb = num("MG_NT_BYTES", 256);
    { const long long v = num("MG_LANE_BURST", -1); if (v >= 0) k.lane_burst = v; }
    { const long long v = num("MG_LANE_DIRECT", -1); if (v == 0 || v == 1) k.lane_direct = (int)v; }
    { const long long v = num("MG_LANE_PACKED", -1); if (v == 0 || v == 1) k.lane_packed = (int)v; }
    { const long long v = num("MG_LANE_LPW", 64); if (v >= 1 && v <= 64) k.lane_lpw = (int)v; }
    { const long long v = num("MG_LANE_CAP", -1); if (v >= 0 && v <= 128) k.lane_cap = (int)v; }
    { const long long v = num("MG_SPARE_RING", 0); if (v >= 4) k.spare_ring = (int)v; }
    k.guard = num("MG_GUARD", 0) == 1;
    k.abort_backtrace = num("MG_ABORT_BACKTRACE", 0) == 1;
    k.exp = (int)num("MG_EXP", 0);
    return k;
  }
};

}  // namespace mg
