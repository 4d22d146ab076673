// mg_host.h — host-side helpers shared by the runtime (mg_api.hip) and the host self-tests (mg_selftest.hip).
#pragma once
#include "../../include/minigrid_hip.h"
#include "mg_gen.h"

namespace mg {

// the generators' view of a level's configuration (also used by mg_selftest_generate, which has no handle)
inline GenParams gen_params_of(const mg_config& c) {
  GenParams g;
  g.kind = c.env_kind; g.W = c.width; g.H = c.height;
  g.start_x = c.agent_start_x; g.start_y = c.agent_start_y; g.start_dir = c.agent_start_dir;
  g.num_crossings = c.num_crossings;
  g.obstacle_cell = c.obstacle_type == (int)T_WALL ? (int)CELL_WALL_GREY : (int)CELL_LAVA;
  g.num_dists = c.num_dists;
  g.strip2_row = c.strip2_row;
  g.room_size = c.room_size;
  g.random_length = c.random_length;
  g.max_steps = c.max_steps; g.instr_off = 0; g.scratch_off = 0;
  return g;
}

}  // namespace mg
