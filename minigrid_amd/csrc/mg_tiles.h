// mg_tiles.h — the RGB tile atlas: every tile Grid.render_tile (core/grid.py:145-198) can produce, rendered ONCE on the
// host at mg_create time (51 cell kinds x {no agent, agent dir 0..3} x {plain, highlighted}); k_render then only blits.
// Product code.  This is the reference's own design (Grid.tile_cache, grid.py:27,161-164) with the cache filled eagerly.
//
// Exactness: the frame bytes are trunc(float64 mean of 3x3 uint8 sub-samples), so the atlas is bit-exact as long as
// every inside/outside decision at the sub-sample points agrees with the reference's arithmetic, which this file
// follows operation by operation (utils/rendering.py; float32 where the reference holds numpy float32 arrays).
// tests/test_abi_cpu.py compares all 510 tiles x tile sizes 4/8/12/16 with tiles rendered by the reference itself.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

#include "mg_device.h"

namespace mg {

constexpr int TILE_KEYS = 51;              // empty, wall/floor/key/ball/box x 6 colours, door x 6 x 3 states, goal, lava
constexpr int TILE_SUBDIVS = 3;            // grid.py:151

// dense tile key of a cell code: empty 0 | wall 1+c | floor 7+c | key 13+c | ball 19+c | box 25+c |
// door 31 + 3c + state | goal 49 | lava 50.  Anything the reference cannot draw falls back to the empty tile.
MG_HD uint32_t cell_tile_key(uint32_t code) {
  const uint32_t t = code & 15u;
  uint32_t c = (code >> 4) & 7u;
  c = c > 5u ? 5u : c;
  switch (t) {
    case T_WALL: return 1u + c;
    case T_FLOOR: return 7u + c;
    case T_KEY: return 13u + c;
    case T_BALL: return 19u + c;
    case T_BOX: return 25u + c;
    case T_BOX_KEY: return 25u + (uint32_t)C_GREY;            // a box shows its own colour, not its content's
    case T_BOX_DOORKEY: return 25u + c;
    case T_DOOR: return 31u + 3u * c;
    case T_DOOR_CLOSED: return 32u + 3u * c;
    case T_DOOR_LOCKED: return 33u + 3u * c;
    case T_GOAL: return 49u;
    case T_LAVA: return 50u;
    default: return 0u;
  }
}

namespace tiles {

struct Rgb { double r, g, b; };
// core/constants.py:8-15 (COLORS), indexed by COLOR_TO_IDX
inline Rgb color_rgb(uint32_t c) {
  static const Rgb k[6] = { {255, 0, 0}, {0, 255, 0}, {0, 0, 255}, {112, 39, 195}, {255, 255, 0}, {100, 100, 100} };
  return k[c > 5u ? 5u : c];
}
inline Rgb scaled(Rgb c, double f) { return { c.r * f, c.g * f, c.b * f }; }

struct Canvas {
  int S;
  std::vector<uint8_t> px;
  explicit Canvas(int s) : S(s), px((size_t)s * s * 3, 0) {}
  // fill_coords (rendering.py:25-37): the colour is stored into a uint8 image, i.e. truncated
  template <class F>
  void fill(F inside, Rgb c) {
    for (int y = 0; y < S; y++)
      for (int x = 0; x < S; x++) {
        const double yf = (y + 0.5) / S, xf = (x + 0.5) / S;
        if (inside(xf, yf)) {
          uint8_t* p = &px[((size_t)y * S + x) * 3];
          p[0] = (uint8_t)c.r; p[1] = (uint8_t)c.g; p[2] = (uint8_t)c.b;
        }
      }
  }
};

// rendering.py:95-99
struct InRect {
  double xmin, xmax, ymin, ymax;
  bool operator()(double x, double y) const { return x >= xmin && x <= xmax && y >= ymin && y <= ymax; }
};
// rendering.py:87-92
struct InCircle {
  double cx, cy, r;
  bool operator()(double x, double y) const { return (x - cx) * (x - cx) + (y - cy) * (y - cy) <= r * r; }
};
// rendering.py:55-84: the end points and the unit direction are float32, the per-point arithmetic float64
struct InLine {
  float p0x, p0y, dx, dy, dist;
  double xmin, xmax, ymin, ymax, r;
  InLine(double x0, double y0, double x1, double y1, double r_) : r(r_) {
    p0x = (float)x0; p0y = (float)y0;
    const float ex = (float)x1 - p0x, ey = (float)y1 - p0y;
    dist = std::sqrt(ex * ex + ey * ey);
    dx = ex / dist; dy = ey / dist;
    xmin = std::fmin(x0, x1) - r; xmax = std::fmax(x0, x1) + r;
    ymin = std::fmin(y0, y1) - r; ymax = std::fmax(y0, y1) + r;
  }
  bool operator()(double x, double y) const {
    if (x < xmin || x > xmax || y < ymin || y > ymax) return false;
    const double pqx = x - (double)p0x, pqy = y - (double)p0y;
    double a = pqx * (double)dx + pqy * (double)dy;
    a = a < 0.0 ? 0.0 : (a > (double)dist ? (double)dist : a);
    const double px = (double)p0x + a * (double)dx, py = (double)p0y + a * (double)dy;
    const double ux = x - px, uy = y - py;
    return std::sqrt(ux * ux + uy * uy) <= r;
  }
};
// rendering.py:102-127 behind rotate_fn (rendering.py:40-52): float32 corners, float64 point
struct InAgentTriangle {
  float ax, ay, v0x, v0y, v1x, v1y, dot00, dot01, dot11, inv;
  double ct, st;
  explicit InAgentTriangle(int agent_dir) {
    // grid.py:176-184: corners (0.12, 0.19), (0.87, 0.50), (0.12, 0.81), rotated by 0.5 * pi * agent_dir about the centre
    ax = 0.12f; ay = 0.19f;
    const float bx = 0.87f, by = 0.50f, cx = 0.12f, cy = 0.81f;
    v0x = cx - ax; v0y = cy - ay; v1x = bx - ax; v1y = by - ay;
    dot00 = v0x * v0x + v0y * v0y; dot01 = v0x * v1x + v0y * v1y; dot11 = v1x * v1x + v1y * v1y;
    inv = 1.0f / (dot00 * dot11 - dot01 * dot01);
    const double theta = 0.5 * M_PI * agent_dir;
    ct = std::cos(-theta); st = std::sin(-theta);
  }
  bool operator()(double x, double y) const {
    x = x - 0.5; y = y - 0.5;
    const double x2 = 0.5 + x * ct - y * st, y2 = 0.5 + y * ct + x * st;
    const double v2x = x2 - (double)ax, v2y = y2 - (double)ay;
    const double dot02 = (double)v0x * v2x + (double)v0y * v2y, dot12 = (double)v1x * v2x + (double)v1y * v2y;
    const double u = ((double)dot11 * dot02 - (double)dot01 * dot12) * (double)inv;
    const double v = ((double)dot00 * dot12 - (double)dot01 * dot02) * (double)inv;
    return u >= 0 && v >= 0 && (u + v) < 1;
  }
};

// the render() of each WorldObj (core/world_object.py:116-288)
inline void draw_object(Canvas& cv, uint32_t key) {
  const Rgb black{0, 0, 0};
  if (key == 0) return;
  if (key <= 6 || key == 49) { cv.fill(InRect{0, 1, 0, 1}, key == 49 ? color_rgb(C_GREEN) : color_rgb(key - 1)); return; }   // Wall :167, Goal :116
  if (key <= 12) { cv.fill(InRect{0.031, 1, 0.031, 1}, scaled(color_rgb(key - 7), 0.5)); return; }                           // Floor :131-134 (COLORS / 2)
  if (key <= 18) {                                                                                                           // Key :246-258
    const Rgb c = color_rgb(key - 13);
    cv.fill(InRect{0.50, 0.63, 0.31, 0.88}, c);
    cv.fill(InRect{0.38, 0.50, 0.59, 0.66}, c);
    cv.fill(InRect{0.38, 0.50, 0.81, 0.88}, c);
    cv.fill(InCircle{0.56, 0.28, 0.190}, c);
    cv.fill(InCircle{0.56, 0.28, 0.064}, black);
    return;
  }
  if (key <= 24) { cv.fill(InCircle{0.5, 0.5, 0.31}, color_rgb(key - 19)); return; }                                         // Ball :268-269
  if (key <= 30) {                                                                                                           // Box :280-288
    const Rgb c = color_rgb(key - 25);
    cv.fill(InRect{0.12, 0.88, 0.12, 0.88}, c);
    cv.fill(InRect{0.18, 0.82, 0.18, 0.82}, black);
    cv.fill(InRect{0.16, 0.84, 0.47, 0.53}, c);
    return;
  }
  if (key <= 48) {                                                                                                           // Door :214-236
    const Rgb c = color_rgb((key - 31) / 3);
    const uint32_t state = (key - 31) % 3;
    if (state == 0) {
      cv.fill(InRect{0.88, 1.00, 0.00, 1.00}, c);
      cv.fill(InRect{0.92, 0.96, 0.04, 0.96}, black);
    } else if (state == 2) {
      cv.fill(InRect{0.00, 1.00, 0.00, 1.00}, c);
      cv.fill(InRect{0.06, 0.94, 0.06, 0.94}, scaled(c, 0.45));
      cv.fill(InRect{0.52, 0.75, 0.50, 0.56}, c);
    } else {
      cv.fill(InRect{0.00, 1.00, 0.00, 1.00}, c);
      cv.fill(InRect{0.04, 0.96, 0.04, 0.96}, black);
      cv.fill(InRect{0.08, 0.92, 0.08, 0.92}, c);
      cv.fill(InRect{0.12, 0.88, 0.12, 0.88}, black);
      cv.fill(InCircle{0.75, 0.50, 0.08}, c);
    }
    return;
  }
  // Lava :144-157
  cv.fill(InRect{0, 1, 0, 1}, Rgb{255, 128, 0});
  for (int i = 0; i < 3; i++) {
    const double ylo = 0.3 + 0.2 * i, yhi = 0.4 + 0.2 * i;
    cv.fill(InLine(0.1, ylo, 0.3, yhi, 0.03), black);
    cv.fill(InLine(0.3, yhi, 0.5, ylo, 0.03), black);
    cv.fill(InLine(0.5, ylo, 0.7, yhi, 0.03), black);
    cv.fill(InLine(0.7, yhi, 0.9, ylo, 0.03), black);
  }
}

// Grid.render_tile (grid.py:145-198) followed by the uint8 store of Grid.render (grid.py:236): out[ts][ts][3].
// agent: 0 = no agent, 1..4 = agent_dir 0..3.
inline void render_tile(uint32_t key, int agent, bool highlight, int ts, uint8_t* out) {
  const int S = ts * TILE_SUBDIVS;
  Canvas cv(S);
  const Rgb line{100, 100, 100};
  cv.fill(InRect{0, 0.031, 0, 1}, line);
  cv.fill(InRect{0, 1, 0, 0.031}, line);
  draw_object(cv, key);
  if (agent > 0) cv.fill(InAgentTriangle(agent - 1), Rgb{255, 0, 0});
  if (highlight)                                   // highlight_img (rendering.py:130-137): img + 0.3 * (255 - img), truncated
    for (auto& p : cv.px) {
      double b = (double)p + 0.30 * (double)(uint8_t)(255 - p);
      b = b < 0 ? 0 : (b > 255 ? 255 : b);
      p = (uint8_t)b;
    }
  // downsample (rendering.py:8-22): mean over the sub-sample columns, then over the sub-sample rows, in float64
  for (int y = 0; y < ts; y++)
    for (int x = 0; x < ts; x++)
      for (int ch = 0; ch < 3; ch++) {
        double rows[TILE_SUBDIVS];
        for (int sy = 0; sy < TILE_SUBDIVS; sy++) {
          double s = 0;
          for (int sx = 0; sx < TILE_SUBDIVS; sx++) s += (double)cv.px[((size_t)(y * TILE_SUBDIVS + sy) * S + x * TILE_SUBDIVS + sx) * 3 + ch];
          rows[sy] = s / TILE_SUBDIVS;
        }
        double s = 0;
        for (int sy = 0; sy < TILE_SUBDIVS; sy++) s += rows[sy];
        out[((size_t)y * ts + x) * 3 + ch] = (uint8_t)(s / TILE_SUBDIVS);
      }
}

// out[TILE_KEYS][5][2][ts][ts][3]
inline void render_all(int ts, uint8_t* out) {
  const size_t tb = (size_t)ts * ts * 3;
  for (int k = 0; k < TILE_KEYS; k++)
    for (int ad = 0; ad < 5; ad++)
      for (int hl = 0; hl < 2; hl++) render_tile((uint32_t)k, ad, hl != 0, ts, out + (((size_t)k * 5 + ad) * 2 + hl) * tb);
}

}  // namespace tiles
}  // namespace mg
