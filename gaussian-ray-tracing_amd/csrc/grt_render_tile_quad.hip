// grt_render_tile_quad.hip — the tile kernel's quad mode (MODE 3: one 4x4 quadrant of a heavy tile per wave, lanes = rays x slots)
// as a translation unit of its own: the same source as grt_render_tile.hip, its four instantiations compiled for 3 waves per SIMD
// (the exact test holds its particle's record per lane).  See the note at `constexpr bool QUAD` and launch_render_tile_quad.
#define GRT_TILE_QUAD_TU 1
#include "grt_render_tile.hip"
