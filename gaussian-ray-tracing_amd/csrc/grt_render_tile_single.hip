// grt_render_tile_single.hip — the tile kernel's one-ray-per-wave mode (MODE 2: the rays of mesh frames whose bundle gave
// up, and the retry queue of k_bounce) as a translation unit of its own: the same source as grt_render_tile.hip, compiled
// with an 8-key window (each lane's window cells carry the events' radiance in LDS in this mode: 14 KB per wave instead of
// 19 KB) for 3 waves per SIMD, and a resident grid to match (256 CUs x 11).  See the note above launch_render_tile_single.
#define GRT_TILE_SINGLE_TU 1
#define GRT_TILE_KS 8
#define GRT_TILE_WAVES2 3
#ifndef GRT_TILE_GRID2
#define GRT_TILE_GRID2 2816u
#endif
#include "grt_render_tile.hip"
