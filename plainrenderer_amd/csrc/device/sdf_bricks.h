// SDF volumes in cache-line bricks (north star: SDF bricks; SDF.inc:101-184 is what reads them). Built, measured, and NOT the default: profiles/r04_not_kept.txt (3).
//
// A 64^3 R16F volume in its image layout has one 128-byte cache line per (y, z) row: the 2 x 2 x 2 texels of a trilinear fetch always lie in FOUR lines, and
// the 64 diverging rays of a wave touch ~16 distinct lines per load instruction (profiles/r03_trace_counters.txt). The brick variant of the fast trace
// (PLR_TRACE_BRICKS=1) marches through a re-laid copy of every volume in which a cache line is a BRICK of 7 x 4 x 2 cells:
//   * 8 texels in x: the brick's 7 cells plus the first texel of the next brick (stored twice), so the two x neighbours of a cell are always adjacent in ONE
//     brick and still come with one 32-bit load;
//   * 4 rows in y, 2 slices in z: a cell's four (y, z) rows lie in 1.9 lines on average instead of 4.
// Texel (x, y, z) of cell column bx = x / 7 lives at
//   (((z >> 1) * nby + (y >> 2)) * nbx + bx) * 64 + ((z & 1) * 4 + (y & 3)) * 8 + (x - 7 bx)         [in texels; a brick is 64 texels = 128 bytes]
// The copy holds the same half floats: results do not change by a bit (tests/test_sdfgi.py). 589 KB instead of 512 KB per 64^3 volume.
// What the measurement said: the four row loads of a fetch miss in parallel, so halving the lines per fetch does not shorten a march step; the trace waits for
// the length of its chain of dependent steps.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace plr {

constexpr int kBrickedFormatFlag = 0x4000; // or-ed into the fmt field of a volume's ImgView as staged by the trace kernel: ptr is the bricked copy

struct BrickGrid { int nbx, nby, nbz; };
__host__ __device__ inline BrickGrid brickGrid(int w, int h, int d) { return {(w - 1 + 6) / 7 > 0 ? (w - 1 + 6) / 7 : 1, (h + 3) / 4, (d + 1) / 2}; }
__host__ __device__ inline size_t brickedTexelCount(int w, int h, int d) { const BrickGrid g = brickGrid(w, h, d); return (size_t)g.nbx * g.nby * g.nbz * 64u; }
// x / 7 for 0 <= x < 8192 without a division
__host__ __device__ inline int div7(int x) { return (x * 9363) >> 16; }

} // namespace plr
