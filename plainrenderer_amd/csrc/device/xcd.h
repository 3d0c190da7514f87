// Block-to-tile mapping for kernels whose neighbouring tiles share most of what they read (the GI spatial filter's ~100-pixel disc).
//
// The dispatcher hands work-groups to the 8 XCDs round-robin by linear id (block b -> XCD b % 8) and every XCD has its own 4 MB L2. With the
// natural mapping the blocks of one tile row are spread over all eight L2s, so each L2 ends up fetching the whole image. xcdWalk cuts the
// tile rows into 8 * chunksPerXcd horizontal chunks, gives XCD k the chunks k, k + 8, ... (interleaved, so that cheap sky rows and expensive
// geometry rows are spread over the XCDs) and walks a chunk column by column: the tiles in flight on an XCD form a compact block whose
// footprint does not grow with the image width. Launch with xcdWalkGrid(...) blocks (one-dimensional).
// Measured on kernels with SMALL footprints (deferred shading, TAA, SDF trace) the same mapping is neutral to harmful (one chunk per XCD:
// shading +33 %, trace +20 % from the load imbalance between bands), and on the streaming passes whose tiles share only a halo row or two (GI
// upscale 51 -> 55 us, temporal GI filter 43.4 -> 44.5 us with 2 chunks per XCD): they keep the natural mapping.
#pragma once
#include <hip/hip_runtime.h>
#include "types.h"

namespace plr {

inline int xcdChunkRows(int tilesY, int chunksPerXcd) { return (tilesY + 8 * chunksPerXcd - 1) / (8 * chunksPerXcd); }
inline dim3 xcdWalkGrid(int tilesX, int tilesY, int chunksPerXcd) { return dim3((unsigned)(tilesX * xcdChunkRows(tilesY, chunksPerXcd) * chunksPerXcd) * 8u); }

// tile of this block; false: the block is padding. chunkRows = xcdChunkRows(tilesY, chunksPerXcd)
PLR_DI bool xcdWalk(int tilesX, int tilesY, int chunkRows, int& tileX, int& tileY) {
    const int local = (int)(blockIdx.x >> 3), perChunk = tilesX * chunkRows;
    const int turn = local / perChunk, within = local - turn * perChunk;
    tileX = within / chunkRows;
    tileY = (turn * 8 + (int)(blockIdx.x & 7u)) * chunkRows + (within - tileX * chunkRows);
    return tileY < tilesY;
}

} // namespace plr
