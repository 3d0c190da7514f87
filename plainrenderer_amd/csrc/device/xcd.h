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

// The same walk with the image cut in BOTH directions: splitX (1, 2, 4 or 8) columns of XCDs, 8 / splitX rows of them; XCD k works on column k % splitX, and on
// the chunks (k / splitX), (k / splitX) + 8 / splitX, ... of that column (interleaved as above). A chunk is then splitX times narrower and, for the same number of
// chunks per XCD, splitX times taller: what an XCD's L2 fetches beyond its own tiles - the discs' reach r around every chunk - is (1 + 2r / height)(1 + 2r / width)
// of the chunk instead of 1 + 2r / height of a full-width one, and the height is what was small. splitX = 1 is xcdWalk.
inline int xcdChunkRows2(int tilesY, int chunksPerXcd, int splitX) { const int v = (8 / splitX) * chunksPerXcd; return (tilesY + v - 1) / v; }
inline int xcdBlockTilesX(int tilesX, int splitX) { return (tilesX + splitX - 1) / splitX; }
inline dim3 xcdWalkGrid2(int tilesX, int tilesY, int chunksPerXcd, int splitX) {
    return dim3((unsigned)(xcdBlockTilesX(tilesX, splitX) * xcdChunkRows2(tilesY, chunksPerXcd, splitX) * chunksPerXcd) * 8u);
}
PLR_DI bool xcdWalk2(int tilesX, int tilesY, int chunkRows, int splitX, int& tileX, int& tileY) {
    const int xcd = (int)(blockIdx.x & 7u), local = (int)(blockIdx.x >> 3);
    const int cx = xcd % splitX, ry = xcd / splitX, rowsSplit = 8 / splitX;
    const int bw = (tilesX + splitX - 1) / splitX, perChunk = bw * chunkRows;
    const int turn = local / perChunk, within = local - turn * perChunk;
    const int lx = within / chunkRows;
    tileX = cx * bw + lx;
    tileY = (turn * rowsSplit + ry) * chunkRows + (within - lx * chunkRows);
    return tileX < tilesX && tileY < tilesY;
}

} // namespace plr
