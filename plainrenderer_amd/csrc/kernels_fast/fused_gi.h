// Shared by the spatial GI filter and its producers (sdfDiffuseTrace, filterIndirectDiffuseTemporal) in the PLR_MATH_FAST set.
//
// The spatial filter gathers ONE 16-byte texel per sample: {Y_SH (4 halves), CoCg (2 halves), a quarter of the linear-depth denominator
// (float)}. On its own the filter pass fills that packed copy with a pre-pass over its inputs (13 us, 124 MB per pass at 4K). When the pass
// that PRODUCES the filter's input is recorded right before it (pass fusion, backend.h), the producer writes the packed texel next to its
// regular outputs - it has the values in registers - and the pre-pass disappears (or shrinks to the rows other GPUs sent, in band rendering).
#pragma once
#include "../backend.h"
#include "../device/shading_common.h"

namespace plr {

// packed texel of one GI pixel: ysh = the RGBA16F texel, cocg = the RG16F texel, depth = the depth-buffer value the filter's depthTexture
// holds at this texel. A texel with a NaN component (filterIndirectDiffuseSpatial.comp:118 skips it) or a non-positive denominator is
// stored as zeros with a negative denominator.
PLR_DI uint4 packGiTexel(uint2 ysh, uint32_t cocg, float depth, float nearPlane, float farPlane) {
#pragma clang fp contract(off) // the same bits whichever file (contraction on or off) the caller is compiled in: fused and unfused frames stay identical
    float den = farPlane + (1.f - depth) * (nearPlane - farPlane);
    const float probe = ((halfBitsToFloat(ysh.x & 0xffffu) + halfBitsToFloat(ysh.x >> 16)) + (halfBitsToFloat(ysh.y & 0xffffu) + halfBitsToFloat(ysh.y >> 16))) +
                        (halfBitsToFloat(cocg & 0xffffu) + halfBitsToFloat(cocg >> 16));
    if (probe != probe || !(den > 0.f)) { ysh = make_uint2(0u, 0u); cocg = 0u; den = -1.f; }
    return make_uint4(ysh.x, ysh.y, cocg, f2u(0.25f * den)); // a quarter of the denominator: the weight's numerator (negative: skip the texel)
}

// where the producer of a spatial filter pass's input has to put the packed texels (gi_spatial_fast.hip)
struct SpatialPackTarget {
    uint4* packed = nullptr; // [h][w] of the filter's input images
    ImgView depth;           // the filter's depthTexture (same texel grid as its inputs): R16F or D32
};
// Called by a producer's launcher (PassCtx::consumer, backend.h): if the consumer is a spatial filter execution that samples the producer's
// storage images outY / outC through packed texels, fills *out and returns 0; kUseGeneralKernel: no packing for this execution.
int spatialPackTargetOfConsumer(const PassCtx& producer, int outY, int outC, SpatialPackTarget* out);
// the producer has launched a kernel that writes the packed texels of rows [y0, y1) of the consumer's input: the filter's own packing
// pre-pass then covers only the rows it can read that nobody packed this frame (none in a whole-frame dispatch; the halo rows a
// neighbouring band sent, in band rendering)
void spatialNotePackedRows(const PassCtx& producer, int y0, int y1);
// the same for a rectangle of the consumer's input (tile rendering: the producer covers the columns of its tile only)
void spatialNotePackedRect(const PassCtx& producer, int x0, int y0, int x1, int y1);

} // namespace plr
