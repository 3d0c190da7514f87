// Internal interface between the backend runtime (backend.cpp) and the pass kernels (kernels/*.hip, kernels_fast/*.hip, kernels_exact/*.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstring>
#include <string>
#include <vector>

#include "device/image.h"
#include "device/types.h"

namespace plr {

constexpr int kMaxBindings = 32;

struct SpecConstant {
    uint32_t location;
    std::vector<uint8_t> data;
};

struct BufferBinding {
    void* ptr = nullptr;
    size_t size = 0;
    bool readOnly = false;
};

// Everything a pass launcher needs, with handles already resolved to HBM addresses.
struct PassCtx {
    hipStream_t stream = nullptr;
    const GlobalUbo* global = nullptr;    // set 0 binding 0 (device memory)
    // host copy of what the device buffer holds when this frame's passes run (the last setUniformBufferData of the global buffer, applied before the
    // first pass): launchers that must decide on the host from a UBO field (the screen resolution) read it here; null if the host never filled it
    const GlobalUbo* globalHost = nullptr;
    const ImgView* bindless = nullptr;    // set 2 (device array of mip-0 views, indexed by global texture index)
    const ImgView* bindlessHost = nullptr; // host copy of the same array (bindlessCount entries): a launcher that knows the index on the host - the frame's
                                          // noise texture, from globalHost - passes the view as a kernel argument instead of two dependent loads per wave
    uint32_t bindlessCount = 0;
    ImgView sampled[kMaxBindings];
    ImgView storage[kMaxBindings];
    uint32_t sampledMask = 0, storageMask = 0, sbufMask = 0, ubufMask = 0;
    BufferBinding sbuf[kMaxBindings];
    BufferBinding ubuf[kMaxBindings];
    std::vector<uint8_t> push;
    uint32_t dispatch[3] = {1, 1, 1};
    uint32_t base[3] = {0, 0, 0};         // first workgroup of the dispatch (vkCmdDispatchBase semantics; band rendering, plr.h)
    // a second range of workgroup rows covered by the same launch (pass fusion of two executions of one pass that differ in their rows only:
    // the edge rows a band renderer produces first, above and below its interior); 0 rows = none. Set by launchOverTwoRowRanges
    uint32_t extraBaseY = 0, extraCountY = 0;
    // pass fusion with elision (plr_set_pass_fusion(2)): bit b set = the image at storage binding b is touched by no execution of this frame outside the
    // fused sequence this context is part of, so a fused launcher that consumes it inside its own kernel may leave it unwritten
    // the frame's buffer fills still waiting to be applied (backend.cpp flushFills): a fused launcher registered with PLR_REGISTER_FUSION_TAKES_FILLS either lets a
    // block of its FIRST kernel apply them (applyFillsBlock; only if that kernel touches none of the destinations) and sets pendingFillsTaken, or calls
    // applyPendingFillsNow() before its first launch. Null: nothing pending.
    uint8_t* pendingFillSlot = nullptr;
    mutable bool pendingFillsTaken = false;
    int (*applyPendingFillsNow)() = nullptr;
    uint32_t elidableStorage = 0;
    mutable uint32_t elidedStorage = 0;   // set by the fused launcher: the storage bindings it really left unwritten (the backend flags those images)
    // rows to produce first + edge signal (plr.h first_rows; band rendering): workgroup rows [base[1], firstRows[0]) and [firstRows[1], base[1] + dispatch[1])
    // come first and edgeSignal is raised to edgeValue when they are complete. A launcher that orders its blocks accordingly (TwoRanges::setEdgeFirst)
    // sets edgeSignalHonoured; otherwise the backend raises the signal behind the launch. edgeSignal == nullptr: nothing to do.
    uint32_t firstRows[2] = {0, 0};
    uint32_t* edgeSignal = nullptr;
    uint32_t* edgeCounter = nullptr;
    uint32_t edgeValue = 0;
    mutable bool edgeSignalHonoured = false;
    uint32_t validRows[2] = {0, 0};       // rows of the input images that hold valid data (band rendering, plr.h); {0, 0} = all
    uint32_t validCols[2] = {0, 0};       // the same for columns (tile rendering, plr.h valid_cols); {0, 0} = all
    // workgroup columns to produce first (plr.h first_cols; tile rendering): [base[0], firstCols[0]) and [firstCols[1], base[0] + dispatch[0]), together with
    // firstRows the frame of the tile its neighbours wait for. {0, 0} = none
    uint32_t firstCols[2] = {0, 0};
    // [lo, hi) for an input image of imageH rows
    void validRowRange(int imageH, int* lo, int* hi) const {
        const bool all = validRows[0] == 0 && validRows[1] == 0;
        *lo = all ? 0 : (int)(validRows[0] < (uint32_t)imageH ? validRows[0] : (uint32_t)imageH);
        *hi = all ? imageH : (int)(validRows[1] < (uint32_t)imageH ? validRows[1] : (uint32_t)imageH);
    }
    // [lo, hi) for an input image of imageW columns
    void validColRange(int imageW, int* lo, int* hi) const {
        const bool all = validCols[0] == 0 && validCols[1] == 0;
        *lo = all ? 0 : (int)(validCols[0] < (uint32_t)imageW ? validCols[0] : (uint32_t)imageW);
        *hi = all ? imageW : (int)(validCols[1] < (uint32_t)imageW ? validCols[1] : (uint32_t)imageW);
    }
    // producer -> consumer link (PLR_REGISTER_CONSUMER_LINK below): the context of the LATER execution of this frame that is the first to read an
    // image this execution writes, possibly with host callbacks (halo exchanges) and further executions of this same pass in between; null if
    // there is none, if another pass writes the image first, in PLR_MATH_EXACT, with pass fusion off or a signature buffer set
    const PassCtx* consumer = nullptr;
    // set by the backend on the contexts of a fused sequence whose EARLY PART (EarlyPart below) it has launched in this frame: the fused launcher issues the rest only
    mutable bool earlyPartDone = false;
    uint64_t frameSerial = 0;             // serial of this plr_render_frame call, unique in the process: lets a pass's host-side bookkeeping tell this frame's entries from stale ones
    const std::vector<SpecConstant>* spec = nullptr;
    std::string* err = nullptr;
    void** scratchSlot = nullptr;         // persistent per-pass scratch (device memory, grow-only)
    size_t* scratchSize = nullptr;
    // decision signatures (plr_debug_set_decision_signature, plr.h): when set, the kernels that support it write one word per output pixel
    uint32_t* debugSig = nullptr;
    size_t debugSigWords = 0;
    // the frame's noise texture (noise.inc: global texture noiseTextureIndices[frameIndexMod4]) resolved on the host; false if the host does not know
    // the global buffer's contents or the texture table (the kernel then chases the three pointers itself)
    bool hostNoiseView(ImgView* out) const {
        if (!globalHost || !bindlessHost || bindlessCount == 0) return false;
        const uint32_t slot = (uint32_t)globalHost->noiseTextureIndices[globalHost->frameIndexMod4 & 3u];
        *out = bindlessHost[slot < bindlessCount - 1u ? slot : bindlessCount - 1u];
        return out->ptr != nullptr && out->w > 0 && out->h > 0;
    }
    uint32_t* sigFor(size_t pixels) const { return debugSig && debugSigWords >= pixels ? debugSig : nullptr; }

    bool hasSampled(int b) const { return (sampledMask >> b) & 1u; }
    bool hasStorage(int b) const { return (storageMask >> b) & 1u; }
    bool hasSbuf(int b) const { return (sbufMask >> b) & 1u; }
    bool hasUbuf(int b) const { return (ubufMask >> b) & 1u; }

    // spec constants are raw bytes with the C++ sizeof of the host type (bool = 1 byte)
    const SpecConstant* findSpec(uint32_t location) const;
    int32_t specInt(uint32_t location, int32_t def) const;
    uint32_t specUint(uint32_t location, uint32_t def) const { return (uint32_t)specInt(location, (int32_t)def); }
    float specFloat(uint32_t location, float def) const;
    bool specBool(uint32_t location, bool def) const;

    // pixel rows [y0, y1) covered by the recorded dispatch for workgroups of wgRows rows, clipped to an image of imageH rows
    struct RowSpan { int y0, y1; };
    RowSpan rowSpan(int imageH, int wgRows = 8) const {
        const long long a = (long long)base[1] * wgRows, b = a + (long long)dispatch[1] * wgRows;
        RowSpan r;
        r.y0 = (int)(a < imageH ? a : imageH);
        r.y1 = (int)(b < imageH ? b : imageH);
        return r;
    }

    // pixel columns [x0, x1) covered by the recorded dispatch for workgroups of wgCols columns, clipped to an image of imageW columns (tile rendering:
    // dispatch_base[0] / dispatch_count[0] restrict a pass to the columns of its tile, as [1] does for rows)
    struct ColSpan { int x0, x1; };
    ColSpan colSpan(int imageW, int wgCols = 8) const {
        const long long a = (long long)base[0] * wgCols, b = a + (long long)dispatch[0] * wgCols;
        ColSpan r;
        r.x0 = (int)(a < imageW ? a : imageW);
        r.x1 = (int)(b < imageW ? b : imageW);
        return r;
    }
    // the recorded dispatch covers every column of an image of imageW columns
    bool wholeRows(int imageW, int wgCols = 8) const { const ColSpan s = colSpan(imageW, wgCols); return s.x0 == 0 && s.x1 == imageW; }

    int fail(int code, const std::string& msg) const;
    // checks presence + format of a binding; returns 0 or records an error
    int needSampled(int binding, int fmt, const char* what) const;
    int needStorage(int binding, int fmt, const char* what) const;
    int needSbuf(int binding, size_t minSize, const char* what) const;
    int needUbuf(int binding, size_t minSize, const char* what) const;
    int needGlobal() const;
    void* scratch(size_t bytes) const;
    // pass timing (plr_set_pass_timing): a pass that launches an auxiliary kernel before its main one calls this between the two, so the
    // auxiliary part is reported as its own entry "<pass name> (<label>)" and the pass entry times the main kernel alone
    void splitTiming(const char* label) const;
    // a fused launcher (pass fusion) that covers its passes with one kernel EACH calls this between two of them: the running timing segment is
    // closed under the first pass's own name and a new one opened under the second's, so per-pass timings stay per pass
    static void splitTimingBetween(const PassCtx& first, const PassCtx& second);
    const char* passName = nullptr; // debug label of the pass (plr_get_renderpass_timings)
};

typedef int (*LaunchFn)(const PassCtx&);
// fused launcher body for "the same pass twice, over two row ranges": if the two executions bind the same resources and the second one's rows lie
// below the first one's, `single` is called once with the second range in extraBaseY / extraCountY; kUseGeneralKernel otherwise (two launches)
int launchOverTwoRowRanges(const PassCtx* const* ctxs, size_t count, LaunchFn single);
// the remap a kernel applies to its block row when a launch covers two row ranges: rows of the second range start `gap` block rows further down
// ... or, exclusively, when the launch produces its EDGE rows first (PassCtx::firstRows): block rows [0, edgeTop) and the last edgeBottom of `total`
// are taken by the first edgeTop + edgeBottom block rows of the grid, the interior by the rest; every wave of an edge block reports in (edgeDone) and
// the last one raises the signal - while the interior blocks of the same launch are still running.
struct TwoRanges {
    int split = 0x7fffffff, gap = 0;
    int edgeTop = 0, edgeBottom = 0, total = 0;
    // tile rendering (plr.h first_cols): the first edgeLeft and the last edgeRight block columns of the block rows between the top and the bottom edge belong
    // to the edge as well - the frame of the tile. The grid is walked as a linear list L = blockIdx.y * gridDim.x + blockIdx.x: top rows, bottom rows, left
    // columns, right columns, interior (blockXY). With no edge columns that is the order of blockRow.
    int edgeLeft = 0, edgeRight = 0;
    uint32_t edgeBlocks = 0; // blocks of the edge = arrivals before the signal (setEdgeFirst)
    uint32_t* edgeCounter = nullptr;
    uint32_t* edgeSignal = nullptr;
    uint32_t edgeValue = 0;
    // block row of the image region this block of the grid works on
    __device__ __forceinline__ int blockRow(int r) const {
        if (edgeTop + edgeBottom == 0) return r + (r >= split ? gap : 0);
        if (r < edgeTop) return r;
        if (r < edgeTop + edgeBottom) return total - edgeBottom + (r - edgeTop);
        return r - edgeBottom;
    }
    // block column and block row of the image region this block of the grid works on (kernels that can run the frame of a tile first; gridDim.x block
    // columns, `total` block rows)
    __device__ __forceinline__ void blockXY(int* bx, int* by) const {
        if (edgeLeft + edgeRight == 0) { *bx = (int)blockIdx.x; *by = blockRow((int)blockIdx.y); return; }
        const int BX = (int)gridDim.x, mid = total - edgeTop - edgeBottom;
        int L = (int)blockIdx.y * BX + (int)blockIdx.x;
        if (L < edgeTop * BX) { *by = L / BX; *bx = L - *by * BX; return; }
        L -= edgeTop * BX;
        if (L < edgeBottom * BX) { const int q = L / BX; *by = total - edgeBottom + q; *bx = L - q * BX; return; }
        L -= edgeBottom * BX;
        if (L < edgeLeft * mid) { const int q = L / edgeLeft; *by = edgeTop + q; *bx = L - q * edgeLeft; return; }
        L -= edgeLeft * mid;
        if (L < edgeRight * mid) { const int q = L / edgeRight; *by = edgeTop + q; *bx = BX - edgeRight + (L - q * edgeRight); return; }
        L -= edgeRight * mid;
        const int iw = BX - edgeLeft - edgeRight, q = L / iw;
        *by = edgeTop + q; *bx = edgeLeft + (L - q * iw);
    }
    __device__ __forceinline__ bool isEdge(int r) const { return edgeLeft + edgeRight == 0 ? r < edgeTop + edgeBottom : (uint32_t)r * gridDim.x + blockIdx.x < edgeBlocks; }
    // edgeDone: called once by EVERY wave of the block at the end of its work (all its stores issued), on every path out of the kernel. An edge block's outputs
    // are stored WRITE-THROUGH (storeOut below with through = isEdge(): `sc1` stores reach memory, MI355X_MICROARCH.md "stores of each flavour"), so a
    // wave only has to wait for its own stores and arrive; an agent-scope release per wave - a write-back of the XCD's whole L2 each time - made a
    // band's trace take 2.5 ms (measured, round 4). The last arrival of the launch resets the counter for the next launch and raises the signal at
    // system scope (the command processor polls it: hipStreamWaitValue32); the exchange's send kernel starts after that, with clean caches.
    // Arrivals are per BLOCK (the block's waves meet at a barrier: every wave calls this exactly once) and sharded over kEdgeShards counters on separate
    // cache lines by the block's place in the list, each of which forwards one arrival to the top counter when its last block is in: one word takes ~88
    // atomics per microsecond (MI355X_MICROARCH.md, dequeue), and 15 000 edge waves of a band arriving on ONE word made the pass 130 us longer (measured, round 4).
    static constexpr uint32_t kEdgeShards = 32, kEdgeShardStride = 16; // counters[(1 + shard) * 16], counters[0] = top
    __device__ __forceinline__ void edgeDone(int r) const {
        if (edgeBlocks == 0) return;
        const uint32_t L = (uint32_t)r * gridDim.x + blockIdx.x; // (r = blockIdx.y: the block's place in the grid, not its block row)
        if (L >= edgeBlocks) return;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t shard = L % kEdgeShards;
            const uint32_t inShard = edgeBlocks / kEdgeShards + (shard < edgeBlocks % kEdgeShards ? 1u : 0u);
            uint32_t* mine = edgeCounter + (1u + shard) * kEdgeShardStride;
            if (__hip_atomic_fetch_add(mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == inShard - 1u) {
                __hip_atomic_store(mine, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t shards = edgeBlocks < kEdgeShards ? edgeBlocks : kEdgeShards;
                if (__hip_atomic_fetch_add(edgeCounter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == shards - 1u) {
                    __hip_atomic_store(edgeCounter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(edgeSignal, edgeValue, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
    }
    // host: turn PassCtx::firstRows / firstCols into the edge-first order for blocks of blockRowsPx x blockColsPx pixels over pixel rows [y0, y1) and pixel
    // columns [x0, x1) (wgRows / wgCols pixels per workgroup row / column), blocksX blocks per row (= gridDim.x of the launch); false (and nothing set) if the
    // edges are not whole blocks - the backend then signals itself
    bool setEdgeFirst(const struct PassCtx& c, int y0, int y1, int blockRowsPx, int wgRows, unsigned blocksX, int x0 = 0, int x1 = 0, int blockColsPx = 0, int wgCols = 8);
};
// output stores of a kernel that may run rows-first: plain, or write-through (relaxed agent-scope atomic stores = `global_store ... sc1`) for an edge block
__device__ __forceinline__ void storeOut(uint32_t* p, uint32_t v, bool through) {
    if (through) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}
__device__ __forceinline__ void storeOut(uint2* p, uint2 v, bool through) {
    if (through) __hip_atomic_store((uint64_t*)p, (uint64_t)v.x | ((uint64_t)v.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}
__device__ __forceinline__ void storeOut(uint4* p, uint4 v, bool through) {
    if (through) {
        __hip_atomic_store((uint64_t*)p, (uint64_t)v.x | ((uint64_t)v.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store((uint64_t*)p + 1, (uint64_t)v.z | ((uint64_t)v.w << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else *p = v;
}
// block rows of the two ranges for blocks of blockRows pixel rows: 0 = fine (*blocks = total block rows, *end = end row of the launch), else not expressible
int twoRangeBlocks(const PassCtx& c, int imageH, int blockRows, int wgRows, TwoRanges* out, int* blocks, int* y0, int* end);
// a PLR_MATH_FAST launcher returns this when the recorded execution is outside the configuration its kernel was built for;
// the backend then runs the general (exact-order) kernel of the same shader
constexpr int kUseGeneralKernel = 1;

struct ShaderRegistrar {
    ShaderRegistrar(const char* name, LaunchFn fn, bool fast = false);
};
#define PLR_REGISTER_SHADER(name, fn) static ::plr::ShaderRegistrar plr_registrar_##fn(name, fn)
// restructured / fast-math variant of a shader, used when the math mode is PLR_MATH_FAST (kernels_fast/*.hip)
#define PLR_REGISTER_SHADER_FAST(name, fn) static ::plr::ShaderRegistrar plr_registrar_fast_##fn(name, fn, true)

// Fusion of ADJACENT recorded executions (PLR_MATH_FAST only; plr_set_pass_fusion). The boundary stays one setComputePassExecution per
// reference dispatch; when the recorded sequence contains the shaders of a fused launcher back to back (no host callback in between), the
// backend hands all their contexts to it and it covers them with fewer kernels. A launcher returns kUseGeneralKernel when the bindings are
// not what it was built for (the executions then run one by one), 0 when it launched everything, < 0 on error. Results must not depend on
// whether a sequence was fused (tests/test_fusion.py compares bytes).
typedef int (*FusedLaunchFn)(const PassCtx* const* ctxs, size_t count);
// The frame's buffer fills as a table in pinned host memory (backend.cpp flushFills): {count, padding, done} header, entries, payloads. One block applies it.
struct FillEntry { uint64_t dst; uint32_t srcOffset, size; };
constexpr size_t kFillTableHeader = 16; // uint32 count, padding, uint64 done (written by the kernel: the serial of the fill it has consumed)
#ifdef __HIPCC__
__device__ inline void applyFillsBlock(uint8_t* __restrict__ slot, uint64_t serial) {
    const uint32_t count = *(const uint32_t*)slot;
    const FillEntry* entries = (const FillEntry*)(slot + kFillTableHeader);
    for (uint32_t i = 0; i < count; i++) {
        const FillEntry f = entries[i];
        uint8_t* dst = (uint8_t*)f.dst;
        const uint8_t* src = slot + f.srcOffset;
        if ((((uint32_t)f.dst | f.srcOffset | f.size) & 3u) == 0u)
            for (uint32_t w = threadIdx.x; w < f.size / 4u; w += blockDim.x) ((uint32_t*)dst)[w] = ((const uint32_t*)src)[w];
        else
            for (uint32_t k = threadIdx.x; k < f.size; k += blockDim.x) dst[k] = src[k];
        __syncthreads(); // call order: a later fill of the same bytes wins
    }
    // every read of the slot is done (the barrier above): tell the host it may re-use it
    // (relaxed: the slot was only READ, and every value read has been consumed by a store above - a release here is a system-scope write-back of the
    //  XCD's L2 in front of every frame, measured at +7 us per frame)
    if (threadIdx.x == 0) __hip_atomic_store((uint64_t*)(slot + 8), serial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
#endif

struct FusionRegistrar {
    // writesSignatures: the fused kernels write the decision signatures of all their passes (plr_debug_set_decision_signature) - such a sequence
    // stays fused while a signature buffer is set; all others run pass by pass then
    FusionRegistrar(const char* label, std::initializer_list<const char*> shaders, FusedLaunchFn fn, bool writesSignatures = false, bool takesFills = false);
};
#define PLR_REGISTER_FUSION(label, fn, ...) static ::plr::FusionRegistrar plr_fusion_##fn(label, {__VA_ARGS__}, fn)
#define PLR_REGISTER_FUSION_WITH_SIGNATURES(label, fn, ...) static ::plr::FusionRegistrar plr_fusion_##fn(label, {__VA_ARGS__}, fn, true)
// the launcher handles PassCtx::pendingFillSlot of its first execution (see there)
#define PLR_REGISTER_FUSION_TAKES_FILLS(label, fn, ...) static ::plr::FusionRegistrar plr_fusion_##fn(label, {__VA_ARGS__}, fn, false, true)

// EARLY PART of a fused sequence (round 6; plr_set_early_parts). A fused launcher may split off the work of its passes that depends on none of the frame's
// intermediate results - the deferred shade's direct lighting needs the G-buffer, the shadow cascades and the LUTs, but nothing of the GI chain recorded in front
// of it - as a launch of its own that the backend issues on the EARLY STREAM as soon as the recorded resources allow: behind the last execution of the frame (or
// pending buffer fill) that writes anything the early kernels read, beside everything recorded between that point and the sequence itself. The boundary does not
// change: the caller records the same executions in the same order; what changes is when a part of one of them starts.
//   Query:  the launcher validates the bindings (no launch, no allocation) and lists the allocations its early kernels READ (image mip-0 addresses, buffer
//           addresses; the global uniform block must be taken from PassCtx::globalHost, by value - the device copy is filled by the frame's first launch).
//           Returns 0, or kUseGeneralKernel: no early part for these bindings (the sequence runs as one fused launch, as without this feature).
//   Launch: PassCtx::stream of the contexts is the early stream. The early kernels write nothing but the pass's scratch memory.
// When the sequence's turn comes, the launch stream waits for the early part and the fused launcher is called with PassCtx::earlyPartDone set on its contexts.
struct EarlyPart {
    enum Mode { Query, Launch } mode = Query;
    std::vector<const void*>* reads = nullptr;
};
typedef int (*EarlyLaunchFn)(const PassCtx* const* ctxs, size_t count, EarlyPart& part);
struct EarlyPartRegistrar {
    EarlyPartRegistrar(FusedLaunchFn fused, EarlyLaunchFn early, const char* label);
};
// after the PLR_REGISTER_FUSION* of `fusedFn`, in the same translation unit
#define PLR_REGISTER_EARLY_PART(fusedFn, earlyFn, label) static ::plr::EarlyPartRegistrar plr_early_part_##earlyFn(fusedFn, earlyFn, label)

// Fusion ACROSS the recorded order: when an execution of `producer` is launched, PassCtx::consumer points at the first later execution of
// `consumer` that samples one of its storage images. The producer's fast launcher may then do part of the consumer's work on the rows it
// produces (sdfDiffuseTrace / filterIndirectDiffuseTemporal write the spatial filter's packed texels, fused_gi.h) - this also works in band
// rendering, where exchange callbacks separate the two and the producer runs as several executions (edge rows, interior rows).
struct ConsumerLinkRegistrar {
    ConsumerLinkRegistrar(const char* producerShader, const char* consumerShader);
};
#define PLR_REGISTER_CONSUMER_LINK(id, producer, consumer) static ::plr::ConsumerLinkRegistrar plr_consumer_link_##id(producer, consumer)
// a launcher that did (part of) another pass's work / found its own pre-pass already done reports it: plr_get_pass_fusion's counter
void countFusedExecutions(uint32_t n);

// Content version of an image allocation (allocationBase = the address of its mip 0), for launchers that keep DERIVED copies of an input in their
// scratch memory (the deferred shade: PCF tap table by noise-texel position, kernels_fast/shading_fast.hip). The value changes whenever the backend
// knows the contents may have changed: creation / resize, plr_upload_image*, plr_write / copy_device_memory into it, an execution or a host callback
// that binds it for writing (counted when that execution is launched, i.e. in recorded order), a host callback recorded without a resource list
// (then every version changes). 0 = do not cache: the allocation's address was handed out (plr_get_image_device_pointer) and may be written
// behind the backend's back, or the address is not an image of this backend.
uint64_t contentVersionOf(const void* allocationBase);

inline unsigned divUp(unsigned a, unsigned b) { return (a + b - 1) / b; }

// records the message plr_last_error() returns on this thread; returns code
int setLastError(int code, const std::string& msg);

#define PLR_CHECK_LAUNCH(ctx)                                                                   \
    do {                                                                                        \
        hipError_t e_ = hipGetLastError();                                                      \
        if (e_ != hipSuccess) return (ctx).fail(-2, std::string("kernel launch failed: ") + hipGetErrorString(e_)); \
    } while (0)

} // namespace plr
