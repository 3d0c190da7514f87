// Halo exchange of band / tile rendering over RCCL, in the C++ host ("RCCL over xGMI only for the TAA / denoise / bloom halo rows").
//
// One process per GPU renders one rectangle of the screen: a band of rows, or - BASELINE config 5 - a tile of a 2 x 2 grid (frame_pipeline.h, BandSettings).
// Where a pass reads texels a neighbouring rectangle produced, FramePipeline calls its exchange callback from inside plr_render_frame, in pass order.
// This file IS that callback when plrf_rccl_attach*() has been called. There is no Python and no torch in the frame loop; the launcher only hands every
// rank the ncclUniqueId once at start-up.
//
// Bands (whole rows): every exchanged region is one contiguous range of an image, so an exchange is a group of ncclSend / ncclRecv straight from / into the
// images with the band above and the band below. Tiles: a region is a rectangle (a strided range), so an exchange is ONE pack kernel that gathers every region
// of every image of the exchange point for all peers into a staging arena, one group with one ncclSend + one ncclRecv per peer, and ONE unpack kernel. A tile
// exchanges with every tile it touches, corners included: on a node every pair of GPUs has its own xGMI link, so the diagonal neighbour is a third peer of the
// same group, not a second phase. The luminance histogram is one 512-byte ncclAllReduce, the depth range two one-float all-reduces.
//
// Ordering. Exchanges with a BEGIN / END phase (band_overlap_exchange) run on a communication stream: BEGIN makes it wait for the producer's edge signal (rows /
// tile frame first, plr.h first_rows) or for an event behind the producer and posts pack + group + unpack there; the producer's interior runs beside them; END
// makes the launch stream wait for the completion event. Exchanges without a phase and the all-reduces are enqueued on the launch stream itself.
//
// Watchdog. An exchange whose peer never posts its side parks the launch stream for ever. Every BEGIN (and every in-line exchange) arms an entry
// {rank, exchange, phase, completion event, time}; overdue entries fail the next exchange callback with a message, and a background thread prints the message
// and aborts the process (a posted collective cannot be cancelled) - plr_frame.h "exchange watchdog".
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include "../../../include/plr_frame.h"
#include "frame_pipeline.h"

using namespace plrhost;

namespace {

thread_local std::string g_xerr;
struct LocalGroup;
thread_local LocalGroup* g_pendingLocalGroup = nullptr;
int xfail(int code, const std::string& msg) { g_xerr = msg; return code; }

constexpr uint32_t kBandAlignment = 64; // rows / columns; edges never split a HiZ / culling / histogram tile or the coarsest bloom texel

// rows [begin, end) of band `index` of `nBands`: multiples of 64 rows, sizes differing by at most 64 (the last band ends at `height`)
void bandRowsOf(uint32_t height, uint32_t nBands, uint32_t index, uint32_t* begin, uint32_t* end) {
    const uint32_t tiles = (height + kBandAlignment - 1) / kBandAlignment;
    const uint32_t base = tiles / nBands, extra = tiles % nBands;
    const uint32_t b = index * base + std::min(index, extra);
    const uint32_t e = b + base + (index < extra ? 1u : 0u);
    *begin = b * kBandAlignment;
    *end = std::min(e * kBandAlignment, height);
}

// rows of band b in an image of imageRows rows that shows the frame at 1 / divisor resolution. bounds: nBands + 1 row boundaries of a
// partition chosen by the caller (load balancing: bands of unequal height), or null for the equal partition of bandRowsOf
void bandRowsInImage(uint32_t frameHeight, uint32_t nBands, const uint32_t* bounds, uint32_t b, uint32_t imageRows, uint32_t* begin, uint32_t* end) {
    const uint32_t divisor = std::max(1u, (frameHeight + imageRows / 2) / std::max(imageRows, 1u));
    uint32_t b0, b1;
    if (bounds) { b0 = bounds[b]; b1 = bounds[b + 1]; }
    else bandRowsOf(frameHeight, nBands, b, &b0, &b1);
    *begin = b0 / divisor;
    *end = std::min((b1 + divisor - 1) / divisor, imageRows);
}

// The transfers of one exchange item for band `band`: it sends its first / last haloRows owned rows up / down and receives the rows
// just above / below its own from the neighbours. A halo taller than the neighbouring band is clipped to that band (rows further
// away belong to the band after it and are not exchanged: stated limit halo <= height of the neighbouring band).
uint32_t planItem(uint32_t frameHeight, uint32_t nBands, const uint32_t* bounds, uint32_t band, uint32_t imageRows, uint32_t haloRows, uint32_t rowBegin, uint32_t rowEnd,
                  plrf_exchange_op* ops) {
    uint32_t n = 0;
    auto add = [&](uint32_t peer, uint32_t send, uint32_t a, uint32_t b) {
        if (b > a) { ops[n].peer = peer; ops[n].send = send; ops[n].row_begin = a; ops[n].row_end = b; n++; }
    };
    if (band > 0) {
        uint32_t ub, ue;
        bandRowsInImage(frameHeight, nBands, bounds, band - 1, imageRows, &ub, &ue);
        add(band - 1, 1, rowBegin, std::min(rowBegin + haloRows, rowEnd));
        add(band - 1, 0, std::max(rowBegin > haloRows ? rowBegin - haloRows : 0u, ub), rowBegin);
    }
    if (band + 1 < nBands) {
        uint32_t db, de;
        bandRowsInImage(frameHeight, nBands, bounds, band + 1, imageRows, &db, &de);
        add(band + 1, 1, std::max(rowEnd > haloRows ? rowEnd - haloRows : 0u, rowBegin), rowEnd);
        add(band + 1, 0, rowEnd, std::min(std::min(rowEnd + haloRows, imageRows), de));
    }
    return n;
}

// What the exchange itself moves for a band: planItem's transfers, and - when a halo is taller than a neighbouring band (the exact mode's whole-image halo) -
// the rows of the bands behind it too: with every band p the rows of the own band within haloRows of p's are sent, p's rows within haloRows of the own band
// are received. For a halo no taller than the neighbours this is planItem, transfer for transfer.
std::vector<plrf_exchange_op> planBandRows(uint32_t frameHeight, uint32_t nBands, const uint32_t* bounds, uint32_t band, uint32_t imageRows, uint32_t haloRows, uint32_t rowBegin,
                                           uint32_t rowEnd) {
    std::vector<plrf_exchange_op> ops;
    auto add = [&](uint32_t peer, uint32_t send, uint32_t a, uint32_t b) { if (b > a) ops.push_back({peer, send, a, b}); };
    const uint32_t lo = rowBegin > haloRows ? rowBegin - haloRows : 0u, hi = (uint32_t)std::min<uint64_t>((uint64_t)rowEnd + haloRows, imageRows);
    for (uint32_t p = 0; p < nBands; p++) {
        if (p == band) continue;
        uint32_t tb, te;
        bandRowsInImage(frameHeight, nBands, bounds, p, imageRows, &tb, &te);
        if (p < band) te = std::min(te, rowBegin); // (a reduced image's shared boundary row belongs to the band below it, as in planItem)
        else tb = std::max(tb, rowEnd);
        const uint32_t plo = tb > haloRows ? tb - haloRows : 0u, phi = (uint32_t)std::min<uint64_t>((uint64_t)te + haloRows, imageRows);
        add(p, 1, std::max(rowBegin, plo), std::min(rowEnd, phi));
        add(p, 0, std::max(tb, lo), std::min(te, hi));
    }
    return ops;
}

// ---------------------------------------------------------------- rectangles (tile rendering)
struct Rect { uint32_t x0, y0, x1, y1; };
bool emptyRect(const Rect& r) { return r.x1 <= r.x0 || r.y1 <= r.y0; }
Rect intersect(const Rect& a, const Rect& b) { return {std::max(a.x0, b.x0), std::max(a.y0, b.y0), std::min(a.x1, b.x1), std::min(a.y1, b.y1)}; }
// (the sums in 64 bits: PLRF_HALO_WHOLE_IMAGE = 0xffffffff must saturate, not wrap - plrf_exchange_plan_rects is public and does not clamp its halo)
Rect grow(const Rect& r, uint32_t k, uint32_t w, uint32_t h) {
    return {r.x0 > k ? r.x0 - k : 0u, r.y0 > k ? r.y0 - k : 0u, (uint32_t)std::min<uint64_t>((uint64_t)r.x1 + k, w), (uint32_t)std::min<uint64_t>((uint64_t)r.y1 + k, h)};
}
// the rectangle of a full-resolution rectangle in an image of cols x rows texels showing the frame at 1 / divisor resolution (begin rounds down, end up:
// the same rule as bandRowsInImage)
Rect scaleRect(const Rect& r, uint32_t frameW, uint32_t frameH, uint32_t cols, uint32_t rows) {
    const uint32_t dx = std::max(1u, (frameW + cols / 2) / std::max(cols, 1u)), dy = std::max(1u, (frameH + rows / 2) / std::max(rows, 1u));
    return {r.x0 / dx, r.y0 / dy, std::min((r.x1 + dx - 1) / dx, cols), std::min((r.y1 + dy - 1) / dy, rows)};
}
uint32_t planRects(uint32_t frameW, uint32_t frameH, uint32_t world, const Rect* rects, uint32_t rank, uint32_t cols, uint32_t rows, uint32_t halo, plrf_rect_op* ops, uint32_t capacity) {
    uint32_t n = 0;
    const Rect mine = scaleRect(rects[rank], frameW, frameH, cols, rows);
    auto add = [&](uint32_t peer, uint32_t send, const Rect& r) {
        if (emptyRect(r)) return;
        if (n < capacity) ops[n] = {peer, send, r.x0, r.y0, r.x1, r.y1};
        n++;
    };
    for (uint32_t p = 0; p < world; p++) {
        if (p == rank) continue; // every rank within the halo's reach is a peer: the tiles that touch (edge or corner) and, when a halo is wider than a neighbour, the tile behind it
        const Rect theirs = scaleRect(rects[p], frameW, frameH, cols, rows);
        add(p, 1, intersect(mine, grow(theirs, halo, cols, rows)));
        add(p, 0, intersect(theirs, grow(mine, halo, cols, rows)));
    }
    return n;
}

int checkRects(uint32_t frameW, uint32_t frameH, uint32_t world, const uint32_t* rects, const char* who) {
    if (!rects || world == 0) return xfail(PLR_ERR_INVALID_ARGUMENT, std::string(who) + ": no rectangles");
    uint64_t area = 0;
    for (uint32_t r = 0; r < world; r++) {
        const uint32_t* q = rects + 4 * r;
        if (q[2] <= q[0] || q[3] <= q[1] || q[2] > frameW || q[3] > frameH) return xfail(PLR_ERR_INVALID_ARGUMENT, std::string(who) + ": rectangle " + std::to_string(r) + " is empty or outside the frame");
        if (q[0] % kBandAlignment || q[1] % kBandAlignment || (q[2] % kBandAlignment && q[2] != frameW) || (q[3] % kBandAlignment && q[3] != frameH))
            return xfail(PLR_ERR_INVALID_ARGUMENT, std::string(who) + ": rectangle edges must be multiples of 64 (or the frame's last column / row)");
        area += (uint64_t)(q[2] - q[0]) * (q[3] - q[1]);
        for (uint32_t o = 0; o < r; o++) {
            const uint32_t* p = rects + 4 * o;
            if (q[0] < p[2] && p[0] < q[2] && q[1] < p[3] && p[1] < q[3]) return xfail(PLR_ERR_INVALID_ARGUMENT, std::string(who) + ": rectangles " + std::to_string(o) + " and " + std::to_string(r) + " overlap");
        }
    }
    if (area != (uint64_t)frameW * frameH) return xfail(PLR_ERR_INVALID_ARGUMENT, std::string(who) + ": the rectangles do not cover the frame");
    return PLR_OK;
}

// ---------------------------------------------------------------- pack / unpack: rectangles of images <-> one staging arena
struct CopyRegion {
    uint8_t* image;   // address of the region's first texel
    uint8_t* buffer;  // where its tightly packed copy lives
    uint32_t pitch, widthBytes, rows, unit; // unit: 16 if addresses, pitch and width allow 16-byte accesses, else 4 (texels are 4 or 8 bytes)
};
constexpr int kMaxRegions = 48; // per launch: (images of an exchange point) x (peers) - six images and three peers in a 2 x 2 grid
struct CopyTable { int n; CopyRegion r[kMaxRegions]; };
// one launch for every region of an exchange point: blockIdx.y = region, a block's threads walk the region's units with a grid stride
template <bool PACK>
__global__ __launch_bounds__(256) void exchangeCopyKernel(CopyTable t) {
    const CopyRegion r = t.r[blockIdx.y];
    const uint32_t perRow = r.widthBytes / r.unit, total = perRow * r.rows;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const uint32_t row = i / perRow, c = i - row * perRow;
        uint8_t* img = r.image + (size_t)row * r.pitch + (size_t)c * r.unit;
        uint8_t* buf = r.buffer + (size_t)i * r.unit;
        if (r.unit == 16u) { if (PACK) *(uint4*)buf = *(const uint4*)img; else *(uint4*)img = *(const uint4*)buf; }
        else { if (PACK) *(uint32_t*)buf = *(const uint32_t*)img; else *(uint32_t*)img = *(const uint32_t*)buf; }
    }
}

// ---------------------------------------------------------------- request lists (plr_frame.h PLRF_HALO_REQUESTED)
// A SLICE is the part of a request bitmap over one rank's rectangle, tightly packed: `rows` texel rows of `wordsPerRow` words, word (0, 0) = texel (x0, y0) of
// the rectangle (x0 a multiple of 32: rectangle edges are multiples of 64 pixels = 32 trace texels). Both sides of a request walk the same slice - the requester's
// copy and the copy it sent - in the same order: word by word, bit by bit; offsets[w] = set bits in front of word w, so texel number offsets[w] + (set bits of
// word w below bit b) of the response belongs to bit b of word w.
// offsets[w] counts within the word's 1024-word BLOCK; blockBase[w / 1024] = set bits in front of that block (two small kernels: a block of 1024 lanes per 1024
// words, then one wave per slice over the <= 128 block sums; ONE 1024-thread block per slice walking 64 words per thread serially took 127 us beside the trace)
struct RequestSlice { const uint32_t* bits; uint32_t* offsets; uint32_t* blockBase; uint32_t* count; uint32_t words; };
constexpr int kMaxSlices = 64;
constexpr uint32_t kScanBlock = 1024, kMaxScanBlocks = 128; // <= 131072 words = 4 M texels per slice
struct SliceTable { int n; RequestSlice s[kMaxSlices]; };
__global__ __launch_bounds__(1024) void requestScanBlocksKernel(SliceTable t) {
    __shared__ uint32_t waveSum[16];
    const RequestSlice sl = t.s[blockIdx.y];
    const uint32_t w = blockIdx.x * kScanBlock + threadIdx.x;
    if (blockIdx.x * kScanBlock >= sl.words) return;
    const uint32_t n = w < sl.words ? (uint32_t)__popc(sl.bits[w]) : 0u;
    uint32_t incl = n; // inclusive scan inside the wave
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1) { const uint32_t v = __shfl_up(incl, d, 64); if (lane >= d) incl += v; }
    if (lane == 63u) waveSum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t k = 0; k < (threadIdx.x >> 6); k++) base += waveSum[k];
    if (w < sl.words) sl.offsets[w] = base + incl - n;
    if (threadIdx.x == 1023u) sl.blockBase[blockIdx.x] = base + incl; // the block's total, turned into a base by the kernel below
}
__global__ __launch_bounds__(64) void requestScanSlicesKernel(SliceTable t) {
    const RequestSlice sl = t.s[blockIdx.x];
    if (threadIdx.x != 0) return;
    const uint32_t blocks = (sl.words + kScanBlock - 1) / kScanBlock;
    uint32_t run = 0;
    for (uint32_t b = 0; b < blocks; b++) { const uint32_t v = sl.blockBase[b]; sl.blockBase[b] = run; run += v; }
    *sl.count = run;
}
// the texels one peer asked this rank for (GATHER: images -> buffer, 16 bytes each: Y_SH, CoCg, the R16F depth) / the texels this rank asked one peer for
// (scatter: buffer -> images); (x0, y0) = the rectangle the slice covers, in trace texels
struct RequestWalk { const uint32_t* bits; const uint32_t* offsets; const uint32_t* blockBase; uint4* buffer; uint32_t words, wordsPerRow, x0, y0; };
constexpr int kMaxWalks = 16;
struct WalkTable { int n; RequestWalk w[kMaxWalks]; uint2* ysh; uint32_t* cocg; uint16_t* depth; uint32_t imageCols; };
// One lane per BIT: a wave takes kWalkWords = 8 words of the slice, two per step (the halves of the wave), four steps unrolled - the words, their offsets and then the
// texels of all four steps are independent loads, a wave's life is two memory latencies whatever its words hold. (One thread per word walked a dense word's 32 texels
// one after the other: the words along a rectangle's edge are dense, and the gather / scatter of 0.4 M texels took 27 - 52 us, on the responses' critical path:
// profiles/r06c_band_timeline.txt.)
constexpr uint32_t kWalkWords = 8;
template <bool GATHER>
__global__ __launch_bounds__(256) void requestWalkKernel(WalkTable t) {
    const RequestWalk k = t.w[blockIdx.y];
    const uint32_t lane = threadIdx.x & 63u, half = lane >> 5, bit = lane & 31u;
    const uint32_t w0 = (blockIdx.x * 4u + (threadIdx.x >> 6)) * kWalkWords; // wave-uniform
    if (w0 >= k.words) return;
    uint32_t word[4];
    bool any = false;
#pragma unroll
    for (uint32_t i = 0; i < 4; i++) {
        const uint32_t wi = w0 + 2u * i + half;
        word[i] = wi < k.words ? k.bits[wi] : 0u;
        any = any || word[i] != 0u;
    }
    if (__builtin_amdgcn_ballot_w64(any) == 0ull) return;
    uint32_t slot[4];
#pragma unroll
    for (uint32_t i = 0; i < 4; i++) {
        const uint32_t wi = w0 + 2u * i + half;
        slot[i] = word[i] ? k.blockBase[wi / kScanBlock] + k.offsets[wi] + (uint32_t)__popc(word[i] & ((1u << bit) - 1u)) : 0u;
    }
    size_t idx[4];
    bool on[4];
#pragma unroll
    for (uint32_t i = 0; i < 4; i++) {
        const uint32_t wi = w0 + 2u * i + half;
        const uint32_t row = wi / k.wordsPerRow, col = wi - row * k.wordsPerRow;
        on[i] = ((word[i] >> bit) & 1u) != 0u;
        idx[i] = (size_t)(k.y0 + row) * t.imageCols + (k.x0 + col * 32u + bit);
    }
    if (GATHER) {
        uint4 v[4];
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) if (on[i]) { const uint2 y = t.ysh[idx[i]]; v[i] = make_uint4(y.x, y.y, t.cocg[idx[i]], (uint32_t)t.depth[idx[i]]); }
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) if (on[i]) k.buffer[slot[i]] = v[i];
    } else {
        uint4 v[4];
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) if (on[i]) v[i] = k.buffer[slot[i]];
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) if (on[i]) { t.ysh[idx[i]] = make_uint2(v[i].x, v[i].y); t.cocg[idx[i]] = v[i].z; t.depth[idx[i]] = (uint16_t)v[i].w; }
    }
}
static unsigned walkBlocks(uint32_t words) { return (words + 4u * kWalkWords - 1u) / (4u * kWalkWords); }

// ---------------------------------------------------------------- waiting for the producer's edge signal on the communication stream
// hipStreamWaitValue32 does it in the command processor - and a PENDING value wait on one stream slows every kernel the chip runs meanwhile: with the host a frame
// ahead the communication stream's head is such a wait nearly all the time, and the kernels of a band measured 5 - 20 % longer (trace 116 -> 141 us, shade 215 -> 227,
// spatial filter 111 -> 125: profiles/r05_tile_vs_band.txt "value wait"). A pending EVENT wait costs nothing, and neither does one sleeping wave: the wait is a
// one-wave kernel on the communication stream that sleeps until the word has reached the value; stream order then holds the transfers back behind it.
// `cancel`: a word of pinned host memory the exchange's destructor sets - a frame that failed between the producer's launch and its edge signal would otherwise
// leave this wave asleep for ever and hipStreamDestroy / hipFree waiting for it
__global__ __launch_bounds__(64) void waitForValueKernel(const uint32_t* __restrict__ word, uint32_t value, const uint32_t* __restrict__ cancel) {
    if (threadIdx.x == 0)
        while ((int32_t)(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - value) < 0) {
            if (cancel && __hip_atomic_load(cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) break;
            __builtin_amdgcn_s_sleep(16);
        }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// ---------------------------------------------------------------- watchdog
struct Watchdog {
    struct Entry { int rank, id, phase; plrf_watchdog_query query; void* user; std::chrono::steady_clock::time_point armed; uint32_t deadlineMs; };
    std::mutex m;
    std::vector<Entry> entries;
    uint32_t deadlineMs = 2000;
    // the first exchanges of a communicator are not steady state: RCCL sets up its point-to-point connections inside the first group with each peer (hundreds of
    // milliseconds to seconds on a node), and ranks leave their set-up at different times. The first graceArms arms (~ the first ten frames) get graceMs instead.
    uint32_t graceMs = 60000, graceArms = 64, arms = 0;
    static const char* phaseName(int phase) { return phase == PLRF_EXCHANGE_BEGIN ? "BEGIN (transfers posted, completion pending)" : (phase == PLRF_EXCHANGE_END ? "END" : "in line on the launch stream"); }
    static const char* idName(int id) {
        static const char* names[PLRF_EXCHANGE_COUNT] = {"histogram all-reduce", "traced GI halo", "temporally filtered GI halo", "GI history halo", "resolved colour + TAA history halo", "depth range all-reduce", "GI sample requests"};
        return id >= 0 && id < PLRF_EXCHANGE_COUNT ? names[id] : "?";
    }
    void arm(int rank, int id, int phase, plrf_watchdog_query query, void* user) {
        if (!deadlineMs) return;
        std::lock_guard<std::mutex> lock(m);
        const uint32_t limit = arms < graceArms ? std::max(deadlineMs, graceMs) : deadlineMs;
        arms++;
        for (Entry& e : entries) if (e.id == id && e.query == query && e.user == user) { e.phase = phase; e.armed = std::chrono::steady_clock::now(); e.deadlineMs = limit; return; } // re-armed every frame
        entries.push_back({rank, id, phase, query, user, std::chrono::steady_clock::now(), limit});
    }
    // drops completed entries; true + message if one is overdue
    bool poll(std::string* msg) {
        if (!deadlineMs) return false;
        std::lock_guard<std::mutex> lock(m);
        const auto now = std::chrono::steady_clock::now();
        for (size_t i = 0; i < entries.size();) {
            const Entry& e = entries[i];
            if (e.query(e.user)) { entries.erase(entries.begin() + (long)i); continue; }
            const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(now - e.armed).count();
            if (ms > (long long)e.deadlineMs) {
                if (msg) *msg = "exchange watchdog: rank " + std::to_string(e.rank) + ", exchange " + std::to_string(e.id) + " (" + idName(e.id) + "), phase " + phaseName(e.phase) +
                                ": not complete after " + std::to_string(ms) + " ms (deadline " + std::to_string(e.deadlineMs) +
                                " ms) - a peer has not posted its side of the group, or the producer's edge signal was never raised (the launch stream may be parked in an earlier in-line exchange: "
                                "the histogram / depth-range all-reduce or the GI history exchange)";
                return true;
            }
            i++;
        }
        return false;
    }
};

struct StageArena { uint8_t* ptr = nullptr; size_t size = 0; };

// ---------------------------------------------------------------- the IN-PROCESS transport (round 6)
// All ranks of a partition in ONE process on one GPU (one host thread + backend + pipeline + exchange each, as tests/test_config5_8k.py runs them): where a
// communicator's ncclSend / ncclRecv go, a rank publishes what it posted - per transfer {peer, direction, address, bytes}, in posting order: the k-th send of rank a
// to rank b pairs with b's k-th receive from a, RCCL's matching rule inside a group - with an event behind its pack kernel, waits (on the host) until the peers it
// receives from have published the same exchange of the same frame, and copies their send ranges into its receive ranges on ITS stream behind their events
// (hipMemcpyAsync device to device). Then it waits until the peers it sends to have enqueued their copies and orders its stream behind them: a send buffer - a
// staging arena, or for bands the image rows themselves - is not overwritten while a peer still reads it, which is what a completed ncclSend guarantees.
// Everything else of an exchange is the code the communicator path runs: plans, pack / unpack kernels, the communication stream, BEGIN behind the producers' edge
// signal (the sleeping-wave wait), END, the watchdog. The all-reduces (histogram sum, depth range min / max) go through per-rank slots of group memory and a
// one-block kernel per rank.
// FROZEN (plrf_local_group_freeze): a rank no longer waits for its peers; it copies from what they published LAST - their buffers keep the texels of their last
// frame. tools/band_cost.py times one partition alone that way, with real neighbour data (one frame old) in its halos instead of its own texels.
struct LocalGroup {
    struct Op { int peer; bool send; uint8_t* ptr; size_t bytes; };
    struct Post { uint64_t generation = 0, received = 0; std::vector<Op> ops; hipEvent_t posted = nullptr, copied = nullptr; };
    std::mutex m;
    std::condition_variable cv;
    int world = 0, device = -1;
    bool aborted = false, frozen = false;
    uint32_t timeoutMs = 120000;
    std::vector<std::vector<Post>> posts; // [rank][exchange id]
    // all-reduce slots: [parity][rank] x 512 bytes of device memory, the event behind each rank's copy into its slot
    uint8_t* slots = nullptr;
    static constexpr size_t kSlotBytes = 512;
    struct Reduce { uint64_t generation = 0; hipEvent_t filled = nullptr; };
    std::vector<std::vector<Reduce>> reduces; // [rank][exchange id]
    std::string error;

    ~LocalGroup() {
        for (auto& r : posts) for (Post& p : r) { if (p.posted) hipEventDestroy(p.posted); if (p.copied) hipEventDestroy(p.copied); }
        for (auto& r : reduces) for (Reduce& q : r) if (q.filled) hipEventDestroy(q.filled);
        if (slots) hipFree(slots);
    }
    // waits until pred() holds; false (with `error` set) on abort or timeout
    template <class Pred> bool waitFor(std::unique_lock<std::mutex>& lock, const char* what, Pred pred) {
        const bool ok = cv.wait_for(lock, std::chrono::milliseconds(timeoutMs), [&] { return aborted || pred(); });
        if (!ok) { aborted = true; error = std::string("in-process exchange: timed out waiting for ") + what; cv.notify_all(); return false; }
        if (aborted && !pred()) { if (error.empty()) error = "in-process exchange: another rank failed"; return false; }
        return true;
    }
};
// every receive of one exchange in ONE launch (a device copy per transfer is a blit launch of ~10 us each: three peers made an exchange take 100 us on the launch stream,
// where a communicator's group is one kernel): blockIdx.y = transfer, 16-byte units where both addresses allow it, else words (sizes are multiples of 4)
struct LocalCopy { uint8_t* dst; const uint8_t* src; size_t bytes; };
constexpr int kMaxLocalCopies = 32;
struct LocalCopyTable { int n; LocalCopy c[kMaxLocalCopies]; };
__global__ __launch_bounds__(256) void localCopyKernel(LocalCopyTable t) {
    const LocalCopy c = t.c[blockIdx.y];
    const bool wide = (((uintptr_t)c.dst | (uintptr_t)c.src) & 15u) == 0u;
    if (wide) {
        const size_t units = c.bytes / 16;
        for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < units; i += (size_t)gridDim.x * 256u) ((uint4*)c.dst)[i] = ((const uint4*)c.src)[i];
        for (size_t i = units * 4 + (size_t)blockIdx.x * 256u + threadIdx.x; i < c.bytes / 4; i += (size_t)gridDim.x * 256u) ((uint32_t*)c.dst)[i] = ((const uint32_t*)c.src)[i];
    } else {
        for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < c.bytes / 4; i += (size_t)gridDim.x * 256u) ((uint32_t*)c.dst)[i] = ((const uint32_t*)c.src)[i];
    }
}
// dst[i] = reduction over the ranks' slots; op 0: uint32 sum (histogram bins), 1: {min, max} of two floats (depth range)
__global__ __launch_bounds__(128) void localAllReduceKernel(const uint8_t* __restrict__ slots, int world, int op, uint32_t words, uint32_t* __restrict__ dst) {
    const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
    if (i >= words) return;
    if (op == 0) {
        uint32_t sum = 0;
        for (int r = 0; r < world; r++) sum += ((const uint32_t*)(slots + (size_t)r * LocalGroup::kSlotBytes))[i];
        dst[i] = sum;
    } else {
        float v = ((const float*)slots)[i];
        for (int r = 1; r < world; r++) { const float o = ((const float*)(slots + (size_t)r * LocalGroup::kSlotBytes))[i]; v = i == 0 ? fminf(v, o) : fmaxf(v, o); }
        ((float*)dst)[i] = v;
    }
}

struct RcclExchange {
    FramePipeline* fp = nullptr;
    ncclComm_t comm = nullptr;
    LocalGroup* local = nullptr;  // the in-process transport (plrf_local_attach_rects): ranks of one process exchange through device copies
    bool loopback = false;
    int rank = 0, world = 1, device = 0;
    uint32_t frameWidth = 0, frameHeight = 0;
    std::vector<uint32_t> bounds; // world + 1 row boundaries, or empty: the equal partition (bands)
    std::vector<Rect> rects;      // tile rendering: every rank's rectangle; empty = bands
    hipStream_t commStream = nullptr;
    uint32_t lastSignalValue = 0; // the edge signal value the previous BEGIN waited for: a BEGIN whose producer raised no new one orders behind the launch stream
    hipEvent_t ready[PLRF_EXCHANGE_COUNT] = {}, done[PLRF_EXCHANGE_COUNT] = {};
    // END without an event wait (PLRF_EXCHANGE_END_BY_VALUE=1; measured, not the default): the communication stream stores the exchange's serial into a word of signal
    // memory behind its transfers (hipStreamWriteValue32) and the launch stream waits for that value (hipStreamWaitValue32) - a satisfied value wait is cheaper than a
    // cross-stream event dependency (~3 against ~10 us), a PENDING one slows every kernel on the chip. One 8-byte signal allocation per exchange point; nullptr: events.
    uint32_t* doneFlag[PLRF_EXCHANGE_COUNT] = {};
    uint32_t doneSerial[PLRF_EXCHANGE_COUNT] = {};
    StageArena sendArena[PLRF_EXCHANGE_COUNT], recvArena[PLRF_EXCHANGE_COUNT];
    uint64_t bytesSent = 0, bytesReceived = 0, exchanges = 0; // of the last frame (reset by the histogram exchange, the first of a frame)
    // request lists (PLRF_HALO_REQUESTED): per peer one buffer of request slices going out (the peer's rectangle, both spatial filter passes) and one coming in (this
    // rank's rectangle, both passes), their per-word offsets, the totals (device, and a pinned copy the host reads behind countsReady)
    struct RequestState {
        bool ready = false;
        plrf_gi_request info{};
        struct Peer { uint32_t outWords = 0, outWordsPerRow = 0, inWords = 0; uint32_t* out = nullptr; uint32_t* in = nullptr; uint32_t* outOff = nullptr; uint32_t* inOff = nullptr;
                      uint32_t* outBase = nullptr; uint32_t* inBase = nullptr; /* [2 points][kMaxScanBlocks] */ Rect rect{}; };
        std::vector<Peer> peers;          // [world]; the entry of this rank is empty
        uint32_t inWordsPerRow = 0;
        Rect mine{};
        uint8_t* arena = nullptr;
        uint32_t* countsDev = nullptr;    // [2 points][world][2: out, in]
        uint32_t* countsHost = nullptr;   // pinned
        hipEvent_t countsReady = nullptr;
        StageArena send[2], recv[2];
    } req;
    int lastOverlapMode = 0;
    bool deferredPost[PLRF_EXCHANGE_COUNT] = {}; // request lists: a BEGIN whose transfers are posted from its END (run())
    bool packedRegions = false;
    int streamWaitValueSupported = 0;
    // watchdog: entries are completion events; a background thread reports (and aborts) when the frame loop itself is stuck
    Watchdog dog;
    struct DogArg { RcclExchange* x; int id; } dogArgs[PLRF_EXCHANGE_COUNT];
    std::thread dogThread;
    std::mutex dogMutex;
    std::condition_variable dogWake;
    bool dogStop = false, dogAbort = true;
    // the watchdog REPORTS an overdue exchange after dog.deadlineMs (stderr, and the next exchange callback fails); it ABORTS the process only when the exchange is
    // still incomplete dogAbortAfterMs later (PLRF_EXCHANGE_WATCHDOG_ABORT_MS; a peer paused in a debugger or an oversubscribed host recovers before that - ADVICE r05)
    uint32_t dogAbortAfterMs = 30000;
    std::chrono::steady_clock::time_point dogFiredAt;
    uint32_t* cancelWord = nullptr; // pinned host word waitForValueKernel polls (set by the destructor)
    std::atomic<bool> dogFired{false};
    std::string dogMessage;

    ~RcclExchange() {
        {
            std::lock_guard<std::mutex> lock(dogMutex);
            dogStop = true;
        }
        dogWake.notify_all();
        if (dogThread.joinable()) dogThread.join();
        if (cancelWord) *(volatile uint32_t*)cancelWord = 1u; // a wave still asleep on an edge signal that will never come lets go
        if (comm) ncclCommDestroy(comm);
        for (auto e : ready) if (e) hipEventDestroy(e);
        for (auto e : done) if (e) hipEventDestroy(e);
        for (auto f : doneFlag) if (f) hipFree(f);
        for (auto& a : sendArena) if (a.ptr) hipFree(a.ptr);
        for (auto& a : recvArena) if (a.ptr) hipFree(a.ptr);
        if (commStream) hipStreamDestroy(commStream);
        if (cancelWord) hipHostFree(cancelWord);
        if (req.arena) hipFree(req.arena);
        if (req.countsDev) hipFree(req.countsDev);
        if (req.countsHost) hipHostFree(req.countsHost);
        if (req.countsReady) hipEventDestroy(req.countsReady);
        for (auto& a : req.send) if (a.ptr) hipFree(a.ptr);
        for (auto& a : req.recv) if (a.ptr) hipFree(a.ptr);
    }

    int nccl(ncclResult_t r, const char* what) { return r == ncclSuccess ? 0 : xfail(PLR_ERR_HIP, std::string(what) + ": " + ncclGetErrorString(r)); }
    int hip(hipError_t e, const char* what) { return e == hipSuccess ? 0 : xfail(PLR_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e)); }

    static int eventDone(void* user) {
        const DogArg* a = (const DogArg*)user;
        return hipEventQuery(a->x->done[a->id]) != hipErrorNotReady ? 1 : 0; // (an error also ends the watch: the failing call reports it)
    }
    void startWatchdog() {
        if (const char* ms = std::getenv("PLRF_EXCHANGE_WATCHDOG_MS")) dog.deadlineMs = (uint32_t)std::max(0, std::atoi(ms));
        if (const char* ms = std::getenv("PLRF_EXCHANGE_WATCHDOG_FIRST_MS")) dog.graceMs = (uint32_t)std::max(0, std::atoi(ms));
        if (const char* ab = std::getenv("PLRF_EXCHANGE_WATCHDOG_ABORT")) dogAbort = std::atoi(ab) != 0;
        if (const char* ms = std::getenv("PLRF_EXCHANGE_WATCHDOG_ABORT_MS")) dogAbortAfterMs = (uint32_t)std::max(0, std::atoi(ms));
        for (int i = 0; i < PLRF_EXCHANGE_COUNT; i++) dogArgs[i] = {this, i};
        if (!dog.deadlineMs) return;
        dogThread = std::thread([this] {
            (void)hipSetDevice(device);
            std::unique_lock<std::mutex> lock(dogMutex);
            while (!dogStop) {
                dogWake.wait_for(lock, std::chrono::milliseconds(50));
                if (dogStop) break;
                std::string msg;
                const bool overdue = dog.poll(&msg);
                if (!dogFired.load() && overdue) {
                    dogMessage = msg;
                    dogFired.store(true);
                    dogFiredAt = std::chrono::steady_clock::now();
                    std::fprintf(stderr, "[plr] %s\n", msg.c_str());
                    std::fflush(stderr);
                } else if (dogFired.load() && !overdue) {
                    dogFired.store(false); // it completed after all (a benign stall): frames go on
                    std::fprintf(stderr, "[plr] exchange watchdog: the overdue exchange of rank %d has completed\n", rank);
                } else if (dogFired.load() && dogAbort && std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - dogFiredAt).count() > (long long)dogAbortAfterMs) {
                    std::fprintf(stderr, "[plr] exchange watchdog: still incomplete %u ms after the report: aborting (a posted collective cannot be cancelled)\n", dogAbortAfterMs);
                    std::fflush(stderr);
                    std::abort();
                }
            }
        });
    }
    // every exchange callback: an overdue entry fails the frame here (the thread above has the case where the frame loop itself is stuck)
    int checkWatchdog() {
        std::string msg;
        if (dogFired.load()) { std::lock_guard<std::mutex> lock(dogMutex); return xfail(PLR_ERR_HIP, dogMessage); }
        if (dog.poll(&msg)) return xfail(PLR_ERR_HIP, msg);
        return 0;
    }

    bool tiledItems(const plrf_exchange_item* items, uint32_t count) const {
        if (!rects.empty()) for (const Rect& r : rects) if (r.x0 != 0 || r.x1 != frameWidth) return true;
        for (uint32_t i = 0; i < count; i++) if (items[i].col_begin != 0 || items[i].col_end != items[i].image_cols) return true;
        return false;
    }

    int arena(StageArena& a, size_t bytes) {
        if (a.size >= bytes) return 0;
        if (a.ptr) { if (int rc = hip(hipDeviceSynchronize(), "hipDeviceSynchronize")) return rc; hipFree(a.ptr); a.ptr = nullptr; a.size = 0; }
        const size_t want = std::max<size_t>(bytes + bytes / 4, 1 << 20);
        if (int rc = hip(hipMalloc((void**)&a.ptr, want), "hipMalloc(exchange staging)")) return rc;
        a.size = want;
        return 0;
    }

    // regions of `table` in launches of at most kMaxRegions
    template <bool PACK>
    int launchCopies(const std::vector<CopyRegion>& regions, hipStream_t stream) {
        for (size_t first = 0; first < regions.size(); first += kMaxRegions) {
            CopyTable t{};
            t.n = (int)std::min<size_t>(kMaxRegions, regions.size() - first);
            uint32_t maxUnits = 1;
            for (int i = 0; i < t.n; i++) { t.r[i] = regions[first + i]; maxUnits = std::max(maxUnits, t.r[i].widthBytes / t.r[i].unit * t.r[i].rows); }
            const dim3 grid(std::min((maxUnits + 1023u) / 1024u, 256u), (unsigned)t.n); // four units per thread, at most 256 blocks per region
            exchangeCopyKernel<PACK><<<grid, 256, 0, stream>>>(t);
            if (int rc = hip(hipGetLastError(), "exchange pack / unpack launch")) return rc;
        }
        return 0;
    }

    // ---- the in-process transport: `ops` = this rank's transfers of exchange `id` in posting order; the send data is final on `stream` when this is called
    // (behind the pack kernel / the producer); returns with the copies of every receive enqueued on `stream` and `stream` ordered behind the peers' reads of the sends
    int localTransfer(int id, const std::vector<LocalGroup::Op>& ops, hipStream_t stream) {
        LocalGroup& g = *local;
        LocalGroup::Post& mine = g.posts[(size_t)rank][(size_t)id];
        if (!mine.posted) {
            if (int rc = hip(hipEventCreateWithFlags(&mine.posted, hipEventDisableTiming), "hipEventCreateWithFlags")) return rc;
            if (int rc = hip(hipEventCreateWithFlags(&mine.copied, hipEventDisableTiming), "hipEventCreateWithFlags")) return rc;
        }
        if (int rc = hip(hipEventRecord(mine.posted, stream), "hipEventRecord(posted)")) return rc;
        std::unique_lock<std::mutex> lock(g.m);
        mine.ops = ops;
        const uint64_t gen = ++mine.generation;
        g.cv.notify_all();
        std::vector<int> from, to;
        for (const LocalGroup::Op& o : ops) { std::vector<int>& v = o.send ? to : from; if (std::find(v.begin(), v.end(), o.peer) == v.end()) v.push_back(o.peer); }
        if (!g.frozen) {
            if (!g.waitFor(lock, "a peer to post its side of the exchange", [&] { for (int p : from) if (g.posts[(size_t)p][(size_t)id].generation < gen) return false; return true; }))
                return xfail(PLR_ERR_HIP, g.error);
        }
        // my k-th receive from p <- p's k-th send to me
        std::vector<LocalCopy> copies;
        auto copy = [&](uint8_t* dst, const uint8_t* src, size_t bytes) -> int {
            if (bytes % 4 == 0 && bytes > 0) { copies.push_back({dst, src, bytes}); return 0; }
            return bytes ? hip(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream), "hipMemcpyAsync(in-process transfer)") : 0;
        };
        for (int p : from) {
            const LocalGroup::Post& theirs = g.posts[(size_t)p][(size_t)id];
            if (!theirs.posted || theirs.generation == 0) return xfail(PLR_ERR_HIP, "in-process exchange: peer " + std::to_string(p) + " has never posted exchange " + std::to_string(id));
            std::vector<const LocalGroup::Op*> sends;
            for (const LocalGroup::Op& o : theirs.ops) if (o.send && o.peer == rank) sends.push_back(&o);
            size_t k = 0;
            if (int rc = hip(hipStreamWaitEvent(stream, theirs.posted, 0), "hipStreamWaitEvent(peer posted)")) return rc;
            for (const LocalGroup::Op& o : ops) {
                if (o.send || o.peer != p) continue;
                if (g.frozen) { // timing replay: the peer's buffers hold its LAST frame - as many bytes as both sides have (request lists change size from frame to frame)
                    if (k < sends.size()) if (int rc = copy(o.ptr, sends[k]->ptr, std::min(o.bytes, sends[k]->bytes))) return rc;
                    k++;
                    continue;
                }
                if (k >= sends.size() || sends[k]->bytes != o.bytes)
                    return xfail(PLR_ERR_HIP, "in-process exchange " + std::to_string(id) + ": rank " + std::to_string(rank) + " expects " + std::to_string(o.bytes) + " bytes from rank " + std::to_string(p) +
                                                  ", which posted " + (k < sends.size() ? std::to_string(sends[k]->bytes) + " bytes" : std::string("fewer sends")) + " (the plans of the two ranks disagree)");
                if (int rc = copy(o.ptr, sends[k]->ptr, o.bytes)) return rc;
                k++;
            }
            if (k != sends.size() && !g.frozen) return xfail(PLR_ERR_HIP, "in-process exchange " + std::to_string(id) + ": rank " + std::to_string(p) + " posted more sends to rank " + std::to_string(rank) + " than it receives");
        }
        for (size_t first = 0; first < copies.size(); first += kMaxLocalCopies) {
            LocalCopyTable t{};
            t.n = (int)std::min<size_t>(kMaxLocalCopies, copies.size() - first);
            size_t largest = 16;
            for (int i = 0; i < t.n; i++) { t.c[i] = copies[first + i]; largest = std::max(largest, t.c[i].bytes); }
            localCopyKernel<<<dim3((unsigned)std::min<size_t>((largest / 16 + 1023) / 1024, 512), (unsigned)t.n), 256, 0, stream>>>(t);
            if (int rc = hip(hipGetLastError(), "localCopyKernel")) return rc;
        }
        if (int rc = hip(hipEventRecord(mine.copied, stream), "hipEventRecord(copied)")) return rc;
        mine.received = gen;
        g.cv.notify_all();
        if (!g.frozen) {
            if (!g.waitFor(lock, "a peer to take what was sent to it", [&] { for (int p : to) if (g.posts[(size_t)p][(size_t)id].received < gen) return false; return true; }))
                return xfail(PLR_ERR_HIP, g.error);
            for (int p : to) if (int rc = hip(hipStreamWaitEvent(stream, g.posts[(size_t)p][(size_t)id].copied, 0), "hipStreamWaitEvent(peer copied)")) return rc;
        }
        return 0;
    }
    // in-process all-reduce of `bytes` (<= 512) at ptr, in place: op 0 = uint32 sum, 1 = {min, max} of two floats
    int localAllReduce(int id, void* ptr, size_t bytes, int op, hipStream_t stream) {
        LocalGroup& g = *local;
        if (bytes > LocalGroup::kSlotBytes) return xfail(PLR_ERR_UNSUPPORTED, "in-process all-reduce: more than 512 bytes");
        LocalGroup::Reduce& mine = g.reduces[(size_t)rank][(size_t)id];
        if (!mine.filled) if (int rc = hip(hipEventCreateWithFlags(&mine.filled, hipEventDisableTiming), "hipEventCreateWithFlags")) return rc;
        std::unique_lock<std::mutex> lock(g.m);
        const uint64_t gen = mine.generation + 1;
        // two sets of slots PER EXCHANGE ID: a rank one exchange ahead does not overwrite what a peer still reads - and the depth-range all-reduce does not land in the
        // histogram's slots (one shared set: a rank already at the depth range overwrote bins another rank was still summing; found by the tile light-matrix test)
        uint8_t* base = g.slots + ((size_t)id * 2u + (size_t)(gen & 1u)) * (size_t)g.world * LocalGroup::kSlotBytes;
        lock.unlock();
        if (int rc = hip(hipMemcpyAsync(base + (size_t)rank * LocalGroup::kSlotBytes, ptr, bytes, hipMemcpyDeviceToDevice, stream), "hipMemcpyAsync(all-reduce slot)")) return rc;
        if (int rc = hip(hipEventRecord(mine.filled, stream), "hipEventRecord(filled)")) return rc;
        lock.lock();
        mine.generation = gen;
        g.cv.notify_all();
        if (g.frozen) return 0; // a rank on its own: its value is the "sum" (timing replay only)
        if (!g.waitFor(lock, "every rank to reach the all-reduce", [&] { for (int p = 0; p < g.world; p++) if (g.reduces[(size_t)p][(size_t)id].generation < gen) return false; return true; }))
            return xfail(PLR_ERR_HIP, g.error);
        for (int p = 0; p < g.world; p++) if (p != rank) if (int rc = hip(hipStreamWaitEvent(stream, g.reduces[(size_t)p][(size_t)id].filled, 0), "hipStreamWaitEvent(filled)")) return rc;
        lock.unlock();
        const uint32_t words = (uint32_t)(bytes / 4);
        localAllReduceKernel<<<(words + 127u) / 128u, 128, 0, stream>>>(base, g.world, op, words, (uint32_t*)ptr);
        return hip(hipGetLastError(), "localAllReduceKernel");
    }

    // tile rendering: pack -> one send / receive per peer -> unpack, on `stream`
    int postPacked(int id, const plrf_exchange_item* items, uint32_t count, hipStream_t stream) {
        struct PeerPlan { size_t sendBytes = 0, recvBytes = 0, sendOffset = 0, recvOffset = 0; };
        std::vector<PeerPlan> peers((size_t)world);
        struct Piece { uint32_t item; plrf_rect_op op; size_t offset; };
        std::vector<Piece> pieces;
        std::vector<plrf_rect_op> ops((size_t)2 * (size_t)std::max(world - 1, 1));
        auto align = [](size_t v) { return (v + 255) & ~(size_t)255; };
        for (uint32_t i = 0; i < count; i++) {
            const plrf_exchange_item& it = items[i];
            const uint32_t n = planRects(frameWidth, frameHeight, (uint32_t)world, rects.data(), (uint32_t)rank, it.image_cols, it.image_rows, it.halo_rows, ops.data(), (uint32_t)ops.size());
            for (uint32_t k = 0; k < n && k < ops.size(); k++) {
                const plrf_rect_op& o = ops[k];
                const size_t bytes = (size_t)(o.x1 - o.x0) * it.texel_bytes * (o.y1 - o.y0);
                PeerPlan& pp = peers[o.peer];
                size_t& total = o.send ? pp.sendBytes : pp.recvBytes;
                pieces.push_back({i, o, total});
                total = (total + bytes + 15) & ~(size_t)15; // the next region of this peer's buffer starts 16-byte aligned
            }
        }
        size_t sendTotal = 0, recvTotal = 0;
        for (PeerPlan& pp : peers) { pp.sendOffset = sendTotal; sendTotal += align(pp.sendBytes); pp.recvOffset = recvTotal; recvTotal += align(pp.recvBytes); }
        if (sendTotal + recvTotal == 0) { exchanges++; return 0; }
        if (int rc = arena(sendArena[id], sendTotal)) return rc;
        if (int rc = arena(recvArena[id], recvTotal)) return rc;
        std::vector<CopyRegion> packs, unpacks;
        for (const Piece& pc : pieces) {
            const plrf_exchange_item& it = items[pc.item];
            const plrf_rect_op& o = pc.op;
            CopyRegion r;
            r.image = (uint8_t*)it.device_ptr + (size_t)o.y0 * it.row_bytes + (size_t)o.x0 * it.texel_bytes;
            r.buffer = (o.send ? sendArena[id].ptr + peers[o.peer].sendOffset : recvArena[id].ptr + peers[o.peer].recvOffset) + pc.offset;
            r.pitch = it.row_bytes; r.widthBytes = (o.x1 - o.x0) * it.texel_bytes; r.rows = o.y1 - o.y0;
            r.unit = ((uintptr_t)r.image % 16 == 0 && r.pitch % 16 == 0 && r.widthBytes % 16 == 0) ? 16u : 4u;
            if (r.widthBytes % 4u) return xfail(PLR_ERR_UNSUPPORTED, "exchange: texel rows that are no multiple of 4 bytes");
            (o.send ? packs : unpacks).push_back(r);
        }
        if (int rc = launchCopies<true>(packs, stream)) return rc;
        if (local) {
            std::vector<LocalGroup::Op> lops;
            for (int p = 0; p < world; p++) { // the order of the communicator path's group: per peer, the send, then the receive
                const PeerPlan& pp = peers[(size_t)p];
                if (pp.sendBytes) lops.push_back({p, true, sendArena[id].ptr + pp.sendOffset, pp.sendBytes});
                if (pp.recvBytes) lops.push_back({p, false, recvArena[id].ptr + pp.recvOffset, pp.recvBytes});
            }
            if (int rc = localTransfer(id, lops, stream)) return rc;
        } else if (loopback) {
            // stand-in for the links: the arena's bytes make one trip through the copy path (the texels that "arrive" are this rank's own: timing only)
            const size_t n = std::min(sendTotal, recvTotal);
            if (n) if (int rc = hip(hipMemcpyAsync(recvArena[id].ptr, sendArena[id].ptr, n, hipMemcpyDeviceToDevice, stream), "hipMemcpyAsync(loopback)")) return rc;
        } else {
            if (int rc = nccl(ncclGroupStart(), "ncclGroupStart")) return rc;
            int rc = 0;
            for (int p = 0; p < world && !rc; p++) {
                const PeerPlan& pp = peers[(size_t)p];
                if (pp.sendBytes) rc = nccl(ncclSend(sendArena[id].ptr + pp.sendOffset, pp.sendBytes, ncclUint8, p, comm, stream), "ncclSend");
                if (pp.recvBytes && !rc) rc = nccl(ncclRecv(recvArena[id].ptr + pp.recvOffset, pp.recvBytes, ncclUint8, p, comm, stream), "ncclRecv");
            }
            const int grc = nccl(ncclGroupEnd(), "ncclGroupEnd");
            if (rc || grc) return rc ? rc : grc;
        }
        for (const PeerPlan& pp : peers) { bytesSent += pp.sendBytes; bytesReceived += pp.recvBytes; }
        exchanges++;
        return launchCopies<false>(unpacks, stream);
    }

    // ---- request lists (plr_frame.h PLRF_HALO_REQUESTED). moves `ops` (posting order; the peers post the mirror image) through the communicator or the in-process transport
    int transfer(int id, const std::vector<LocalGroup::Op>& ops, hipStream_t stream) {
        for (const LocalGroup::Op& o : ops) (o.send ? bytesSent : bytesReceived) += o.bytes;
        exchanges++;
        if (local) return localTransfer(id, ops, stream);
        if (loopback) return 0; // (one rank on its own: nothing to ask anybody for)
        if (int rc = nccl(ncclGroupStart(), "ncclGroupStart")) return rc;
        int rc = 0;
        for (size_t k = 0; k < ops.size() && !rc; k++)
            rc = ops[k].send ? nccl(ncclSend(ops[k].ptr, ops[k].bytes, ncclUint8, ops[k].peer, comm, stream), "ncclSend") : nccl(ncclRecv(ops[k].ptr, ops[k].bytes, ncclUint8, ops[k].peer, comm, stream), "ncclRecv");
        const int grc = nccl(ncclGroupEnd(), "ncclGroupEnd");
        return rc ? rc : grc;
    }
    Rect traceRect(int p) const { // rank p's rectangle in trace texels
        Rect full;
        if (!rects.empty()) full = rects[(size_t)p];
        else { uint32_t b0, b1; if (bounds.empty()) bandRowsOf(frameHeight, (uint32_t)world, (uint32_t)p, &b0, &b1); else { b0 = bounds[(size_t)p]; b1 = bounds[(size_t)p + 1]; } full = {0u, b0, frameWidth, b1}; }
        return scaleRect(full, frameWidth, frameHeight, req.info.image_cols, req.info.image_rows);
    }
    int prepareRequests(hipStream_t stream) {
        plrf_gi_request info;
        if (int rc = plrf_get_gi_request_exchange(fp, &info)) return xfail(rc, plrf_last_error());
        if (!info.enabled) return xfail(PLR_ERR_INVALID_ARGUMENT, "exchange: a request exchange for a pipeline without band_gi_halo = PLRF_HALO_REQUESTED");
        if (req.ready && std::memcmp(&info, &req.info, sizeof(info)) == 0) return 0;
        if (req.ready) return xfail(PLR_ERR_UNSUPPORTED, "exchange: the request buffers of the pipeline changed");
        req.info = info;
        req.mine = traceRect(rank);
        if (req.mine.x0 != info.x0 || req.mine.y0 != info.y0 || req.mine.x1 != info.x1 || req.mine.y1 != info.y1) return xfail(PLR_ERR_INVALID_ARGUMENT, "exchange: the pipeline's rectangle is not this rank's");
        auto wordsPerRow = [](const Rect& r) { return (r.x1 + 31u) / 32u - r.x0 / 32u; };
        req.inWordsPerRow = wordsPerRow(req.mine);
        const uint32_t inWords = 2u * req.inWordsPerRow * (req.mine.y1 - req.mine.y0);
        req.peers.assign((size_t)world, RequestState::Peer{});
        size_t total = 0;
        for (int p = 0; p < world; p++) {
            if (p == rank) continue;
            RequestState::Peer& pe = req.peers[(size_t)p];
            pe.rect = traceRect(p);
            if (pe.rect.x0 % 32u || req.mine.x0 % 32u) return xfail(PLR_ERR_UNSUPPORTED, "exchange: request lists need rectangle edges on multiples of 32 trace texels");
            pe.outWordsPerRow = wordsPerRow(pe.rect);
            pe.outWords = 2u * pe.outWordsPerRow * (pe.rect.y1 - pe.rect.y0);
            pe.inWords = inWords;
            if (pe.outWords / 2 > kScanBlock * kMaxScanBlocks || pe.inWords / 2 > kScanBlock * kMaxScanBlocks) return xfail(PLR_ERR_UNSUPPORTED, "exchange: request lists over rectangles of more than 4 M trace texels");
            total += 2 * ((size_t)pe.outWords + pe.inWords) * 4 + 4 * 256 + 2 * (2 * kMaxScanBlocks * 4 + 256);
        }
        // (cleared ON THE STREAM the exchange runs on: hipMemset goes to the null stream, which does not order against non-blocking streams - the clear could land on
        //  top of the first frame's slices)
        if (int rc = hip(hipMalloc((void**)&req.arena, std::max<size_t>(total, 256)), "hipMalloc(request slices)")) return rc;
        if (int rc = hip(hipMemsetAsync(req.arena, 0, std::max<size_t>(total, 256), stream), "hipMemsetAsync(request slices)")) return rc;
        uint8_t* at = req.arena;
        auto take = [&](uint32_t words) { uint32_t* p0 = (uint32_t*)at; at += (((size_t)words * 4 + 255) & ~(size_t)255); return p0; };
        for (int p = 0; p < world; p++) {
            if (p == rank) continue;
            RequestState::Peer& pe = req.peers[(size_t)p];
            pe.out = take(pe.outWords); pe.in = take(pe.inWords); pe.outOff = take(pe.outWords); pe.inOff = take(pe.inWords);
            pe.outBase = take(2 * kMaxScanBlocks); pe.inBase = take(2 * kMaxScanBlocks);
        }
        const size_t countBytes = (size_t)2 * (size_t)world * 2 * sizeof(uint32_t);
        if (int rc = hip(hipMalloc((void**)&req.countsDev, countBytes), "hipMalloc(request counts)")) return rc;
        if (int rc = hip(hipMemsetAsync(req.countsDev, 0, countBytes, stream), "hipMemsetAsync(request counts)")) return rc;
        if (int rc = hip(hipHostMalloc((void**)&req.countsHost, countBytes, hipHostMallocDefault), "hipHostMalloc(request counts)")) return rc;
        std::memset(req.countsHost, 0, countBytes);
        if (int rc = hip(hipEventCreateWithFlags(&req.countsReady, hipEventDisableTiming), "hipEventCreateWithFlags")) return rc;
        req.ready = true;
        return 0;
    }
    uint32_t* countSlot(uint32_t* base, int point, int peer, int in) const { return base + (((size_t)point * (size_t)world + (size_t)peer) * 2 + (size_t)in); }
    // PLRF_EXCHANGE_GI_REQUESTS: the marks of both spatial filter passes are in the bitmaps (the launch stream is behind the giSampleRequests passes); every peer gets the
    // part over its rectangle, this rank the parts over its own, and the totals go to the host
    int postRequests(hipStream_t stream) {
        if (int rc = prepareRequests(stream)) return rc;
        std::vector<CopyRegion> packs;
        std::vector<LocalGroup::Op> ops;
        for (int p = 0; p < world; p++) {
            if (p == rank) continue;
            const RequestState::Peer& pe = req.peers[(size_t)p];
            for (int q = 0; q < 2; q++) {
                CopyRegion r;
                r.image = (uint8_t*)req.info.bitmap[q] + ((size_t)pe.rect.y0 * req.info.row_words + pe.rect.x0 / 32u) * 4;
                r.buffer = (uint8_t*)(pe.out + (size_t)q * (pe.outWords / 2));
                r.pitch = req.info.row_words * 4; r.widthBytes = pe.outWordsPerRow * 4; r.rows = pe.rect.y1 - pe.rect.y0; r.unit = 4u;
                packs.push_back(r);
            }
            ops.push_back({p, true, (uint8_t*)pe.out, (size_t)pe.outWords * 4});
            ops.push_back({p, false, (uint8_t*)pe.in, (size_t)pe.inWords * 4});
        }
        if (int rc = launchCopies<true>(packs, stream)) return rc;
        if (int rc = transfer(PLRF_EXCHANGE_GI_REQUESTS, ops, stream)) return rc;
        SliceTable t{};
        for (int p = 0; p < world; p++) {
            if (p == rank) continue;
            const RequestState::Peer& pe = req.peers[(size_t)p];
            for (int q = 0; q < 2; q++) {
                if (t.n + 2 > kMaxSlices) return xfail(PLR_ERR_UNSUPPORTED, "exchange: request lists for more than 16 peers");
                t.s[t.n++] = {pe.out + (size_t)q * (pe.outWords / 2), pe.outOff + (size_t)q * (pe.outWords / 2), pe.outBase + (size_t)q * kMaxScanBlocks, countSlot(req.countsDev, q, p, 0), pe.outWords / 2};
                t.s[t.n++] = {pe.in + (size_t)q * (pe.inWords / 2), pe.inOff + (size_t)q * (pe.inWords / 2), pe.inBase + (size_t)q * kMaxScanBlocks, countSlot(req.countsDev, q, p, 1), pe.inWords / 2};
            }
        }
        if (t.n) {
            uint32_t maxWords = 1;
            for (int i = 0; i < t.n; i++) maxWords = std::max(maxWords, t.s[i].words);
            requestScanBlocksKernel<<<dim3((maxWords + kScanBlock - 1) / kScanBlock, (unsigned)t.n), 1024, 0, stream>>>(t);
            if (int rc = hip(hipGetLastError(), "requestScanBlocksKernel")) return rc;
            requestScanSlicesKernel<<<(unsigned)t.n, 64, 0, stream>>>(t);
            if (int rc = hip(hipGetLastError(), "requestScanSlicesKernel")) return rc;
        }
        if (int rc = hip(hipMemcpyAsync(req.countsHost, req.countsDev, (size_t)2 * (size_t)world * 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream), "hipMemcpyAsync(request counts)")) return rc;
        return hip(hipEventRecord(req.countsReady, stream), "hipEventRecord(request counts)");
    }
    // PLRF_EXCHANGE_GI_TRACE / _GI_TEMPORAL with request lists: this rank's input images of spatial filter pass `point` are complete on its rectangle (the launch stream is
    // behind their producer): gather what the peers asked for, trade, scatter what this rank asked for into the images and the depth texture
    int postRequested(int id, int point, hipStream_t stream) {
        if (!req.ready) return xfail(PLR_ERR_INVALID_ARGUMENT, "exchange: requested texels before the request exchange of the frame");
        static const bool debugWait = std::getenv("PLRF_EXCHANGE_DEBUG_WAIT") != nullptr; // how long does the launch thread wait for the counts? (0: it is the GPU that waits for the host)
        const auto w0 = std::chrono::steady_clock::now();
        if (int rc = hip(hipEventSynchronize(req.countsReady), "hipEventSynchronize(request counts)")) return rc; // the host needs the sizes; the GPU is long past this point
        if (debugWait && point == 0) {
            static thread_local double total = 0; static thread_local int n = 0;
            total += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count();
            if (++n % 20 == 0) { fprintf(stderr, "[exchange rank %d] launch thread waited %.1f us per frame for the request counts (last 20 frames)\n", rank, total / 20); total = 0; }
        }
        if (int rc = hip(hipStreamWaitEvent(stream, req.countsReady, 0), "hipStreamWaitEvent(request counts)")) return rc; // (the slices and their offsets were made on the communication stream)
        size_t sendTotal = 0, recvTotal = 0;
        std::vector<size_t> sendAt((size_t)world, 0), recvAt((size_t)world, 0);
        for (int p = 0; p < world; p++) {
            if (p == rank) continue;
            sendAt[(size_t)p] = sendTotal; sendTotal += (size_t)*countSlot(req.countsHost, point, p, 1) * 16;
            recvAt[(size_t)p] = recvTotal; recvTotal += (size_t)*countSlot(req.countsHost, point, p, 0) * 16;
        }
        if (int rc = arena(req.send[point], std::max<size_t>(sendTotal, 16))) return rc;
        if (int rc = arena(req.recv[point], std::max<size_t>(recvTotal, 16))) return rc;
        WalkTable gather{}, scatter{};
        gather.ysh = scatter.ysh = (uint2*)req.info.ysh[point]; gather.cocg = scatter.cocg = (uint32_t*)req.info.cocg[point]; gather.depth = scatter.depth = (uint16_t*)req.info.depth;
        gather.imageCols = scatter.imageCols = req.info.image_cols;
        std::vector<LocalGroup::Op> ops;
        uint32_t maxGather = 1, maxScatter = 1;
        for (int p = 0; p < world; p++) {
            if (p == rank) continue;
            const RequestState::Peer& pe = req.peers[(size_t)p];
            const uint32_t nIn = *countSlot(req.countsHost, point, p, 1), nOut = *countSlot(req.countsHost, point, p, 0);
            if (gather.n >= kMaxWalks) return xfail(PLR_ERR_UNSUPPORTED, "exchange: request lists for more than 16 peers");
            if (nIn) {
                gather.w[gather.n++] = {pe.in + (size_t)point * (pe.inWords / 2), pe.inOff + (size_t)point * (pe.inWords / 2), pe.inBase + (size_t)point * kMaxScanBlocks, (uint4*)(req.send[point].ptr + sendAt[(size_t)p]), pe.inWords / 2, req.inWordsPerRow, req.mine.x0, req.mine.y0};
                maxGather = std::max(maxGather, pe.inWords / 2);
                ops.push_back({p, true, req.send[point].ptr + sendAt[(size_t)p], (size_t)nIn * 16});
            }
            if (nOut) {
                scatter.w[scatter.n++] = {pe.out + (size_t)point * (pe.outWords / 2), pe.outOff + (size_t)point * (pe.outWords / 2), pe.outBase + (size_t)point * kMaxScanBlocks, (uint4*)(req.recv[point].ptr + recvAt[(size_t)p]), pe.outWords / 2, pe.outWordsPerRow, pe.rect.x0, pe.rect.y0};
                maxScatter = std::max(maxScatter, pe.outWords / 2);
                ops.push_back({p, false, req.recv[point].ptr + recvAt[(size_t)p], (size_t)nOut * 16});
            }
        }
        if (gather.n) {
            requestWalkKernel<true><<<dim3(walkBlocks(maxGather), (unsigned)gather.n), 256, 0, stream>>>(gather);
            if (int rc = hip(hipGetLastError(), "requestWalkKernel(gather)")) return rc;
        }
        if (int rc = transfer(id, ops, stream)) return rc;
        if (scatter.n) {
            requestWalkKernel<false><<<dim3(walkBlocks(maxScatter), (unsigned)scatter.n), 256, 0, stream>>>(scatter);
            if (int rc = hip(hipGetLastError(), "requestWalkKernel(scatter)")) return rc;
        }
        return 0;
    }

    // all transfers of exchange `id`, one ncclGroup, on `stream`
    int post(int id, hipStream_t stream) {
        if (req.ready && (id == PLRF_EXCHANGE_GI_TRACE || id == PLRF_EXCHANGE_GI_TEMPORAL)) return postRequested(id, id == PLRF_EXCHANGE_GI_TRACE ? 0 : 1, stream);
        plrf_exchange_item items[16];
        uint32_t count = 16;
        if (int rc = plrf_get_exchange_items(fp, id, items, &count)) return xfail(rc, plrf_last_error());
        count = std::min(count, 16u);
        packedRegions = tiledItems(items, count);
        if (packedRegions) {
            if (rects.empty()) return xfail(PLR_ERR_INVALID_ARGUMENT, "exchange: the pipeline renders a tile but the exchange was attached with rows only (plrf_rccl_attach_rects)");
            return postPacked(id, items, count, stream);
        }
        if (local) { // the transfers of the communicator path below, through the in-process transport: straight from / into the images
            std::vector<LocalGroup::Op> lops;
            for (uint32_t i = 0; i < count; i++) {
                const plrf_exchange_item& it = items[i];
                for (const plrf_exchange_op& o : planBandRows(frameHeight, (uint32_t)world, bounds.empty() ? nullptr : bounds.data(), (uint32_t)rank, it.image_rows, it.halo_rows, it.row_begin, it.row_end)) {
                    const size_t bytes = (size_t)(o.row_end - o.row_begin) * it.row_bytes;
                    lops.push_back({(int)o.peer, o.send != 0, (uint8_t*)it.device_ptr + (size_t)o.row_begin * it.row_bytes, bytes});
                    (o.send ? bytesSent : bytesReceived) += bytes;
                }
            }
            exchanges++;
            return localTransfer(id, lops, stream);
        }
        if (loopback) { // bands send straight from the images: nothing local to stand in for; the bytes a real exchange would move are still counted
            for (uint32_t i = 0; i < count; i++) {
                for (const plrf_exchange_op& o : planBandRows(frameHeight, (uint32_t)world, bounds.empty() ? nullptr : bounds.data(), (uint32_t)rank, items[i].image_rows, items[i].halo_rows, items[i].row_begin, items[i].row_end))
                    (o.send ? bytesSent : bytesReceived) += (uint64_t)(o.row_end - o.row_begin) * items[i].row_bytes;
            }
            exchanges++;
            return 0;
        }
        if (int rc = nccl(ncclGroupStart(), "ncclGroupStart")) return rc;
        int rc = 0;
        for (uint32_t i = 0; i < count && !rc; i++) {
            const plrf_exchange_item& it = items[i];
            const std::vector<plrf_exchange_op> ops = planBandRows(frameHeight, (uint32_t)world, bounds.empty() ? nullptr : bounds.data(), (uint32_t)rank, it.image_rows, it.halo_rows, it.row_begin, it.row_end);
            for (size_t k = 0; k < ops.size() && !rc; k++) {
                uint8_t* p = (uint8_t*)it.device_ptr + (size_t)ops[k].row_begin * it.row_bytes;
                const size_t bytes = (size_t)(ops[k].row_end - ops[k].row_begin) * it.row_bytes;
                if (ops[k].send) { rc = nccl(ncclSend(p, bytes, ncclUint8, (int)ops[k].peer, comm, stream), "ncclSend"); bytesSent += bytes; }
                else { rc = nccl(ncclRecv(p, bytes, ncclUint8, (int)ops[k].peer, comm, stream), "ncclRecv"); bytesReceived += bytes; }
            }
        }
        const int grc = nccl(ncclGroupEnd(), "ncclGroupEnd");
        exchanges++;
        return rc ? rc : grc;
    }

    int run(int idWithPhase, hipStream_t launchStream) {
        const int id = idWithPhase & PLRF_EXCHANGE_ID_MASK, phase = idWithPhase & (PLRF_EXCHANGE_BEGIN | PLRF_EXCHANGE_END);
        if (id < 0 || id >= PLRF_EXCHANGE_COUNT) return xfail(PLR_ERR_INVALID_ARGUMENT, "exchange id out of range");
        if (int rc = checkWatchdog()) return rc;
        if (id == PLRF_EXCHANGE_HISTOGRAM) {
            bytesSent = bytesReceived = exchanges = 0;
            void* ptr = nullptr;
            size_t bytes = 0;
            if (int rc = plrf_get_histogram_exchange(fp, &ptr, &bytes)) return xfail(rc, plrf_last_error());
            if (local) return localAllReduce(id, ptr, bytes, 0, launchStream);
            if (loopback) return 0;
            // 128 bin counts, each < 2^32 pixels in total: the unsigned sum is exact, so every band derives the same exposure
            if (int rc = nccl(ncclAllReduce(ptr, ptr, bytes / 4, ncclUint32, ncclSum, comm, launchStream), "ncclAllReduce(histogram)")) return rc;
            return watch(id, 0, launchStream);
        }
        if (id == PLRF_EXCHANGE_DEPTH_APEX) {
            // SURVEY 8e, collective 2: {min, max} of the bands' depth ranges -> the apex of the unpartitioned depth pyramid (min / max are exact)
            void* ptr = nullptr;
            size_t bytes = 0;
            if (int rc = plrf_get_depth_apex_exchange(fp, &ptr, &bytes)) return xfail(rc, plrf_last_error());
            if (local) return localAllReduce(id, ptr, 8, 1, launchStream);
            if (loopback) return 0;
            float* f = (float*)ptr;
            if (int rc = nccl(ncclGroupStart(), "ncclGroupStart")) return rc;
            const int r0 = nccl(ncclAllReduce(f, f, 1, ncclFloat, ncclMin, comm, launchStream), "ncclAllReduce(depth min)");
            const int r1 = nccl(ncclAllReduce(f + 1, f + 1, 1, ncclFloat, ncclMax, comm, launchStream), "ncclAllReduce(depth max)");
            const int grc = nccl(ncclGroupEnd(), "ncclGroupEnd");
            if (r0 || r1 || grc) return r0 ? r0 : (r1 ? r1 : grc);
            return watch(id, 0, launchStream);
        }
        if (id == PLRF_EXCHANGE_GI_REQUESTS) {
            // on the COMMUNICATION stream, behind the two request passes: the bitmaps travel, are scanned and counted while the launch stream goes on with the culling
            // and the trace; the responses (on the launch stream) wait for countsReady
            if (int rc = hip(hipEventRecord(ready[id], launchStream), "hipEventRecord")) return rc;
            if (int rc = hip(hipStreamWaitEvent(commStream, ready[id], 0), "hipStreamWaitEvent")) return rc;
            if (int rc = postRequests(commStream)) return rc;
            if (dog.deadlineMs) { if (int rc = hip(hipEventRecord(done[id], commStream), "hipEventRecord")) return rc; dog.arm(rank, id, PLRF_EXCHANGE_BEGIN, &RcclExchange::eventDone, &dogArgs[id]); }
            return 0;
        }
        if (phase == PLRF_EXCHANGE_BEGIN && req.ready && (id == PLRF_EXCHANGE_GI_TRACE || id == PLRF_EXCHANGE_GI_TEMPORAL)) {
            // Request lists: the sizes of the responses are the counts of the request exchange, which the HOST must have read (postRequested) - a wait of the launch
            // thread for the GPU. Here, right behind the producer's launch, the launch stream would run dry behind that wait (54 us between the trace and the filter's first
            // phase, profiles/r06c_band_timeline.txt). So the BEGIN only marks the point on the launch stream; the responses are posted from the END, which the launch thread
            // reaches with the filter's first phase - the waves that need no requested texel - already enqueued: the GPU works through it while the host waits.
            lastOverlapMode = 1;
            if (int rc = hip(hipEventRecord(ready[id], launchStream), "hipEventRecord")) return rc;
            if (int rc = hip(hipStreamWaitEvent(commStream, ready[id], 0), "hipStreamWaitEvent")) return rc;
            deferredPost[id] = true;
            return 0;
        }
        if (phase == PLRF_EXCHANGE_BEGIN) {
            // When may the transfers start? Edges-first producers (band_overlap_exchange 2): as soon as the launch that is still running has written its edge
            // rows / columns - the communication stream waits for the backend's edge signal (a word the kernel's last edge block stores, hipStreamWaitValue32),
            // NOT for the launch stream. Otherwise (split producers, or no stream memory operations on this device): an event behind the producer.
            void* signal = nullptr;
            uint32_t value = 0;
            const bool rowsFirst = plrf_band_rows_first(fp) != 0 && streamWaitValueSupported && plr_get_edge_signal(&signal, &value) == PLR_OK && signal != nullptr && value != lastSignalValue;
            if (rowsFirst) {
                lastSignalValue = value;
                lastOverlapMode = 2;
                static const bool waitInKernel = !std::getenv("PLRF_EXCHANGE_BEGIN_WAIT") || std::string(std::getenv("PLRF_EXCHANGE_BEGIN_WAIT")) != "value"; // experiment hook
                if (waitInKernel) {
                    waitForValueKernel<<<1, 64, 0, commStream>>>((const uint32_t*)signal, value, cancelWord);
                    if (int rc = hip(hipGetLastError(), "waitForValueKernel")) return rc;
                } else if (int rc = hip(hipStreamWaitValue32(commStream, signal, value, hipStreamWaitValueGte, 0xffffffffu), "hipStreamWaitValue32")) return rc;
            } else {
                lastOverlapMode = 1;
                if (int rc = hip(hipEventRecord(ready[id], launchStream), "hipEventRecord")) return rc;
                if (int rc = hip(hipStreamWaitEvent(commStream, ready[id], 0), "hipStreamWaitEvent")) return rc;
            }
            return postBehindBegin(id);
        }
        if (phase == PLRF_EXCHANGE_END) {
            if (deferredPost[id]) { deferredPost[id] = false; if (int rc = postBehindBegin(id)) return rc; }
            if (doneFlag[id]) return hip(hipStreamWaitValue32(launchStream, doneFlag[id], doneSerial[id], hipStreamWaitValueGte, 0xffffffffu), "hipStreamWaitValue32(END)");
            return hip(hipStreamWaitEvent(launchStream, done[id], 0), "hipStreamWaitEvent");
        }
        if (int rc = post(id, launchStream)) return rc;
        return watch(id, 0, launchStream);
    }
    // the transfers of exchange `id` on the communication stream (which is behind the BEGIN's wait), their completion flag / event, the watchdog's entry
    int postBehindBegin(int id) {
        if (int rc = post(id, commStream)) return rc;
        if (doneFlag[id]) if (int rc = hip(hipStreamWriteValue32(commStream, doneFlag[id], ++doneSerial[id], 0), "hipStreamWriteValue32")) return rc;
        // (the event stays: it is what the watchdog queries, and the END's order when there are no stream memory operations)
        if (!doneFlag[id] || dog.deadlineMs) if (int rc = hip(hipEventRecord(done[id], commStream), "hipEventRecord")) return rc;
        dog.arm(rank, id, PLRF_EXCHANGE_BEGIN, &RcclExchange::eventDone, &dogArgs[id]);
        return 0;
    }
    // an exchange enqueued on the launch stream itself gets no completion event by default: an event record is a barrier packet of ~6 us on the launch stream,
    // and a launch stream parked in such an exchange shows up anyway - the next overlapped exchange's producer never runs, its edge signal is never raised, and
    // that BEGIN entry goes overdue (the message says so). PLRF_EXCHANGE_WATCH_INLINE=1 records the event and names the in-line exchange itself.
    int watch(int id, int phase, hipStream_t stream) {
        static const bool watchInline = std::getenv("PLRF_EXCHANGE_WATCH_INLINE") && std::atoi(std::getenv("PLRF_EXCHANGE_WATCH_INLINE")) != 0;
        if (!watchInline || loopback || !dog.deadlineMs || world < 2) return 0;
        if (int rc = hip(hipEventRecord(done[id], stream), "hipEventRecord")) return rc;
        dog.arm(rank, id, phase, &RcclExchange::eventDone, &dogArgs[id]);
        return 0;
    }

    static int callback(void* user, int id, void* stream) { return ((RcclExchange*)user)->run(id, (hipStream_t)stream); }
};

int attachCommon(void* pipeline, const void* unique_id_128_bytes, int rank, int world, RcclExchange* x, void** out_exchange) {
    x->fp = (FramePipeline*)pipeline;
    x->rank = rank; x->world = world;
    x->loopback = unique_id_128_bytes == nullptr && !x->local;
    int rc = x->hip(hipGetDevice(&x->device), "hipGetDevice");
    if (!rc && !x->loopback && !x->local) {
        ncclUniqueId id;
        std::memcpy(&id, unique_id_128_bytes, sizeof(id));
        rc = x->nccl(ncclCommInitRank(&x->comm, world, id, rank), "ncclCommInitRank");
    }
    if (!rc) rc = x->hip(hipStreamCreateWithFlags(&x->commStream, hipStreamNonBlocking), "hipStreamCreateWithFlags");
    if (!rc) { rc = x->hip(hipHostMalloc((void**)&x->cancelWord, 64, hipHostMallocMapped), "hipHostMalloc(cancel word)"); if (!rc) *x->cancelWord = 0u; }
    for (int i = 0; i < PLRF_EXCHANGE_COUNT && !rc; i++) {
        rc = x->hip(hipEventCreateWithFlags(&x->ready[i], hipEventDisableTiming), "hipEventCreateWithFlags");
        if (!rc) rc = x->hip(hipEventCreateWithFlags(&x->done[i], hipEventDisableTiming), "hipEventCreateWithFlags");
    }
    if (!rc) {
        // stream memory operations (hipStreamWaitValue32 on the producer's edge signal): asked of the device, not assumed
        int can = 0;
        if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, x->device) != hipSuccess) { (void)hipGetLastError(); can = 0; }
        x->streamWaitValueSupported = can;
        // OFF by default: with the BEGIN's wait in a kernel the two forms of the END measure the same in the replay (slowest band 0.841 / 0.838 ms), and a value wait that
        // stays pending - an END on a real link - slows whatever runs beside it (the transfers' kernels, the asynchronous tail): profiles/r05_not_kept.txt (6)
        static const bool endByValue = std::getenv("PLRF_EXCHANGE_END_BY_VALUE") && std::atoi(std::getenv("PLRF_EXCHANGE_END_BY_VALUE")) != 0; // experiment hook: 1 = stream value wait
        for (int i = 0; can && endByValue && i < PLRF_EXCHANGE_COUNT; i++) {
            if (hipExtMallocWithFlags((void**)&x->doneFlag[i], 8, hipMallocSignalMemory) != hipSuccess) { (void)hipGetLastError(); x->doneFlag[i] = nullptr; continue; }
            if (hipMemset(x->doneFlag[i], 0, 8) != hipSuccess) { (void)hipGetLastError(); hipFree(x->doneFlag[i]); x->doneFlag[i] = nullptr; }
        }
    }
    if (rc) { delete x; return rc; }
    if (plrf_set_exchange_callback(pipeline, &RcclExchange::callback, x) != PLR_OK) { delete x; return xfail(PLR_ERR_INVALID_ARGUMENT, plrf_last_error()); }
    x->startWatchdog();
    *out_exchange = x;
    return PLR_OK;
}

} // namespace

extern "C" {

const char* plrf_rccl_last_error(void) { return g_xerr.c_str(); }

int plrf_band_rows(uint32_t frame_height, uint32_t n_bands, uint32_t band, uint32_t* out_row_begin, uint32_t* out_row_end) {
    if (!out_row_begin || !out_row_end || n_bands == 0 || band >= n_bands) return xfail(PLR_ERR_INVALID_ARGUMENT, "plrf_band_rows: invalid argument");
    if (n_bands > (frame_height + kBandAlignment - 1) / kBandAlignment) return xfail(PLR_ERR_INVALID_ARGUMENT, "plrf_band_rows: more bands than 64-row tiles");
    bandRowsOf(frame_height, n_bands, band, out_row_begin, out_row_end);
    return PLR_OK;
}

static int checkBounds(uint32_t frame_height, uint32_t n_bands, const uint32_t* row_bounds, const char* who) {
    if (!row_bounds) return PLR_OK;
    if (row_bounds[0] != 0 || row_bounds[n_bands] != frame_height) return xfail(PLR_ERR_INVALID_ARGUMENT, std::string(who) + ": row_bounds must start at 0 and end at frame_height");
    for (uint32_t b = 0; b < n_bands; b++) {
        if (row_bounds[b + 1] <= row_bounds[b]) return xfail(PLR_ERR_INVALID_ARGUMENT, std::string(who) + ": row_bounds must be strictly increasing");
        if (b + 1 < n_bands && row_bounds[b + 1] % kBandAlignment) return xfail(PLR_ERR_INVALID_ARGUMENT, std::string(who) + ": interior row_bounds must be multiples of 64");
    }
    return PLR_OK;
}

int plrf_exchange_plan_rows(uint32_t frame_height, uint32_t n_bands, const uint32_t* row_bounds, uint32_t band, uint32_t image_rows, uint32_t halo_rows, uint32_t row_begin,
                            uint32_t row_end, plrf_exchange_op* out_ops, uint32_t* out_count) {
    if (!out_ops || !out_count || n_bands == 0 || band >= n_bands || image_rows == 0) return xfail(PLR_ERR_INVALID_ARGUMENT, "plrf_exchange_plan: invalid argument");
    if (int rc = checkBounds(frame_height, n_bands, row_bounds, "plrf_exchange_plan_rows")) return rc;
    *out_count = planItem(frame_height, n_bands, row_bounds, band, image_rows, halo_rows, row_begin, row_end, out_ops);
    return PLR_OK;
}
int plrf_exchange_plan_all_bands(uint32_t frame_height, uint32_t n_bands, const uint32_t* row_bounds, uint32_t band, uint32_t image_rows, uint32_t halo_rows, uint32_t row_begin,
                                 uint32_t row_end, plrf_exchange_op* out_ops, uint32_t capacity, uint32_t* out_count) {
    if (!out_ops || !out_count || n_bands == 0 || band >= n_bands || image_rows == 0) return xfail(PLR_ERR_INVALID_ARGUMENT, "plrf_exchange_plan_all_bands: invalid argument");
    if (int rc = checkBounds(frame_height, n_bands, row_bounds, "plrf_exchange_plan_all_bands")) return rc;
    const std::vector<plrf_exchange_op> ops = planBandRows(frame_height, n_bands, row_bounds, band, image_rows, halo_rows, row_begin, row_end);
    for (size_t i = 0; i < ops.size() && i < capacity; i++) out_ops[i] = ops[i];
    *out_count = (uint32_t)ops.size();
    return PLR_OK;
}
int plrf_exchange_plan(uint32_t frame_height, uint32_t n_bands, uint32_t band, uint32_t image_rows, uint32_t halo_rows, uint32_t row_begin, uint32_t row_end,
                       plrf_exchange_op* out_ops, uint32_t* out_count) {
    return plrf_exchange_plan_rows(frame_height, n_bands, nullptr, band, image_rows, halo_rows, row_begin, row_end, out_ops, out_count);
}

int plrf_exchange_plan_rects(uint32_t frame_width, uint32_t frame_height, uint32_t world, const uint32_t* rects, uint32_t rank, uint32_t image_cols, uint32_t image_rows,
                             uint32_t halo, plrf_rect_op* out_ops, uint32_t capacity, uint32_t* out_count) {
    if (!out_count || (!out_ops && capacity) || rank >= world || image_cols == 0 || image_rows == 0) return xfail(PLR_ERR_INVALID_ARGUMENT, "plrf_exchange_plan_rects: invalid argument");
    if (int rc = checkRects(frame_width, frame_height, world, rects, "plrf_exchange_plan_rects")) return rc;
    *out_count = planRects(frame_width, frame_height, world, (const Rect*)rects, rank, image_cols, image_rows, halo, out_ops, capacity);
    return PLR_OK;
}

int plrf_tile_rects(uint32_t frame_width, uint32_t frame_height, uint32_t gx, uint32_t gy, const uint32_t* col_bounds, const uint32_t* row_bounds, uint32_t* out_rects) {
    if (!out_rects || gx == 0 || gy == 0) return xfail(PLR_ERR_INVALID_ARGUMENT, "plrf_tile_rects: invalid argument");
    if (gx > (frame_width + kBandAlignment - 1) / kBandAlignment || gy > (frame_height + kBandAlignment - 1) / kBandAlignment)
        return xfail(PLR_ERR_INVALID_ARGUMENT, "plrf_tile_rects: more tiles than 64-pixel cells");
    if (int rc = checkBounds(frame_width, gx, col_bounds, "plrf_tile_rects (columns)")) return rc;
    if (int rc = checkBounds(frame_height, gy, row_bounds, "plrf_tile_rects (rows)")) return rc;
    for (uint32_t ty = 0; ty < gy; ty++)
        for (uint32_t tx = 0; tx < gx; tx++) {
            uint32_t* q = out_rects + 4 * (ty * gx + tx);
            if (col_bounds) { q[0] = col_bounds[tx]; q[2] = col_bounds[tx + 1]; } else bandRowsOf(frame_width, gx, tx, &q[0], &q[2]);
            if (row_bounds) { q[1] = row_bounds[ty]; q[3] = row_bounds[ty + 1]; } else bandRowsOf(frame_height, gy, ty, &q[1], &q[3]);
        }
    return PLR_OK;
}

int plrf_rccl_get_unique_id(void* out_128_bytes) {
    static_assert(sizeof(ncclUniqueId) == PLRF_RCCL_UNIQUE_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    const ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return xfail(PLR_ERR_HIP, std::string("ncclGetUniqueId: ") + ncclGetErrorString(r));
    std::memcpy(out_128_bytes, &id, sizeof(id));
    return PLR_OK;
}

int plrf_rccl_attach(void* pipeline, const void* unique_id_128_bytes, int rank, int world, uint32_t frame_height, void** out_exchange) {
    return plrf_rccl_attach_rows(pipeline, unique_id_128_bytes, rank, world, frame_height, nullptr, out_exchange);
}

int plrf_rccl_attach_rows(void* pipeline, const void* unique_id_128_bytes, int rank, int world, uint32_t frame_height, const uint32_t* row_bounds, void** out_exchange) {
    if (!pipeline || !unique_id_128_bytes || !out_exchange || world < 1 || rank < 0 || rank >= world) return xfail(PLR_ERR_INVALID_ARGUMENT, "plrf_rccl_attach: invalid argument");
    if (int rc = checkBounds(frame_height, (uint32_t)world, row_bounds, "plrf_rccl_attach_rows")) return rc;
    RcclExchange* x = new RcclExchange();
    x->frameHeight = frame_height;
    x->frameWidth = ((FramePipeline*)pipeline)->settings.width;
    if (row_bounds) x->bounds.assign(row_bounds, row_bounds + world + 1);
    return attachCommon(pipeline, unique_id_128_bytes, rank, world, x, out_exchange);
}

int plrf_rccl_attach_rects(void* pipeline, const void* unique_id_128_bytes, int rank, int world, uint32_t frame_width, uint32_t frame_height, const uint32_t* rects,
                           void** out_exchange) {
    if (!pipeline || !out_exchange || world < 1 || rank < 0 || rank >= world) return xfail(PLR_ERR_INVALID_ARGUMENT, "plrf_rccl_attach_rects: invalid argument");
    if (int rc = checkRects(frame_width, frame_height, (uint32_t)world, rects, "plrf_rccl_attach_rects")) return rc;
    const FramePipeline* fp = (const FramePipeline*)pipeline;
    const BandSettings& b = fp->settings.band;
    const uint32_t* mine = rects + 4 * rank;
    const uint32_t c0 = b.tiled() ? b.colBegin : 0u, c1 = b.tiled() ? b.colEnd : fp->settings.width;
    if (!b.enabled() || mine[0] != c0 || mine[2] != c1 || mine[1] != b.rowBegin || mine[3] != b.rowEnd || fp->settings.width != frame_width || fp->settings.height != frame_height)
        return xfail(PLR_ERR_INVALID_ARGUMENT, "plrf_rccl_attach_rects: the pipeline was not created with this rank's rectangle");
    RcclExchange* x = new RcclExchange();
    x->local = g_pendingLocalGroup; // plrf_local_attach_rects: the in-process transport takes the communicator's place
    x->frameWidth = frame_width; x->frameHeight = frame_height;
    x->rects.assign((const Rect*)rects, (const Rect*)rects + world);
    // (whole-row rectangles: the band path needs the row boundaries, which the rectangles carry when they are listed top to bottom)
    bool rowsOnly = true;
    for (const Rect& r : x->rects) rowsOnly = rowsOnly && r.x0 == 0 && r.x1 == frame_width;
    if (rowsOnly) {
        for (int r = 0; r < world; r++) if ((r == 0 ? 0u : x->rects[(size_t)r - 1].y1) != x->rects[(size_t)r].y0) { delete x; return xfail(PLR_ERR_INVALID_ARGUMENT, "plrf_rccl_attach_rects: bands must be listed top to bottom"); }
        x->bounds.push_back(0);
        for (const Rect& r : x->rects) x->bounds.push_back(r.y1);
    }
    return attachCommon(pipeline, unique_id_128_bytes, rank, world, x, out_exchange);
}

// ---- the in-process transport (see LocalGroup above)
int plrf_local_group_create(int world, void** out_group) {
    if (!out_group || world < 1 || world > 64) return xfail(PLR_ERR_INVALID_ARGUMENT, "plrf_local_group_create: invalid argument");
    LocalGroup* g = new LocalGroup();
    g->world = world;
    g->posts.assign((size_t)world, std::vector<LocalGroup::Post>(PLRF_EXCHANGE_COUNT));
    g->reduces.assign((size_t)world, std::vector<LocalGroup::Reduce>(PLRF_EXCHANGE_COUNT));
    if (const char* ms = std::getenv("PLRF_LOCAL_EXCHANGE_TIMEOUT_MS")) g->timeoutMs = (uint32_t)std::max(1, std::atoi(ms));
    *out_group = g;
    return PLR_OK;
}
int plrf_local_group_destroy(void* group) { delete (LocalGroup*)group; return PLR_OK; }
int plrf_local_group_abort(void* group) {
    if (!group) return PLR_OK;
    LocalGroup* g = (LocalGroup*)group;
    std::lock_guard<std::mutex> lock(g->m);
    g->aborted = true;
    g->cv.notify_all();
    return PLR_OK;
}
int plrf_local_group_freeze(void* group, int frozen) {
    if (!group) return xfail(PLR_ERR_INVALID_ARGUMENT, "plrf_local_group_freeze: null group");
    LocalGroup* g = (LocalGroup*)group;
    std::lock_guard<std::mutex> lock(g->m);
    g->frozen = frozen != 0;
    g->cv.notify_all();
    return PLR_OK;
}
int plrf_local_attach_rects(void* pipeline, void* group, int rank, int world, uint32_t frame_width, uint32_t frame_height, const uint32_t* rects, void** out_exchange) {
    if (!group) return xfail(PLR_ERR_INVALID_ARGUMENT, "plrf_local_attach_rects: null group");
    LocalGroup* g = (LocalGroup*)group;
    if (world != g->world) return xfail(PLR_ERR_INVALID_ARGUMENT, "plrf_local_attach_rects: the group was created for another number of ranks");
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return xfail(PLR_ERR_HIP, "hipGetDevice");
    {
        std::lock_guard<std::mutex> lock(g->m);
        if (g->device < 0) g->device = device;
        if (g->device != device) return xfail(PLR_ERR_UNSUPPORTED, "plrf_local_attach_rects: the in-process transport copies device to device on ONE GPU; ranks on several GPUs use the communicator (plrf_rccl_attach_rects)");
        if (!g->slots) {
            const size_t slotBytes = 2 * (size_t)PLRF_EXCHANGE_COUNT * (size_t)world * LocalGroup::kSlotBytes;
            if (hipMalloc((void**)&g->slots, slotBytes) != hipSuccess) return xfail(PLR_ERR_HIP, "hipMalloc(all-reduce slots)");
            if (hipMemset(g->slots, 0, slotBytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return xfail(PLR_ERR_HIP, "hipMemset(all-reduce slots)");
        }
    }
    void* exchange = nullptr;
    // the rectangle checks and the band bookkeeping of the communicator path; the group takes the communicator's place
    g_pendingLocalGroup = g;
    const int rc = plrf_rccl_attach_rects(pipeline, nullptr, rank, world, frame_width, frame_height, rects, &exchange);
    g_pendingLocalGroup = nullptr;
    if (rc) return rc;
    *out_exchange = exchange;
    return PLR_OK;
}

int plrf_rccl_detach(void* pipeline, void* exchange) {
    if (pipeline) plrf_set_exchange_callback(pipeline, nullptr, nullptr);
    delete (RcclExchange*)exchange;
    return PLR_OK;
}

int plrf_rccl_get_stats(void* exchange, uint64_t* out_bytes_sent, uint64_t* out_bytes_received, uint64_t* out_exchanges) {
    if (!exchange) return xfail(PLR_ERR_INVALID_ARGUMENT, "plrf_rccl_get_stats: null exchange");
    const RcclExchange* x = (const RcclExchange*)exchange;
    if (out_bytes_sent) *out_bytes_sent = x->bytesSent;
    if (out_bytes_received) *out_bytes_received = x->bytesReceived;
    if (out_exchanges) *out_exchanges = x->exchanges;
    return PLR_OK;
}

int plrf_rccl_get_info(void* exchange, plrf_rccl_info* out) {
    if (!exchange || !out) return xfail(PLR_ERR_INVALID_ARGUMENT, "plrf_rccl_get_info: null argument");
    RcclExchange* x = (RcclExchange*)exchange;
    std::memset(out, 0, sizeof(*out));
    int count = 0, version = 0;
    if (x->comm && ncclCommCount(x->comm, &count) != ncclSuccess) count = -1;
    (void)ncclGetVersion(&version);
    out->rccl_ranks = count; out->rccl_version = version; out->overlap_mode = x->lastOverlapMode; out->packed_regions = x->packedRegions ? 1 : 0;
    out->stream_wait_value_supported = x->streamWaitValueSupported; out->watchdog_ms = (int32_t)x->dog.deadlineMs;
    return PLR_OK;
}

// Moves rows [src_row, src_row + rows) of `image` onto rows [dst_row, dst_row + rows) of the same image with an ncclSend / ncclRecv pair
// addressed to this rank itself, through the same group / stream / event sequence as a BEGIN + END exchange. A one-GPU box cannot
// host two ranks (RCCL refuses duplicate devices), so this is how the transport's mechanics are exercised there.
int plrf_rccl_self_test(void* exchange, void* device_ptr, uint32_t row_bytes, uint32_t src_row, uint32_t dst_row, uint32_t rows, void* launch_stream) {
    if (!exchange || !device_ptr) return xfail(PLR_ERR_INVALID_ARGUMENT, "plrf_rccl_self_test: null argument");
    RcclExchange* x = (RcclExchange*)exchange;
    if (!x->comm) return xfail(PLR_ERR_INVALID_ARGUMENT, "plrf_rccl_self_test: a loopback exchange has no communicator");
    hipStream_t ls = (hipStream_t)launch_stream;
    const size_t bytes = (size_t)rows * row_bytes;
    if (int rc = x->hip(hipEventRecord(x->ready[1], ls), "hipEventRecord")) return rc;
    if (int rc = x->hip(hipStreamWaitEvent(x->commStream, x->ready[1], 0), "hipStreamWaitEvent")) return rc;
    if (int rc = x->nccl(ncclGroupStart(), "ncclGroupStart")) return rc;
    int rc = x->nccl(ncclSend((uint8_t*)device_ptr + (size_t)src_row * row_bytes, bytes, ncclUint8, x->rank, x->comm, x->commStream), "ncclSend(self)");
    if (!rc) rc = x->nccl(ncclRecv((uint8_t*)device_ptr + (size_t)dst_row * row_bytes, bytes, ncclUint8, x->rank, x->comm, x->commStream), "ncclRecv(self)");
    const int grc = x->nccl(ncclGroupEnd(), "ncclGroupEnd");
    if (rc || grc) return rc ? rc : grc;
    if (int rc2 = x->hip(hipEventRecord(x->done[1], x->commStream), "hipEventRecord")) return rc2;
    return x->hip(hipStreamWaitEvent(ls, x->done[1], 0), "hipStreamWaitEvent");
}

int plrf_rccl_self_test_rect(void* exchange, void* device_ptr, uint32_t pitch_bytes, uint32_t texel_bytes, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t dst_x,
                             uint32_t dst_y, void* launch_stream) {
    if (!exchange || !device_ptr || x1 <= x0 || y1 <= y0 || (texel_bytes != 4 && texel_bytes != 8 && texel_bytes != 16)) return xfail(PLR_ERR_INVALID_ARGUMENT, "plrf_rccl_self_test_rect: invalid argument");
    RcclExchange* x = (RcclExchange*)exchange;
    hipStream_t ls = (hipStream_t)launch_stream;
    const size_t bytes = (size_t)(x1 - x0) * texel_bytes * (y1 - y0);
    if (int rc = x->arena(x->sendArena[1], bytes)) return rc;
    if (int rc = x->arena(x->recvArena[1], bytes)) return rc;
    auto region = [&](uint32_t rx, uint32_t ry, uint8_t* buffer) {
        CopyRegion r;
        r.image = (uint8_t*)device_ptr + (size_t)ry * pitch_bytes + (size_t)rx * texel_bytes; r.buffer = buffer;
        r.pitch = pitch_bytes; r.widthBytes = (x1 - x0) * texel_bytes; r.rows = y1 - y0;
        r.unit = ((uintptr_t)r.image % 16 == 0 && r.pitch % 16 == 0 && r.widthBytes % 16 == 0) ? 16u : 4u;
        return r;
    };
    if (int rc = x->hip(hipEventRecord(x->ready[1], ls), "hipEventRecord")) return rc;
    if (int rc = x->hip(hipStreamWaitEvent(x->commStream, x->ready[1], 0), "hipStreamWaitEvent")) return rc;
    if (int rc = x->launchCopies<true>({region(x0, y0, x->sendArena[1].ptr)}, x->commStream)) return rc;
    if (x->comm) {
        if (int rc = x->nccl(ncclGroupStart(), "ncclGroupStart")) return rc;
        int rc = x->nccl(ncclSend(x->sendArena[1].ptr, bytes, ncclUint8, x->rank, x->comm, x->commStream), "ncclSend(self)");
        if (!rc) rc = x->nccl(ncclRecv(x->recvArena[1].ptr, bytes, ncclUint8, x->rank, x->comm, x->commStream), "ncclRecv(self)");
        const int grc = x->nccl(ncclGroupEnd(), "ncclGroupEnd");
        if (rc || grc) return rc ? rc : grc;
    } else if (int rc = x->hip(hipMemcpyAsync(x->recvArena[1].ptr, x->sendArena[1].ptr, bytes, hipMemcpyDeviceToDevice, x->commStream), "hipMemcpyAsync(loopback)")) return rc;
    if (int rc = x->launchCopies<false>({region(dst_x, dst_y, x->recvArena[1].ptr)}, x->commStream)) return rc;
    if (int rc = x->hip(hipEventRecord(x->done[1], x->commStream), "hipEventRecord")) return rc;
    return x->hip(hipStreamWaitEvent(ls, x->done[1], 0), "hipStreamWaitEvent");
}

// ---- the watchdog with a caller-supplied completion query (no GPU needed: tests/test_bands.py)
int plrf_watchdog_create(uint32_t deadline_ms, void** out_watchdog) {
    if (!out_watchdog) return xfail(PLR_ERR_INVALID_ARGUMENT, "plrf_watchdog_create: null argument");
    Watchdog* w = new Watchdog();
    w->deadlineMs = deadline_ms;
    w->graceArms = 0; // the bare mechanism: every arm has the stated deadline (the native exchange gives its first arms PLRF_EXCHANGE_WATCHDOG_FIRST_MS)
    *out_watchdog = w;
    return PLR_OK;
}
int plrf_watchdog_destroy(void* watchdog) { delete (Watchdog*)watchdog; return PLR_OK; }
int plrf_watchdog_arm(void* watchdog, int rank, int exchange_id, int phase, plrf_watchdog_query query, void* user) {
    if (!watchdog || !query) return xfail(PLR_ERR_INVALID_ARGUMENT, "plrf_watchdog_arm: null argument");
    ((Watchdog*)watchdog)->arm(rank, exchange_id, phase, query, user);
    return PLR_OK;
}
int plrf_watchdog_poll(void* watchdog, char* out_message, size_t capacity) {
    if (!watchdog) return xfail(PLR_ERR_INVALID_ARGUMENT, "plrf_watchdog_poll: null watchdog");
    std::string msg;
    const bool overdue = ((Watchdog*)watchdog)->poll(&msg);
    if (out_message && capacity) {
        const size_t n = std::min(capacity - 1, msg.size());
        std::memcpy(out_message, msg.data(), n);
        out_message[n] = 0;
    }
    return overdue ? 1 : 0;
}

} // extern "C"
