// Halo exchange of band rendering over RCCL, in the C++ host ("RCCL over xGMI only for the TAA / denoise / bloom halo rows").
//
// One process per GPU renders one band of screen rows (frame_pipeline.h, BandSettings). Where a pass reads rows a neighbouring band
// produced, FramePipeline calls its exchange callback from inside plr_render_frame, in pass order. This file IS that callback when
// plrf_rccl_attach() has been called: every exchange is a group of ncclSend / ncclRecv with the band above and the band below (one
// direct xGMI link per neighbour - no ring, no all-gather), the luminance histogram is one 512-byte ncclAllReduce. There is no Python
// and no torch in the frame loop; the launcher only hands every rank the ncclUniqueId once at start-up.
//
// Ordering. Exchanges with a BEGIN / END phase (band_overlap_exchange) run on a communication stream: BEGIN records an event on the
// launch stream (the producer's edge rows are already queued there), the communication stream waits for it and carries the transfers;
// the producer's interior rows, recorded next, run beside them; END makes the launch stream wait for the transfers' completion event.
// Exchanges without a phase and the histogram all-reduce are enqueued on the launch stream itself (in order, nothing to wait for).
#include <algorithm>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include "../../../include/plr_frame.h"
#include "frame_pipeline.h"

using namespace plrhost;

namespace {

thread_local std::string g_xerr;
int xfail(int code, const std::string& msg) { g_xerr = msg; return code; }

constexpr uint32_t kBandAlignment = 64; // rows; band edges never split a HiZ / culling / histogram tile or the coarsest bloom texel

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

struct RcclExchange {
    FramePipeline* fp = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    uint32_t frameHeight = 0;
    std::vector<uint32_t> bounds; // world + 1 row boundaries, or empty: the equal partition
    hipStream_t commStream = nullptr;
    uint32_t lastSignalValue = 0; // the edge signal value the previous BEGIN waited for: a BEGIN whose producer raised no new one orders behind the launch stream
    hipEvent_t ready[PLRF_EXCHANGE_COUNT] = {}, done[PLRF_EXCHANGE_COUNT] = {};
    uint64_t bytesSent = 0, bytesReceived = 0, exchanges = 0; // of the last frame (reset by the histogram exchange, the first of a frame)

    ~RcclExchange() {
        if (comm) ncclCommDestroy(comm);
        for (auto e : ready) if (e) hipEventDestroy(e);
        for (auto e : done) if (e) hipEventDestroy(e);
        if (commStream) hipStreamDestroy(commStream);
    }

    int nccl(ncclResult_t r, const char* what) { return r == ncclSuccess ? 0 : xfail(PLR_ERR_HIP, std::string(what) + ": " + ncclGetErrorString(r)); }
    int hip(hipError_t e, const char* what) { return e == hipSuccess ? 0 : xfail(PLR_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e)); }

    // all transfers of exchange `id`, one ncclGroup, on `stream`
    int post(int id, hipStream_t stream) {
        plrf_exchange_item items[16];
        uint32_t count = 16;
        if (int rc = plrf_get_exchange_items(fp, id, items, &count)) return xfail(rc, plrf_last_error());
        count = std::min(count, 16u);
        if (int rc = nccl(ncclGroupStart(), "ncclGroupStart")) return rc;
        int rc = 0;
        for (uint32_t i = 0; i < count && !rc; i++) {
            const plrf_exchange_item& it = items[i];
            plrf_exchange_op ops[4];
            const uint32_t n = planItem(frameHeight, (uint32_t)world, bounds.empty() ? nullptr : bounds.data(), (uint32_t)rank, it.image_rows, it.halo_rows, it.row_begin, it.row_end, ops);
            for (uint32_t k = 0; k < n && !rc; k++) {
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
        if (id == PLRF_EXCHANGE_HISTOGRAM) {
            bytesSent = bytesReceived = exchanges = 0;
            void* ptr = nullptr;
            size_t bytes = 0;
            if (int rc = plrf_get_histogram_exchange(fp, &ptr, &bytes)) return xfail(rc, plrf_last_error());
            // 128 bin counts, each < 2^32 pixels in total: the unsigned sum is exact, so every band derives the same exposure
            return nccl(ncclAllReduce(ptr, ptr, bytes / 4, ncclUint32, ncclSum, comm, launchStream), "ncclAllReduce(histogram)");
        }
        if (id == PLRF_EXCHANGE_DEPTH_APEX) {
            // SURVEY 8e, collective 2: {min, max} of the bands' depth ranges -> the apex of the unpartitioned depth pyramid (min / max are exact)
            void* ptr = nullptr;
            size_t bytes = 0;
            if (int rc = plrf_get_depth_apex_exchange(fp, &ptr, &bytes)) return xfail(rc, plrf_last_error());
            float* f = (float*)ptr;
            if (int rc = nccl(ncclGroupStart(), "ncclGroupStart")) return rc;
            const int r0 = nccl(ncclAllReduce(f, f, 1, ncclFloat, ncclMin, comm, launchStream), "ncclAllReduce(depth min)");
            const int r1 = nccl(ncclAllReduce(f + 1, f + 1, 1, ncclFloat, ncclMax, comm, launchStream), "ncclAllReduce(depth max)");
            const int grc = nccl(ncclGroupEnd(), "ncclGroupEnd");
            return r0 ? r0 : (r1 ? r1 : grc);
        }
        if (phase == PLRF_EXCHANGE_BEGIN) {
            // When may the transfers start? Rows-first producers (band_overlap_exchange 2): as soon as the launch that is still running has written its edge
            // rows - the communication stream waits for the backend's edge signal (a word the kernel's last edge wave stores, hipStreamWaitValue32),
            // NOT for the launch stream. Otherwise (split producers, or no stream memory operations): an event behind the edge launch.
            void* signal = nullptr;
            uint32_t value = 0;
            const bool rowsFirst = plrf_band_rows_first(fp) != 0 && plr_get_edge_signal(&signal, &value) == PLR_OK && signal != nullptr && value != lastSignalValue;
            if (rowsFirst) {
                lastSignalValue = value;
                if (int rc = hip(hipStreamWaitValue32(commStream, signal, value, hipStreamWaitValueGte, 0xffffffffu), "hipStreamWaitValue32")) return rc;
            } else {
                if (int rc = hip(hipEventRecord(ready[id], launchStream), "hipEventRecord")) return rc;
                if (int rc = hip(hipStreamWaitEvent(commStream, ready[id], 0), "hipStreamWaitEvent")) return rc;
            }
            if (int rc = post(id, commStream)) return rc;
            return hip(hipEventRecord(done[id], commStream), "hipEventRecord");
        }
        if (phase == PLRF_EXCHANGE_END) return hip(hipStreamWaitEvent(launchStream, done[id], 0), "hipStreamWaitEvent");
        return post(id, launchStream);
    }

    static int callback(void* user, int id, void* stream) { return ((RcclExchange*)user)->run(id, (hipStream_t)stream); }
};

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
int plrf_exchange_plan(uint32_t frame_height, uint32_t n_bands, uint32_t band, uint32_t image_rows, uint32_t halo_rows, uint32_t row_begin, uint32_t row_end,
                       plrf_exchange_op* out_ops, uint32_t* out_count) {
    return plrf_exchange_plan_rows(frame_height, n_bands, nullptr, band, image_rows, halo_rows, row_begin, row_end, out_ops, out_count);
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
    x->fp = (FramePipeline*)pipeline;
    x->rank = rank; x->world = world; x->frameHeight = frame_height;
    if (row_bounds) x->bounds.assign(row_bounds, row_bounds + world + 1);
    ncclUniqueId id;
    std::memcpy(&id, unique_id_128_bytes, sizeof(id));
    int rc = x->nccl(ncclCommInitRank(&x->comm, world, id, rank), "ncclCommInitRank");
    if (!rc) rc = x->hip(hipStreamCreateWithFlags(&x->commStream, hipStreamNonBlocking), "hipStreamCreateWithFlags");
    for (int i = 0; i < PLRF_EXCHANGE_COUNT && !rc; i++) {
        rc = x->hip(hipEventCreateWithFlags(&x->ready[i], hipEventDisableTiming), "hipEventCreateWithFlags");
        if (!rc) rc = x->hip(hipEventCreateWithFlags(&x->done[i], hipEventDisableTiming), "hipEventCreateWithFlags");
    }
    if (rc) { delete x; return rc; }
    if (plrf_set_exchange_callback(pipeline, &RcclExchange::callback, x) != PLR_OK) { delete x; return xfail(PLR_ERR_INVALID_ARGUMENT, plrf_last_error()); }
    *out_exchange = x;
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

// Moves rows [src_row, src_row + rows) of `image` onto rows [dst_row, dst_row + rows) of the same image with an ncclSend / ncclRecv pair
// addressed to this rank itself, through the same group / stream / event sequence as a BEGIN + END exchange. A one-GPU box cannot
// host two ranks (RCCL refuses duplicate devices), so this is how the transport's mechanics are exercised there.
int plrf_rccl_self_test(void* exchange, void* device_ptr, uint32_t row_bytes, uint32_t src_row, uint32_t dst_row, uint32_t rows, void* launch_stream) {
    if (!exchange || !device_ptr) return xfail(PLR_ERR_INVALID_ARGUMENT, "plrf_rccl_self_test: null argument");
    RcclExchange* x = (RcclExchange*)exchange;
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

} // extern "C"
