// HIP backend runtime behind the C-ABI of include/plr.h.
//
// Replaces the Vulkan body of the reference's RenderBackend (Plain/src/Runtime/Rendering/Backend/
// RenderBackend.cpp) for compute passes: images and buffers are plain HBM allocations, a "pass" is a
// precompiled HIP kernel selected by the shader file name, and a frame is the recorded list of
// executions launched in order on one HIP stream (in-order execution gives the reference's
// write->read barrier rule, RenderBackend.cpp:632-767, for free).
#include "backend.h"
#include "kernels_fast/pcf_taps.h"

#include <dlfcn.h>
#include <deque>
#include <algorithm>
#include <chrono>
#include <mutex>
#include <cstdlib>
#include <atomic>
#include <set>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>

#include "../../include/plr.h"

namespace plr {

// ---------------------------------------------------------------- registry
struct ShaderEntry { std::string name; LaunchFn fn = nullptr; LaunchFn fast = nullptr; };
// shaders whose kernels index the global texture array (set 2): they may read any non-transient image
static bool shaderReadsBindless(const std::string& shader) {
    return shader == "deferredShading.comp" || shader == "sdfDiffuseTrace.comp" || shader == "sdfDebugVisualisation.comp";
}
// storage-image bindings a shader only reads (imageLoad): recorded as reads, so that readers of the same image are not ordered against each other
// lightMatrix.comp:57-138 loads the apex texel of the depth pyramid through binding 1 and writes the cascade buffer only
// shaders whose launchers honour dispatch_base[0] (plr.h): the passes a tile renderer restricts to the columns of its tile
static bool shaderTakesColumns(const std::string& shader) {
    static const char* const names[] = {"histogramPerTile.comp", "histogramCombineTiles.comp", "depthHiZPyramid.comp", "depthPyramidApex.comp", "depthDownscale.comp",
                                        "sdfCameraTileCulling.comp", "sdfDiffuseTrace.comp", "filterIndirectDiffuseSpatial.comp", "filterIndirectDiffuseTemporal.comp",
                                        "indirectLightUpscale.comp", "deferredShading.comp", "temporalFilter.comp", "bloomDownsample.comp", "bloomUpsample.comp",
                                        "applyBloom.comp", "tonemapping.comp", "giSampleRequests.comp"};
    for (const char* n : names) if (shader == n) return true;
    return false;
}
static bool storageBindingIsReadOnly(const std::string& shader, uint32_t binding) { return shader == "lightMatrix.comp" && binding == 1; }
// The registry is written by static initialisers: libplr.so's own before anything runs, libplr_exact.so's when a thread loads it - possibly while another thread
// (one backend per host thread) looks a shader up. A deque keeps entries in place; every access takes the lock.
static std::deque<ShaderEntry>& registry() {
    static std::deque<ShaderEntry> r;
    return r;
}
static std::mutex& registryMutex() {
    static std::mutex m;
    return m;
}
ShaderRegistrar::ShaderRegistrar(const char* name, LaunchFn fn, bool fast) {
    std::lock_guard<std::mutex> lock(registryMutex());
    for (auto& e : registry())
        if (e.name == name) { (fast ? e.fast : e.fn) = fn; return; }
    ShaderEntry e;
    e.name = name;
    (fast ? e.fast : e.fn) = fn;
    registry().push_back(e);
}

struct FusionEntry {
    std::string label;
    std::vector<std::string> shaders;
    FusedLaunchFn fn;
    bool writesSignatures = false;
    bool takesFills = false; // handles PassCtx::pendingFillSlot of its first execution itself (backend.h)
    EarlyLaunchFn early = nullptr; // the sequence's early part (backend.h EarlyPart), or null
    std::string earlyLabel;
};
static std::vector<FusionEntry>& fusions() {
    static std::vector<FusionEntry> r;
    return r;
}
FusionRegistrar::FusionRegistrar(const char* label, std::initializer_list<const char*> shaders, FusedLaunchFn fn, bool writesSignatures, bool takesFills) {
    FusionEntry e;
    e.label = label;
    e.writesSignatures = writesSignatures;
    e.takesFills = takesFills;
    for (const char* sname : shaders) e.shaders.push_back(sname);
    e.fn = fn;
    // longer sequences first: a chain of three is tried before a pair that is its prefix
    auto& list = fusions();
    auto pos = list.begin();
    while (pos != list.end() && pos->shaders.size() >= e.shaders.size()) ++pos;
    list.insert(pos, std::move(e));
}

EarlyPartRegistrar::EarlyPartRegistrar(FusedLaunchFn fused, EarlyLaunchFn early, const char* label) {
    for (FusionEntry& e : fusions()) if (e.fn == fused) { e.early = early; e.earlyLabel = label ? label : "early part"; }
}

struct ConsumerLink { std::string producer, consumer; };
static std::vector<ConsumerLink>& consumerLinks() {
    static std::vector<ConsumerLink> r;
    return r;
}
ConsumerLinkRegistrar::ConsumerLinkRegistrar(const char* producerShader, const char* consumerShader) { consumerLinks().push_back({producerShader, consumerShader}); }

// ---------------------------------------------------------------- PassCtx helpers
const SpecConstant* PassCtx::findSpec(uint32_t location) const {
    if (!spec) return nullptr;
    for (const auto& s : *spec)
        if (s.location == location) return &s;
    return nullptr;
}
int32_t PassCtx::specInt(uint32_t location, int32_t def) const {
    const SpecConstant* s = findSpec(location);
    if (!s || s->data.empty()) return def;
    if (s->data.size() >= 4) { int32_t v; std::memcpy(&v, s->data.data(), 4); return v; }
    return (int32_t)s->data[0];
}
float PassCtx::specFloat(uint32_t location, float def) const {
    const SpecConstant* s = findSpec(location);
    if (!s || s->data.size() < 4) return def;
    float v; std::memcpy(&v, s->data.data(), 4); return v;
}
bool PassCtx::specBool(uint32_t location, bool def) const {
    const SpecConstant* s = findSpec(location);
    if (!s || s->data.empty()) return def;
    // host passes sizeof(bool) == 1 byte (Backend/VulkanShader.cpp:4-23); accept 4-byte VkBool32 too
    if (s->data.size() >= 4) { uint32_t v; std::memcpy(&v, s->data.data(), 4); return v != 0; }
    return s->data[0] != 0;
}
int PassCtx::fail(int code, const std::string& msg) const {
    if (err) *err = msg;
    return code;
}
static const char* formatName(int f) {
    static const char* names[] = {"R8", "RG8", "RGBA8", "R16_sFloat", "RG16_sFloat", "RG32_sFloat", "RG16_sNorm", "RGBA16_sFloat",
                                  "RGBA16_sNorm", "RGBA32_sFloat", "R11G11B10_uFloat", "Depth16", "Depth32", "BC1", "BC3", "BC5", "BGRA8_uNorm"};
    return (f >= 0 && f <= 16) ? names[f] : "?";
}
int PassCtx::needSampled(int b, int fmt, const char* what) const {
    if (!hasSampled(b)) return fail(PLR_ERR_BINDING, std::string("missing sampled image at binding ") + std::to_string(b) + " (" + what + ")");
    if (fmt >= 0 && sampled[b].fmt != fmt)
        return fail(PLR_ERR_BINDING, std::string(what) + ": sampled binding " + std::to_string(b) + " expects " + formatName(fmt) + ", got " + formatName(sampled[b].fmt));
    return 0;
}
int PassCtx::needStorage(int b, int fmt, const char* what) const {
    if (!hasStorage(b)) return fail(PLR_ERR_BINDING, std::string("missing storage image at binding ") + std::to_string(b) + " (" + what + ")");
    if (fmt >= 0 && storage[b].fmt != fmt)
        return fail(PLR_ERR_BINDING, std::string(what) + ": storage binding " + std::to_string(b) + " expects " + formatName(fmt) + ", got " + formatName(storage[b].fmt));
    return 0;
}
int PassCtx::needSbuf(int b, size_t minSize, const char* what) const {
    if (!hasSbuf(b)) return fail(PLR_ERR_BINDING, std::string("missing storage buffer at binding ") + std::to_string(b) + " (" + what + ")");
    if (sbuf[b].size < minSize)
        return fail(PLR_ERR_BINDING, std::string(what) + ": storage buffer at binding " + std::to_string(b) + " has " + std::to_string(sbuf[b].size) + " bytes, needs " + std::to_string(minSize));
    return 0;
}
int PassCtx::needUbuf(int b, size_t minSize, const char* what) const {
    if (!hasUbuf(b)) return fail(PLR_ERR_BINDING, std::string("missing uniform buffer at binding ") + std::to_string(b) + " (" + what + ")");
    if (ubuf[b].size < minSize)
        return fail(PLR_ERR_BINDING, std::string(what) + ": uniform buffer at binding " + std::to_string(b) + " too small");
    return 0;
}
int PassCtx::needGlobal() const {
    if (!global) return fail(PLR_ERR_BINDING, "global uniform buffer (set 0 binding 0) not set: call plr_set_global_descriptor_set_resources");
    return 0;
}
void* PassCtx::scratch(size_t bytes) const {
    if (!scratchSlot) return nullptr;
    if (*scratchSize < bytes) {
        if (*scratchSlot) { hipStreamSynchronize(stream); hipFree(*scratchSlot); }
        *scratchSlot = nullptr;
        if (hipMalloc(scratchSlot, bytes) != hipSuccess) { *scratchSize = 0; return nullptr; }
        hipMemsetAsync(*scratchSlot, 0, bytes, stream);
        *scratchSize = bytes;
    }
    return *scratchSlot;
}

// ---------------------------------------------------------------- resources
static int formatBytes(uint32_t f) {
    switch (f) {
        case PLR_FORMAT_R8: return 1; case PLR_FORMAT_RG8: return 2; case PLR_FORMAT_RGBA8: return 4;
        case PLR_FORMAT_R16_SFLOAT: return 2; case PLR_FORMAT_RG16_SFLOAT: return 4; case PLR_FORMAT_RG32_SFLOAT: return 8;
        case PLR_FORMAT_RG16_SNORM: return 4; case PLR_FORMAT_RGBA16_SFLOAT: return 8; case PLR_FORMAT_RGBA16_SNORM: return 8;
        case PLR_FORMAT_RGBA32_SFLOAT: return 16; case PLR_FORMAT_R11G11B10_UFLOAT: return 4; case PLR_FORMAT_DEPTH16: return 2;
        case PLR_FORMAT_DEPTH32: return 4; case PLR_FORMAT_BGRA8_UNORM: return 4;
        default: return 0; // BCn: not on the hot path
    }
}

struct MipInfo { uint32_t w, h, d; size_t offset, bytes; };

struct ImageRes {
    plr_image_desc desc{};
    std::vector<MipInfo> mips;
    void* dev = nullptr;
    size_t bytes = 0;
    bool inUse = false; // transient pool bookkeeping
    // a fused launch consumed this image inside its kernel and did not write it (pass fusion level 2): the allocation holds an OLDER frame's texels. The flag is
    // persistent - cleared only when an execution really writes the image - and an execution (or host callback, or host download) that would READ the stale
    // texels fails loudly (a next frame recorded differently used to read them silently). The one reader that may bind it: the consumer the
    // eliding producer handed its result to through another channel (elidedReader, valid for the frame elidedSerial), or a member of the same fused launch
    bool elided = false;
    const void* elidedReader = nullptr;
    uint64_t elidedSerial = 0;
};

struct BufferRes {
    void* dev = nullptr;
    size_t size = 0;
};

struct PassRes {
    std::string shader, name;
    std::vector<SpecConstant> spec;
    LaunchFn fn = nullptr, fast = nullptr;
    bool readsBindless = false;
    void* scratch = nullptr;
    size_t scratchSize = 0;
};

// one resource an execution touches, for the hazard analysis of the stream scheduler: key = base address of the allocation
// (an image with all its mips, a buffer), or one of the pseudo keys below
struct Access { const void* key; bool write; };
static const char kBindlessKeyStorage = 0;
static const void* const kBindlessKey = &kBindlessKeyStorage; // "some image of the global texture array (set 2)"

struct Execution {
    uint32_t pass;
    PassCtx ctx;
    std::vector<Access> access;
    plr_host_callback callback = nullptr; // host callback execution (pass is unused)
    void* callbackUser = nullptr;
    const char* callbackName = ""; // interned in Backend::callbackNames
    bool asyncTail = false;        // plr_compute_pass_execution::async_tail
    uint32_t firstRows[2] = {0, 0}; // plr_compute_pass_execution::first_rows
    uint32_t firstCols[2] = {0, 0}; // plr_compute_pass_execution::first_cols
    bool edgesFirst() const { return firstRows[0] || firstRows[1] || firstCols[0] || firstCols[1]; }
    bool callbackAccessKnown = false; // host callback recorded with its resource list (plr_set_host_callback_execution_on): `access` is complete
};

struct FillOrder {
    void* dst;
    size_t size;
    size_t stagingOffset;
};

struct Backend {
    int device = 0;
    hipStream_t stream = nullptr;
    std::vector<ImageRes> images;
    std::vector<ImageRes> transient;
    ImageRes swapchain;
    std::vector<BufferRes> ubufs, sbufs;
    std::vector<plr_sampler_desc> samplers;
    std::vector<std::unique_ptr<PassRes>> passes;
    std::vector<Execution> executions;
    std::vector<FillOrder> fills;
    std::vector<uint8_t> fillData;
    // pinned staging of the deferred buffer fills: a ring of slots, one per frame in flight, each with the event that says the GPU has read it.
    // (Round 3 had ONE buffer and waited for its event every frame: the host could never be more than a frame ahead - VERDICT r03 weak 7.)
    static constexpr int kPinnedSlots = 3;
    // `serial`: what the fill kernel writes into the slot's header when it has read everything (the host polls the pinned word: no event record - a
    // barrier packet of ~7 us on the launch stream - per frame); `free` is only recorded on the rare paths that end with a copy-engine transfer
    struct PinnedSlot { void* host = nullptr; size_t size = 0; hipEvent_t free = nullptr; bool busy = false, eventPending = false; uint64_t serial = 0; } pinnedSlots[kPinnedSlots];
    uint64_t fillSerial = 0;
    // edge signal (plr.h first_rows): one word of signal memory the command processor can wait on (hipStreamWaitValue32), a device counter of arrived
    // edge waves, the value the last signalled launch raises it to; edgeSignal == nullptr: no stream memory operations on this platform
    uint32_t* edgeSignal = nullptr;
    uint32_t* edgeCounter = nullptr;
    uint32_t edgeSerial = 0;
    uint32_t pinnedNext = 0;
    uint32_t globalUbo = PLR_INVALID_INDEX;
    // plr_set_global_descriptor_set_layout: bit b = binding b of set 0 is declared as a uniform buffer / sampler / sampled image; layoutSet: a layout was given
    uint32_t layoutUbufs = 0, layoutSamplers = 0, layoutSampled = 0;
    bool layoutSet = false;
    ImgView* bindlessDev = nullptr;
    std::vector<ImgView> bindlessHost;   // what bindlessDev holds (PassCtx::bindlessHost)
    uint32_t bindlessCapacity = 0;
    bool bindlessDirty = true;
    uint64_t allocated = 0;
    bool passTiming = false;
    int mathMode = PLR_MATH_FAST;
    std::vector<hipEvent_t> passEvents; // pool; a timed segment uses two consecutive events
    struct Segment { size_t firstEvent; const char* name; };
    std::vector<Segment> segments;      // of the frame being launched / last launched with timing on
    size_t eventsUsed = 0;
    const char* currentPassName = nullptr;
    bool timingNow = false;
    std::vector<plr_renderpass_time> lastTimings;
    std::set<std::string> callbackNames; // stable storage for the labels of host callback executions
    size_t timedExecutions = 0;
    hipEvent_t frameStart = nullptr, frameEnd = nullptr;
    bool frameRecorded = false;
    bool frameTimeAsked = false; // plr_get_last_frame_gpu_time was called since the last plr_render_frame: the next frame is bracketed by events
    float lastFrameGpuMs = 0.f;  // of the most recent bracketed frame
    float lastCpuMs = 0.f;
    // stream scheduler (launchAll): independent passes of a frame run on side streams
    static constexpr int kSideStreams = 3;
    hipStream_t sideStreams[kSideStreams] = {nullptr, nullptr, nullptr};
    std::vector<hipEvent_t> orderEvents; // pool of timing-less events for cross-stream dependencies
    size_t orderEventsUsed = 0;
    bool overlap = false;                // off by default: see the measurement in the scheduler comment
    int activeSideStreams = 1;           // side streams the scheduler uses (PLR_SIDE_STREAMS, 1..kSideStreams)
    hipStream_t curStream = nullptr;     // stream of the execution being launched (timing events go there)
    uint32_t lastOverlapped = 0;         // executions of the last frame that were placed on a side stream
    uint32_t* debugSig = nullptr;        // decision-signature buffer (plr_debug_set_decision_signature)
    size_t debugSigWords = 0;
    int fusion = 2;                      // plr_set_pass_fusion: 0 off, 1 on, 2 on + a fused launcher may leave an intermediate image unwritten
    GlobalUbo globalShadow{};            // host copy of the global uniform buffer as of the last flushed fill (PassCtx::globalHost)
    bool globalShadowValid = false;
    uint32_t lastFused = 0;              // executions of the last frame that ran inside a fused launch
    // executions of the last frame that ran the GENERAL (exact-order) kernel of their shader although the math mode is PLR_MATH_FAST - no fast kernel is
    // registered for the shader, or its launcher declined the bindings (kUseGeneralKernel) - and their names (plr_get_general_kernel_executions)
    uint32_t lastGeneral = 0;
    std::string lastGeneralNames;
    uint64_t frameSerial = 0;            // serial of the running launchAll call, unique in the process (PassCtx::frameSerial)
    // asynchronous frame tail (plr_compute_pass_execution::async_tail, plr.h): executions launched on tailStream that the main stream has not
    // waited for yet, as the union of what they touch; tailDone is recorded behind the last of them
    hipStream_t tailStream = nullptr;
    hipEvent_t tailDone = nullptr, tailStart = nullptr;
    uint8_t* pendingFillSlot = nullptr;     // this frame's fill table, not applied yet (applyPendingFillsNow)
    hipEvent_t pendingFillEvent = nullptr;  // its slot's completion event, recorded behind the launch that applies it

    // early parts of fused sequences (backend.h EarlyPart; plr_set_early_parts): their stream, the event the launch stream records for it when the early part has to
    // start behind executions of this frame, the event behind the early part. mainOps counts what has been put on the launch stream (launches, callbacks, copies):
    // the tail's start event of the previous frame (tailStartOps = the count when it was recorded, tailStartCoversLate = it was recorded behind that frame's late
    // part) orders the early stream behind the previous frame's late part for free when nothing has gone onto the launch stream since
    hipStream_t earlyStream = nullptr;
    int earlyParts = 0;                  // plr_set_early_parts: 0 off (default: measured slower, profiles/r06_overlap.txt), 1 on, 2 on even with nothing to run beside (tests)
    uint32_t lastEarly = 0;              // early parts launched in the last frame
    uint64_t mainOps = 0, tailStartOps = ~0ull;
    bool tailStartCoversLate = false, lateLaunchedThisFrame = false;
    std::vector<const void*> lastFillDsts; // destinations of the buffer fills flushed for the frame being launched

    std::vector<Access> tailPending;
    bool asyncTail = true;               // plr_set_async_tail
    bool fusionReorder = true;           // plr_set_pass_fusion_reorder
    uint32_t lastAsync = 0;              // executions of the last frame that ran on the tail stream
    void* globalCopies[2] = {nullptr, nullptr}; // rotating device copies of the global uniform buffer: a tail that still reads frame N's must not see frame N + 1's fill
    uint32_t globalCopyIndex = 0;
    std::set<std::string> fusedNames;    // stable storage for the timing labels of fused launches
    // content versions of image allocations (backend.h contentVersionOf): key = allocation base; absent = not seen yet (gets a version on first query)
    std::unordered_map<const void*, uint64_t> contentVersion;
    std::unordered_set<const void*> externallyWritable; // allocations whose address was handed out: never cached
};

// A thread that calls plr_setup owns its backend: a process that drives several GPUs (or several bands on one GPU, as the partition tests do)
// uses one thread per backend. A thread that never called plr_setup ADOPTS the process's first backend - the reference's single process-global
// gRenderBackend (RenderBackend.cpp:39): a host that sets up on one thread and records from a worker thread works as it does there (calls must
// not overlap in time, as in the reference). The adoption ends when that backend is shut down (epoch check). Adopting makes the backend's device
// the thread's current one. A thread that has had a backend of its own does not adopt: after its plr_shutdown it gets PLR_ERR_NOT_INITIALISED
// again instead of silently operating on another thread's backend.
static thread_local Backend* g = nullptr;
static thread_local std::string g_err;

// ---- The PLR_MATH_EXACT launch paths of the hot path's big passes (GI trace, GI filters, deferred shade, TAA, bloom: csrc/kernels_exact/) are a library of their
// own, libplr_exact.so next to this one (plainrenderer_amd/build.py): the shipped path is the fast set, the exact set is the parity instrument. It is loaded the first
// time it is needed - plr_set_math_mode(PLR_MATH_EXACT), or an execution whose fast launcher declines its bindings - and registers its launchers like any other
// translation unit (ShaderRegistrar). Fails loudly when the file is missing.
static std::mutex g_exactSetMutex;
static bool g_exactSetTried = false;
static void* g_exactSetHandle = nullptr;
static std::string g_exactSetError;
static int ensureExactSet() {
    std::lock_guard<std::mutex> lock(g_exactSetMutex);
    if (!g_exactSetTried) {
        g_exactSetTried = true;
        Dl_info info{};
        std::string path;
        if (dladdr((const void*)&ensureExactSet, &info) && info.dli_fname) {
            path = info.dli_fname; // .../libplr.so or .../libplr_<tag>.so
            const size_t slash = path.find_last_of('/');
            const size_t name = slash == std::string::npos ? 0 : slash + 1;
            if (path.compare(name, 6, "libplr") == 0) path.insert(name + 6, "_exact");
            else path.clear();
        }
        if (path.empty()) g_exactSetError = "cannot locate libplr.so to find libplr_exact.so next to it";
        else {
            g_exactSetHandle = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (!g_exactSetHandle) g_exactSetError = "the PLR_MATH_EXACT kernel set is a separate library and it did not load (" + path + "): " + (dlerror() ? dlerror() : "?") +
                                                     " - build it with plainrenderer_amd/build.py";
        }
    }
    if (!g_exactSetHandle) { g_err = g_exactSetError; return PLR_ERR_UNSUPPORTED; }
    return PLR_OK;
}

static std::string shaderBaseName(const std::string& path) {
    const size_t slash = path.find_last_of("/\\");
    return slash == std::string::npos ? path : path.substr(slash + 1);
}
static bool findShader(const std::string& path, ShaderEntry* out) {
    const std::string base = shaderBaseName(path);
    for (int attempt = 0; attempt < 2; attempt++) {
        {
            std::lock_guard<std::mutex> lock(registryMutex());
            for (const auto& e : registry())
                if (e.name == base && (e.fn || e.fast)) { *out = e; return true; }
        }
        if (attempt == 0 && ensureExactSet() != PLR_OK) break; // a shader only the exact set has
    }
    return false;
}
// the general launcher of a pass whose shader's exact launch path lives in libplr_exact.so
static int resolveExactLauncher(LaunchFn* fn, const std::string& shader) {
    if (int rc = ensureExactSet()) return rc;
    const std::string base = shaderBaseName(shader);
    {
        std::lock_guard<std::mutex> lock(registryMutex());
        for (const auto& e : registry())
            if (e.name == base && e.fn) { *fn = e.fn; return PLR_OK; }
    }
    g_err = "no PLR_MATH_EXACT kernel for shader '" + shader + "'";
    return PLR_ERR_UNKNOWN_SHADER;
}

static thread_local bool g_adopted = false;
static thread_local bool g_ownedOnce = false; // this thread has set up (and possibly shut down) a backend of its own: it never adopts another thread's
static thread_local uint64_t g_adoptedEpoch = 0;
static std::atomic<Backend*> g_first{nullptr};
static std::atomic<uint64_t> g_firstEpoch{0};
static bool resolveBackend() {
    if (g && g_adopted && g_adoptedEpoch != g_firstEpoch.load()) { g = nullptr; g_adopted = false; }
    if (!g && !g_ownedOnce) {
        Backend* first = g_first.load();
        if (first) {
            g = first; g_adopted = true; g_adoptedEpoch = g_firstEpoch.load();
            // the current device is per thread: allocations, events and launches of this thread must go to the backend's GPU, not to device 0
            (void)hipSetDevice(first->device);
        }
    }
    return g != nullptr;
}

static int setErr(int code, const std::string& msg) { g_err = msg; return code; }
int setLastError(int code, const std::string& msg) { return setErr(code, msg); }
#define HIP_TRY(x)                                                                                      \
    do {                                                                                                \
        hipError_t e_ = (x);                                                                            \
        if (e_ != hipSuccess) return setErr(PLR_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define NEED_INIT() \
    if (!resolveBackend()) return setErr(PLR_ERR_NOT_INITIALISED, "plr_setup has not been called")
// entry points that read or write device memory, or wait for the GPU, from the host: the asynchronous frame tail has to be behind them
static int joinAsyncTail();
#define NEED_INIT_JOINED() \
    NEED_INIT();           \
    if (int jrc_ = joinAsyncTail()) return jrc_

// the main stream waits for everything launched on the tail stream so far (a no-op when nothing is pending)
static int joinAsyncTail() {
    if (!g || g->tailPending.empty()) return PLR_OK;
    HIP_TRY(hipStreamWaitEvent(g->stream, g->tailDone, 0));
    g->tailPending.clear();
    return PLR_OK;
}
static bool hazardWithTail(const std::vector<Access>& access) {
    // (the pseudo key "some image of the global texture array" does not take part: the array is read for images no pass writes - SDF volumes, noise
    //  textures - and an image a pass does write is ordered through its explicit binding, which is all the reference's barrier tracking sees as well,
    //  RenderBackend.cpp:632-767. With it every pass that writes any default image would collide with the tail's write of the TAA target.)
    for (const Access& a : access) {
        if (a.key == kBindlessKey) continue;
        for (const Access& t : g->tailPending)
            if (a.key == t.key && (a.write || t.write)) return true;
    }
    return false;
}

// ---- content versions (backend.h contentVersionOf)
static uint64_t nextContentVersion() { static std::atomic<uint64_t> counter{0}; return ++counter; }
static void touchAllocation(const void* base) { if (g && base) g->contentVersion[base] = nextContentVersion(); }
static void touchAccesses(const std::vector<Access>& access) {
    for (const Access& a : access) if (a.write && a.key != kBindlessKey) touchAllocation(a.key);
}
uint64_t contentVersionOf(const void* base) {
    if (!g || !base || g->externallyWritable.count(base)) return 0;
    bool ours = false;
    for (const ImageRes& im : g->images) if (im.dev == base) { ours = true; break; }
    if (!ours) for (const ImageRes& im : g->transient) if (im.dev == base) { ours = true; break; }
    if (!ours) return 0;
    auto it = g->contentVersion.find(base);
    if (it == g->contentVersion.end()) it = g->contentVersion.emplace(base, nextContentVersion()).first;
    return it->second;
}

// a raw device address was written (plr_write / copy_device_memory): the image allocation it lies in, if any, has new contents
// a HOST write into an image (upload, raw copy / write into its allocation): new contents, and the texels are the host's now - an image a fused launch had left
// unwritten (ImageRes::elided) is readable again (a host that re-initialised such an image got the "not written" error on its next read)
static void hostWroteImage(ImageRes& im) {
    touchAllocation(im.dev);
    im.elided = false; im.elidedReader = nullptr; im.elidedSerial = 0;
}
static void touchAddress(const void* p) {
    auto inside = [&](const ImageRes& im) { return im.dev && (const uint8_t*)p >= (const uint8_t*)im.dev && (const uint8_t*)p < (const uint8_t*)im.dev + im.bytes; };
    for (ImageRes& im : g->images) if (inside(im)) { hostWroteImage(im); return; }
    for (ImageRes& im : g->transient) if (inside(im)) { hostWroteImage(im); return; }
}

static uint32_t mipCountFromResolution(uint32_t w, uint32_t h, uint32_t d) {
    // Common/Utilities/MathUtils.cpp:17-19
    uint32_t m = std::max(std::max(w, h), d);
    uint32_t c = 1;
    while (m > 1) { m >>= 1; c++; }
    return c;
}

static int layoutImage(ImageRes& im, const plr_image_desc& d) {
    const int bpp = formatBytes(d.format);
    if (bpp == 0) return setErr(PLR_ERR_UNSUPPORTED, "image format not supported by the HIP backend (BCn formats are outside the hot path)");
    if (d.width == 0) return setErr(PLR_ERR_INVALID_ARGUMENT, "image width is 0");
    im.desc = d;
    const uint32_t w = d.width, h = std::max(d.height, 1u), dep = std::max(d.depth, 1u);
    uint32_t levels = 1;
    if (d.mip_count == PLR_MIP_FULL_CHAIN || d.mip_count == PLR_MIP_FULL_CHAIN_ALREADY_IN_DATA) levels = mipCountFromResolution(w, h, dep);
    else if (d.mip_count == PLR_MIP_MANUAL) levels = std::max(d.manual_mip_count, 1u);
    im.mips.clear();
    size_t off = 0;
    for (uint32_t m = 0; m < levels; m++) {
        MipInfo mi;
        mi.w = std::max(w >> m, 1u); mi.h = std::max(h >> m, 1u); mi.d = std::max(dep >> m, 1u);
        mi.offset = off;
        mi.bytes = (size_t)mi.w * mi.h * mi.d * bpp;
        off += (mi.bytes + 255) & ~(size_t)255; // 256-byte aligned mip bases keep dwordx4 accesses aligned
        im.mips.push_back(mi);
    }
    im.bytes = off;
    return PLR_OK;
}

static int allocImage(ImageRes& im, const plr_image_desc& d) {
    int rc = layoutImage(im, d);
    if (rc) return rc;
    HIP_TRY(hipMalloc(&im.dev, im.bytes));
    HIP_TRY(hipMemsetAsync(im.dev, 0, im.bytes, g->stream));
    g->allocated += im.bytes;
    touchAllocation(im.dev);
    return PLR_OK;
}

static void freeImage(ImageRes& im) {
    if (im.dev) { hipFree(im.dev); g->allocated -= im.bytes; g->contentVersion.erase(im.dev); g->externallyWritable.erase(im.dev); }
    im.dev = nullptr; im.bytes = 0; im.mips.clear();
}

static ImageRes* resolveImage(plr_image_handle h) {
    if (h.type == PLR_IMAGE_DEFAULT) return h.index < g->images.size() && g->images[h.index].dev ? &g->images[h.index] : nullptr;
    if (h.type == PLR_IMAGE_TRANSIENT) return h.index < g->transient.size() && g->transient[h.index].inUse ? &g->transient[h.index] : nullptr;
    if (h.type == PLR_IMAGE_SWAPCHAIN) return g->swapchain.dev ? &g->swapchain : nullptr;
    return nullptr;
}

static ImgView makeView(const ImageRes& im, uint32_t mip) {
    ImgView v;
    const MipInfo& mi = im.mips[mip];
    v.ptr = (uint8_t*)im.dev + mi.offset;
    v.w = (int32_t)mi.w; v.h = (int32_t)mi.h; v.d = (int32_t)mi.d;
    v.fmt = (int32_t)im.desc.format;
    return v;
}

static bool sameDesc(const plr_image_desc& a, const plr_image_desc& b) { return std::memcmp(&a, &b, sizeof(a)) == 0; }

int launchSkyLutProbe(const ImgView& lut, const float* dirs, float* out, int64_t n); // kernels_fast/stream_fast.hip
int launchSamplerProbe(const ImgView& view, int filter, int address, const float* coords, float* out, int64_t n); // kernels/probes.hip


static bool sameView(const ImgView& a, const ImgView& b) { return a.ptr == b.ptr && a.w == b.w && a.h == b.h && a.d == b.d && a.fmt == b.fmt; }
int launchOverTwoRowRanges(const PassCtx* const* ctxs, size_t count, LaunchFn single) {
    if (count != 2) return kUseGeneralKernel;
    const PassCtx& a = *ctxs[0];
    const PassCtx& b = *ctxs[1];
    if (a.sampledMask != b.sampledMask || a.storageMask != b.storageMask || a.sbufMask != b.sbufMask || a.ubufMask != b.ubufMask || a.push != b.push || a.spec != b.spec ||
        a.dispatch[0] != b.dispatch[0] || a.base[0] != b.base[0] || a.dispatch[2] != b.dispatch[2] || a.validRows[0] != b.validRows[0] || a.validRows[1] != b.validRows[1] ||
        a.validCols[0] != b.validCols[0] || a.validCols[1] != b.validCols[1] || a.extraCountY || b.extraCountY || a.scratchSlot != b.scratchSlot)
        return kUseGeneralKernel;
    for (int i = 0; i < kMaxBindings; i++) {
        if (a.hasSampled(i) && !sameView(a.sampled[i], b.sampled[i])) return kUseGeneralKernel;
        if (a.hasStorage(i) && !sameView(a.storage[i], b.storage[i])) return kUseGeneralKernel;
        if (a.hasSbuf(i) && (a.sbuf[i].ptr != b.sbuf[i].ptr || a.sbuf[i].size != b.sbuf[i].size)) return kUseGeneralKernel;
        if (a.hasUbuf(i) && (a.ubuf[i].ptr != b.ubuf[i].ptr || a.ubuf[i].size != b.ubuf[i].size)) return kUseGeneralKernel;
    }
    if (b.base[1] < a.base[1] + a.dispatch[1] || a.dispatch[1] == 0 || b.dispatch[1] == 0) return kUseGeneralKernel; // the second range lies below the first
    PassCtx both = a;
    both.extraBaseY = b.base[1];
    both.extraCountY = b.dispatch[1];
    return single(both);
}

int twoRangeBlocks(const PassCtx& c, int imageH, int blockRows, int wgRows, TwoRanges* out, int* blocks, int* y0, int* end) {
    const PassCtx::RowSpan a = c.rowSpan(imageH, wgRows);
    *out = TwoRanges{};
    *y0 = a.y0; *end = a.y1;
    *blocks = a.y1 > a.y0 ? (a.y1 - a.y0 + blockRows - 1) / blockRows : 0;
    if (!c.extraCountY) return 0;
    const long long b0 = (long long)c.extraBaseY * wgRows, b1 = b0 + (long long)c.extraCountY * wgRows;
    const int e0 = (int)(b0 < imageH ? b0 : imageH), e1 = (int)(b1 < imageH ? b1 : imageH);
    // the first range must end on a block boundary (its last block would otherwise run into rows of neither range) and the second start on one
    if ((a.y1 - a.y0) % blockRows || (e0 - a.y0) % blockRows || e0 < a.y1 || e1 <= e0) return 1;
    out->split = *blocks;
    out->gap = (e0 - a.y0) / blockRows - *blocks;
    *blocks += (e1 - e0 + blockRows - 1) / blockRows;
    *end = e1;
    return 0;
}

bool TwoRanges::setEdgeFirst(const PassCtx& c, int y0, int y1, int blockRowsPx, int wgRows, unsigned blocksX, int xOrigin, int x1, int blockColsPx, int wgCols) {
    const bool rowsAsked = c.firstRows[0] || c.firstRows[1], colsAsked = c.firstCols[0] || c.firstCols[1];
    if (!c.edgeSignal || !(rowsAsked || colsAsked) || c.extraCountY || y1 <= y0 || blocksX == 0) return false;
    const int totalRows = (y1 - y0 + blockRowsPx - 1) / blockRowsPx;
    int top = 0, bottom = 0, left = 0, right = 0;
    if (rowsAsked) {
        const long long topEnd = std::min<long long>((long long)c.firstRows[0] * wgRows, y1), bottomBegin = std::min<long long>((long long)c.firstRows[1] * wgRows, y1);
        if (topEnd < y0 || bottomBegin < topEnd) return false;
        if ((topEnd - y0) % blockRowsPx || (bottomBegin - y0) % blockRowsPx) return false; // an edge must be whole block rows (the last block row of the launch may be partial)
        top = (int)((topEnd - y0) / blockRowsPx);
        bottom = totalRows - (int)((bottomBegin - y0) / blockRowsPx);
    }
    if (colsAsked) {
        // a launcher that cannot order its block columns (blockColsPx == 0) leaves the signal to the backend: raised behind the whole launch
        if (blockColsPx <= 0 || x1 <= xOrigin) return false;
        const long long leftEnd = std::min<long long>((long long)c.firstCols[0] * wgCols, x1), rightBegin = std::min<long long>((long long)c.firstCols[1] * wgCols, x1);
        if (rightBegin < leftEnd) return false;
        // block columns that hold a pixel column of an edge (a block column that is only partly edge is taken as a whole: more is written through, nothing is late)
        left = leftEnd > xOrigin && leftEnd > (long long)c.base[0] * wgCols ? (int)((leftEnd - xOrigin + blockColsPx - 1) / blockColsPx) : 0;
        right = rightBegin < x1 ? (int)blocksX - (int)((rightBegin - xOrigin) / blockColsPx) : 0;
        left = std::min(std::max(left, 0), (int)blocksX);
        right = std::min(std::max(right, 0), (int)blocksX - left);
    }
    if (top + bottom > totalRows) return false;
    if (top + bottom + left + right == 0) return false;
    edgeTop = top; edgeBottom = bottom; total = totalRows;
    edgeLeft = left; edgeRight = right;
    edgeBlocks = (uint32_t)(top + bottom) * blocksX + (uint32_t)(left + right) * (uint32_t)(totalRows - top - bottom);
    edgeCounter = c.edgeCounter; edgeSignal = c.edgeSignal; edgeValue = c.edgeValue;
    if (edgeBlocks == 0) { edgeTop = edgeBottom = edgeLeft = edgeRight = 0; total = 0; return false; }
    // experiment hook (profiles/r05_not_kept.txt (6)): the edges-first block ORDER alone - no write-through, no arrivals, the backend raises the signal behind the launch
    static const bool orderOnly = std::getenv("PLR_EDGE_ORDER_ONLY") && std::atoi(std::getenv("PLR_EDGE_ORDER_ONLY")) != 0;
    if (orderOnly) { edgeBlocks = 0; return false; }
    c.edgeSignalHonoured = true;
    return true;
}

void countFusedExecutions(uint32_t n) { if (g) g->lastFused += n; }

} // namespace plr

using namespace plr;

extern "C" {

const char* plr_last_error(void) { return g_err.c_str(); }

int plr_setup(int device_ordinal, uint32_t width, uint32_t height) {
    if (g && g_adopted) { g = nullptr; g_adopted = false; } // a thread that had adopted the process's first backend now gets its own
    if (g) return setErr(PLR_ERR_INVALID_ARGUMENT, "plr_setup called twice; call plr_shutdown first");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        return setErr(PLR_ERR_HIP, std::string("no HIP device available: ") + hipGetErrorString(e) + " (this backend has no CPU fallback)");
    if (device_ordinal < 0 || device_ordinal >= count) return setErr(PLR_ERR_INVALID_ARGUMENT, "device ordinal out of range");
    HIP_TRY(hipSetDevice(device_ordinal));
    g_ownedOnce = true;
    g = new Backend();
    {
        Backend* none = nullptr;
        g_first.compare_exchange_strong(none, g); // the process's first backend: what threads without a backend of their own use
    }
    g->device = device_ordinal;
    int leastPriority = 0, greatestPriority = 0; // (numerically: greatest <= least)
    if (hipDeviceGetStreamPriorityRange(&leastPriority, &greatestPriority) != hipSuccess) { (void)hipGetLastError(); leastPriority = greatestPriority = 0; }
    static const int mainPriority = std::getenv("PLR_MAIN_PRIORITY") ? std::atoi(std::getenv("PLR_MAIN_PRIORITY")) : 0;   // experiment hooks: 1 = the launch stream at the greatest priority
    static const int earlyPriority = std::getenv("PLR_EARLY_PRIORITY") ? std::atoi(std::getenv("PLR_EARLY_PRIORITY")) : 0; // 1 = the early stream at the least priority
    if (mainPriority) HIP_TRY(hipStreamCreateWithPriority(&g->stream, hipStreamNonBlocking, greatestPriority));
    else HIP_TRY(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&g->frameStart));
    HIP_TRY(hipEventCreate(&g->frameEnd));
    for (auto& slot : g->pinnedSlots) HIP_TRY(hipEventCreateWithFlags(&slot.free, hipEventDisableTiming));
    for (auto& st : g->sideStreams) HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&g->tailStream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&g->tailDone, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&g->tailStart, hipEventDisableTiming));
    if (earlyPriority) HIP_TRY(hipStreamCreateWithPriority(&g->earlyStream, hipStreamNonBlocking, leastPriority));
    else HIP_TRY(hipStreamCreateWithFlags(&g->earlyStream, hipStreamNonBlocking));
    if (std::getenv("PLR_STREAM_DEBUG")) fprintf(stderr, "[plr streams] priority range least %d greatest %d; main %s, early %s\n", leastPriority, greatestPriority, mainPriority ? "greatest" : "default", earlyPriority ? "least" : "default");
    if (const char* ep = std::getenv("PLR_EARLY_PARTS")) g->earlyParts = std::min(std::max(std::atoi(ep), 0), 2);
    // the edge signal is waited for with hipStreamWaitValue32: asked of the device, not assumed (without it plr_get_edge_signal reports no signal and a caller orders
    // behind the launch stream - VERDICT r04 item 5)
    int canWaitValue = 0;
    if (hipDeviceGetAttribute(&canWaitValue, hipDeviceAttributeCanUseStreamWaitValue, device_ordinal) != hipSuccess) { (void)hipGetLastError(); canWaitValue = 0; }
    if (canWaitValue && hipExtMallocWithFlags((void**)&g->edgeSignal, 8, hipMallocSignalMemory) == hipSuccess && hipMalloc((void**)&g->edgeCounter, 4096) == hipSuccess) { // TwoRanges::edgeDone: top counter + 32 shard counters, 64 bytes apart
        HIP_TRY(hipMemset(g->edgeSignal, 0, 8));
        HIP_TRY(hipMemset(g->edgeCounter, 0, 4096));
    } else {
        (void)hipGetLastError();
        if (g->edgeSignal) hipFree(g->edgeSignal);
        g->edgeSignal = nullptr; g->edgeCounter = nullptr;
    }
    if (const char* es = std::getenv("PLR_EDGE_SIGNAL")) if (std::atoi(es) == 0 && g->edgeSignal) { hipFree(g->edgeSignal); hipFree(g->edgeCounter); g->edgeSignal = g->edgeCounter = nullptr; }
    if (const char* at = std::getenv("PLR_ASYNC_TAIL")) g->asyncTail = std::atoi(at) != 0;
    if (const char* ov = std::getenv("PLR_STREAM_OVERLAP")) g->overlap = std::atoi(ov) != 0;
    if (const char* pf = std::getenv("PLR_PASS_FUSION")) g->fusion = std::min(std::max(std::atoi(pf), 0), 2);
    if (const char* ns = std::getenv("PLR_SIDE_STREAMS")) g->activeSideStreams = std::min(std::max(std::atoi(ns), 1), (int)Backend::kSideStreams);
    return plr_recreate_swapchain(width, height);
}

int plr_shutdown(void) {
    if (!resolveBackend()) return PLR_OK;
    if (g_adopted) return setErr(PLR_ERR_INVALID_ARGUMENT, "plr_shutdown: this thread uses the backend another thread set up; that thread shuts it down");
    hipSetDevice(g->device);
    hipDeviceSynchronize();
    g->tailPending.clear();
    if (g->tailStream) hipStreamDestroy(g->tailStream);
    if (g->tailDone) hipEventDestroy(g->tailDone);
    if (g->tailStart) hipEventDestroy(g->tailStart);
    if (g->earlyStream) hipStreamDestroy(g->earlyStream);
    for (void* c : g->globalCopies) if (c) hipFree(c);
    if (g->edgeSignal) hipFree(g->edgeSignal);
    if (g->edgeCounter) hipFree(g->edgeCounter);
    for (auto& im : g->images) freeImage(im);
    for (auto& im : g->transient) freeImage(im);
    freeImage(g->swapchain);
    for (auto& b : g->ubufs) if (b.dev) hipFree(b.dev);
    for (auto& b : g->sbufs) if (b.dev) hipFree(b.dev);
    for (auto& p : g->passes) if (p->scratch) hipFree(p->scratch);
    if (g->debugSig) hipFree(g->debugSig);
    for (auto ev : g->passEvents) hipEventDestroy(ev);
    for (auto ev : g->orderEvents) hipEventDestroy(ev);
    for (auto st : g->sideStreams) if (st) hipStreamDestroy(st);
    if (g->bindlessDev) hipFree(g->bindlessDev);
    for (auto& slot : g->pinnedSlots) { if (slot.host) hipHostFree(slot.host); if (slot.free) hipEventDestroy(slot.free); }
    hipEventDestroy(g->frameStart); hipEventDestroy(g->frameEnd);
    hipStreamDestroy(g->stream);
    {
        Backend* mine = g;
        if (g_first.compare_exchange_strong(mine, nullptr)) g_firstEpoch++; // adopters let go at their next call
    }
    delete g;
    g = nullptr;
    g_adopted = false;
    return PLR_OK;
}

int plr_recreate_swapchain(uint32_t width, uint32_t height) {
    NEED_INIT_JOINED();
    if (width == 0 || height == 0) return setErr(PLR_ERR_INVALID_ARGUMENT, "swapchain size must be non-zero");
    HIP_TRY(hipStreamSynchronize(g->stream));
    freeImage(g->swapchain);
    plr_image_desc d{};
    d.width = width; d.height = height; d.depth = 1; d.type = PLR_IMAGE_2D; d.format = PLR_FORMAT_BGRA8_UNORM;
    d.usage_flags = PLR_USAGE_STORAGE; d.mip_count = PLR_MIP_ONE; d.manual_mip_count = 1;
    return allocImage(g->swapchain, d);
}

int plr_wait_for_gpu_idle(void) {
    NEED_INIT_JOINED();
    HIP_TRY(hipStreamSynchronize(g->stream));
    return PLR_OK;
}

// raw copies on the calling thread's backend (band exchange inside one process, histogram reduction on the host): the host side
// never has to load a HIP runtime of its own
int plr_copy_device_memory(void* dst, const void* src, size_t size) {
    NEED_INIT_JOINED();
    if (size == 0) return PLR_OK;
    if (!dst || !src) return setErr(PLR_ERR_INVALID_ARGUMENT, "plr_copy_device_memory: null pointer");
    HIP_TRY(hipMemcpyAsync(dst, src, size, hipMemcpyDeviceToDevice, g->stream));
    g->mainOps++;
    touchAddress(dst);
    return PLR_OK;
}
int plr_copy_device_memory_2d(void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t width_bytes, size_t rows) {
    NEED_INIT_JOINED();
    if (width_bytes == 0 || rows == 0) return PLR_OK;
    if (!dst || !src || dst_pitch < width_bytes || src_pitch < width_bytes) return setErr(PLR_ERR_INVALID_ARGUMENT, "plr_copy_device_memory_2d: null pointer or a pitch smaller than the width");
    HIP_TRY(hipMemcpy2DAsync(dst, dst_pitch, src, src_pitch, width_bytes, rows, hipMemcpyDeviceToDevice, g->stream));
    g->mainOps++;
    touchAddress(dst);
    return PLR_OK;
}
int plr_read_device_memory(void* dst_host, const void* src, size_t size) {
    NEED_INIT_JOINED();
    if (size == 0) return PLR_OK;
    if (!dst_host || !src) return setErr(PLR_ERR_INVALID_ARGUMENT, "plr_read_device_memory: null pointer");
    HIP_TRY(hipMemcpyAsync(dst_host, src, size, hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    return PLR_OK;
}
int plr_write_device_memory(void* dst, const void* src_host, size_t size) {
    NEED_INIT_JOINED();
    if (size == 0) return PLR_OK;
    if (!dst || !src_host) return setErr(PLR_ERR_INVALID_ARGUMENT, "plr_write_device_memory: null pointer");
    HIP_TRY(hipMemcpyAsync(dst, src_host, size, hipMemcpyHostToDevice, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    touchAddress(dst);
    return PLR_OK;
}

int plr_update_shader_code(void) { NEED_INIT(); return PLR_OK; }

int plr_resize_images(const plr_image_handle* images, uint32_t count, uint32_t width, uint32_t height) {
    NEED_INIT_JOINED();
    HIP_TRY(hipStreamSynchronize(g->stream));
    for (uint32_t i = 0; i < count; i++) {
        if (images[i].type != PLR_IMAGE_DEFAULT) return setErr(PLR_ERR_INVALID_ARGUMENT, "only default images can be resized");
        ImageRes* im = resolveImage(images[i]);
        if (!im) return setErr(PLR_ERR_INVALID_ARGUMENT, "plr_resize_images: invalid image handle");
        plr_image_desc d = im->desc;
        d.width = width; d.height = height;
        freeImage(*im);
        int rc = allocImage(*im, d);
        if (rc) return rc;
    }
    g->bindlessDirty = true;
    return PLR_OK;
}

int plr_new_frame(void) {
    NEED_INIT();
    g->executions.clear();
    for (auto& t : g->transient) t.inUse = false;
    return PLR_OK;
}

static int resolveResources(const plr_pass_resources& r, PassCtx& ctx) {
    for (uint32_t i = 0; i < r.sampled_image_count; i++) {
        const plr_image_resource& ir = r.sampled_images[i];
        if (ir.binding >= (uint32_t)kMaxBindings) return setErr(PLR_ERR_BINDING, "sampled image binding index too large");
        ImageRes* im = resolveImage(ir.image);
        if (!im) return setErr(PLR_ERR_BINDING, "sampled image at binding " + std::to_string(ir.binding) + ": invalid image handle");
        if (ir.mip_level >= im->mips.size()) return setErr(PLR_ERR_BINDING, "sampled image mip level out of range");
        ctx.sampled[ir.binding] = makeView(*im, ir.mip_level);
        ctx.sampledMask |= 1u << ir.binding;
    }
    for (uint32_t i = 0; i < r.storage_image_count; i++) {
        const plr_image_resource& ir = r.storage_images[i];
        if (ir.binding >= (uint32_t)kMaxBindings) return setErr(PLR_ERR_BINDING, "storage image binding index too large");
        ImageRes* im = resolveImage(ir.image);
        if (!im) return setErr(PLR_ERR_BINDING, "storage image at binding " + std::to_string(ir.binding) + ": invalid image handle");
        if (ir.mip_level >= im->mips.size()) return setErr(PLR_ERR_BINDING, "storage image mip level out of range");
        ctx.storage[ir.binding] = makeView(*im, ir.mip_level);
        ctx.storageMask |= 1u << ir.binding;
    }
    for (uint32_t i = 0; i < r.storage_buffer_count; i++) {
        const plr_storage_buffer_resource& br = r.storage_buffers[i];
        if (br.binding >= (uint32_t)kMaxBindings || br.buffer >= g->sbufs.size()) return setErr(PLR_ERR_BINDING, "invalid storage buffer resource");
        ctx.sbuf[br.binding] = {g->sbufs[br.buffer].dev, g->sbufs[br.buffer].size, br.read_only != 0};
        ctx.sbufMask |= 1u << br.binding;
    }
    for (uint32_t i = 0; i < r.uniform_buffer_count; i++) {
        const plr_uniform_buffer_resource& br = r.uniform_buffers[i];
        if (br.binding >= (uint32_t)kMaxBindings || br.buffer >= g->ubufs.size()) return setErr(PLR_ERR_BINDING, "invalid uniform buffer resource");
        ctx.ubuf[br.binding] = {g->ubufs[br.buffer].dev, g->ubufs[br.buffer].size, true};
        ctx.ubufMask |= 1u << br.binding;
    }
    return PLR_OK;
}

int plr_set_compute_pass_execution(const plr_compute_pass_execution* e) {
    NEED_INIT();
    if (!e) return setErr(PLR_ERR_INVALID_ARGUMENT, "execution is null");
    if (e->handle >= g->passes.size()) return setErr(PLR_ERR_INVALID_ARGUMENT, "invalid pass handle");
    g->executions.emplace_back();
    Execution& x = g->executions.back();
    x.pass = e->handle;
    int rc = resolveResources(e->resources, x.ctx);
    if (rc) { g->executions.pop_back(); return rc; }
    if (e->push_constant_size) x.ctx.push.assign((const uint8_t*)e->push_constants, (const uint8_t*)e->push_constants + e->push_constant_size);
    for (int i = 0; i < 3; i++) { x.ctx.dispatch[i] = e->dispatch_count[i]; x.ctx.base[i] = e->dispatch_base[i]; }
    x.ctx.validRows[0] = e->valid_rows[0]; x.ctx.validRows[1] = e->valid_rows[1];
    x.ctx.validCols[0] = e->valid_cols[0]; x.ctx.validCols[1] = e->valid_cols[1];
    x.asyncTail = e->async_tail != 0;
    x.firstRows[0] = e->first_rows[0]; x.firstRows[1] = e->first_rows[1];
    x.firstCols[0] = e->first_cols[0]; x.firstCols[1] = e->first_cols[1];
    {
        // what the execution may touch (stream scheduler): whole allocations, so a kernel that walks the mip chain of a bound image or
        // addresses rows outside its dispatch is covered; uniform buffers are only written between frames
        const plr_pass_resources& r = e->resources;
        bool writesDefaultImage = false;
        for (uint32_t i = 0; i < r.sampled_image_count; i++) x.access.push_back({resolveImage(r.sampled_images[i].image)->dev, false});
        for (uint32_t i = 0; i < r.storage_image_count; i++) {
            const bool written = !storageBindingIsReadOnly(g->passes[e->handle]->shader, r.storage_images[i].binding);
            x.access.push_back({resolveImage(r.storage_images[i].image)->dev, written});
            writesDefaultImage = writesDefaultImage || (written && r.storage_images[i].image.type == PLR_IMAGE_DEFAULT);
        }
        for (uint32_t i = 0; i < r.storage_buffer_count; i++) x.access.push_back({g->sbufs[r.storage_buffers[i].buffer].dev, r.storage_buffers[i].read_only == 0});
        // uniform buffers are read: a fill of one (plr_set_uniform_buffer_data, applied at the next plr_render_frame) must wait for an asynchronous
        // tail that still reads it (flushFills; ADVICE r03: without these entries a tail execution with a UBO of its own raced with the next fill)
        for (uint32_t i = 0; i < r.uniform_buffer_count; i++) x.access.push_back({g->ubufs[r.uniform_buffers[i].buffer].dev, false});
        x.access.push_back({g->passes[e->handle].get(), true}); // the pass's scratch memory: executions of one pass never overlap
        if (g->passes[e->handle]->readsBindless) x.access.push_back({kBindlessKey, false});
        if (writesDefaultImage) x.access.push_back({kBindlessKey, true});
    }
    if (e->dispatch_base[2] != 0) { g->executions.pop_back(); return setErr(PLR_ERR_INVALID_ARGUMENT, "dispatch_base[2] must be 0"); }
    // a first workgroup COLUMN (tile rendering) is honoured by the passes of the per-pixel frame path, whose launchers turn it into a column span
    // (PassCtx::colSpan), and by histogramCombineTiles (first tile); every other pass covers whole rows: not silently ignored
    if (e->dispatch_base[0] != 0 && !shaderTakesColumns(g->passes[e->handle]->shader)) {
        const std::string shader = g->passes[e->handle]->shader;
        g->executions.pop_back();
        return setErr(PLR_ERR_UNSUPPORTED, "dispatch_base[0] is not honoured by " + shader + ": its kernels cover whole rows");
    }
    return PLR_OK;
}

int plr_set_host_callback_execution_on(plr_host_callback callback, void* user, const char* name, const plr_image_handle* images, uint32_t n_images,
                                       const plr_storage_buffer_handle* buffers, uint32_t n_buffers) {
    NEED_INIT();
    if ((n_images && !images) || (n_buffers && !buffers)) return setErr(PLR_ERR_INVALID_ARGUMENT, "null resource list");
    for (uint32_t i = 0; i < n_images; i++) if (!resolveImage(images[i])) return setErr(PLR_ERR_INVALID_ARGUMENT, "host callback: invalid image handle");
    for (uint32_t i = 0; i < n_buffers; i++) if (buffers[i] >= g->sbufs.size()) return setErr(PLR_ERR_INVALID_ARGUMENT, "host callback: invalid storage buffer handle");
    if (int rc = plr_set_host_callback_execution(callback, user, name)) return rc;
    Execution& x = g->executions.back();
    x.callbackAccessKnown = true;
    for (uint32_t i = 0; i < n_images; i++) x.access.push_back({resolveImage(images[i])->dev, true});
    for (uint32_t i = 0; i < n_buffers; i++) x.access.push_back({g->sbufs[buffers[i]].dev, true});
    return PLR_OK;
}

int plr_set_host_callback_execution(plr_host_callback callback, void* user, const char* name) {
    NEED_INIT();
    if (!callback) return setErr(PLR_ERR_INVALID_ARGUMENT, "callback is null");
    g->executions.emplace_back();
    Execution& x = g->executions.back();
    x.pass = PLR_INVALID_INDEX;
    x.callback = callback;
    x.callbackUser = user;
    x.callbackName = g->callbackNames.insert(name ? name : "host callback").first->c_str();
    return PLR_OK;
}

int plr_prepare_for_drawcall_recording(void) { NEED_INIT(); return PLR_OK; }

static int queueFill(void* dst, size_t cap, const void* data, size_t size) {
    if (!data && size) return setErr(PLR_ERR_INVALID_ARGUMENT, "buffer data is null");
    if (size > cap) return setErr(PLR_ERR_INVALID_ARGUMENT, "buffer data (" + std::to_string(size) + " B) larger than the buffer (" + std::to_string(cap) + " B)");
    if (size == 0) return PLR_OK;
    const size_t off = (g->fillData.size() + 15) & ~(size_t)15;
    g->fillData.resize(off + size);
    std::memcpy(g->fillData.data() + off, data, size);
    g->fills.push_back({dst, size, off});
    return PLR_OK;
}

int plr_set_uniform_buffer_data(plr_uniform_buffer_handle buffer, const void* data, size_t size) {
    NEED_INIT();
    if (buffer >= g->ubufs.size()) return setErr(PLR_ERR_INVALID_ARGUMENT, "invalid uniform buffer handle");
    return queueFill(g->ubufs[buffer].dev, g->ubufs[buffer].size, data, size);
}

int plr_set_storage_buffer_data(plr_storage_buffer_handle buffer, const void* data, size_t size) {
    NEED_INIT();
    if (buffer >= g->sbufs.size()) return setErr(PLR_ERR_INVALID_ARGUMENT, "invalid storage buffer handle");
    return queueFill(g->sbufs[buffer].dev, g->sbufs[buffer].size, data, size);
}

int plr_set_global_descriptor_set_layout(const plr_shader_layout* l) {
    NEED_INIT();
    if (!l) return setErr(PLR_ERR_INVALID_ARGUMENT, "layout is null");
    if ((l->sampler_binding_count && !l->sampler_bindings) || (l->sampled_image_binding_count && !l->sampled_image_bindings) || (l->uniform_buffer_binding_count && !l->uniform_buffer_bindings) ||
        (l->storage_image_binding_count && !l->storage_image_bindings) || (l->storage_buffer_binding_count && !l->storage_buffer_bindings))
        return setErr(PLR_ERR_INVALID_ARGUMENT, "layout: a binding list is null");
    // set 0 as every kernel reads it (global.inc:4-42): binding 0 = the `global` uniform buffer, 1..8 = samplers, 9 = the noise texture of the graphics passes
    uint32_t ub = 0, smp = 0, img = 0;
    for (uint32_t i = 0; i < l->uniform_buffer_binding_count; i++) {
        if (l->uniform_buffer_bindings[i] != 0) return setErr(PLR_ERR_BINDING, "global descriptor set layout: the only uniform buffer of set 0 is `global` at binding 0 (global.inc:4)");
        ub |= 1u;
    }
    for (uint32_t i = 0; i < l->sampler_binding_count; i++) {
        const uint32_t b = l->sampler_bindings[i];
        if (b < 1 || b > 8) return setErr(PLR_ERR_BINDING, "global descriptor set layout: samplers live at bindings 1..8 (global.inc:35-42), got " + std::to_string(b));
        smp |= 1u << b;
    }
    for (uint32_t i = 0; i < l->sampled_image_binding_count; i++) {
        const uint32_t b = l->sampled_image_bindings[i];
        if (b >= (uint32_t)kMaxBindings || b <= 8) return setErr(PLR_ERR_BINDING, "global descriptor set layout: sampled image at binding " + std::to_string(b) + " collides with the uniform buffer / samplers");
        img |= 1u << b;
    }
    if (l->storage_image_binding_count || l->storage_buffer_binding_count) return setErr(PLR_ERR_BINDING, "global descriptor set layout: set 0 holds no storage images or storage buffers");
    if (!(ub & 1u)) return setErr(PLR_ERR_BINDING, "global descriptor set layout: the `global` uniform buffer at binding 0 is missing");
    g->layoutUbufs = ub; g->layoutSamplers = smp; g->layoutSampled = img; g->layoutSet = true;
    return PLR_OK;
}

int plr_set_global_descriptor_set_resources(const plr_pass_resources* r) {
    NEED_INIT();
    if (!r) return setErr(PLR_ERR_INVALID_ARGUMENT, "resources is null");
    if (g->layoutSet) { // a resource at a binding the declared layout does not have (RenderBackend.cpp validates against the set's layout the same way)
        for (uint32_t i = 0; i < r->uniform_buffer_count; i++)
            if (r->uniform_buffers[i].binding >= 32u || !((g->layoutUbufs >> r->uniform_buffers[i].binding) & 1u))
                return setErr(PLR_ERR_BINDING, "global descriptor set: uniform buffer at binding " + std::to_string(r->uniform_buffers[i].binding) + " is not in the layout");
        for (uint32_t i = 0; i < r->sampler_count; i++)
            if (r->samplers[i].binding >= 32u || !((g->layoutSamplers >> r->samplers[i].binding) & 1u))
                return setErr(PLR_ERR_BINDING, "global descriptor set: sampler at binding " + std::to_string(r->samplers[i].binding) + " is not in the layout");
        for (uint32_t i = 0; i < r->sampled_image_count; i++)
            if (r->sampled_images[i].binding >= 32u || !((g->layoutSampled >> r->sampled_images[i].binding) & 1u))
                return setErr(PLR_ERR_BINDING, "global descriptor set: sampled image at binding " + std::to_string(r->sampled_images[i].binding) + " is not in the layout");
    }
    for (uint32_t i = 0; i < r->sampler_count; i++)
        if (r->samplers[i].sampler >= g->samplers.size()) return setErr(PLR_ERR_INVALID_ARGUMENT, "global descriptor set: invalid sampler handle");
    for (uint32_t i = 0; i < r->uniform_buffer_count; i++) {
        if (r->uniform_buffers[i].binding == 0) {
            if (r->uniform_buffers[i].buffer >= g->ubufs.size()) return setErr(PLR_ERR_INVALID_ARGUMENT, "invalid global uniform buffer handle");
            if (g->ubufs[r->uniform_buffers[i].buffer].size < sizeof(GlobalUbo)) return setErr(PLR_ERR_INVALID_ARGUMENT, "global uniform buffer smaller than 340 bytes");
            if (g->globalUbo != r->uniform_buffers[i].buffer) {
                g->globalShadowValid = false;
                if (int rc = joinAsyncTail()) return rc;
                for (void*& c : g->globalCopies) { if (c) hipFree(c); c = nullptr; } // copies of another buffer: the kernels read the real one until its next fill
            }
            g->globalUbo = r->uniform_buffers[i].buffer;
        }
    }
    // samplers at bindings 1..8: their filter/address semantics are fixed by the binding number in every kernel,
    // exactly as each GLSL shader names g_sampler_* explicitly (resources/shaders/global.inc:35-42)
    return PLR_OK;
}

static int fillPass(PassRes& p, const plr_compute_pass_desc* desc) {
    if (!desc || !desc->src_path_relative) return setErr(PLR_ERR_INVALID_ARGUMENT, "pass description / shader path is null");
    ShaderEntry found;
    const ShaderEntry* entry = findShader(desc->src_path_relative, &found) ? &found : nullptr;
    if (!entry) return setErr(PLR_ERR_UNKNOWN_SHADER, std::string("no HIP kernel for shader '") + desc->src_path_relative + "'");
    p.shader = desc->src_path_relative;
    if (desc->name) p.name = desc->name;
    p.fn = entry->fn;
    p.fast = entry->fast;
    p.readsBindless = shaderReadsBindless(entry->name);
    p.spec.clear();
    for (uint32_t i = 0; i < desc->specialisation_constant_count; i++) {
        const auto& s = desc->specialisation_constants[i];
        SpecConstant c;
        c.location = s.location;
        if (s.size) c.data.assign((const uint8_t*)s.data, (const uint8_t*)s.data + s.size);
        p.spec.push_back(std::move(c));
    }
    return PLR_OK;
}

int plr_update_compute_pass_shader_description(plr_pass_handle pass, const plr_compute_pass_desc* desc) {
    NEED_INIT();
    if (pass >= g->passes.size()) return setErr(PLR_ERR_INVALID_ARGUMENT, "invalid pass handle");
    plr_compute_pass_desc d = *desc;
    std::string keepName = g->passes[pass]->name;
    int rc = fillPass(*g->passes[pass], &d);
    if (!desc->name) g->passes[pass]->name = keepName;
    return rc;
}

int plr_create_compute_pass(const plr_compute_pass_desc* desc, plr_pass_handle* out_pass) {
    NEED_INIT();
    if (!out_pass) return setErr(PLR_ERR_INVALID_ARGUMENT, "out_pass is null");
    auto p = std::make_unique<PassRes>();
    int rc = fillPass(*p, desc);
    if (rc) return rc;
    g->passes.push_back(std::move(p));
    *out_pass = (plr_pass_handle)(g->passes.size() - 1);
    return PLR_OK;
}

// ---- deferred buffer fills (plr_set_uniform / storage_buffer_data), applied in call order at the start of plr_render_frame.
// Round 3 issued one hipMemcpyAsync per fill: five 4..340-byte copies + the rotating copy of the global buffer = six blit kernels of ~6 us each in
// front of every frame's first pass, on the launch stream (profiles/r04_frame_timeline.txt: 34 us of copies + two event records between the last
// kernel of frame N and the first of frame N + 1). Now the frame's fills are ONE kernel: the host writes a table {destination, offset, size} and the
// payloads into a pinned slot, the kernel reads the slot over PCIe (a kilobyte: one round trip) and stores to every destination, in call order.
constexpr uint32_t kFillKernelMaxBytes = 65536; // larger fills (scene set-up) keep the copy engine
__global__ __launch_bounds__(256) void applyFillsKernel(uint8_t* __restrict__ slot, uint64_t serial) { applyFillsBlock(slot, serial); } // backend.h

// The small fills of a frame are not applied by a launch of their own when the frame's first launch can host them: flushFills leaves the table pending, and whoever
// launches first either takes it (a fused launcher registered for that: the frame front's first kernel touches no host-filled buffer, so one more block of it applies
// the table - one dependent launch less in front of every frame) or applies it now.
static int applyPendingFillsNow() {
    if (!g->pendingFillSlot) return PLR_OK;
    applyFillsKernel<<<1, 256, 0, g->stream>>>(g->pendingFillSlot, 0);
    HIP_TRY(hipGetLastError());
    g->mainOps++;
    g->pendingFillSlot = nullptr;
    if (g->pendingFillEvent) { HIP_TRY(hipEventRecord(g->pendingFillEvent, g->stream)); g->pendingFillEvent = nullptr; }
    return PLR_OK;
}

static int flushFills() {
    if (g->fills.empty()) return PLR_OK;
    // a table left pending by an earlier flush (fills flushed twice without a launch in between: an early error return of a frame) goes first, before any
    // kernel or copy of the newer fills is enqueued - call order
    if (int rc = applyPendingFillsNow()) return rc;
    const void* globalDev = g->globalUbo != PLR_INVALID_INDEX ? g->ubufs[g->globalUbo].dev : nullptr;
    bool globalFilled = false;
    for (const auto& f : g->fills) globalFilled = globalFilled || (globalDev && f.dst == globalDev);
    const bool copyGlobal = globalFilled && g->asyncTail;
    // slot layout: [count | entries (one more for the rotating copy of the global buffer) | payloads as queued (16-byte aligned offsets)]
    const size_t entryBytes = (g->fills.size() + 1) * sizeof(FillEntry);
    const size_t payloadBase = (kFillTableHeader + entryBytes + 15) & ~(size_t)15;
    const size_t need = payloadBase + g->fillData.size();
    Backend::PinnedSlot& slot = g->pinnedSlots[g->pinnedNext];
    g->pinnedNext = (g->pinnedNext + 1) % Backend::kPinnedSlots;
    if (slot.busy) { // the frame that used this slot, kPinnedSlots frames ago
        if (slot.eventPending) HIP_TRY(hipEventSynchronize(slot.free));
        else {
            volatile uint64_t* done = (volatile uint64_t*)((uint8_t*)slot.host + 8);
            for (uint32_t spins = 0; *done != slot.serial; spins++) {
                if (spins > 2000) { HIP_TRY(hipStreamSynchronize(g->stream)); break; } // (a kernel that never ran: an earlier launch error)
                std::this_thread::yield();
            }
        }
        slot.busy = slot.eventPending = false;
    }
    if (slot.size < need) {
        if (slot.host) hipHostFree(slot.host);
        slot.size = std::max<size_t>(need * 2, 64 << 10);
        HIP_TRY(hipHostMalloc(&slot.host, slot.size, hipHostMallocDefault));
    }
    uint8_t* host = (uint8_t*)slot.host;
    std::memcpy(host + payloadBase, g->fillData.data(), g->fillData.size());
    // a fill of a buffer the asynchronous tail of the previous frame still uses waits for it - except the global uniform buffer, which every pass
    // reads: the kernels read rotating device copies of it (below), so the tail keeps the values of its own frame
    for (const auto& f : g->fills)
        if (f.dst != globalDev && hazardWithTail({Access{f.dst, true}})) { if (int rc = joinAsyncTail()) return rc; break; }
    uint32_t count = 0;
    bool usedCopyEngine = false;
    FillEntry* entries = (FillEntry*)(host + kFillTableHeader);
    const FillOrder* lastGlobal = nullptr;
    for (const auto& f : g->fills) {
        if (f.size > kFillKernelMaxBytes || payloadBase + f.stagingOffset > 0xffffffffull) {
            // a big fill (scene set-up) keeps the copy engine. Call order: the small fills queued so far go first; the table header is re-used
            // afterwards, so the kernel that reads it has to be done (rare path: a synchronisation here costs nothing that matters)
            if (count) {
                *(uint32_t*)host = count;
                applyFillsKernel<<<1, 256, 0, g->stream>>>(host, 0);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipStreamSynchronize(g->stream));
                count = 0;
            }
            HIP_TRY(hipMemcpyAsync(f.dst, host + payloadBase + f.stagingOffset, f.size, hipMemcpyHostToDevice, g->stream));
            g->mainOps++;
            usedCopyEngine = true;
        } else {
            entries[count++] = FillEntry{(uint64_t)(uintptr_t)f.dst, (uint32_t)(payloadBase + f.stagingOffset), (uint32_t)f.size};
        }
        if (globalDev && f.dst == globalDev) { // fills are applied in call order: the shadow ends up as the buffer does
            std::memcpy(&g->globalShadow, (uint8_t*)slot.host + payloadBase + f.stagingOffset, std::min(f.size, sizeof(GlobalUbo)));
            if (f.size >= sizeof(GlobalUbo)) g->globalShadowValid = true;
            lastGlobal = &f;
        }
    }
    if (copyGlobal) {
        // next rotating copy of the global uniform buffer = the buffer as it will be; a pending tail may still read that copy (two frames back): join first
        const uint32_t next = g->globalCopyIndex ^ 1u;
        if (!g->globalCopies[next]) HIP_TRY(hipMalloc(&g->globalCopies[next], sizeof(GlobalUbo)));
        if (hazardWithTail({Access{g->globalCopies[next], true}})) if (int rc = joinAsyncTail()) return rc;
        if (lastGlobal && lastGlobal->size >= sizeof(GlobalUbo) && lastGlobal->size <= kFillKernelMaxBytes) {
            entries[count++] = FillEntry{(uint64_t)(uintptr_t)g->globalCopies[next], (uint32_t)(payloadBase + lastGlobal->stagingOffset), (uint32_t)sizeof(GlobalUbo)};
        } else {
            // a partial fill of the global buffer: the copy must be the whole buffer as it is after the fills
            if (count) { *(uint32_t*)host = count; applyFillsKernel<<<1, 256, 0, g->stream>>>(host, 0); HIP_TRY(hipGetLastError()); HIP_TRY(hipStreamSynchronize(g->stream)); count = 0; }
            HIP_TRY(hipMemcpyAsync(g->globalCopies[next], globalDev, sizeof(GlobalUbo), hipMemcpyDeviceToDevice, g->stream));
            g->mainOps++;
        }
        g->globalCopyIndex = next;
    }
    slot.serial = ++g->fillSerial;
    // completion of the slot: an event behind the fill kernel (default), or - PLR_FILL_POLL=1 - a word the kernel stores into the slot and the host polls
    static const bool pollSlot = std::getenv("PLR_FILL_POLL") && std::atoi(std::getenv("PLR_FILL_POLL")) != 0;
    if (count && !usedCopyEngine && pollSlot) {
        *(uint32_t*)host = count;
        *(volatile uint64_t*)(host + 8) = 0;
        applyFillsKernel<<<1, 256, 0, g->stream>>>(host, slot.serial);
        HIP_TRY(hipGetLastError());
        g->mainOps++;
    } else {
        slot.eventPending = true; // its event is recorded behind whatever reads the slot last
        if (count && !usedCopyEngine) { *(uint32_t*)host = count; g->pendingFillSlot = host; g->pendingFillEvent = slot.free; } // applied by the frame's first launch
        else {
            if (count) { *(uint32_t*)host = count; applyFillsKernel<<<1, 256, 0, g->stream>>>(host, 0); HIP_TRY(hipGetLastError()); }
            g->mainOps++;
            HIP_TRY(hipEventRecord(slot.free, g->stream)); // a copy-engine transfer read the slot too (set-up frames): an event covers both
        }
    }
    slot.busy = true;
    for (const auto& f : g->fills) g->lastFillDsts.push_back(f.dst);
    g->fills.clear();
    g->fillData.clear();
    return PLR_OK;
}

static int flushBindless() {
    if (!g->bindlessDirty) return PLR_OK;
    const uint32_t n = (uint32_t)g->images.size();
    if (n == 0) { g->bindlessHost.clear(); g->bindlessDirty = false; return PLR_OK; }
    if (g->bindlessCapacity < n) {
        HIP_TRY(hipStreamSynchronize(g->stream));
        if (g->bindlessDev) hipFree(g->bindlessDev);
        g->bindlessCapacity = std::max(n * 2, 64u);
        HIP_TRY(hipMalloc((void**)&g->bindlessDev, sizeof(ImgView) * g->bindlessCapacity));
    }
    std::vector<ImgView>& table = g->bindlessHost;
    table.resize(n);
    for (uint32_t i = 0; i < n; i++) {
        if (g->images[i].dev) table[i] = makeView(g->images[i], 0);
        else { table[i].ptr = nullptr; table[i].w = table[i].h = table[i].d = 0; table[i].fmt = -1; }
    }
    HIP_TRY(hipMemcpyAsync(g->bindlessDev, table.data(), sizeof(ImgView) * n, hipMemcpyHostToDevice, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    g->bindlessDirty = false;
    return PLR_OK;
}

static int timingEvent(hipEvent_t* out) {
    if (g->eventsUsed == g->passEvents.size()) {
        hipEvent_t ev;
        HIP_TRY(hipEventCreate(&ev));
        g->passEvents.push_back(ev);
    }
    *out = g->passEvents[g->eventsUsed++];
    return PLR_OK;
}
static int beginSegment(const char* name) {
    hipEvent_t ev;
    if (int rc = timingEvent(&ev)) return rc;
    g->segments.push_back({g->eventsUsed - 1, name});
    HIP_TRY(hipEventRecord(ev, g->curStream ? g->curStream : g->stream));
    return PLR_OK;
}
static int endSegment() {
    hipEvent_t ev;
    if (int rc = timingEvent(&ev)) return rc;
    HIP_TRY(hipEventRecord(ev, g->curStream ? g->curStream : g->stream));
    return PLR_OK;
}
void PassCtx::splitTiming(const char* label) const {
    if (!g || !g->timingNow || g->segments.empty()) return;
    // close the running segment under "<pass> (<label>)" and open a new one under the pass name
    const char* pass = g->currentPassName ? g->currentPassName : "";
    g->segments.back().name = g->callbackNames.insert(std::string(pass) + " (" + label + ")").first->c_str();
    if (endSegment() != PLR_OK) return;
    (void)beginSegment(pass);
}

void PassCtx::splitTimingBetween(const PassCtx& first, const PassCtx& second) {
    if (!g || !g->timingNow || g->segments.empty() || !first.passName || !second.passName) return;
    g->segments.back().name = first.passName;
    if (endSegment() != PLR_OK) return;
    (void)beginSegment(second.passName);
}


// ---------------------------------------------------------------- stream scheduler
// The recorded executions are launched in order, but not all on one stream: an execution only has to wait for the earlier ones it
// has a hazard with (read-after-write, write-after-read, write-after-write on an allocation it binds). Executions without such a
// dependency on the tail of the main stream go to one of kSideStreams side streams and overlap with it - in the frame of this
// hot path the luminance histogram -> exposure chain (small, latency-bound kernels) runs beside the depth pyramid / culling chain.
// Cross-stream dependencies are events; every side stream joins the main stream before a host callback and at the end of the
// list, so a frame as a whole is still ordered on the main stream (what plr_wait_for_gpu_idle, readbacks and the band exchange
// callbacks rely on).
// Off by default (plr_set_stream_overlap(1) / PLR_STREAM_OVERLAP=1 turns it on): on MI355X / ROCm 7.2 a cross-stream dependency
// (event record + barrier packet on another hardware queue) costs 15-20 us, more than the 4K frame's independent chain (four small
// exposure kernels, 53 us) can win back. Measured ms per 4K frame: one in-order stream 1.187; exposure chain on one side stream with
// two waits per frame 1.206; two side streams 1.277; per-execution placement on three side streams 1.332. Results are byte-identical
// either way (tests/test_full_frame.py); the scheduler pays off only for longer independent chains (sky LUT / froxel producers).
constexpr int kMaxStreams = 1 + Backend::kSideStreams;
struct PlanNode {
    int stream = 0;            // 0 = main, 1.. = side stream
    std::vector<int> waits;    // executions on other streams to wait for
    bool signal = false;       // somebody on another stream waits for this execution
    int after[kMaxStreams];    // once this execution has started, everything up to index after[s] on stream s has completed or precedes it
};

static int orderEvent(hipEvent_t* out) {
    if (g->orderEventsUsed == g->orderEvents.size()) {
        hipEvent_t ev;
        HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        g->orderEvents.push_back(ev);
    }
    *out = g->orderEvents[g->orderEventsUsed++];
    return PLR_OK;
}

// executions [first, last) contain no host callback; returns in mainAfter what the main stream is ordered after at the end of the run
static void planStreams(size_t first, size_t last, std::vector<PlanNode>& plan, int nStreams, int* mainAfter) {
    struct ResState { const void* key; int lastWriter; std::vector<int> readers; };
    std::vector<ResState> states;
    int tail[kMaxStreams], load[kMaxStreams];
    for (int st = 0; st < kMaxStreams; st++) { tail[st] = -1; load[st] = 0; }
    std::vector<int> deps;
    auto stateOf = [&](const void* key) -> ResState& {
        for (auto& s : states) if (s.key == key) return s;
        states.push_back({key, -1, {}});
        return states.back();
    };
    for (size_t i = first; i < last; i++) {
        const Execution& x = g->executions[i];
        deps.clear();
        auto addDep = [&](int d) { if (d >= 0 && std::find(deps.begin(), deps.end(), d) == deps.end()) deps.push_back(d); };
        for (const Access& a : x.access) {
            const ResState& st = stateOf(a.key);
            addDep(st.lastWriter);
            if (a.write) for (int r : st.readers) addDep(r);
        }
        for (const Access& a : x.access) {
            ResState& st = stateOf(a.key);
            if (a.write) { st.lastWriter = (int)i; st.readers.clear(); }
            else if (st.lastWriter != (int)i && (st.readers.empty() || st.readers.back() != (int)i)) st.readers.push_back((int)i);
        }
        PlanNode& node = plan[i];
        // The stream of the latest dependency: stream order then covers every dependency on that stream and the chain stays together.
        // An execution that depends on nothing launched in this run starts a new chain on the stream with the fewest executions so far
        // (the main stream on a tie). A cross-stream wait costs several microseconds on this hardware, so chains are what is spread
        // over streams, not single executions.
        int chosen = 0;
        if (!deps.empty()) chosen = plan[*std::max_element(deps.begin(), deps.end())].stream;
        else
            for (int st = 1; st < nStreams; st++) if (load[st] < load[chosen]) chosen = st;
        node.stream = chosen;
        for (int st = 0; st < kMaxStreams; st++) node.after[st] = tail[chosen] >= 0 ? plan[tail[chosen]].after[st] : -1;
        node.after[chosen] = (int)i;
        // wait for the latest dependency on every other stream unless an earlier wait of this stream already covers it
        int latest[kMaxStreams];
        for (int& l : latest) l = -1;
        for (int d : deps) if (plan[d].stream != chosen) latest[plan[d].stream] = std::max(latest[plan[d].stream], d);
        for (int st = 0; st < kMaxStreams; st++) {
            const int d = latest[st];
            if (d < 0 || d <= node.after[st]) continue;
            node.waits.push_back(d);
            plan[d].signal = true;
            for (int t = 0; t < kMaxStreams; t++) node.after[t] = std::max(node.after[t], plan[d].after[t]);
        }
        tail[chosen] = (int)i;
        load[chosen]++;
    }
    for (int st = 0; st < kMaxStreams; st++) mainAfter[st] = tail[0] >= 0 ? plan[tail[0]].after[st] : -1;
}

static void prepareCtx(Execution& x, hipStream_t stream, const GlobalUbo* globalPtr) {
    PassRes& p = *g->passes[x.pass];
    x.ctx.stream = stream;
    x.ctx.global = globalPtr;
    x.ctx.globalHost = globalPtr && g->globalShadowValid ? &g->globalShadow : nullptr;
    x.ctx.elidableStorage = 0;
    x.ctx.elidedStorage = 0;
    x.ctx.earlyPartDone = false;
    x.ctx.firstRows[0] = x.firstRows[0]; x.ctx.firstRows[1] = x.firstRows[1];
    x.ctx.firstCols[0] = x.firstCols[0]; x.ctx.firstCols[1] = x.firstCols[1];
    x.ctx.edgeSignal = nullptr; x.ctx.edgeCounter = nullptr; x.ctx.edgeValue = 0; x.ctx.edgeSignalHonoured = false;
    x.ctx.frameSerial = g->frameSerial;
    x.ctx.bindless = g->bindlessDev;
    x.ctx.bindlessHost = g->bindlessHost.size() == g->images.size() ? g->bindlessHost.data() : nullptr;
    x.ctx.bindlessCount = (uint32_t)g->images.size();
    x.ctx.spec = &p.spec;
    x.ctx.err = &g_err;
    x.ctx.scratchSlot = &p.scratch;
    x.ctx.scratchSize = &p.scratchSize;
    x.ctx.debugSig = g->debugSig;
    x.ctx.debugSigWords = g->debugSigWords;
    x.ctx.passName = p.name.c_str();
}

// images a launcher left unwritten (PassCtx::elidedStorage): a host download must not return last frame's bytes silently
static ImageRes* imageOfAllocation(const void* base) {
    for (ImageRes& im : g->images) if (im.dev && im.dev == base) return &im;
    for (ImageRes& im : g->transient) if (im.dev && im.dev == base) return &im;
    return nullptr;
}
// after a launch: the storage images the execution binds are written now (stale flag cleared) - or, where the launcher said so, left unwritten (stale from here on)
static void noteElidedImages(const Execution& x) {
    const PassCtx& cx = x.ctx;
    for (int b = 0; b < kMaxBindings; b++) {
        if (!cx.hasStorage(b)) continue;
        bool written = false;
        for (const Access& a : x.access) if (a.write && a.key == cx.storage[b].ptr) written = true; // (mip 0 of an image the pass writes; a read-only storage binding is no write)
        if (!written) continue;
        ImageRes* im = imageOfAllocation(cx.storage[b].ptr);
        if (!im) continue;
        if ((cx.elidedStorage >> b) & 1u) { im->elided = true; im->elidedReader = (const void*)cx.consumer; im->elidedSerial = g->frameSerial; }
        else im->elided = false;
    }
}
// before a launch: does the execution read an image that holds stale texels? `group`: the executions launched together (a fused launch reads what its own members
// elide in registers). Host callbacks with a resource list are checked for every image they touch (a halo exchange sends what it "writes")
static int refuseStaleReads(const Execution& x, const Execution* group, size_t groupCount) {
    for (const Access& a : x.access) {
        if (a.key == kBindlessKey || (a.write && !x.callback)) continue;
        const ImageRes* im = imageOfAllocation(a.key);
        if (!im || !im->elided) continue;
        if (im->elidedSerial == g->frameSerial && im->elidedReader == (const void*)&x.ctx) continue; // the consumer the producer packed its result for
        bool insideGroup = false;
        for (size_t k = 0; k < groupCount && !insideGroup; k++)
            for (const Access& w : group[k].access) if (w.write && w.key == a.key) { insideGroup = true; break; }
        if (insideGroup) continue;
        const std::string who = x.callback ? std::string("host callback '") + x.callbackName + "'" : "pass '" + g->passes[x.pass]->name + "' (" + g->passes[x.pass]->shader + ")";
        return setErr(PLR_ERR_UNSUPPORTED, who + " reads an image that was not written: an earlier fused launch kept its result in registers / packed texels (pass fusion level 2) "
                                           "and this frame is recorded differently; plr_set_pass_fusion(1) keeps intermediates");
    }
    return PLR_OK;
}

// Pass fusion level 2 for a producer whose consumer takes its output through another channel (PassCtx::consumer: the spatial GI filter gathers the packed
// texels its producer writes, fused_gi.h): a storage image of the producer is elidable when nothing can ever read what the producer would have written to it:
//  * behind the producer, nothing but the consumer touches it in this frame (a later "write" may read as well: a storage binding, an exchange callback);
//  * the frame's first access to it is a write by a producer of a consumer link - this execution itself or another (the trace): those launchers only store to
//    their outputs, so the next frame, recorded alike, overwrites the image before anything reads it.
// (A host that asks for the image gets the loud "not written" error, as for a fused launch.)
static void markElidableBehindConsumer(Execution& x) {
    const size_t index = (size_t)(&x - g->executions.data());
    for (const Execution& y : g->executions) if (y.callback && !y.callbackAccessKnown) return; // a host callback that may read anything
    auto linkProducer = [](const Execution& y) {
        if (y.callback) return false;
        for (const ConsumerLink& link : consumerLinks()) if (link.producer == g->passes[y.pass]->shader) return true;
        return false;
    };
    for (int b = 0; b < kMaxBindings; b++) {
        if (!x.ctx.hasStorage(b)) continue;
        const void* key = nullptr;
        for (const Access& a : x.access) if (a.write && a.key == x.ctx.storage[b].ptr) key = a.key;
        if (!key) continue;
        bool ok = true, first = true;
        for (size_t j = 0; j < g->executions.size() && ok; j++)
            for (const Access& a : g->executions[j].access) {
                if (a.key != key) continue;
                if (first) { first = false; if (!a.write || !linkProducer(g->executions[j])) ok = false; }
                if (j > index && &g->executions[j].ctx != x.ctx.consumer) ok = false;
            }
        if (ok) x.ctx.elidableStorage |= 1u << b;
    }
}

static int launchExecution(Execution& x, hipStream_t stream, const GlobalUbo* globalPtr, bool timed) {
    PassRes& p = *g->passes[x.pass];
    if (int frc = applyPendingFillsNow()) return frc; // the frame's fills, if its first launch is this one
    prepareCtx(x, stream, globalPtr);
    if (g->fusion >= 2 && g->mathMode == PLR_MATH_FAST && !g->debugSig && x.ctx.consumer) markElidableBehindConsumer(x);
    if (int src = refuseStaleReads(x, &x, 1)) return src;
    g->currentPassName = p.name.c_str();
    g->curStream = stream;
    if (timed) if (int trc = beginSegment(p.name.c_str())) return trc;
    touchAccesses(x.access); // the images it writes have new contents from here on (contentVersionOf)
    const bool signalled = x.edgesFirst() && g->edgeSignal;
    if (signalled) { x.ctx.edgeSignal = g->edgeSignal; x.ctx.edgeCounter = g->edgeCounter; x.ctx.edgeValue = ++g->edgeSerial; }
    if (stream == g->stream) g->mainOps++;
    int rc = (g->mathMode == PLR_MATH_FAST && p.fast) ? p.fast(x.ctx) : kUseGeneralKernel;
    if (rc == kUseGeneralKernel) {
        x.ctx.edgeSignalHonoured = false; // (a fast launcher that ordered its blocks and then declined: the general kernel raises nothing - ADVICE r04)
        if (g->mathMode == PLR_MATH_FAST) { // not silently: the caller can ask
            g->lastGeneral++;
            if (g->lastGeneralNames.size() < 2048) g->lastGeneralNames += (g->lastGeneralNames.empty() ? "" : ", ") + p.name + " [" + p.shader + "]";
        }
        if (!p.fn) if (int lrc = resolveExactLauncher(&p.fn, p.shader)) return lrc;
        rc = p.fn(x.ctx);
    }
    if (rc) { g_err = "pass '" + p.name + "' (" + p.shader + "): " + g_err; return rc; }
    // a kernel that did not order its rows (general kernel, edges that are not whole block rows): the signal is raised behind the whole launch
    if (signalled && !x.ctx.edgeSignalHonoured) HIP_TRY(hipStreamWriteValue32(stream, g->edgeSignal, x.ctx.edgeValue, 0));
    if (timed) if (int trc = endSegment()) return trc;
    noteElidedImages(x);
    return PLR_OK;
}

// ---- early parts of fused sequences (backend.h EarlyPart). planEarlyParts runs once per launchAll, after the executions have their final order: for every
// sequence whose fused launcher has an early part and accepts the bindings it finds the earliest position `at` of the recorded list where that part may start -
// behind the last execution that writes (or the last host callback that may write) anything the early kernels read - and keeps the plan only if at least one
// compute execution lies between that position and the sequence: something to run beside. launchEarlyPartsAt(i) issues the plans of position i on the early
// stream before execution i is launched; tryFusedLaunch makes the launch stream wait for the part's event and tells the launcher (PassCtx::earlyPartDone).
struct EarlyPlan { size_t at = 0, group = 0, count = 0; const FusionEntry* f = nullptr; hipEvent_t done = nullptr; bool launched = false; };
static thread_local std::vector<EarlyPlan> g_earlyPlans;
static bool fusionNamesMatch(const FusionEntry& f, size_t i, size_t n) {
    const size_t m = f.shaders.size();
    if (i + m > n || m > 8) return false;
    for (size_t k = 0; k < m; k++) {
        const Execution& y = g->executions[i + k];
        if (y.callback || y.edgesFirst() || g->passes[y.pass]->shader != f.shaders[k]) return false;
    }
    return true;
}
static void planEarlyParts(const GlobalUbo* globalPtr) {
    g_earlyPlans.clear();
    if (!g->earlyParts || !g->fusion || g->mathMode != PLR_MATH_FAST || g->debugSig || g->overlap || !g->earlyStream) return;
    const size_t n = g->executions.size();
    std::vector<const void*> reads;
    for (size_t i = 0; i < n; i++) {
        if (g->executions[i].callback) continue;
        const FusionEntry* f = nullptr;
        for (const FusionEntry& e : fusions()) if (fusionNamesMatch(e, i, n)) { f = &e; break; } // the entry tryFusedLaunch tries first
        if (!f || !f->early) continue;
        const size_t m = f->shaders.size();
        bool plain = true;
        for (size_t k = 0; k < m; k++) plain = plain && !g->executions[i + k].asyncTail;
        if (!plain) continue;
        const PassCtx* ctxs[8];
        for (size_t k = 0; k < m; k++) { prepareCtx(g->executions[i + k], g->earlyStream, globalPtr); ctxs[k] = &g->executions[i + k].ctx; }
        reads.clear();
        EarlyPart query;
        query.mode = EarlyPart::Query;
        query.reads = &reads;
        if (f->early(ctxs, m, query) != 0) { g_err.clear(); continue; } // no early part for these bindings (an error is reported by the fused launch itself)
        auto isRead = [&](const void* key) { return key && std::find(reads.begin(), reads.end(), key) != reads.end(); };
        size_t at = i;
        while (at > 0) {
            const Execution& y = g->executions[at - 1];
            if (y.callback && !y.callbackAccessKnown) break; // may write anything
            bool writes = false;
            for (const Access& a : y.access) writes = writes || (a.write && a.key != kBindlessKey && isRead(a.key));
            if (writes) break;
            at--;
        }
        // a pending buffer fill of something the part reads is applied by (or in front of) the frame's first launch: the part starts behind that
        bool filled = false;
        for (const void* dst : g->lastFillDsts) filled = filled || isRead(dst);
        if (filled && at == 0) at = 1;
        // what the asynchronous tail of the previous frame still writes: not worth a join
        bool tailHazard = false;
        for (const void* key : reads) tailHazard = tailHazard || (key && hazardWithTail({Access{key, false}}));
        size_t beside = 0;
        for (size_t j = at; j < i; j++) beside += g->executions[j].callback ? 0 : 1;
        // experiment hook (tools/early_overlap.sh): PLR_EARLY_AT=<substring of a pass name> starts the part in front of the first execution at or behind the legal
        // position whose pass name contains it ("self": in front of the sequence itself = the two launches back to back)
        static const char* forcedAt = std::getenv("PLR_EARLY_AT");
        if (forcedAt && *forcedAt) {
            size_t j = at;
            while (j < i && (g->executions[j].callback || g->passes[g->executions[j].pass]->name.find(forcedAt) == std::string::npos)) j++;
            at = j;
        }
        beside = 0;
        for (size_t j = at; j < i; j++) beside += g->executions[j].callback ? 0 : 1;
        if (tailHazard || (g->earlyParts < 2 && (at >= i || beside == 0))) continue;
        EarlyPlan plan;
        plan.at = at; plan.group = i; plan.count = m; plan.f = f;
        g_earlyPlans.push_back(plan);
        i += m - 1;
    }
}
static int launchEarlyPartsAt(size_t index, const GlobalUbo* globalPtr, bool timed) {
    for (EarlyPlan& plan : g_earlyPlans) {
        if (plan.at != index || plan.launched) continue;
        // order the early stream behind what the part depends on: everything launched so far in this frame - or, at the frame's start, the previous frame's late
        // part, which read the scratch memory this part overwrites: the tail's start event of that frame covers it for free (an event record is a barrier packet
        // of several microseconds on the launch stream) when nothing has gone onto the launch stream since
        if (index == 0 && g->tailStartCoversLate && g->tailStartOps == g->mainOps) HIP_TRY(hipStreamWaitEvent(g->earlyStream, g->tailStart, 0));
        else {
            hipEvent_t start;
            if (int rc = orderEvent(&start)) return rc;
            HIP_TRY(hipEventRecord(start, g->stream));
            HIP_TRY(hipStreamWaitEvent(g->earlyStream, start, 0));
        }
        const PassCtx* ctxs[8];
        for (size_t k = 0; k < plan.count; k++) { prepareCtx(g->executions[plan.group + k], g->earlyStream, globalPtr); ctxs[k] = &g->executions[plan.group + k].ctx; }
        hipStream_t previous = g->curStream;
        g->curStream = g->earlyStream;
        if (timed) {
            std::string name;
            for (size_t k = 0; k < plan.count; k++) name += (k ? " + " : "") + g->passes[g->executions[plan.group + k].pass]->name;
            name += " (" + plan.f->earlyLabel + ", early stream)";
            if (int trc = beginSegment(g->fusedNames.insert(name).first->c_str())) return trc;
        }
        EarlyPart part;
        part.mode = EarlyPart::Launch;
        const int rc = plan.f->early(ctxs, plan.count, part);
        if (rc) { g_err = "early part of fused launch '" + plan.f->label + "': " + g_err; return rc < 0 ? rc : PLR_ERR_HIP; }
        if (timed) if (int trc = endSegment()) return trc;
        g->curStream = previous;
        if (int erc = orderEvent(&plan.done)) return erc;
        HIP_TRY(hipEventRecord(plan.done, g->earlyStream));
        plan.launched = true;
        g->lastEarly++;
    }
    return PLR_OK;
}

// executions [i, last) are compute passes (no host callback). If a fused launcher matches the shaders starting at i and accepts the
// bindings, it is launched and the number of executions it covered is returned; 0: nothing fused (launch execution i on its own)
static int tryFusedLaunch(size_t i, size_t last, hipStream_t stream, const GlobalUbo* globalPtr, bool timed, size_t* covered) {
    *covered = 0;
    if (!g->fusion || g->mathMode != PLR_MATH_FAST) return PLR_OK;
    for (const FusionEntry& f : fusions()) {
        const size_t n = f.shaders.size();
        if (i + n > last) continue;
        if (g->debugSig && !f.writesSignatures) continue;
        bool match = true;
        for (size_t k = 0; k < n && match; k++) match = g->passes[g->executions[i + k].pass]->shader == f.shaders[k];
        for (size_t k = 0; k < n && match; k++) match = !g->executions[i + k].edgesFirst(); // rows-first executions are launched on their own
        if (!match) continue;
        const PassCtx* ctxs[8];
        if (n > 8) continue;
        for (size_t k = 0; k < n; k++) { prepareCtx(g->executions[i + k], stream, globalPtr); ctxs[k] = &g->executions[i + k].ctx; }
        EarlyPlan* early = nullptr;
        for (EarlyPlan& plan : g_earlyPlans) if (plan.launched && plan.group == i && plan.f == &f) early = &plan;
        if (early) {
            HIP_TRY(hipStreamWaitEvent(stream, early->done, 0));
            for (size_t k = 0; k < n; k++) g->executions[i + k].ctx.earlyPartDone = true;
        }
        if (g->fusion >= 2) {
            // which storage images of the sequence does nothing else in this frame touch? (keys are allocation bases: an image with all its mips)
            for (size_t k = 0; k < n; k++) {
                Execution& x = g->executions[i + k];
                for (int b = 0; b < kMaxBindings; b++) {
                    if (!x.ctx.hasStorage(b)) continue;
                    const void* key = nullptr;
                    for (const Access& a : x.access) if (a.write && a.key == x.ctx.storage[b].ptr) key = a.key; // mip 0 of an image the pass writes
                    if (!key) continue;
                    bool outside = false;
                    for (size_t j = 0; j < g->executions.size() && !outside; j++) {
                        if (j >= i && j < i + n) continue;
                        for (const Access& a : g->executions[j].access) if (a.key == key) { outside = true; break; }
                    }
                    if (!outside) x.ctx.elidableStorage |= 1u << b;
                }
            }
        }
        const char* label = nullptr;
        g->curStream = stream;
        if (timed) {
            std::string name;
            for (size_t k = 0; k < n; k++) name += (k ? " + " : "") + g->passes[g->executions[i + k].pass]->name;
            label = g->fusedNames.insert(name).first->c_str();
            if (int trc = beginSegment(label)) return trc;
        }
        g->currentPassName = label ? label : f.label.c_str();
        for (size_t k = 0; k < n; k++) touchAccesses(g->executions[i + k].access);
        // the frame's pending fills: a launcher registered for it gets the table with its first execution (and either hosts it in its first kernel or applies it
        // itself before launching), for every other launcher they are applied now
        Execution& first = g->executions[i];
        first.ctx.pendingFillSlot = nullptr; first.ctx.pendingFillsTaken = false; first.ctx.applyPendingFillsNow = nullptr;
        if (g->pendingFillSlot) {
            if (f.takesFills && stream == g->stream) { first.ctx.pendingFillSlot = g->pendingFillSlot; first.ctx.applyPendingFillsNow = &applyPendingFillsNow; }
            else if (int frc = applyPendingFillsNow()) return frc;
        }
        for (size_t k = 0; k < n; k++) if (int src = refuseStaleReads(g->executions[i + k], &g->executions[i], n)) return src;
        const int rc = f.fn(ctxs, n);
        if (first.ctx.pendingFillSlot) {
            if (rc == 0 && first.ctx.pendingFillsTaken) { // applied by a block of the launcher's first kernel: the slot is free behind this launch
                g->pendingFillSlot = nullptr;
                if (g->pendingFillEvent) { HIP_TRY(hipEventRecord(g->pendingFillEvent, g->stream)); g->pendingFillEvent = nullptr; }
            } else if (rc == 0 && g->pendingFillSlot) { g_err = "fused launch '" + f.label + "' neither took nor applied the frame's pending fills"; return PLR_ERR_HIP; }
            first.ctx.pendingFillSlot = nullptr; first.ctx.applyPendingFillsNow = nullptr;
        }
        if (rc == kUseGeneralKernel) {
            if (early) { g_err = "fused launch '" + f.label + "' declined the bindings its early part accepted"; return PLR_ERR_HIP; }
            if (timed) { g->segments.pop_back(); g->eventsUsed -= 1; } // the opening event stays recorded on the stream; its slot is reused
            continue;
        }
        if (rc) { g_err = "fused launch '" + f.label + "': " + g_err; return rc; }
        if (timed) if (int trc = endSegment()) return trc;
        for (size_t k = 0; k < n; k++) noteElidedImages(g->executions[i + k]);
        *covered = n;
        g->lastFused += (uint32_t)n;
        if (stream == g->stream) g->mainOps++;
        if (early) g->lateLaunchedThisFrame = true;
        return PLR_OK;
    }
    return PLR_OK;
}

// how many executions starting at i a fused launch could cover (shader names only; the launcher may still decline)
static size_t fusionWindow(size_t i, size_t last) {
    size_t best = 1;
    if (!g->fusion || g->mathMode != PLR_MATH_FAST) return best;
    for (const FusionEntry& f : fusions()) {
        const size_t n = f.shaders.size();
        if (n <= best || i + n > last) continue;
        bool match = true;
        for (size_t k = 0; k < n && match; k++) match = g->passes[g->executions[i + k].pass]->shader == f.shaders[k];
        if (match) best = n;
    }
    return best;
}


// ---- fusion across the caller's pass order. A fused launch needs its executions back to back, and the caller records in ITS order: with the input
// producers recorded as compute passes (RenderFrontend.cpp:342-405) the sky LUT passes and the light matrix sit between the members of the frame front
// (histogram chain, pyramid, culling), which then ran as six launches instead of two. The executions in between are moved out of the way when the
// recorded resources say that nothing changes: an execution is hoisted in front of the group if it shares no resource (with a write on either side) with
// the members recorded before it, or sunk behind the group if it shares none with the members recorded after it; executions moved to the same side keep
// their order, and one that moves in front of an earlier one that moves behind must not share a resource with it either. Rows-first executions and host
// callbacks end the search - except a callback recorded with its resource list, which may be sunk behind the group like an execution (round 5: band
// rendering's histogram all-reduce sits between the histogram chain and the pyramid; behind the group, the band's front is two launches).
// The order is changed in place (a replayed frame keeps it); what is launched is the same set of executions on the same resources.
static bool executionsConflict(const Execution& a, const Execution& b) {
    for (const Access& x : a.access)
        for (const Access& y : b.access) {
            if (x.key != y.key || !(x.write || y.write)) continue;
            if (x.key == kBindlessKey && x.write && y.write) continue; // two writers of two images of the global array: their own keys decide
            return true;
        }
    return false;
}
static void gatherFusionGroups() {
    if (!g->fusion || g->mathMode != PLR_MATH_FAST || g->overlap || g->debugSig) return;
    if (!g->fusionReorder) return;
    std::vector<Execution>& ex = g->executions;
    constexpr size_t kMaxMoved = 8;
    for (size_t i = 0; i < ex.size(); i++) {
        if (ex[i].callback) continue;
        for (const FusionEntry& f : fusions()) { // longest first
            const size_t m = f.shaders.size();
            if (m < 2 || g->passes[ex[i].pass]->shader != f.shaders[0] || ex[i].edgesFirst()) continue;
            std::vector<size_t> members{i}, moved;
            for (size_t j = i + 1; j < ex.size() && members.size() < m && moved.size() <= kMaxMoved; j++) {
                const Execution& y = ex[j];
                // a rows-first execution belongs to the callback recorded behind it (edge signal): it is neither fused nor moved
                if ((y.callback && !y.callbackAccessKnown) || y.asyncTail != ex[i].asyncTail || y.edgesFirst()) break;
                // a host callback recorded WITH its resource list (plr_set_host_callback_execution_on: band rendering's histogram all-reduce names the histogram
                // buffer) is an execution like any other for this purpose, except that it is only ever sunk behind the group, never hoisted in front of it
                if (y.callback) { moved.push_back(j); continue; }
                if (g->passes[y.pass]->shader == f.shaders[members.size()]) members.push_back(j);
                else moved.push_back(j);
            }
            if (members.size() != m || moved.size() > kMaxMoved) continue;
            if (moved.empty()) break; // back to back already (and the longest pattern that starts here)
            std::vector<size_t> front, back;
            bool legal = true;
            for (size_t s : moved) {
                bool hoist = !ex[s].callback, sink = true;
                for (size_t k : members) {
                    if (!executionsConflict(ex[s], ex[k])) continue;
                    if (k < s) hoist = false; else sink = false;
                }
                for (size_t b : back) if (executionsConflict(ex[s], ex[b])) hoist = false; // b was recorded before s and stays behind the group
                if (hoist) front.push_back(s);
                else if (sink) back.push_back(s);
                else { legal = false; break; }
            }
            if (!legal) continue;
            std::vector<Execution> order;
            order.reserve(members.size() + moved.size());
            for (size_t s : front) order.push_back(std::move(ex[s]));
            for (size_t k : members) order.push_back(std::move(ex[k]));
            for (size_t s : back) order.push_back(std::move(ex[s]));
            for (size_t k = 0; k < order.size(); k++) ex[i + k] = std::move(order[k]);
            i += front.size() + members.size() - 1; // the executions behind the group are looked at next
            break;
        }
    }
}

// PassCtx::consumer of every execution (backend.h PLR_REGISTER_CONSUMER_LINK)
static void linkConsumers(const GlobalUbo* globalPtr) {
    const size_t n = g->executions.size();
    for (size_t i = 0; i < n; i++) g->executions[i].ctx.consumer = nullptr;
    if (!g->fusion || g->mathMode != PLR_MATH_FAST || g->debugSig || g->overlap || consumerLinks().empty()) return;
    for (size_t i = 0; i < n; i++) {
        Execution& x = g->executions[i];
        if (x.callback) continue;
        const std::string& shader = g->passes[x.pass]->shader;
        for (const ConsumerLink& link : consumerLinks()) {
            if (link.producer != shader || x.ctx.consumer) continue;
            // keys of the images this execution writes
            std::vector<const void*> written;
            for (const Access& a : x.access) if (a.write && a.key != kBindlessKey && a.key != (const void*)g->passes[x.pass].get()) written.push_back(a.key);
            bool stop = false;
            for (size_t j = i + 1; j < n && !stop && !x.ctx.consumer; j++) {
                Execution& y = g->executions[j];
                if (y.callback || y.pass == x.pass) continue; // exchange callbacks and the pass's other row ranges do not end the search
                bool reads = false;
                for (const Access& a : y.access)
                    for (const void* key : written)
                        if (a.key == key) { if (a.write) stop = true; else reads = true; }
                if (reads && !stop && g->passes[y.pass]->shader == link.consumer) {
                    prepareCtx(y, g->stream, globalPtr);
                    x.ctx.consumer = &y.ctx;
                } else if (reads) stop = true; // somebody else reads it first: the order of packing and reading is no longer ours to reason about
            }
        }
    }
}

static int launchAll(bool timed) {
    const size_t n = g->executions.size();
    // unique across backends: host-side bookkeeping of the launchers (the spatial filter's packed rows) is keyed by device addresses, which a
    // later backend on the same thread gets handed again - an entry of this serial was written by this backend in this frame
    static std::atomic<uint64_t> serialCounter{0};
    g->frameSerial = ++serialCounter;
    g->timingNow = timed;
    if (timed) { g->segments.clear(); g->eventsUsed = 0; }
    g->orderEventsUsed = 0;
    g->lastOverlapped = 0;
    g->lastFused = 0;
    g->lastGeneral = 0;
    g->lastGeneralNames.clear();
    const GlobalUbo* globalPtr = g->globalUbo != PLR_INVALID_INDEX ? (const GlobalUbo*)g->ubufs[g->globalUbo].dev : nullptr;
    const bool tailAllowed = g->asyncTail && !g->overlap;
    if (globalPtr && tailAllowed && g->globalCopies[g->globalCopyIndex]) globalPtr = (const GlobalUbo*)g->globalCopies[g->globalCopyIndex];
    g->lastAsync = 0;
    bool tailOpen = false, tailDirty = false; // tailOpen: the tail stream is ordered behind the main stream's work so far; tailDirty: tailDone is stale
    auto closeTail = [&]() -> int { if (tailDirty) { HIP_TRY(hipEventRecord(g->tailDone, g->tailStream)); tailDirty = false; } return PLR_OK; };
    gatherFusionGroups();
    linkConsumers(globalPtr);
    g->lastEarly = 0;
    g->lateLaunchedThisFrame = false;
    planEarlyParts(globalPtr);
    std::vector<PlanNode> plan(n);
    std::vector<hipEvent_t> done(n, nullptr);
    size_t i = 0;
    while (i < n) {
        Execution& x = g->executions[i];
        if (x.callback) {
            if (int erc = launchEarlyPartsAt(i, globalPtr, timed)) return erc;
            // everything before a callback has joined the main stream (end of the previous run)
            // a host callback may touch anything (halo exchange on raw pointers): the asynchronous tail joins first - unless the callback was
            // recorded with the resources it touches and shares none with the tail
            if (!x.callbackAccessKnown || hazardWithTail(x.access)) {
                if (int rc = closeTail()) return rc;
                if (int rc = joinAsyncTail()) return rc;
            }
            tailOpen = false;
            g->curStream = g->stream;
            if (int rc = applyPendingFillsNow()) return rc;
            if (timed) if (int rc = beginSegment(x.callbackName)) return rc;
            if (x.callbackAccessKnown) { if (int rc = refuseStaleReads(x, nullptr, 0)) return rc; }
            if (x.callbackAccessKnown) touchAccesses(x.access);
            else g->contentVersion.clear(); // may have written anything: every image gets a new version at its next query
            g->mainOps++;
            const int crc = x.callback(x.callbackUser, (void*)g->stream);
            if (crc) return setErr(crc, "host callback '" + std::string(x.callbackName) + "' failed with code " + std::to_string(crc));
            if (timed) if (int rc = endSegment()) return rc;
            i++;
            continue;
        }
        size_t last = i;
        while (last < n && !g->executions[last].callback) last++;
        if (!g->overlap) {
            while (i < last) {
                if (int erc = launchEarlyPartsAt(i, globalPtr, timed)) return erc;
                // executions flagged async_tail (plr.h) go to the tail stream: behind everything launched before them, beside everything launched
                // after them that shares no resource with them - the next frame's passes included
                const bool async = tailAllowed && g->executions[i].asyncTail;
                if (async) if (int rc = applyPendingFillsNow()) return rc; // (a frame whose first launch is a tail launch: the fills go in front of the tail's start)
                size_t runEnd = i + 1; // [i, runEnd): executions of the same kind (a fused launch never mixes the two)
                while (runEnd < last && (tailAllowed && g->executions[runEnd].asyncTail) == async) runEnd++;
                hipStream_t stream = g->stream;
                if (async) {
                    if (!tailOpen) {
                        HIP_TRY(hipEventRecord(g->tailStart, g->stream));
                        HIP_TRY(hipStreamWaitEvent(g->tailStream, g->tailStart, 0));
                        g->tailStartOps = g->mainOps; // (launchEarlyPartsAt: may the next frame's early part order itself behind this event?)
                        g->tailStartCoversLate = g->lateLaunchedThisFrame;
                        tailOpen = true;
                    }
                    stream = g->tailStream;
                }
                if (!async && !g->tailPending.empty()) {
                    // anything this launch (or the fused launch it may become) shares with the pending tail: the main stream waits for the tail first
                    const size_t window = fusionWindow(i, runEnd);
                    bool hazard = false;
                    for (size_t k = i; k < i + window && !hazard; k++) hazard = hazardWithTail(g->executions[k].access);
                    if (hazard) { if (int rc = closeTail()) return rc; if (int rc = joinAsyncTail()) return rc; }
                }
                size_t covered = 0;
                if (int rc = tryFusedLaunch(i, runEnd, stream, globalPtr, timed, &covered)) return rc;
                const size_t count = covered ? covered : 1;
                if (async) {
                    // the union of what the pending tail touches, one entry per (allocation, strongest access): a host that never joins the tail
                    // (no download, no shared resource) would otherwise grow this list by a frame's worth of entries per frame
                    auto note = [&](const Access& a) {
                        for (Access& t : g->tailPending) if (t.key == a.key) { t.write = t.write || a.write; return; }
                        g->tailPending.push_back(a);
                    };
                    for (size_t k = i; k < i + count; k++)
                        for (const Access& a : g->executions[k].access) note(a);
                    note(Access{(const void*)globalPtr, false});
                    tailDirty = true;
                    g->lastAsync += (uint32_t)count;
                } else tailOpen = false; // the main stream moves on: a later tail execution must be ordered behind this one
                if (!covered) if (int rc = launchExecution(g->executions[i], stream, globalPtr, timed)) return rc;
                for (size_t k = i + 1; k < i + count; k++) if (int erc = launchEarlyPartsAt(k, globalPtr, timed)) return erc; // positions inside a fused launch: right behind it
                i += count;
            }
            continue;
        }
        if (int rc = applyPendingFillsNow()) return rc; // side streams start behind the launch stream's work so far: the fills belong to it
        int mainAfter[kMaxStreams];
        planStreams(i, last, plan, 1 + g->activeSideStreams, mainAfter);
        static const bool debugPlan = std::getenv("PLR_STREAM_DEBUG") != nullptr;
        if (debugPlan) {
            for (size_t k = i; k < last; k++) {
                std::string w;
                for (int d : plan[k].waits) w += " " + std::to_string(d);
                fprintf(stderr, "[plr streams] %2zu %-45s stream %d%s%s%s\n", k, g->passes[g->executions[k].pass]->name.c_str(), plan[k].stream,
                        plan[k].signal ? " signals" : "", w.empty() ? "" : " waits for", w.c_str());
            }
        }
        // side streams start after everything launched on the main stream so far (earlier frames, uploads, callbacks); the event is
        // recorded before the first launch of the run, so a side stream does not wait for this run's main-stream executions
        hipEvent_t runStart = nullptr;
        for (size_t k = i; k < last && !runStart; k++)
            if (plan[k].stream != 0) {
                if (int rc = orderEvent(&runStart)) return rc;
                HIP_TRY(hipEventRecord(runStart, g->stream));
            }
        bool joined[1 + Backend::kSideStreams] = {true, false, false, false};
        int tail[1 + Backend::kSideStreams] = {-1, -1, -1, -1};
        for (size_t k = i; k < last; k++) {
            const PlanNode& node = plan[k];
            hipStream_t stream = node.stream == 0 ? g->stream : g->sideStreams[node.stream - 1];
            if (!joined[node.stream]) {
                HIP_TRY(hipStreamWaitEvent(stream, runStart, 0));
                joined[node.stream] = true;
            }
            for (int d : node.waits) HIP_TRY(hipStreamWaitEvent(stream, done[d], 0));
            if (int rc = launchExecution(g->executions[k], stream, globalPtr, timed)) return rc;
            if (node.signal) {
                if (int rc = orderEvent(&done[k])) return rc;
                HIP_TRY(hipEventRecord(done[k], stream));
            }
            tail[node.stream] = (int)k;
            if (node.stream != 0) g->lastOverlapped++;
        }
        // join: the main stream continues after the tails of the side streams
        for (int st = 1; st <= Backend::kSideStreams; st++) {
            if (tail[st] < 0 || tail[st] <= mainAfter[st]) continue; // unused, or a main-stream execution already waited for its tail
            hipEvent_t ev = done[tail[st]];
            if (!ev) {
                if (int rc = orderEvent(&ev)) return rc;
                HIP_TRY(hipEventRecord(ev, g->sideStreams[st - 1]));
            }
            HIP_TRY(hipStreamWaitEvent(g->stream, ev, 0));
        }
        i = last;
    }
    if (int rc = closeTail()) return rc;
    g->curStream = g->stream;
    g->timingNow = false;
    return PLR_OK;
}

int plr_render_frame(int /*present_to_screen*/) {
    NEED_INIT();
    const auto t0 = std::chrono::steady_clock::now();
    g->lastFillDsts.clear();
    int rc = flushFills();
    if (rc) return rc;
    rc = flushBindless();
    if (rc) return rc;
    // the frame's two timing events only when somebody asked for timings (plr_set_pass_timing): an event record is a barrier packet of ~6 us on the
    // launch stream (profiles/r04_frame_timeline.txt), and there were four of them between two frames
    // (a caller that polls the getter every frame, as a frame-time overlay does, gets every frame bracketed from its second frame on)
    const bool frameTimed = g->passTiming || g->frameTimeAsked;
    g->frameTimeAsked = false;
    if (frameTimed) HIP_TRY(hipEventRecord(g->frameStart, g->stream));
    rc = launchAll(g->passTiming);
    if (rc) return rc;
    if (int frc = applyPendingFillsNow()) return frc; // a frame without a launch
    if (frameTimed) HIP_TRY(hipEventRecord(g->frameEnd, g->stream));
    g->frameRecorded = frameTimed;
    g->timedExecutions = g->passTiming ? g->segments.size() : 0;
    if (g->passTiming) {
        g->lastTimings.clear();
        for (auto& sg : g->segments) g->lastTimings.push_back({0.f, sg.name});
    }
    g->lastCpuMs = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return PLR_OK;
}

int plr_set_pass_fusion(int enabled) {
    NEED_INIT();
    g->fusion = std::min(std::max(enabled, 0), 2);
    return PLR_OK;
}
int plr_set_pass_fusion_reorder(int enabled) {
    NEED_INIT();
    g->fusionReorder = enabled != 0;
    return PLR_OK;
}
int plr_get_pass_fusion(int* out_enabled, uint32_t* out_fused_executions) {
    NEED_INIT();
    if (out_enabled) *out_enabled = g->fusion;
    if (out_fused_executions) *out_fused_executions = g->lastFused;
    return PLR_OK;
}

int plr_get_edge_signal(void** out_signal, uint32_t* out_value) {
    NEED_INIT();
    if (out_signal) *out_signal = (void*)g->edgeSignal;
    if (out_value) *out_value = g->edgeSerial;
    return PLR_OK;
}
int plr_get_general_kernel_executions(uint32_t* out_count, char* out_names, size_t names_capacity) {
    NEED_INIT();
    if (out_count) *out_count = g->lastGeneral;
    if (out_names && names_capacity) {
        const size_t n = std::min(names_capacity - 1, g->lastGeneralNames.size());
        std::memcpy(out_names, g->lastGeneralNames.data(), n);
        out_names[n] = 0;
    }
    return PLR_OK;
}
int plr_set_async_tail(int enabled) {
    NEED_INIT_JOINED();
    if (enabled && !g->asyncTail) {
        // the rotating copies of the global uniform buffer were not refreshed while the tail was off: drop them, the kernels read the buffer itself
        // until its next fill makes a new copy
        HIP_TRY(hipStreamSynchronize(g->stream));
        for (void*& c : g->globalCopies) { if (c) hipFree(c); c = nullptr; }
    }
    g->asyncTail = enabled != 0;
    return PLR_OK;
}
int plr_get_async_tail(int* out_enabled, uint32_t* out_async_executions) {
    NEED_INIT();
    if (out_enabled) *out_enabled = g->asyncTail ? 1 : 0;
    if (out_async_executions) *out_async_executions = g->lastAsync;
    return PLR_OK;
}

int plr_set_early_parts(int enabled) {
    NEED_INIT();
    g->earlyParts = std::min(std::max(enabled, 0), 2);
    return PLR_OK;
}
int plr_get_early_parts(int* out_enabled, uint32_t* out_early_launches) {
    NEED_INIT();
    if (out_enabled) *out_enabled = g->earlyParts;
    if (out_early_launches) *out_early_launches = g->lastEarly;
    return PLR_OK;
}

int plr_set_stream_overlap(int enabled) {
    NEED_INIT_JOINED();
    g->overlap = enabled != 0;
    return PLR_OK;
}
int plr_get_stream_overlap(int* out_enabled, uint32_t* out_overlapped_executions) {
    NEED_INIT();
    if (out_enabled) *out_enabled = g->overlap ? 1 : 0;
    if (out_overlapped_executions) *out_overlapped_executions = g->lastOverlapped;
    return PLR_OK;
}

int plr_replay_frame(uint32_t count, float* out_total_gpu_ms) {
    NEED_INIT_JOINED();
    g->lastFillDsts.clear();
    int rc = flushFills();
    if (rc) return rc;
    rc = flushBindless();
    if (rc) return rc;
    HIP_TRY(hipEventRecord(g->frameStart, g->stream));
    if (int frc = applyPendingFillsNow()) return frc; // (a replayed frame re-launches its first launch: the table is applied once, in front)
    for (uint32_t i = 0; i < count; i++) {
        rc = launchAll(false);
        if (rc) return rc;
    }
    HIP_TRY(hipEventRecord(g->frameEnd, g->stream));
    HIP_TRY(hipEventSynchronize(g->frameEnd));
    g->frameRecorded = true;
    g->timedExecutions = 0;
    if (out_total_gpu_ms) HIP_TRY(hipEventElapsedTime(out_total_gpu_ms, g->frameStart, g->frameEnd));
    return PLR_OK;
}

int plr_get_image_global_texture_array_index(plr_image_handle image, uint32_t* out_index) {
    NEED_INIT();
    if (image.type != PLR_IMAGE_DEFAULT || !resolveImage(image)) return setErr(PLR_ERR_INVALID_ARGUMENT, "invalid image handle");
    *out_index = image.index;
    return PLR_OK;
}

int plr_create_image(const plr_image_desc* desc, const void* initial_data, size_t initial_data_size, plr_image_handle* out_image) {
    NEED_INIT();
    if (!desc || !out_image) return setErr(PLR_ERR_INVALID_ARGUMENT, "null argument");
    ImageRes im;
    int rc = allocImage(im, *desc);
    if (rc) return rc;
    if (initial_data && initial_data_size) {
        // initial data covers mip 0 (or the whole tightly packed chain for FullChainAlreadyInData)
        size_t consumed = 0;
        for (size_t m = 0; m < im.mips.size() && consumed < initial_data_size; m++) {
            const size_t nBytes = std::min(im.mips[m].bytes, initial_data_size - consumed);
            HIP_TRY(hipMemcpyAsync((uint8_t*)im.dev + im.mips[m].offset, (const uint8_t*)initial_data + consumed, nBytes, hipMemcpyHostToDevice, g->stream));
            consumed += nBytes;
            if (desc->mip_count != PLR_MIP_FULL_CHAIN_ALREADY_IN_DATA) break;
        }
        HIP_TRY(hipStreamSynchronize(g->stream));
    }
    g->images.push_back(std::move(im));
    g->bindlessDirty = true;
    out_image->type = PLR_IMAGE_DEFAULT;
    out_image->index = (uint32_t)(g->images.size() - 1);
    return PLR_OK;
}

static int createBuffer(std::vector<BufferRes>& list, size_t size, const void* init, uint32_t* out) {
    if (size == 0 || !out) return setErr(PLR_ERR_INVALID_ARGUMENT, "buffer size is 0 or out handle is null");
    BufferRes b;
    b.size = size;
    HIP_TRY(hipMalloc(&b.dev, (size + 15) & ~(size_t)15));
    HIP_TRY(hipMemsetAsync(b.dev, 0, (size + 15) & ~(size_t)15, g->stream));
    if (init) HIP_TRY(hipMemcpyAsync(b.dev, init, size, hipMemcpyHostToDevice, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    g->allocated += size;
    list.push_back(b);
    *out = (uint32_t)(list.size() - 1);
    return PLR_OK;
}

int plr_create_uniform_buffer(size_t size, const void* initial_data, plr_uniform_buffer_handle* out_buffer) {
    NEED_INIT();
    return createBuffer(g->ubufs, size, initial_data, out_buffer);
}
int plr_create_storage_buffer(size_t size, const void* initial_data, plr_storage_buffer_handle* out_buffer) {
    NEED_INIT();
    return createBuffer(g->sbufs, size, initial_data, out_buffer);
}

int plr_create_sampler(const plr_sampler_desc* desc, plr_sampler_handle* out_sampler) {
    NEED_INIT();
    if (!desc || !out_sampler) return setErr(PLR_ERR_INVALID_ARGUMENT, "null argument");
    g->samplers.push_back(*desc);
    *out_sampler = (uint32_t)(g->samplers.size() - 1);
    return PLR_OK;
}

int plr_create_temporary_image(const plr_image_desc* desc, plr_image_handle* out_image) {
    NEED_INIT();
    if (!desc || !out_image) return setErr(PLR_ERR_INVALID_ARGUMENT, "null argument");
    // pooled by description; the reference aliases by first/last use (RenderBackend.cpp:1026-1123), with 288 GB of
    // HBM the pool simply keeps one allocation per concurrently live description
    for (size_t i = 0; i < g->transient.size(); i++) {
        if (!g->transient[i].inUse && g->transient[i].dev && sameDesc(g->transient[i].desc, *desc)) {
            g->transient[i].inUse = true;
            out_image->type = PLR_IMAGE_TRANSIENT; out_image->index = (uint32_t)i;
            return PLR_OK;
        }
    }
    ImageRes im;
    int rc = allocImage(im, *desc);
    if (rc) return rc;
    im.inUse = true;
    g->transient.push_back(std::move(im));
    out_image->type = PLR_IMAGE_TRANSIENT; out_image->index = (uint32_t)(g->transient.size() - 1);
    return PLR_OK;
}

int plr_get_swapchain_input_image(plr_image_handle* out_image) {
    NEED_INIT();
    out_image->type = PLR_IMAGE_SWAPCHAIN; out_image->index = 0;
    return PLR_OK;
}

int plr_get_memory_stats(uint64_t* out_allocated_size, uint64_t* out_used_size) {
    NEED_INIT();
    if (out_allocated_size) *out_allocated_size = g->allocated;
    if (out_used_size) *out_used_size = g->allocated;
    return PLR_OK;
}

int plr_debug_set_decision_signature(size_t words) {
    NEED_INIT_JOINED();
    HIP_TRY(hipStreamSynchronize(g->stream));
    if (g->debugSig) { hipFree(g->debugSig); g->debugSig = nullptr; g->debugSigWords = 0; }
    if (words == 0) return PLR_OK;
    HIP_TRY(hipMalloc((void**)&g->debugSig, words * 4));
    HIP_TRY(hipMemset(g->debugSig, 0xff, words * 4));
    g->debugSigWords = words;
    return PLR_OK;
}
int plr_debug_read_decision_signature(uint32_t* out_words, size_t words) {
    NEED_INIT_JOINED();
    if (!g->debugSig || !out_words || words > g->debugSigWords) return setErr(PLR_ERR_INVALID_ARGUMENT, "plr_debug_read_decision_signature: no buffer of that size is set");
    HIP_TRY(hipStreamSynchronize(g->stream));
    HIP_TRY(hipMemcpy(out_words, g->debugSig, words * 4, hipMemcpyDeviceToHost));
    return PLR_OK;
}

int plr_debug_sky_lut_eval(plr_image_handle sky_lut, const float* directions, float* out_rgb, int64_t n) {
    NEED_INIT_JOINED();
    ImageRes* im = resolveImage(sky_lut);
    if (!im || !directions || !out_rgb || n <= 0) return setErr(PLR_ERR_INVALID_ARGUMENT, "plr_debug_sky_lut_eval: invalid argument");
    HIP_TRY(hipStreamSynchronize(g->stream));
    return launchSkyLutProbe(makeView(*im, 0), directions, out_rgb, n);
}

int plr_debug_pcf_tap_table(float* out_xy, size_t floats) {
    NEED_INIT_JOINED();
    if (!out_xy || floats != kPcfTapTableBytes / sizeof(float)) return setErr(PLR_ERR_INVALID_ARGUMENT, "plr_debug_pcf_tap_table: out_xy must hold 256 x 12 x 2 floats");
    void* table = nullptr;
    HIP_TRY(hipMalloc(&table, kPcfTapTableBytes));
    hipError_t e = buildPcfTapTable((float2*)table, g->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
    if (e == hipSuccess) e = hipMemcpy(out_xy, table, kPcfTapTableBytes, hipMemcpyDeviceToHost);
    hipFree(table);
    HIP_TRY(e);
    return PLR_OK;
}

int plr_debug_sampler_eval(plr_image_handle image, uint32_t mip_level, int filter, int address, const float* coords, float* out, int64_t n) {
    NEED_INIT_JOINED();
    ImageRes* im = resolveImage(image);
    if (!im || mip_level >= im->mips.size()) return setErr(PLR_ERR_INVALID_ARGUMENT, "plr_debug_sampler_eval: invalid image handle or mip level");
    if (!coords || !out || n <= 0) return setErr(PLR_ERR_INVALID_ARGUMENT, "plr_debug_sampler_eval: null argument");
    HIP_TRY(hipStreamSynchronize(g->stream));
    return launchSamplerProbe(makeView(*im, mip_level), filter, address, coords, out, n);
}

int plr_set_math_mode(int mode) {
    NEED_INIT_JOINED();
    if (mode != PLR_MATH_EXACT && mode != PLR_MATH_FAST) return setErr(PLR_ERR_INVALID_ARGUMENT, "math mode must be PLR_MATH_EXACT or PLR_MATH_FAST");
    if (mode == PLR_MATH_EXACT) if (int rc = ensureExactSet()) return rc; // (loudly, here - not at the first launch)
    g->mathMode = mode;
    return PLR_OK;
}
int plr_get_math_mode(int* out_mode) { NEED_INIT(); *out_mode = g->mathMode; return PLR_OK; }

int plr_set_pass_timing(int enabled) { NEED_INIT(); g->passTiming = enabled != 0; return PLR_OK; }

int plr_get_renderpass_timings(plr_renderpass_time* out_times, uint32_t* inout_count) {
    NEED_INIT();
    if (!inout_count) return setErr(PLR_ERR_INVALID_ARGUMENT, "inout_count is null");
    const uint32_t n = (uint32_t)g->timedExecutions;
    if (!out_times) { *inout_count = n; return PLR_OK; }
    if (n) HIP_TRY(hipEventSynchronize(g->frameEnd));
    if (n && g->tailStream) HIP_TRY(hipStreamSynchronize(g->tailStream)); // segments of the asynchronous tail end on its own stream
    const uint32_t m = std::min(n, *inout_count);
    for (uint32_t i = 0; i < m; i++) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, g->passEvents[g->segments[i].firstEvent], g->passEvents[g->segments[i].firstEvent + 1]));
        out_times[i].time_ms = ms;
        out_times[i].name = g->lastTimings[i].name;
    }
    *inout_count = m;
    return PLR_OK;
}

int plr_get_last_frame_cpu_time(float* out_ms) { NEED_INIT(); *out_ms = g->lastCpuMs; return PLR_OK; }

int plr_get_last_frame_gpu_time(float* out_ms) {
    NEED_INIT_JOINED();
    // Asking is what turns the bracket on: the next frame is timed, and the value handed out is that of the most recent bracketed frame (0 before there
    // is one) - like the reference's getRenderpassTimings (RenderBackend.h:107), whose timestamp queries are read back a frame late.
    g->frameTimeAsked = true;
    if (g->frameRecorded) {
        HIP_TRY(hipEventSynchronize(g->frameEnd));
        HIP_TRY(hipEventElapsedTime(&g->lastFrameGpuMs, g->frameStart, g->frameEnd));
    }
    *out_ms = g->lastFrameGpuMs;
    return PLR_OK;
}

int plr_get_image_description(plr_image_handle image, plr_image_desc* out_desc) {
    NEED_INIT();
    ImageRes* im = resolveImage(image);
    if (!im) return setErr(PLR_ERR_INVALID_ARGUMENT, "invalid image handle");
    *out_desc = im->desc;
    return PLR_OK;
}

static int imageMip(plr_image_handle image, uint32_t mip, ImageRes** im, MipInfo** mi) {
    *im = resolveImage(image);
    if (!*im) return setErr(PLR_ERR_INVALID_ARGUMENT, "invalid image handle");
    if (mip >= (*im)->mips.size()) return setErr(PLR_ERR_INVALID_ARGUMENT, "mip level out of range");
    *mi = &(*im)->mips[mip];
    return PLR_OK;
}

int plr_upload_image(plr_image_handle image, uint32_t mip_level, const void* data, size_t size) {
    NEED_INIT_JOINED();
    ImageRes* im; MipInfo* mi;
    int rc = imageMip(image, mip_level, &im, &mi);
    if (rc) return rc;
    if (size != mi->bytes) return setErr(PLR_ERR_INVALID_ARGUMENT, "upload size " + std::to_string(size) + " != mip size " + std::to_string(mi->bytes));
    HIP_TRY(hipMemcpyAsync((uint8_t*)im->dev + mi->offset, data, size, hipMemcpyHostToDevice, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    hostWroteImage(*im);
    return PLR_OK;
}

int plr_upload_image_rows(plr_image_handle image, uint32_t mip_level, uint32_t row_begin, uint32_t row_count, const void* data, size_t size) {
    NEED_INIT_JOINED();
    ImageRes* im; MipInfo* mi;
    int rc = imageMip(image, mip_level, &im, &mi);
    if (rc) return rc;
    if (mi->d != 1) return setErr(PLR_ERR_INVALID_ARGUMENT, "row upload needs a 2D image");
    if ((uint64_t)row_begin + row_count > mi->h) return setErr(PLR_ERR_INVALID_ARGUMENT, "rows outside the image");
    const size_t rowBytes = mi->bytes / mi->h;
    if (size != rowBytes * row_count) return setErr(PLR_ERR_INVALID_ARGUMENT, "upload size " + std::to_string(size) + " != rows * " + std::to_string(rowBytes));
    if (size == 0) return PLR_OK;
    HIP_TRY(hipMemcpyAsync((uint8_t*)im->dev + mi->offset + rowBytes * row_begin, data, size, hipMemcpyHostToDevice, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    hostWroteImage(*im);
    return PLR_OK;
}

int plr_download_image(plr_image_handle image, uint32_t mip_level, void* out_data, size_t size) {
    NEED_INIT_JOINED();
    ImageRes* im; MipInfo* mi;
    int rc = imageMip(image, mip_level, &im, &mi);
    if (rc) return rc;
    if (size != mi->bytes) return setErr(PLR_ERR_INVALID_ARGUMENT, "download size " + std::to_string(size) + " != mip size " + std::to_string(mi->bytes));
    if (im->elided)
        return setErr(PLR_ERR_UNSUPPORTED, "this image was not written in the last frame: the fused launch that consumes it kept it in registers (pass fusion level 2); "
                                           "plr_set_pass_fusion(1) keeps intermediates");
    HIP_TRY(hipMemcpyAsync(out_data, (uint8_t*)im->dev + mi->offset, size, hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    return PLR_OK;
}

static int downloadBuffer(std::vector<BufferRes>& list, uint32_t h, void* out, size_t offset, size_t size) {
    if (h >= list.size()) return setErr(PLR_ERR_INVALID_ARGUMENT, "invalid buffer handle");
    if (offset + size > list[h].size) return setErr(PLR_ERR_INVALID_ARGUMENT, "buffer download range out of bounds");
    HIP_TRY(hipMemcpyAsync(out, (uint8_t*)list[h].dev + offset, size, hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    return PLR_OK;
}
int plr_download_storage_buffer(plr_storage_buffer_handle buffer, void* out_data, size_t offset, size_t size) {
    NEED_INIT_JOINED();
    return downloadBuffer(g->sbufs, buffer, out_data, offset, size);
}
int plr_download_uniform_buffer(plr_uniform_buffer_handle buffer, void* out_data, size_t offset, size_t size) {
    NEED_INIT_JOINED();
    return downloadBuffer(g->ubufs, buffer, out_data, offset, size);
}

int plr_get_image_device_pointer(plr_image_handle image, uint32_t mip_level, void** out_ptr, size_t* out_size) {
    // No ordering is implied by asking for an address (the C++ host does it at record time, every frame, for its exchange items: a join here would
    // serialise the asynchronous tail of every band frame - measured, +60 us per band). A caller orders its ACCESSES: on plr_get_stream(), which joins.
    NEED_INIT();
    ImageRes* im; MipInfo* mi;
    int rc = imageMip(image, mip_level, &im, &mi);
    if (rc) return rc;
    if (im->elided)
        return setErr(PLR_ERR_UNSUPPORTED, "this image was not written in the last frame: the fused launch that consumes it kept it in registers (pass fusion level 2); "
                                           "plr_set_pass_fusion(1) keeps intermediates");
    *out_ptr = (uint8_t*)im->dev + mi->offset;
    if (out_size) *out_size = mi->bytes;
    g->externallyWritable.insert(im->dev); // the caller may write through the pointer at any time: nothing derived from this image is cached any more
    return PLR_OK;
}

int plr_get_storage_buffer_device_pointer(plr_storage_buffer_handle buffer, void** out_ptr, size_t* out_size) {
    NEED_INIT();
    if (buffer >= g->sbufs.size()) return setErr(PLR_ERR_INVALID_ARGUMENT, "invalid buffer handle");
    *out_ptr = g->sbufs[buffer].dev;
    if (out_size) *out_size = g->sbufs[buffer].size;
    return PLR_OK;
}

int plr_get_stream(void** out_hip_stream) { NEED_INIT_JOINED(); *out_hip_stream = (void*)g->stream; return PLR_OK; }
int plr_get_launch_stream(void** out_hip_stream) { NEED_INIT(); *out_hip_stream = (void*)g->stream; return PLR_OK; }

int plr_get_supported_shaders(const char** out_names, uint32_t capacity) {
    std::lock_guard<std::mutex> lock(registryMutex());
    const auto& r = registry();
    for (uint32_t i = 0; i < capacity && i < r.size(); i++) out_names[i] = r[i].name.c_str();
    return (int)r.size();
}

} // extern "C"
