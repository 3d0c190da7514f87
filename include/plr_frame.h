/* plr_frame.h - C entry points of the C++ host-side frame pipeline (plainrenderer_amd/csrc/frontend/frame_pipeline.*).
 *
 * The pipeline is the compute part of the reference's RenderFrontend::prepareRenderpasses (RenderFrontend.cpp:313-406) and its
 * technique classes, recorded through the RenderBackend boundary of plr.h. These functions exist so a benchmark / test harness
 * in another language can drive whole frames; the reference itself would call the C++ classes directly.
 * plr_setup() must have been called first. All functions return PLR_OK or a negative code (message: plrf_last_error()).
 */
#ifndef PLR_FRAME_H
#define PLR_FRAME_H
#include "plr.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct plrf_settings {
    uint32_t width, height;
    uint32_t shadow_map_res;      /* 2048 in the reference (RenderFrontend.cpp:40) */
    uint32_t brdf_lut_res;        /* 512 (RenderFrontend.cpp:45) */
    uint32_t max_sdf_instances;   /* 1200 (SceneConfig.h:3) */
    uint32_t froxel_depth;        /* 64 */
    /* TAASettings, Techniques/TAA.h:8-17 */
    uint32_t taa_enabled, taa_use_clipping, taa_use_motion_vector_dilation, taa_history_sampling_tech, taa_filter_use_tonemapping;
    /* BloomSettings, Techniques/Bloom.h:5-9 */
    uint32_t bloom_enabled; float bloom_strength, bloom_radius;
    /* SDFTraceSettings, Techniques/SDFGI.h:17-29 */
    uint32_t sdf_half_res_trace, sdf_strict_influence_radius_cutoff; float sdf_trace_influence_radius;
    /* ShadingConfig, RenderFrontend.h:32-38 */
    uint32_t diffuse_brdf, direct_multiscatter, indirect_lighting_tech, use_geometry_aa, sun_shadow_cascade_count;
    /* which pass groups of the frame are recorded (all 1 = full frame) */
    uint32_t run_exposure, run_hiz, run_gi, run_shading, run_taa, run_bloom, run_tonemap;
    /* band rendering (multi-GPU partition by screen rows; no reference counterpart). width/height stay the whole frame's; this
     * instance renders full-resolution rows [band_row_begin, band_row_end) (multiples of 64, or the last row). band_row_end == 0:
     * off. Halos: rows exchanged / recomputed around the band, see BandSettings in csrc/frontend/frame_pipeline.h */
    uint32_t band_row_begin, band_row_end, band_gi_halo, band_gi_history_halo, band_color_halo, band_post_halo;
    /* band_gi_halo decides what a partitioned frame IS. The GI denoiser's disc is 1.5 m in world space - on near geometry wider than any bounded halo - and a
     * sample beyond the exchanged halo gets weight 0. The denoised signal is next frame's history (SDFGI.cpp:421-536), so that deviation feeds back and spreads:
     * with the default (64 trace rows per 2160 frame rows) the share of pixels more than one R11G11B10 code from the unpartitioned frame grows for hundreds of
     * frames and has no steady state within 256 (profiles/r05_config5_series.txt). band_gi_halo = PLRF_HALO_WHOLE_IMAGE exchanges every GI texel with every
     * rank: the partitioned frame then EQUALS the unpartitioned one, bit for bit (tests/test_config5_8k.py, 64 frames at 8K), at the price in the same file. */
#define PLRF_HALO_WHOLE_IMAGE 0xffffffffu
    /* band_gi_halo = PLRF_HALO_REQUESTED (round 6; half-resolution trace only): REQUEST LISTS instead of a halo in front of the two spatial filter passes. Where a disc
     * sample lands depends on depth, camera and frame index only (filterIndirectDiffuseSpatial.comp:53-105), so right after the depth downscale a giSampleRequests pass per
     * filter marks, in a bitmap over the trace image, every texel outside this rectangle a sample of this rectangle's pixels lands on; the exchange callback
     * PLRF_EXCHANGE_GI_REQUESTS trades the bitmaps (to each peer the part over its rectangle) while the trace runs, and at PLRF_EXCHANGE_GI_TRACE /
     * PLRF_EXCHANGE_GI_TEMPORAL every owner sends exactly the requested texels - Y_SH, CoCg and the half-resolution depth, 16 bytes each - which the receiver scatters
     * into its images. Nothing is masked, nothing is approximated: the partitioned frame EQUALS the unpartitioned one bit for bit (tests/test_config5_8k.py), like
     * PLRF_HALO_WHOLE_IMAGE, for 6 - 27 MB received per rank and frame at 8K instead of 200 MB (profiles/r06_gi_request_count.txt). plrf_get_gi_request_exchange
     * describes the buffers to an exchange callback; the native exchange (plrf_rccl_attach_rects / plrf_local_attach_rects) implements it. */
#define PLRF_HALO_REQUESTED 0xfffffffeu
    /* input producers recorded as compute passes instead of uploaded (0 = uploaded): lightMatrix.comp after the depth pyramid */
    uint32_t run_light_matrix; float volumetrics_max_distance;
    uint32_t taa_use_separate_supersampling, taa_supersample_use_tonemapping; /* TAASettings::useSeparateSupersampling (off), supersampleUseTonemapping */
    uint32_t sdf_debug_mode, sdf_debug_tile_usage_with_hiz, sdf_debug_use_influence_radius; /* SDFDebugSettings (SDFGI.h:9-15): mode != 0 replaces the frame by the debug view */
    uint32_t band_taa_history_halo; /* rows of the TAA history exchanged between bands (default 32) */
    uint32_t run_volumetrics; /* the four froxel passes produce volumetricIntegrationVolume (reference default VolumetricsSettings; noise volume "perlinNoise3D" is an input) */
    uint32_t run_sky_luts; /* sky transmission / multiscatter / sky LUT compute passes with the reference's default AtmosphereSettings */
    uint32_t band_overlap_exchange; /* 0: one call per exchange. 1: producers of exchanged images run their edge rows first and the exchange callback is called
                                       twice, with PLRF_EXCHANGE_BEGIN (start, do not wait) and PLRF_EXCHANGE_END (wait).
                                       2 (the default): as 1, but the producer is ONE launch that writes the edge rows first and raises plr_get_edge_signal when they
                                       are complete (plr.h first_rows) - the BEGIN callback may wait for that signal on its own stream instead of ordering
                                       behind the launch stream; 1: an edge launch and an interior launch with the BEGIN callback between them (round 3) */
    /* tile rendering (round 5; BASELINE config 5: the frame as 2 x 2 screen tiles over 4 GPUs): this instance renders columns [band_col_begin, band_col_end) of
     * its rows (multiples of 64, or the last column). band_col_end == 0: whole rows (a band). The halos are the same numbers of columns; exchange items then
     * carry a column range and the transfers are rectangles (plrf_exchange_plan_rects). With band_overlap_exchange 1 a tile behaves as with 0 (its producers
     * are never split); 2 produces the FRAME of the tile first (plr.h first_rows + first_cols). */
    uint32_t band_col_begin, band_col_end;
} plrf_settings;

/* ---- band rendering: halo exchange hooks ----
 * While plrf_frame launches the passes it calls the exchange callback, in pass order, at the points where rows produced by
 * neighbouring bands are needed. The callback moves the rows (RCCL send/recv on the stream it is handed, or any other transport)
 * of every item plrf_get_exchange_items reports for that exchange id; PLRF_EXCHANGE_HISTOGRAM instead sums the 128-bin luminance
 * histogram over all bands in place (plrf_get_histogram_exchange). */
enum plrf_exchange_id { PLRF_EXCHANGE_HISTOGRAM = 0, PLRF_EXCHANGE_GI_TRACE = 1, PLRF_EXCHANGE_GI_TEMPORAL = 2, PLRF_EXCHANGE_GI_HISTORY = 3,
                        PLRF_EXCHANGE_POST = 4,
                        /* only with run_light_matrix: the depth range of the whole frame for the cascade fit (lightMatrix.comp:76-78 reads the apex of the
                         * depth pyramid; a band has none). plrf_get_depth_apex_exchange names two floats {min, max} in device memory holding the band's
                         * range: replace them IN PLACE by the minimum of all bands' first and the maximum of all bands' second float (exact: the result
                         * equals the unpartitioned pyramid's apex bit for bit) */
                        PLRF_EXCHANGE_DEPTH_APEX = 5, PLRF_EXCHANGE_GI_REQUESTS = 6, PLRF_EXCHANGE_COUNT = 7 };
/* phase bits or-ed into exchange_id when band_overlap_exchange is on (ids 1..4; the histogram is always one call): after the producer's
 * edge rows are launched the callback gets id | PLRF_EXCHANGE_BEGIN and must only START the transfers (stream-ordered after what is already
 * on hip_stream); the producer's interior rows are launched next and run beside the transfers; before the first consumer of the halo rows
 * the callback gets id | PLRF_EXCHANGE_END and must make hip_stream wait for their completion. No bits: start and wait in one call. */
enum plrf_exchange_phase { PLRF_EXCHANGE_BEGIN = 0x100, PLRF_EXCHANGE_END = 0x200, PLRF_EXCHANGE_ID_MASK = 0xff };
typedef int (*plrf_exchange_callback)(void* user, int exchange_id, void* hip_stream);
/* an image of image_rows rows of row_bytes bytes at device_ptr; this band owns rows [row_begin, row_end): it sends its first
 * halo_rows owned rows to the band above and its last halo_rows to the band below, and receives rows
 * [row_begin - halo_rows, row_begin) from above and [row_end, row_end + halo_rows) from below (clipped to the image) */
typedef struct plrf_exchange_item {
    plr_image_handle image;
    void* device_ptr;
    uint32_t row_begin, row_end, halo_rows, row_bytes, image_rows;
    /* tile rendering: the rectangle this instance owns is columns [col_begin, col_end) of those rows, of an image of image_cols texels of texel_bytes bytes per row
     * (a band: 0 .. image_cols); the halo is halo_rows texels on every side that has a neighbour, corners included */
    uint32_t col_begin, col_end, image_cols, texel_bytes;
} plrf_exchange_item;
int plrf_set_exchange_callback(void* pipeline, plrf_exchange_callback callback, void* user);
/* items of the frame being launched (valid inside the callback and until the next plrf_frame); *inout_count = capacity in, count out */
int plrf_get_exchange_items(void* pipeline, int exchange_id, plrf_exchange_item* out_items, uint32_t* inout_count);
int plrf_get_histogram_exchange(void* pipeline, void** out_device_ptr, size_t* out_bytes);
/* the request-list exchange (band_gi_halo = PLRF_HALO_REQUESTED): the trace image's size, this rank's rectangle in trace texels, and per spatial filter pass (0: its
 * input is the traced GI, exchanged at PLRF_EXCHANGE_GI_TRACE; 1: the temporally filtered GI, PLRF_EXCHANGE_GI_TEMPORAL) the request bitmap (row_words 32-bit words
 * per texel row; bit x % 32 of word x / 32) and the input images; depth = the filters' R16F depth texture. enabled = 0: the pipeline does not use request lists. */
typedef struct plrf_gi_request {
    int32_t enabled;
    uint32_t image_cols, image_rows, row_words;
    uint32_t x0, y0, x1, y1;
    void* bitmap[2];
    void* ysh[2];
    void* cocg[2];
    void* depth;
} plrf_gi_request;
int plrf_get_gi_request_exchange(void* pipeline, plrf_gi_request* out);
/* non-zero: this pipeline records rows-first producers (band_overlap_exchange 2): a BEGIN callback may wait for plr_get_edge_signal */
int plrf_band_rows_first(void* pipeline);
int plrf_get_depth_apex_exchange(void* pipeline, void** out_device_ptr, size_t* out_bytes); /* 8 bytes: float min, float max */

/* ---- the same exchange, natively over RCCL (csrc/frontend/band_exchange.cpp): one process per GPU, one band per process ----
 * plrf_rccl_attach creates this rank's communicator (ncclCommInitRank with the id rank 0 obtained from plrf_rccl_get_unique_id and handed to
 * every rank by the launcher, once) and installs a C++ exchange callback on the pipeline: per exchange one group of ncclSend / ncclRecv with
 * the band above and the band below on a communication stream ordered against the launch stream by events (BEGIN / END phases), and one
 * 512-byte ncclAllReduce for the luminance histogram. frame_height = rows of the whole frame (bands = plrf_band_rows of it). */
#define PLRF_RCCL_UNIQUE_ID_BYTES 128
int plrf_rccl_get_unique_id(void* out_128_bytes);
int plrf_rccl_attach(void* pipeline, const void* unique_id_128_bytes, int rank, int world, uint32_t frame_height, void** out_exchange);
/* the same for a partition chosen by the caller (load balancing: bands of unequal height): row_bounds = world + 1 row boundaries, 0 ... frame_height,
 * interior ones multiples of 64; band r owns rows [row_bounds[r], row_bounds[r + 1]) and must have been created with exactly those rows. NULL = plrf_band_rows */
int plrf_rccl_attach_rows(void* pipeline, const void* unique_id_128_bytes, int rank, int world, uint32_t frame_height, const uint32_t* row_bounds, void** out_exchange);
int plrf_rccl_detach(void* pipeline, void* exchange);
/* bytes this rank sent / received and the number of point-to-point exchange groups of the last frame */
int plrf_rccl_get_stats(void* exchange, uint64_t* out_bytes_sent, uint64_t* out_bytes_received, uint64_t* out_exchanges);
/* single-GPU check of the transport: rows [src_row, src_row + rows) of an image are sent to and received from this rank itself onto
 * rows [dst_row, ...) through the same group / stream / event sequence as an overlapped exchange */
int plrf_rccl_self_test(void* exchange, void* device_ptr, uint32_t row_bytes, uint32_t src_row, uint32_t dst_row, uint32_t rows, void* launch_stream);
const char* plrf_rccl_last_error(void);
/* the partition and the per-item transfer plan the exchange uses (pure functions, no GPU): band `band` of `n_bands` owns full-resolution rows
 * [begin, end), multiples of 64; for one exchange item (an image of image_rows rows showing the frame at frame_height / image_rows scale, of
 * which this band owns [row_begin, row_end)) the plan lists up to 4 transfers: send / receive with the band above and below */
typedef struct plrf_exchange_op { uint32_t peer, send, row_begin, row_end; } plrf_exchange_op;
int plrf_band_rows(uint32_t frame_height, uint32_t n_bands, uint32_t band, uint32_t* out_row_begin, uint32_t* out_row_end);
int plrf_exchange_plan_rows(uint32_t frame_height, uint32_t n_bands, const uint32_t* row_bounds, uint32_t band, uint32_t image_rows, uint32_t halo_rows, uint32_t row_begin,
                            uint32_t row_end, plrf_exchange_op* out_ops, uint32_t* out_count);
/* what the native exchange moves for a band (up to 2 (n_bands - 1) transfers, out_count = how many there are): the plan above, plus - when a halo is taller
 * than a neighbouring band (band_gi_halo = PLRF_HALO_WHOLE_IMAGE) - the rows of the bands behind the neighbours */
int plrf_exchange_plan_all_bands(uint32_t frame_height, uint32_t n_bands, const uint32_t* row_bounds, uint32_t band, uint32_t image_rows, uint32_t halo_rows, uint32_t row_begin,
                                 uint32_t row_end, plrf_exchange_op* out_ops, uint32_t capacity, uint32_t* out_count);
int plrf_exchange_plan(uint32_t frame_height, uint32_t n_bands, uint32_t band, uint32_t image_rows, uint32_t halo_rows, uint32_t row_begin, uint32_t row_end,
                       plrf_exchange_op* out_ops_4, uint32_t* out_count);

/* ---- tile rendering (round 5): the partition is a list of pixel rectangles, one per rank (rects = world x {x0, y0, x1, y1} at full resolution; edges between
 * rectangles on multiples of 64). A band partition is the special case of full-width rectangles, and for it the plan below equals plrf_exchange_plan_rows.
 * For one exchange item (an image of image_cols x image_rows texels showing the frame at frame_width / image_cols scale) rank `rank` exchanges with every
 * rank whose rectangle TOUCHES its own (shares an edge or a corner - on a 2 x 2 grid all three others; xGMI connects every pair of GPUs of a node directly, so the
 * diagonal neighbour is one more peer of the same group, not a second hop): it sends the part of its own rectangle that lies within halo texels of the peer's
 * and receives the part of the peer's that lies within halo texels of its own (corners included). Ops are listed by ascending peer, send before receive;
 * capacity = 2 x (world - 1) covers every case. */
typedef struct plrf_rect_op { uint32_t peer, send, x0, y0, x1, y1; } plrf_rect_op;
int plrf_exchange_plan_rects(uint32_t frame_width, uint32_t frame_height, uint32_t world, const uint32_t* rects, uint32_t rank, uint32_t image_cols, uint32_t image_rows,
                             uint32_t halo, plrf_rect_op* out_ops, uint32_t capacity, uint32_t* out_count);
/* gx x gy grid of tiles over the frame, row-major (rank = ty * gx + tx), every edge a multiple of 64; col_bounds (gx + 1) / row_bounds (gy + 1) choose the cuts
 * (load balancing), NULL = equal parts. out_rects: gx * gy * 4 values */
int plrf_tile_rects(uint32_t frame_width, uint32_t frame_height, uint32_t gx, uint32_t gy, const uint32_t* col_bounds, const uint32_t* row_bounds, uint32_t* out_rects);
/* the native exchange for a partition into rectangles: as plrf_rccl_attach_rows, but rank r owns rects[4 r .. 4 r + 3] and must have been created with exactly that
 * rectangle (band_row_* / band_col_*). Rectangles that are not whole rows are moved through staging buffers: ONE pack kernel gathers every region of every image
 * of the exchange point for all peers, one group of ncclSend / ncclRecv (one pair per peer) moves the buffers, ONE unpack kernel scatters what arrived.
 * unique_id_128_bytes == NULL: LOOPBACK - no communicator; pack, a device copy standing in for the links, unpack, on the same streams and events (what a
 * single-GPU replay of a partition's frame costs, tools/band_cost.py; the received texels are meaningless). */
int plrf_rccl_attach_rects(void* pipeline, const void* unique_id_128_bytes, int rank, int world, uint32_t frame_width, uint32_t frame_height, const uint32_t* rects,
                           void** out_exchange);
/* ---- the IN-PROCESS transport (round 6): every rank of the partition in ONE process on one GPU - one host thread, backend, pipeline and exchange each, as the
 * partition tests run them. The exchange is the communicator path's code (plans, pack / unpack kernels, communication stream, BEGIN behind the producers' edge
 * signal, END, watchdog); where ncclSend / ncclRecv go, a rank publishes its transfers with an event behind its pack, waits on the host until its peers have posted
 * the same exchange, copies their send ranges into its receive ranges on its own stream (the k-th send of a to b pairs with b's k-th receive from a, RCCL's rule
 * inside a group) and orders its stream behind the peers' reads of what it sent. The all-reduces go through slots of group memory. One group per partition; it
 * must outlive its exchanges (plrf_rccl_detach). plrf_local_group_abort: wake ranks waiting for a peer that failed. plrf_local_group_freeze(1): a rank no longer
 * waits for its peers and copies from what they posted LAST (their buffers keep their last frame's texels): tools/band_cost.py times ONE partition alone with real
 * neighbour data in its halos. */
int plrf_local_group_create(int world, void** out_group);
int plrf_local_group_destroy(void* group);
int plrf_local_group_abort(void* group);
int plrf_local_group_freeze(void* group, int frozen);
int plrf_local_attach_rects(void* pipeline, void* group, int rank, int world, uint32_t frame_width, uint32_t frame_height, const uint32_t* rects, void** out_exchange);
/* what the exchange runs on: ranks of the communicator (ncclCommCount; 0 in loopback), the RCCL version code (ncclGetVersion), the ordering mode of the overlapped
 * exchanges (2 = the producer's edge signal + hipStreamWaitValue32, 1 = an event behind the producer - also what mode 2 falls back to on a device without
 * stream memory operations, 0 = not overlapped), whether the regions go through pack / unpack kernels (a tile partition) or straight from the images (bands) */
typedef struct plrf_rccl_info { int32_t rccl_ranks, rccl_version, overlap_mode, packed_regions, stream_wait_value_supported, watchdog_ms; } plrf_rccl_info;
int plrf_rccl_get_info(void* exchange, plrf_rccl_info* out);
/* single-GPU check of the pack / unpack kernels and the packed transport: the rectangle [x0, x1) x [y0, y1) (texels of texel_bytes bytes) of an image of pitch_bytes
 * per row goes through pack -> ncclSend / ncclRecv with this rank as its own peer (loopback: a device copy) -> unpack onto the rectangle of the same size at (dst_x, dst_y) */
int plrf_rccl_self_test_rect(void* exchange, void* device_ptr, uint32_t pitch_bytes, uint32_t texel_bytes, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t dst_x,
                             uint32_t dst_y, void* launch_stream);

/* ---- exchange watchdog (round 5): an overlapped exchange is a wait the launch stream parks on; if a peer never posts its side the frame
 * hangs and says nothing. Every BEGIN arms an entry {rank, exchange id, phase, completion query, time}; entries are checked at every later exchange
 * callback (the frame then fails with the message) and by a background thread every 50 ms, which prints the message to stderr when an entry goes overdue - a REPORT:
 * a peer paused in a debugger or an oversubscribed host recovers, and the thread says so and frames go on (round 6) - and aborts the process only when
 * the entry is still incomplete PLRF_EXCHANGE_WATCHDOG_ABORT_MS later (default 30000; PLRF_EXCHANGE_WATCHDOG_ABORT=0: never): a hung collective cannot be
 * cancelled, only reported. Deadline: PLRF_EXCHANGE_WATCHDOG_MS (default 2000, 0 = off); the first 64 exchanges of a communicator (about ten frames: RCCL sets up its point-to-point connections inside the first group with each peer, and ranks
 * leave their set-up at different times) get PLRF_EXCHANGE_WATCHDOG_FIRST_MS (default 60000) instead. The functions below expose the mechanism with a caller-supplied completion query, so it can be tested without a GPU. */
typedef int (*plrf_watchdog_query)(void* user); /* 0 = still running, 1 = complete */
int plrf_watchdog_create(uint32_t deadline_ms, void** out_watchdog);
int plrf_watchdog_destroy(void* watchdog);
int plrf_watchdog_arm(void* watchdog, int rank, int exchange_id, int phase, plrf_watchdog_query query, void* user);
/* 0: nothing overdue (completed entries are dropped); 1: an entry is past its deadline - out_message names rank, exchange and phase */
int plrf_watchdog_poll(void* watchdog, char* out_message, size_t capacity);

typedef struct plrf_camera { float position[3], forward[3], up[3], right[3]; } plrf_camera;

int plrf_default_settings(plrf_settings* out, uint32_t width, uint32_t height);
int plrf_create(const plrf_settings* settings, void** out_pipeline);
int plrf_destroy(void* pipeline);
const char* plrf_last_error(void);
/* named resources: see FramePipeline::image / storageBuffer / uniformBuffer */
int plrf_get_image(void* pipeline, const char* name, plr_image_handle* out);
int plrf_get_storage_buffer(void* pipeline, const char* name, plr_storage_buffer_handle* out);
int plrf_get_uniform_buffer(void* pipeline, const char* name, plr_uniform_buffer_handle* out);
/* registers one R16F res^3 SDF volume, returns its global texture array index (SDFInstance.sdfTextureIndex) */
int plrf_add_sdf_volume(void* pipeline, uint32_t res, const void* half_data, size_t bytes, uint32_t* out_texture_index);
/* the same from a DDS file written by the asset pipeline (plr_write_dds_file / the reference's writeDDSFile): any width x height x depth */
int plrf_add_sdf_volume_dds(void* pipeline, const char* path, uint32_t* out_texture_index, uint32_t out_size[3]);
/* {uint count; uint pad[3]; SDFInstance[count]} and {vec3 min; float; vec3 max; float}[count] (SDFGI::updateSDFScene) */
int plrf_set_sdf_scene(void* pipeline, const void* instance_buffer, size_t instance_bytes, const void* world_bbs, size_t bb_bytes);
int plrf_set_sun_direction(void* pipeline, const float direction[3]);
int plrf_set_camera_intrinsic(void* pipeline, float fov_degrees, float near_plane, float far_plane);
int plrf_set_camera_cut(void* pipeline);
/* one iteration of the reference's main loop: record the frame, update camera/UBOs, submit (does not wait for the GPU) */
int plrf_frame(void* pipeline, const plrf_camera* camera, float delta_time, float time);
/* host copies of what the last plrf_frame submitted (340-byte global UBO image, 9 TAA resolve weights) */
int plrf_get_submitted_globals(void* pipeline, void* out_340_bytes);
int plrf_get_resolve_weights(void* pipeline, float* out_9);
int plrf_get_cpu_frame_index(void* pipeline, uint64_t* out);

#ifdef __cplusplus
}
#endif
#endif
