/*
 * gsplat_hip.h -- C ABI of libgsplat_hip.so, the MI355X (gfx950) Gaussian-splat
 * rasterizer that replaces the device half (and the per-camera-move CPU sort)
 * of the reference's GSplatRenderer.  No HDK, GL or torch types cross this
 * boundary: plain pointers, sizes and PODs only.  Every function returns 0 on
 * success or a negative GSR_E_* code and never throws; gsr_last_error() gives
 * the text.  A context is bound to one GPU and is NOT thread-safe (the
 * reference runs everything on Houdini's single draw thread).
 *
 * Reference interfaces replaced (paths relative to /root/reference/gsplat_plugin):
 *   gsr_upload*        <- GSplatRenderer::generateRenderGeometry  src/GSplatRenderer.C:420-532
 *                         (TBB pack into 1 RGBA32F + 2 RGB16F textures, setTexture x3)
 *   gsr_render         <- GSplatRenderer::render                  src/GSplatRenderer.C:534-658
 *                         = argsortByDistance :176-216 (CPU, TBB) + index-texture upload :586-592
 *                         + GL state/uniforms :605-645 + drawInstanced :647, which runs
 *                         shaders/GSplatShaderSource.h:190-288 (VS), :304-312 (FS)
 *                         and the fixed-function blend :613-621
 *   gsr_camera         <- the uniform block of the main program   shaders/GSplatShaderSource.h:119-133,153-159
 */
#ifndef GSPLAT_HIP_H
#define GSPLAT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSR_OK                0
#define GSR_E_INVALID        -1   /* bad argument */
#define GSR_E_HIP            -2   /* a HIP runtime call failed */
#define GSR_E_NO_DEVICE      -3   /* no gfx950 device / device ordinal out of range */
#define GSR_E_NO_GEOMETRY    -4   /* render before any upload */
#define GSR_E_TOO_MANY_PAIRS -5   /* (tile, splat) pair count exceeds GSR_MAX_PAIRS */
#define GSR_E_OOM            -6
#define GSR_E_COMM           -7   /* RCCL could not be loaded, or a communicator call failed */

#define GSR_TILE              16          /* tile edge in pixels */
#define GSR_MAX_DIM           16384       /* max framebuffer width/height (1024 x 1024 tiles) */
#define GSR_MAX_PAIRS         0x7fffff00ll

typedef struct gsr_context gsr_context;

/* Per-frame uniforms.  Field names = the GLSL uniforms of the reference's main
 * program.  Matrices: 16 floats, GL column-major (m[c*4+r]); this is the byte
 * layout of Houdini's UT_Matrix4F::data(), so HDK glue passes them through. */
typedef struct gsr_camera {
    float obj_view[16];    /* glH_ObjViewMatrix    */
    float object[16];      /* glH_ObjectMatrix     */
    float inv_object[16];  /* glH_InvObjectMatrix  */
    float view[16];        /* glH_ViewMatrix       */
    float proj[16];        /* glH_ProjectMatrix    */
    float cam_pos[3];      /* WorldSpaceCameraPos: sort reference point and SH eye
                              (src/GSplatRenderer.C:551-563) */
    int32_t width;         /* glH_ScreenSize.x  (<= GSR_MAX_DIM) */
    int32_t height;        /* glH_ScreenSize.y  (<= GSR_MAX_DIM) */
    int32_t sh_order;      /* GSplatShOrder 0..3; forced to 0 when no SH was uploaded
                              (doSH gate, src/GSplatRenderer.C:623,628) */
} gsr_camera;

/* Counters and timings of the most recent gsr_render plus running totals
 * (totals feed bench.py's roofline: blend_ms_total / blend_launches). */
typedef struct gsr_stats {
    int64_t n_splats;          /* uploaded */
    int64_t n_visible;         /* survived w/z culling and have >=1 tile */
    int64_t pairs_total;       /* D: (super-tile, splat) list entries of the frame */
    int64_t pairs_consumed;    /* records gathered by the blend kernel before early-out */
    int32_t tiles_x, tiles_y;  /* tile grid of this context's shard */
    int32_t record_bytes;      /* bytes of one projected record as gathered by the blend kernel (48) */
    int32_t pair_bytes;        /* bytes of one list entry as scanned by the blend kernel (idx + rect = 8) */
    float ms_preprocess, ms_depth_sort, ms_emit, ms_tile_sort, ms_blend, ms_total; /* last frame, HIP events (timing
                                  level 2; level 1 fills ms_blend only): ms_emit = binning count + scans + pair count
                                  to the host, ms_tile_sort = binning placement */
    double blend_ms_total;     /* sum over all frames since gsr_stats_reset */
    int64_t blend_launches;
    int64_t blend_pairs_consumed_total;
    double frame_ms_total;
    int64_t frames;
    int64_t entries_scanned;               /* list entries (idx+rect) read by the blend kernel, last frame */
    int64_t blend_entries_scanned_total;   /* ... running total */
    int32_t super_tile;                    /* super-tile edge in tiles */
    int32_t stiles_x, stiles_y;            /* super-tile grid (whole image) */
    int32_t reserved_;
    int64_t blend_wave_evals_total;        /* (wave, record) evaluations by the blend kernel = 64 pixel evaluations each, running total */
    double stage_ms_total[5];              /* timing level 2: running totals of ms_preprocess, ms_depth_sort, ms_emit,
                                              ms_tile_sort, ms_blend over stage_frames frames */
    int64_t stage_frames;
    int64_t sorts_skipped;                 /* frames that reused the cached depth order (GSR_OPT_SORT_CACHE) */
    int64_t frames_requeued;               /* frames whose back end ran twice: the pair count outgrew the list buffer */
    int64_t lazy_redo_tiles;               /* lazy colour, last frame: tiles composited by the on-demand fallback */
    int64_t lazy_colours_total;            /* lazy colour: SH evaluations by the ahead-of-time pass, running total */
    int64_t frames_truncated;              /* GSR_OPT_DEFERRED_CHECK only: frames handed over with clamped lists (must stay 0
                                              for exact pixels; the buffer is regrown for the following frames) */
    int64_t frames_culled;                 /* GSR_OPT_OCCLUSION_CULL: frames rendered against the previous frame's depth horizons */
    int64_t frames_repaired;               /* ... of which this many broke a horizon and were rendered again without culling */
    int64_t clusters_total;                /* clusters of 64 storage-ordered splats (k_cluster.h) */
    int64_t clusters_kept;                 /* ... that survived the cluster culling of the last frame whose count reached the host */
    int32_t policy_bits;                   /* what the kernels told the host about the last frames: 1 = lazy colour pays, 2 = a heaviest-first
                                              tile order pays, 4 = occlusion culling has something to work with, 8 = the last culled frame
                                              kept more than 70 % of what an unculled one keeps (culling suspended), 16 = colouring list prefixes is
                                              cheaper than colouring every kept splat */
    int32_t cull_dilate;                   /* occlusion culling: current dilation radius in tiles (GSR_OPT_CULL_DILATE, grown by repairs) */
    int32_t cull_holdoff;                  /* ... frames for which it stays switched off */
    int32_t reserved2_;
    int64_t frames_resorted;               /* GSR_OPT_LOCAL_SORT: frames whose small-frame sort met a bucket far beyond its prediction and were
                                              rendered again with the three global passes */
    int64_t frames_slab;                   /* GSR_OPT_FRONT_SLAB: frames rendered in two phases (front slab, then the rest behind the tiles still open) */
    int64_t frames_jumped;                 /* frames whose camera had jumped since the frame that left the depth horizons: rendered without them (policy mode only) */
    int64_t frames_lazy;                   /* frames whose K1 left the SH colours pending (k_colour.h: list prefixes + on-demand fallback); the others shaded in K1 */
    int64_t uploads;                       /* complete uploads (gsr_upload_end) since gsr_create / gsr_stats_reset */
    double upload_ms[6];                   /* the LAST upload: [0] host -> device copies (wall clock spent inside gsr_upload_append*, quantisation of raw
                                              attributes included), [1] bounding box + Morton codes + their sort (HIP events), [2] k_pack: the arrays into
                                              storage order + cluster bounds (HIP events), [3] wall clock gsr_upload_begin .. gsr_upload_end, [4], [5] reserved */
} gsr_stats;

/* ---- lifetime ----------------------------------------------------------- */
int  gsr_device_count(void);                              /* 0 if no GPU is visible */
int  gsr_create(int device, gsr_context** out);           /* owns a HIP stream on `device` */
void gsr_destroy(gsr_context* ctx);
const char* gsr_last_error(void);                         /* thread-local, never NULL */
const char* gsr_version(void);

/* The context's PUBLIC stream: gsr_render orders its result on it (work queued there afterwards sees
 * the finished frame; the frame's output write waits for work queued there before).  Internally the
 * kernels run on per-frame-slot streams.  Pass a caller-owned hipStream_t (e.g. torch's current
 * stream); NULL restores the context's own stream. */
int  gsr_set_stream(gsr_context* ctx, void* hip_stream);

/* ---- geometry staging (active set changed) ------------------------------ */
/* Arrays are HOST pointers in the reference's registerUpdate() layout
 * (include/GSplatRenderer.h:34-47): P float[3n]; Cd half[3n]; alpha float[n];
 * scale half[3n]; orient half[4n] (x,y,z,w); shx/shy/shz half[16n] (row-major
 * 4x4 per splat, coefficient j at (j/4, j%4)) or all three NULL.  halves are raw
 * binary16 bits.  Data is copied; the caller may free on return.
 * begin/append/end lets the shim concatenate several registry entries without a
 * host-side merge (src/GSplatRenderer.C:420-513). origin = GSplatOrigin. */
int  gsr_upload_begin(gsr_context* ctx, int64_t total_splats, int has_sh, const float origin[3]);
int  gsr_upload_append(gsr_context* ctx, int64_t n,
                       const float* P, const uint16_t* Cd, const float* alpha,
                       const uint16_t* scale, const uint16_t* orient,
                       const uint16_t* shx, const uint16_t* shy, const uint16_t* shz);
/* The same, from RAW float32 point attributes (what a Houdini detail holds): the GPU quantises to fp16 (round to nearest even,
 * overflow to infinity: HDK's fpreal16) and packs -- the work GR_PrimGsplat::update does in a tbb::parallel_for on the CPU
 * (src/GR_GSplat.C:302-372).  NULL Cd / alpha / scale / orient = the reference's defaults (0 / 1 / 1 / (0,0,0,1), :233-272,309-312).
 * SH in any of the three naming schemes (:93-189): sh_scheme 0 = none; 1 = sh_array: sh_vec3_per_point vec3 per point (the first 16
 * are used); 2 = sh_ptr[0..14] = sh1..sh15, float[3n] each; 3 = sh_ptr[0..44] = f_rest_0..44, float[n] each (channel-major: coefficient
 * j = (f_rest_j, f_rest_{j+15}, f_rest_{j+30})).  In schemes 2 and 3 the arrays after the first NULL one count as absent.
 * Attribute precedence (Alpha over opacity), which scheme applies and the gsplat__sh_order rule stay with the caller (GSplatPrim).
 * The SoA left in HBM is bit-identical to gsr_upload_append of the host-quantised arrays. */
typedef struct gsr_raw_attrs {
    const float* P;            /* float[3n], required */
    const float* Cd;           /* float[3n] or NULL */
    const float* alpha;        /* float[n] or NULL */
    const float* scale;        /* float[3n] or NULL */
    const float* orient;       /* float[4n] (x, y, z, w) or NULL */
    int32_t sh_scheme;
    int32_t sh_vec3_per_point;
    const float* sh_array;
    const float* const* sh_ptr;
} gsr_raw_attrs;
int  gsr_upload_append_raw(gsr_context* ctx, int64_t n, const gsr_raw_attrs* attrs);
int  gsr_upload_end(gsr_context* ctx);
/* gives up an upload that failed between begin and end: the context holds no geometry afterwards */
int  gsr_upload_abort(gsr_context* ctx);
/* begin + append + end for a single entry */
int  gsr_upload(gsr_context* ctx, int64_t n,
                const float* P, const uint16_t* Cd, const float* alpha,
                const uint16_t* scale, const uint16_t* orient,
                const uint16_t* shx, const uint16_t* shy, const uint16_t* shz,
                const float origin[3]);

/* ---- multi-GPU: tile-row shard ------------------------------------------ */
/* This context renders only the tile rows of shard `index` of `count`: rows r with r % count == index (layout 0,
 * interleaved: balances any scene) or the contiguous band [index*rpb, (index+1)*rpb), rpb = ceil(tile rows / count)
 * (layout 1, GSR_OPT_SHARD_LAYOUT: a rank keeps ~1/count of the splats, so its sort/binning/colour work shrinks too).
 * Its output is the compact band image: the owned tile rows stacked bottom-up, gsr_band_rows() pixel rows of `width`
 * RGBA-f32 pixels.  The stitched frame is bit-identical to the unsharded one in either layout.  The band is padded to the
 * same height on every rank: pixel rows behind the rank's last image row (a last tile row the image does not fill, whole
 * tile rows of a rank that owns fewer than the others, all of it for a rank beyond the image) are NEVER written in a
 * device target -- clear it once if they are to read as zeros -- and read as zeros in a host target. */
int  gsr_set_row_shard(gsr_context* ctx, int index, int count);
int  gsr_band_rows(int height, int index, int count);      /* pixel rows in that band image */
/* Root side: bands[count] gathered back to back (each padded to
 * gsr_band_rows(height, 0, count) rows) -> full image.  Device pointers. */
int  gsr_stitch_bands(gsr_context* ctx, const float* gathered, int count,
                      int width, int height, float* rgba_out);

/* ---- several GPUs, one caller thread -------------------------------------- */
/* The reference draws from Houdini's single draw thread (src/DM_GSplatHook.C:30-39); gsr_multi drives G contexts -- one
 * per GPU, splats replicated, rank g owning a contiguous band of tile rows (GSR_OPT_SHARD_LAYOUT = 1 is gsr_multi's
 * default; 0 = interleaved rows) -- on behalf of that one thread: every rank's frame is queued by a worker thread of its
 * own (GSR_MULTI_THREADS=0 in the environment: by the caller's thread, one rank after the other), then the caller's thread
 * issues the frame's ONE collective: ncclRecv x (G-1) + ncclSend per peer in one group (RCCL over xGMI) on per-rank
 * transfer streams.  With the band layout the peers' bands are received straight into rgba_out (a band is a block of rows
 * of the final image); with interleaved rows they are de-interleaved on the root.  Bands are double-buffered, so the
 * gather of frame f overlaps the kernels of frame f+1.  The result is bit-identical to the 1-GPU frame. */
typedef struct gsr_multi gsr_multi;
#define GSR_TRANSPORT_AUTO   0   /* RCCL when the devices are distinct (and librccl loads), else COPY */
#define GSR_TRANSPORT_RCCL   1   /* single-process communicator (ncclCommInitAll) */
#define GSR_TRANSPORT_COPY   2   /* hipMemcpyPeerAsync / device-to-device copies ordered by events: several contexts on ONE
                                    GPU (how the 1-GPU test box runs the whole path), or a fallback without RCCL */
int  gsr_multi_create(const int* devices, int count, int transport, gsr_multi** out);   /* devices may repeat for COPY */
void gsr_multi_destroy(gsr_multi* m);
int  gsr_multi_count(gsr_multi* m);
int  gsr_multi_transport(gsr_multi* m);                       /* the transport in use (GSR_TRANSPORT_RCCL / _COPY) */
gsr_context* gsr_multi_context(gsr_multi* m, int rank);      /* rank's context (stats, debug access); do not destroy */
int  gsr_multi_set_stream(gsr_multi* m, void* hip_stream);   /* the stream on devices[0] that frames are ORDERED on (the
                                                                 ranks' kernels and the gather run on streams of the library) */
int  gsr_multi_set_option(gsr_multi* m, int option, int value);
/* what RCCL itself reports: ranks_out[g] = ncclCommUserRank of rank g's communicator, *nranks_out = ncclCommCount of the
 * root's (-1 / 0 with the COPY transport) */
int  gsr_multi_comm_info(gsr_multi* m, int* ranks_out, int* nranks_out);
/* the gather on the root's transfer stream, from the first receive to "frame complete", measured with HIP events:
 * enable = 1 / 0 switches the measurement (and clears the sums when it changes), -1 leaves it; returns the sums so far.
 * Synchronises. */
int  gsr_multi_gather_stats(gsr_multi* m, int enable, double* ms_total, int64_t* gathers);
int  gsr_multi_upload_begin(gsr_multi* m, int64_t total_splats, int has_sh, const float origin[3]);
int  gsr_multi_upload_append(gsr_multi* m, int64_t n, const float* P, const uint16_t* Cd, const float* alpha,
                             const uint16_t* scale, const uint16_t* orient,
                             const uint16_t* shx, const uint16_t* shy, const uint16_t* shz);
int  gsr_multi_upload_end(gsr_multi* m);
int  gsr_multi_upload_abort(gsr_multi* m);
int  gsr_multi_upload(gsr_multi* m, int64_t n, const float* P, const uint16_t* Cd, const float* alpha,
                      const uint16_t* scale, const uint16_t* orient,
                      const uint16_t* shx, const uint16_t* shy, const uint16_t* shz, const float origin[3]);
/* full frame on devices[0] (device pointer there, asynchronous, ordered on the stream of gsr_multi_set_stream) or in host
 * memory (synchronous) */
int  gsr_multi_render(gsr_multi* m, const gsr_camera* cam, float* rgba_out, int out_is_device);
int  gsr_multi_render_depth(gsr_multi* m, const gsr_camera* cam, const float* depth, int depth_is_device,
                            float* rgba_out, int out_is_device);
int  gsr_multi_synchronize(gsr_multi* m);
int  gsr_multi_get_stats(gsr_multi* m, int rank, gsr_stats* out);

/* ---- one process per GPU (torchrun-style launches): the same gather ----------- */
/* The launcher hands every rank the 128-byte id rank 0 obtained (any side channel); after gsr_comm_init a frame is one
 * call per rank: the rank's band is rendered, sent (ncclSend) or received (root: straight into rgba_out_device with the
 * band layout, stitched there with interleaved rows).  With world > 1 the context's kernels move to a stream of the
 * library and the collective to a second one (the gather of frame f overlaps frame f+1); the frame is ORDERED on the
 * context's public stream (gsr_set_stream, before or after gsr_comm_init).  rgba_out_device is the FULL frame on the root
 * and ignored elsewhere. */
#define GSR_COMM_ID_BYTES 128
int  gsr_comm_available(void);                       /* 1 if librccl could be loaded in this process (no GPU work) */
int  gsr_comm_get_unique_id(void* id);
int  gsr_comm_init(gsr_context* ctx, const void* id, int rank, int world);   /* collective; sets the row shard (rank, world) */
int  gsr_comm_info(gsr_context* ctx, int* rank, int* nranks);               /* ncclCommUserRank / ncclCommCount */
int  gsr_comm_destroy(gsr_context* ctx);
int  gsr_comm_render(gsr_context* ctx, const gsr_camera* cam, const float* depth, int depth_is_device,
                     float* rgba_out_device);

/* ---- per frame ---------------------------------------------------------- */
/* Renders the uploaded splats.  rgba_out: float[rows*width*4], premultiplied
 * RGBA, row 0 = BOTTOM row (GL window coordinates), cleared to 0 -- what the
 * reference's blend leaves in an initially transparent float target.  rows =
 * height, or gsr_band_rows() when sharded.  out_is_device: 0 = host pointer
 * (synchronous), 1 = device pointer: asynchronous, ordered on the context's public stream
 * (gsr_set_stream); the call itself only waits for the frame's 4-byte pair count, which the GPU
 * delivers mid-frame while it keeps working (GSR_OPT_DEFERRED_CHECK removes that wait too). */
int  gsr_render(gsr_context* ctx, const gsr_camera* cam, float* rgba_out, int out_is_device);

/* Same frame, depth-tested against what is already in the viewport: the reference draws
 * after Houdini's opaque pass with the depth test on and depth writes off (src/GSplatRenderer.C:595-610).
 * depth: float[height*width] window-space depth (0..1, row 0 = bottom) of the FULL image (also when
 * sharded), host or device pointer; NULL = no test.  A fragment survives iff the splat's window depth
 * (one value per quad, ndc.z*0.5+0.5) <= depth[pixel]. */
int  gsr_render_depth(gsr_context* ctx, const gsr_camera* cam, const float* depth, int depth_is_device,
                      float* rgba_out, int out_is_device);

/* Wireframe overlay (the reference's wire program, shaders/GSplatShaderSource.h:22-110 drawn in
 * src/GR_GSplat.C:477-483): the outline of every splat's +-2 quad in colour Cd, alpha 1, nearest line wins,
 * background 0.  Whole image (ignores the row shard), synchronous.  Like the reference's wire program it uses
 * P without the GSplatOrigin round trip and no object matrix in the covariance. */
int  gsr_render_wire(gsr_context* ctx, const gsr_camera* cam, float* rgba_out, int out_is_device);
/* Wire-OVER display: the reference draws the outlines and still includes the primitive in the splat pass
 * (src/GR_GSplat.C:471-486), so shaded + wire shows both.  The outlines are written on top of the frame that is
 * already in rgba_inout (a finished gsr_render / gsr_render_depth target of the same size); pixels no outline
 * covers keep their value. */
int  gsr_render_wire_over(gsr_context* ctx, const gsr_camera* cam, float* rgba_inout, int is_device);

int  gsr_synchronize(gsr_context* ctx);
int  gsr_get_stats(gsr_context* ctx, gsr_stats* out);     /* synchronizes the stream */
int  gsr_stats_reset(gsr_context* ctx);

/* ---- knobs (performance only; never change pixels) ---------------------- */
#define GSR_OPT_XCD_SWIZZLE     1   /* tile -> workgroup mapping of the blend kernel: 0 = raster order, 1 = every XCD gets whole
                                       super-tiles, 2 (default) = 1 + heaviest tiles first (from the previous frame's per-tile work) whenever the
                                       kernels find the tiles unequal enough for it to pay, 3 = heaviest first always */
#define GSR_OPT_STAGE_TIMING    2   /* HIP events on the frame's stream: 0 = none, 1 = around the blend kernel only
                                       (default; feeds gsr_stats.blend_ms_total), 2 = around every stage (ms_* fields) */
#define GSR_OPT_SORT_CACHE      3   /* 0/1 (default 1): skip the depth sort when the frame description (camera, shard, geometry) is
                                       unchanged -- argsortByDistance's caching (src/GSplatRenderer.C:179-186) for the case
                                       that matters, a static viewport redraw; the sorted list holds only the splats visible
                                       to the camera that sorted, so a pure rotation re-sorts.
                                       2: the reference's rule in full (src/GSplatRenderer.C:165-186) -- the order depends on the camera
                                       POSITION only: the second frame in a row from one position sorts ALL splats for it once, and until
                                       the position moves every frame walks the splats in that order and skips its sort (pixels identical;
                                       single context, not deferred).  Off by default: on MI355X re-sorting the few hundred thousand splats
                                       a frame keeps is cheaper than walking six million in depth order */
#define GSR_OPT_SUPER_TILE      4   /* super-tile edge in tiles: 0 = auto (smallest power of two giving
                                       <= 256 super-tiles), or a lower bound 1,2,4,8,16 (raised as needed
                                       to stay within 256 super-tiles) */
#define GSR_OPT_FRAMES_IN_FLIGHT 6  /* 1 (default) or 2: with 2, frame f+1's front end (preprocess, sorts, binning) overlaps frame f's
                                       blend kernel on the GPU.  Per-frame results and their order on the context stream are
                                       unchanged.  Measured on MI355X: +3..6 % on frames that are not occlusion-culled, -4..10 % on
                                       culled ones (a slot's depth horizons are then two frames old, and the hand-over costs events) */
#define GSR_OPT_DEBUG_FLAGS     5   /* A/B switches for profiling: 1 = no alpha-support shrink of the bboxes,
                                       2 = bbox-only quadrant masks (no separating-axis test), 4 = sort all 32 key bits,
                                       8 = lazy colour without the ahead-of-time pass (every tile takes the fallback), 16 = cluster culling in the
                                       several-rounds-per-workgroup form of clouds beyond 33 M splats */
#define GSR_OPT_DEFERRED_CHECK   7   /* 0 (default) / 1: with a DEVICE target, gsr_render returns as soon as the frame is queued --
                                       no host wait at all -- and the frame's pair count is looked at by the next call that
                                       touches the context.  The back end always runs against the list buffer sized from
                                       earlier frames (+25 % headroom); a frame whose pair count outgrows it is composited
                                       from clamped lists and counted in gsr_stats.frames_truncated (the buffer is regrown
                                       for the next frame).  The first frame after a buffer-less start is never deferred. */
#define GSR_OPT_SHARD_LAYOUT     9   /* 0 (default) = interleaved tile rows, 1 = contiguous bands; set on every rank AND on the
                                       context that stitches */
#define GSR_OPT_TIMING_EVERY    11   /* timing level 1 brackets the blend kernel of every N-th frame only (default 1): a pair of events in
                                       the stream costs the GPU ~12 us of idle queue, and an average wants a sample, not a census */
#define GSR_OPT_OCCLUSION_CULL  10   /* 0 / 1 (default: when the kernels find horizons for most lists, and not for a while after a
                                       horizon broke) / 2 (whenever possible): splats behind the depth at which every tile of the super-tiles they reach went
                                       opaque in the previous frame are dropped before projection, sorting and binning.  Exact: the lists
                                       are cut at those horizons, a tile that runs off a cut list without going opaque reports the
                                       frame, and gsr_render renders it again (as a front-slab frame where that pays: GSR_OPT_FRONT_SLAB) before it
                                       returns (gsr_stats.frames_repaired).  3 = never against a previous frame: every frame is a front-slab frame
                                       (occlusion culling inside the frame only: no prediction, no repairs, a frame time that does not depend on
                                       how the camera moved).
                                       Off for GSR_OPT_DEFERRED_CHECK frames.  Every rank of gsr_multi / gsr_comm culls and checks its own band. */
#define GSR_OPT_LAZY_COLOUR      8   /* SH colours only for the splats a frame can composite (the front of every super-tile list, as
                                       deep as the previous frame scanned, with an on-demand fallback) instead of for every visible
                                       splat: 0 = never, 2 = always, 1 (default) = when it pays -- the kernels compare, every frame,
                                       what the colour pass would evaluate with what eager evaluation does (large dense clouds: yes;
                                       small or sparse ones: no).  Same pixels, bit for bit, in every mode. */
#define GSR_OPT_CLUSTER_CULL    12   /* 1 (default) / 0: every frame starts by testing CLUSTERS of 64 spatially adjacent splats (their box + largest
                                       extent) against the clip planes, the screen, this rank's band of tile rows and the depth
                                       horizons; the per-splat stage runs over the surviving clusters only.  Conservative: same pixels. */
#define GSR_OPT_STORAGE_ORDER   13   /* 1 (default) = the splats are stored in Morton order of their positions (takes effect at the next
                                       upload), 0 = in upload order.  The storage order is what breaks ties in the depth sort (the
                                       reference's own tie order is unspecified: unstable tbb::parallel_sort, src/GSplatRenderer.C:206-207);
                                       gsr_debug_read_storage_order returns it. */
#define GSR_OPT_CULL_DILATE     14   /* occlusion culling: tiles by which a splat's (or cluster's) tile rect is widened before it is compared with the
                                       depth horizons of the previous frame (default 2; 0..64).  The view moves between frames: a wider
                                       neighbourhood culls less but breaks less often.  The library doubles it whenever a frame had to be
                                       repaired and lets it shrink back to this value while frames hold. */
#define GSR_OPT_LOCAL_SORT      15   /* depth sort of frames that keep few splats: 1 (default) = when the slot's previous frame kept <= 0.5 M, one global
                                       scatter into 1024 buckets over the key range that frame kept + one kernel that sorts every bucket locally
                                       (2 launches instead of 9); 0 = always three global LSD passes; 2 = the local form whenever a previous
                                       frame's key range is known.  Same order either way. */
#define GSR_OPT_FRONT_SLAB      16   /* occlusion culling WITHOUT a previous frame.  A frame that cannot use the previous frame's depth horizons (the
                                       first frames of a cloud, a camera jump, the re-render of a frame that broke a horizon) is rendered in two
                                       phases: the nearest splats first (a slab holding ~a tenth of the surviving clusters, picked from a histogram of
                                       their distances), then -- the tiles that are opaque by then need nothing more, exactly, with no prediction and
                                       no check -- the rest only where a tile is still open, continuing from the stored colour and transmittance.
                                       Bit-identical to the one-pass frame.  1 (default) = where occlusion culling pays; 0 = off; 2 = every frame
                                       that is not culled against a previous frame. */
int  gsr_set_option(gsr_context* ctx, int option, int value);

/* ---- debug / test access (device -> host copies of intermediates) -------- */
/* One projected record as tests read it back (not the packed device layout). */
typedef struct gsr_debug_record {
    float cx, cy, a1x, a1y, b1x, b1y, hx, hy, r, g, b, la;   /* a1 = kappa e/s1, b1 = kappa e_perp/s2, kappa = sqrt(log2 e); la = log2(opacity) (contract v3) */
    float key;
    int32_t visible;   /* 0 = culled */
} gsr_debug_record;
int  gsr_debug_read_records(gsr_context* ctx, gsr_debug_record* out, int64_t n);
/* depth order (nearest first) of the splats that survived culling in the last frame; writes min(cap, count)
 * indices and the count */
int  gsr_debug_read_depth_order(gsr_context* ctx, int32_t* perm, int64_t cap, int64_t* n_sorted);
/* the storage order of the uploaded splats: perm[j] = upload index of the splat in storage slot j (n = uploaded count).
 * Equal sort keys leave the depth sort in this order. */
int  gsr_debug_read_storage_order(gsr_context* ctx, int32_t* perm, int64_t n);
/* per-SUPER-tile [start,end) into the sorted pair list + the list itself (splat indices);
 * n_lists = stiles_x*stiles_y, n_pairs = pairs_total of the last frame */
int  gsr_debug_read_tile_lists(gsr_context* ctx, int32_t* list_start, int32_t* list_end, int64_t n_lists,
                               int32_t* pair_splat, int64_t n_pairs);

/* per tile of the last frame, four uint32: {list entries scanned, records gathered, wave-record evaluations, bit 0: the tile
 * stopped because every pixel was opaque} as the blend kernel counted them */
int  gsr_debug_read_tile_work(gsr_context* ctx, uint32_t* work4, int64_t n_tiles);
/* per tile of the WHOLE image (tiles_x * tiles_y of the frame, not the shard), four planes of n_tiles floats: the depth horizons the slot's
 * next frame would be culled against (distance^2, dilated; +inf = none), the raw horizons (sign bit = the tile's status), the dilated
 * status, and the covered depths of the last depth-tested frame as K1 saw them */
int  gsr_debug_read_horizons(gsr_context* ctx, float* out4, int64_t n_tiles);

/* Stand-alone device radix sort of (key,value) u32 pairs on bits [0, key_bits):
 * the sort the pipeline uses, exposed for parity tests (host pointers). */
int  gsr_debug_sort_pairs(gsr_context* ctx, uint32_t* keys, uint32_t* vals, int64_t n, int key_bits);
/* ... the form small frames use: one scatter into 1024 buckets of width 2^bucket_shift starting at bucket_lo (keys outside
 * land in the first / last bucket), then every bucket sorted by one workgroup (k_radix_local).  Any lo / shift gives the same order;
 * GSR_E_INVALID if a bucket outgrows its region of 8192 keys (the pipeline then falls back to the global sort). */
int  gsr_debug_sort_pairs_local(gsr_context* ctx, uint32_t* keys, uint32_t* vals, int64_t n, int key_bits,
                                uint32_t bucket_lo, int bucket_shift);
/* The host-side policies (csrc/gsr_policy.h; DESIGN.md section 4: state table) as a pure function: state16 holds
 * [0] cull.pays [1] cull.weak [2] cull.vis_unculled [3] cull.holdoff [4] cull.backoff [5] cull.streak [6] cull.dilate [7] cull.opt_dilate
 * [8] slab.holdoff [9] local.fails [10] local.holdoff [11] (out) the event's answer; event = GsrPolicyEvent (0 upload, 1 cull allows?(a = opt),
 * 2 cull tick, 3 kernel verdict(a), 4 kept(a = splats, b = culled | opt << 1), 5 frame held, 6 horizon broke, 7 slab allows?(a = forced),
 * 8 slab tick, 9 slab done(a = kept), 10 local-sort begin(a = opt, b = static redraw), 11 local-sort result(a = failed), 12 set dilate(a)).
 * No context, no GPU: what tests/test_policy.py drives.  gsr_debug_policy_state reads a live context's state in the same layout. */
int  gsr_debug_policy(int32_t* state16, int event, long long a, long long b);
int  gsr_debug_policy_state(gsr_context* ctx, int32_t* state16);

#ifdef __cplusplus
}
#endif
#endif
