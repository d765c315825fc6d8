/*
 * frr.h -- C ABI of libfrr_hip.so: the MI355X (gfx950) rasterization path behind
 * vmskisme/f_renderer's Renderer / FrameBuffer surface.
 *
 * The reference API is per-triangle and closure-based (Rust generics):
 *   Renderer::geometry_processing   /root/reference/f_renderer/src/renderer.rs:96-112
 *   Renderer::rasterization         /root/reference/f_renderer/src/renderer.rs:269-284
 *   FrameBuffer::{new,fill,clear,get_data,get_size,set_pixel,get_pixel,sample_2d}
 *                                   /root/reference/f_renderer/src/renderer.rs:418-538
 * driven by the two-pass draw loop   /root/reference/examples/src/bin/phong.rs:314-387.
 * Closures and generic varyings cannot cross an FFI to a GPU, so this ABI is batched at the
 * granularity of that draw loop and shaders are table-selected; varyings are a flat float[K]
 * (the trait bounds Add+Sub+Mul<f32>+Copy+Default, renderer.rs:97-102, say "K-dim real vector").
 * Plain pointers and sizes only; no torch / C++ types.  All functions return FRR_OK (0) or a
 * negative frr_status; nothing unwinds across this boundary.  A frr_ctx is driven by one host
 * thread (the reference is single-threaded, vulkan_base.rs:698-723).
 *
 * There is NO CPU fallback behind these symbols: every entry point that computes needs a gfx950
 * device and fails with FRR_ERR_HIP otherwise.
 */
#ifndef FRR_H
#define FRR_H
#ifndef __HIPCC_RTC__
#include <stdint.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define FRR_ABI_VERSION 4
#define FRR_MAX_VARYINGS 16
#define FRR_MAX_TEXTURES 4
#define FRR_MAX_USER_UNIFORMS 64 /* f32 a user shader receives (frr_set_user_uniforms) */
#define FRR_SHADER_USER_BASE 64  /* ids of user shaders (frr_shader_register) start here */
#define FRR_MAX_OUT_TRIS 19 /* 3 + 18 clip vertices -> 19 fan triangles (renderer.rs:150-171,245-264) */

typedef struct frr_ctx frr_ctx;

typedef enum frr_status {
    FRR_OK = 0,
    FRR_ERR_INVALID = -1,     /* bad argument (the reference would panic: clamp min>max, OOB index) */
    FRR_ERR_HIP = -2,         /* HIP runtime error / no gfx950 device; see frr_last_error */
    FRR_ERR_NOMEM = -3,
    FRR_ERR_UNSUPPORTED = -4,
    FRR_ERR_CAPACITY = -5     /* not something a caller has to handle: a device work list (fan slots, (triangle, tile)
                                 records) that turns out too small is grown and the commands since the failing one are
                                 replayed INSIDE the library at the next synchronisation point (frr_sync, frr_readback,
                                 frr_get_stats ...) -- Renderer::rasterization cannot fail (renderer.rs:269-384), neither
                                 can frr_draw.  This code is returned only if eight replays in a row were not enough. */
} frr_status;

/* Vertex-shader table.  Replaces the `vertex_shader: &F` closure argument (renderer.rs:105,110). */
typedef enum frr_vs {
    FRR_VS_CLIP = 0,       /* VSInput = clip xyzw (4 f32);               K = 0 */
    FRR_VS_CLIP_COLOR = 1, /* VSInput = clip xyzw + rgb (7 f32);         K = 3 */
    FRR_VS_PHONG = 2,      /* VSInput = pos3,uv2,normal3 (phong.rs:49-54); body phong.rs:114-126; K = 8 */
    FRR_VS_GOURAUD = 3     /* same input; per-vertex Lambert colour;     K = 3 */
} frr_vs;

/* Pixel-shader table.  Replaces the `pixel_shader: &F` closure argument (renderer.rs:273,283). */
typedef enum frr_ps {
    FRR_PS_DEPTH = 0, /* depth (+triangle id) only, colour untouched */
    FRR_PS_FLAT = 1,  /* uniforms.flat_color */
    FRR_PS_COLOR = 2, /* (ctx[0],ctx[1],ctx[2],1) */
    FRR_PS_PHONG = 3, /* phong.rs:133-154 (reflect-vector Phong x bilinear texture) */
    FRR_PS_BLINN = 4  /* half-vector variant of the same (not in the reference) */
} frr_ps;

/* ---- user shaders -------------------------------------------------------------------------------------------
 * The reference's shaders are closures: `vertex_shader: &F` with F: Fn(&VSUniform, &VSInput, &mut ShaderContext) -> Vec4
 * (renderer.rs:105,110,116) and `pixel_shader: &F` with F: Fn(&PSUniform, &ShaderContext) -> Vec4 (:273,283,380), over any
 * varying type that is a real vector space (Add + Sub + Mul<f32> + Copy + Default, :97-102).  A closure cannot cross this
 * boundary as a function pointer -- it has to run on the GPU inside the geometry and tile kernels -- so it crosses as
 * TEXT: HIP device source that the library compiles at run time (hiprtc, gfx950, -ffp-contract=off like the library
 * itself) into its own kernels.  The source defines, at global scope,
 *
 *   __device__ void frr_user_vs(const frr::DevUniforms &u, const float *in, float pos[4], float *ctx);
 *       in: vs_input_floats f32 of one vertex;  ctx: num_varyings f32, zero on entry (T::default());  pos: clip xyzw
 *   __device__ void frr_user_ps(const frr::DevUniforms &u, const float *ctx, float out[4], const float *u8lut);
 *       ctx: the interpolated varyings (renderer.rs:374-378);  out: the Vec4 that vec4_to_u8_array quantises (:7-14)
 *
 * and may use what the built-in shaders are written with (f_renderer_amd/csrc/frr_device.h, frr_exact.h, namespace frr):
 * u.mvp / u.model (column-major), u.view_pos, u.light_pos, u.light_color, u.ambient_strength, u.specular_strength,
 * u.flat_color, u.user[FRR_MAX_USER_UNIFORMS] (frr_set_user_uniforms: what the closure would have captured),
 * frr::mat4_mul_vec4, frr::dot3, frr::normalize3 (glam's operation order), frr::f32_max, frr::recip_exact, and
 * frr::sample_2d(u, uu, vv, rgba_out, u8lut) (FrameBuffer::sample_2d of the texture in uniforms.texture_slot, :516-538) and
 * frr::sample_2d_slot(u, slot, uu, vv, rgba_out, u8lut) (the same of ANY uploaded slot 0..FRR_MAX_TEXTURES-1: the
 * reference's PSUniform holds three textures and its closure may sample each, phong.rs:41-47,147-151; an empty slot
 * samples as zero).  fp32 expressions keep their written association (no FMA contraction), as in the reference's Rust.
 *
 * frr_shader_register returns one id >= FRR_SHADER_USER_BASE that stands for the pair: pass it as vs_id to
 * frr_mesh_upload / frr_mesh_bind_device and as ps_id to frr_raster / frr_draw.  The two halves also combine with the
 * tables: a mesh uploaded with a user id may be drawn with a built-in ps_id, and a mesh with a built-in (or another
 * user) vs_id with a user ps_id -- provided the pixel shader's varyings are the vertex shader's (frr_vs_num_varyings;
 * FRR_ERR_INVALID otherwise).  ctx may be NULL (compile only: the registry is per process; a ctx loads the code object
 * on first use).  A source that does not compile: FRR_ERR_UNSUPPORTED, with the compiler's log in frr_last_error(ctx). */
int frr_shader_register(frr_ctx *ctx, const char *hip_source, int vs_input_floats, int num_varyings, int *shader_id);
/* u.user[0..n) for the draws issued from now on (n <= FRR_MAX_USER_UNIFORMS; travels with each draw like frr_uniforms) */
int frr_set_user_uniforms(frr_ctx *ctx, const float *values, int n);

/* VSUniform (phong.rs:26-31) + PSUniform (phong.rs:41-47) + light consts (phong.rs:128-132).
 * Matrices column-major like glam::Mat4::from_cols_array (matrix_util.rs:5-7). */
typedef struct frr_uniforms {
    float model[16], view[16], proj[16];
    float view_pos[3];
    float light_pos[3];
    float light_color[3];
    float ambient_strength;
    float specular_strength;
    float flat_color[4];
    int32_t texture_slot; /* PSUniform.place (phong.rs:34-38,147-151) */
} frr_uniforms;

/* One post-setup vertex as the parity tests read it back: the caller-visible content of
 * Vertex<T> (renderer.rs:387-394) minus the NDC `pos`, in EMISSION order (before the
 * orientation swap of renderer.rs:309-312). */
typedef struct frr_setup_vertex {
    float spf[2];
    int32_t spi[2];
    float rhw;
    float ctx[FRR_MAX_VARYINGS];
} frr_setup_vertex;

typedef struct frr_stats {
    uint64_t tris_in;      /* input triangles submitted since the last frr_clear */
    uint64_t tris_setup;   /* triangles after clip + fan */
    uint64_t bin_entries;  /* (triangle, tile) pairs */
    uint64_t frag_covered; /* pass the edge tests renderer.rs:333-341 (exact only while counting is enabled) */
    uint64_t frag_nan;     /* fragments whose rhw interpolated to NaN; they always pass the depth test and so does the next
                              fragment on their pixel (renderer.rs:363-366): reproduced exactly, by a second pass over the
                              tiles that saw one (the NaN bit pattern itself is the device's) */
    uint32_t draws;
    uint32_t replays;      /* how often the library had to grow a work list and replay commands of this frame (see
                              FRR_ERR_CAPACITY): a diagnostic, the results are those of a frame that never failed */
} frr_stats;

/* ---- context ------------------------------------------------------------------------------ */

/* Creates a context on HIP device `device` with a width x height FrameBuffer (renderer.rs:419-425),
 * an f32 depth buffer (phong.rs:208) and a u32 triangle-id buffer.  `stream` is a hipStream_t the
 * caller owns (e.g. torch's current stream), or NULL for a private stream: everything that touches the frame targets
 * (tile kernels, clears, read-backs) runs on it, in call order.  The ctx owns a second, private stream on which the
 * geometry and binning kernels of the next draw run beside the tile kernel of the current one (option "overlap");
 * it is ordered against `stream` with events, so the caller sees one in-order queue -- with one rule: the contents of
 * a device-bound mesh (frr_mesh_bind_device) are read some time between the frr_draw call and the end of the frame's
 * tile kernels, on the library's streams.  A caller that rewrites such a mesh in place therefore (1) orders the rewrite
 * BEHIND the draws issued so far -- frr_frame_fence(ctx, the stream it rewrites on) -- and (2) binds the mesh again
 * (frr_mesh_bind_device: the library's streams then wait for that stream once) before the next draw.  frr_sync in place
 * of (1) and (2) is always sufficient. */
int frr_create(int device, uint32_t width, uint32_t height, void *stream, frr_ctx **out);
void frr_destroy(frr_ctx *ctx);
const char *frr_last_error(const frr_ctx *ctx);
int frr_abi_version(void);

/* Screen-tile partition for multi-GPU runs: this ctx rasterizes only tile rows ty with
 * ty % world == rank (geometry is replicated).  Default (0,1) = everything. */
int frr_set_partition(frr_ctx *ctx, int rank, int world);
/* Which tile rows a rank owns: 0 (default) = interleaved, ty % world == rank -- balances scenes whose
 * load varies down the screen; 1 = blocked, rank owns a contiguous run of tile_rows / world rows, the first
 * tile_rows % world ranks one more (34 rows over 8 ranks: 5,5,4,4,4,4,4,4; frr_owned_rows tells) -- its part of a
 * row-major image is then ONE contiguous slab, so the final-image gather needs no staging copies (bench.py uses this). */
int frr_set_partition_layout(frr_ctx *ctx, int blocked);
/* The pixel rows of a raster window height_range = (y0, y1) (renderer.rs:271: the sub-window argument this whole
 * partition rests on) that this rank owns, as bands [row0, row1) of window-local rows: ONE band in the blocked
 * layout, one per owned 32-pixel tile row in the interleaved layout; a rank that owns nothing has none.  With
 * frr_target_ptrs this is all a non-Python host needs for the final-image exchange: the rank's part of a
 * row-major target of row stride S bytes per row is [row0 * S, row1 * S) of each plane (ncclSend / ncclRecv or
 * peer stores).  frr_owned_band_count returns the number of bands (>= 0) or a negative
 * frr_status. */
int frr_owned_band_count(const frr_ctx *ctx, int32_t y0, int32_t y1);
int frr_owned_rows(const frr_ctx *ctx, int32_t y0, int32_t y1, int32_t band, int32_t *row0, int32_t *row1);
/* The same rule for ANY rank, without a ctx: what the root of the final-image exchange posts its receives with
 * (examples/gather_rccl.cpp).  Returns the number of bands rank `rank` of `world` owns in a window of height_range
 * (y0, y1) (negative: bad argument); if row0/row1 are not NULL and band is in range they receive band `band`. */
int frr_partition_rows(int32_t y0, int32_t y1, int rank, int world, int blocked, int32_t band, int32_t *row0, int32_t *row1);
/* The final-image exchange of one plane as a list of operations -- a pure function of (window, partition, rank), no ctx and
 * no device: what examples/gather_rccl.cpp posts inside one ncclGroupStart / ncclGroupEnd per frame, and what a test can
 * check for every rank of any world size without owning that many GPUs.  The plane is row-major with `row_elems` elements
 * per pixel row; offsets and counts are in elements.  For rank `rank`: on a rank other than `root` one FRR_XFER_SEND per
 * owned band (to root); on root one FRR_XFER_RECV per owned band of every other rank, and one FRR_XFER_COPY per band of its
 * own (render target -> final image, a device copy).  Every element of the window's plane is the destination of exactly
 * one RECV or COPY of root's list, and each SEND of rank p equals (offset, count) the RECV root posts for p, in the same
 * order.  Returns the number of operations (written up to `cap`), or a negative frr_status. */
typedef enum frr_xfer_kind { FRR_XFER_SEND = 0, FRR_XFER_RECV = 1, FRR_XFER_COPY = 2 } frr_xfer_kind;
typedef struct frr_xfer {
    int32_t kind;     /* frr_xfer_kind */
    int32_t peer;     /* SEND: root; RECV: the sending rank; COPY: root itself */
    uint64_t offset;  /* first element, in the rank's render target AND in the final image (same layout) */
    uint64_t count;   /* elements */
} frr_xfer;
int frr_exchange_plan(int32_t y0, int32_t y1, uint32_t row_elems, int rank, int world, int blocked, int root, frr_xfer *ops, int cap);
/* frr_stats.frag_covered is exact while counting is enabled (default).  Disabling it lets the tile
 * kernel drop whole triangles by hierarchical early-z before their coverage is known (images are
 * identical either way; only the statistic stops being maintained). */
int frr_set_count_fragments(frr_ctx *ctx, int enable);

/* Use caller-owned DEVICE buffers (e.g. torch tensors) as the frame targets instead of the
 * internally allocated ones; any may be NULL to keep the internal one (with option bound_targets_in_flight: all three
 * or none).  On a partitioned ctx
 * (frr_set_partition, world > 1) only the tile rows the rank owns are defined in caller-owned targets
 * (a clear that was performed inside a draw never touches the other ranks' rows); the ctx's own targets
 * are brought up to date in full whenever they are read back. */
int frr_bind_targets(frr_ctx *ctx, void *color_rgba8, void *depth_f32, void *tri_id_u32);
/* The device pointers of the current frame's targets, for reads queued on the ctx's stream (which first waits for the
 * frames issued so far).  With two frames in flight the ctx's own targets are two sets: the pointers are those of the
 * CURRENT frame and stay that frame's until the second frr_clear from now, which -- like every frr_clear that reuses a
 * set whose pointers were handed out -- orders the new frame behind what that stream holds by then (the caller's reads). */
int frr_target_ptrs(frr_ctx *ctx, void **color_rgba8, void **depth_f32, void **tri_id_u32);

/* ---- scene ------------------------------------------------------------------------------- */

/* Vec<[VSInput;3]> (phong.rs:187-205): ntris x 3 x floats_per_vertex(vs) f32, host memory. */
int frr_mesh_upload(frr_ctx *ctx, const float *vs_inputs, uint64_t ntris, int vs_id, int *mesh_out);
/* Same, data already in device memory (borrowed until frr_mesh_free). */
int frr_mesh_bind_device(frr_ctx *ctx, const void *dev_vs_inputs, uint64_t ntris, int vs_id, int *mesh_out);
int frr_mesh_free(frr_ctx *ctx, int mesh);
/* FrameBuffer used as texture (PSUniform.sample_2d_*, phong.rs:43-45): RGBA8 row-major.
 * height >= width is required: sample_2d clamps y with width (renderer.rs:523,525), so a
 * shorter texture is an out-of-bounds panic in the reference. */
int frr_texture_upload(frr_ctx *ctx, int slot, const uint8_t *rgba, uint32_t width, uint32_t height);
int frr_set_uniforms(frr_ctx *ctx, const frr_uniforms *u);
int frr_vs_input_floats(int vs_id);
int frr_vs_num_varyings(int vs_id);

/* ---- frame ------------------------------------------------------------------------------- */

/* frame_buffer.fill(rgba) + depth_buffer.fill(depth)  (phong.rs:316-317, renderer.rs:485-494).
 * Also resets the per-frame statistics and the setup list of a preceding frr_geometry.  The device
 * work is deferred: the next full-framebuffer frr_raster / frr_draw performs the clear inside its tile
 * kernel; every other call that can observe the targets or the statistics (frr_readback, frr_sync,
 * frr_get_stats, frr_target_ptrs, frr_bind_targets, a sub-window raster) settles it first, so the
 * observable behaviour is that of an immediate clear.  (Option clear_eager makes it immediate.) */
int frr_clear(frr_ctx *ctx, const uint8_t rgba[4], float depth);

/* Loop A (phong.rs:321-331): Renderer::geometry_processing over every input triangle of `mesh`,
 * results concatenated in submission order on the device.  *ntris_setup (optional) forces a
 * stream sync to return the count. */
int frr_geometry(frr_ctx *ctx, int mesh, uint64_t *ntris_setup);
/* Loop B (phong.rs:361-381): Renderer::rasterization of the triangles of the last frr_geometry
 * with width_range=(x0,x1), height_range=(y0,y1) (renderer.rs:270-271).  As in the reference the
 * window is addressed locally: pixel (cx,cy) lands at colour (cx-x0, cy-y0) with row stride
 * `width` and at depth index (cy-y0)*x1 + (cx-x0) (renderer.rs:323,326,362,381). */
int frr_raster(frr_ctx *ctx, int ps_id, int32_t x0, int32_t x1, int32_t y0, int32_t y1);
/* frr_geometry + frr_raster.  On a partitioned ctx (frr_set_partition, world > 1) the setup list frr_draw builds
 * keeps only the triangles that touch this rank's tile rows of THIS window: a later frr_raster with another
 * window, or after the partition changed, fails with FRR_ERR_INVALID (the reference may reuse one geometry for
 * several ranges, renderer.rs:269-271 -- call frr_geometry for that, it never filters), and so does
 * frr_readback_setup. */
int frr_draw(frr_ctx *ctx, int mesh, int ps_id, int32_t x0, int32_t x1, int32_t y0, int32_t y1);

/* Stream-side fence, no host wait: `stream` (a hipStream_t of the caller; NULL = the ctx's stream) waits for every frame
 * issued so far, so that what the caller enqueues on it next sees their targets.  Needed with option bound_targets_in_flight
 * (below) and by callers that read the ctx's own targets (frr_target_ptrs) on a stream other than the ctx's; also the way to
 * order an in-place rewrite of a device-bound mesh behind the draws that still read it (frr_create).
 * The frames it fences are whole: like the reference's draw (renderer.rs:269-384 has no failure path) a draw of this
 * library cannot fail -- a raster pass whose need of the internal work lists is not known to fit is checked on the host,
 * and repaired, before frr_raster / frr_draw returns (the call then waits for the pass's binning launch, not for its tile
 * kernel; a pass that repeats a mesh, uniforms, window and partition already seen to fit is not waited for).
 * `stream` has to stay alive until the second frr_clear from now when it fenced the ctx's OWN targets (that frr_clear
 * records an event on it: frr_target_ptrs). */
int frr_frame_fence(frr_ctx *ctx, void *stream);
/* The reverse edge: the next kernel that WRITES the frame targets (the pending frr_clear, the next tile kernel) waits for
 * what `stream` (NULL = the ctx's stream) holds now -- e.g. an exchange that still reads a caller-bound target set which is
 * about to be bound for a new frame under option bound_targets_in_flight, where no frame work runs on the ctx's stream.
 * (The ctx's OWN targets need no such call: the library orders a set's next frame behind whatever was queued on the stream
 * the set's pointers were handed to by frr_target_ptrs / frr_frame_fence, up to the frr_clear that starts that frame.) */
int frr_frame_wait(frr_ctx *ctx, void *stream);
/* Synchronisation point: waits for everything issued so far (both streams).  Caller-bound targets are defined, and a
 * draw that needed a larger work list has been replayed (FRR_ERR_CAPACITY), when this -- or frr_readback,
 * frr_get_stats, frr_readback_setup, frr_geometry with a count -- returns. */
int frr_sync(frr_ctx *ctx);
/* FrameBuffer::get_data (renderer.rs:473-475) + depth + triangle ids -> host; NULLs skipped.
 * tri_id holds, per depth-buffer index, the global emission index (over the draws since the last
 * frr_clear) of the triangle owning the pixel, 0xFFFFFFFF where nothing was drawn. */
int frr_readback(frr_ctx *ctx, uint8_t *rgba, float *depth, uint32_t *tri_id);
/* Vec<[Vertex;3]> of the last frr_geometry, [n][3] frr_setup_vertex, emission order */
int frr_readback_setup(frr_ctx *ctx, frr_setup_vertex *out, uint64_t cap_tris, uint64_t *ntris);
int frr_get_stats(frr_ctx *ctx, frr_stats *out);

/* ---- timing on the ctx stream (HIP events) -------------------------------------------------- */
/* slot in [0,16): record an event now; elapsed in ms between two recorded slots (syncs on `b`). */
int frr_event_record(frr_ctx *ctx, int slot);
int frr_event_elapsed_ms(frr_ctx *ctx, int a, int b, float *ms);
/* Development and test switches of a ctx (frr_set_option(ctx, name, value); unknown name or bad value: FRR_ERR_INVALID).
 * None changes a result: every path is held to the same oracle by the tests.  The library itself reads no environment
 * variable; the Python binding maps FRR_* variables onto these for the test-suite and the tools.
 *   "raster_sweep"            1: brute-force tile kernel (one triangle per wavefront) instead of the span kernel
 *   "raster_nw", "raster_occ" waves per tile workgroup (3, 4, 6, 8, 16) / waves per SIMD its registers are budgeted for
 *                             (4, 6, 8); 0 = chosen per launch
 *   "bin_chunks"              number of chunk workgroups of the segmented binning (0 = by mesh size)
 *   "bin_atomics"             1: global-atomic CSR binning (the path for windows of more than 36,864 tiles)
 *   "bin_capacity"            initial capacity of the (triangle, tile) lists in records (tests of the replay)
 *   "fan_capacity"            initial capacity of the fan space in triangles (the same)
 *   "frames_in_flight"        2 (default): with the ctx's OWN targets a frame that starts with frr_clear takes the other of two target
 *                             sets and the other of two streams, so that two frames are in flight (frr_readback / frr_target_ptrs join
 *                             them; frr_target_ptrs returns the current frame's pointers); 1: one target set, one stream
 *   "bound_targets_in_flight" 1: the same for caller-bound targets -- every frame runs on one of two PRIVATE streams (none of it on
 *                             the ctx's stream); the caller binds a different target set for each of two consecutive frames and orders
 *                             its reads with frr_frame_fence (bench.py's multi-GPU loop).  Default 0: everything that touches
 *                             caller-bound targets runs on the ctx's stream, in call order
 *   "overlap"                 frames that are NOT in flight as above: 2 (default) geometry + binning of a pass WITH VARYINGS run on a
 *                             further private stream beside the previous pass's tile kernel (measured to pay there only); 1: of every
 *                             pass; 0: one stream, one kernel after the other
 *   "tile_slot_records"       records per tile slot of the near-first copy (tests of its overflow arena)
 *   "clip_queue"              1: clipped inputs beyond four per 256-triangle block are expanded by a second launch
 *                             (k_geom_clip, one wavefront each over the whole chip) instead of by their block; 0: never;
 *                             -1 (default): when the counters last read back (frr_sync, frr_readback, frr_get_stats)
 *                             showed a block with more than 16 clipped inputs
 *   "clear_eager"             1: frr_clear runs its own kernel at once instead of riding on the next full-window draw */
int frr_set_option(frr_ctx *ctx, const char *name, int64_t value);
/* per-kernel accumulated device time (ms) and launch count since frr_profile_reset.  `mask`:
 * 0 = off, -1 = every kernel, else OR of (1 << index) with index in the order k_clear,
 * k_geom, k_geom_scan, k_bin_count, k_tile_scan, k_bin_fill, k_raster,
 * k_bin_seg (k_geom covers the clip kernel too when the clip queue is in use).  A profiled launch is bracketed by two
 * HIP events on the ctx stream. */
int frr_profile_enable(frr_ctx *ctx, int mask);
/* bracket only every `period`-th launch of each selected kernel (default 1): an event pair costs the
 * stream ~4 us, which matters when the whole frame is 150 us; frr_profile_get then reports the sampled
 * launches. */
int frr_profile_set_period(frr_ctx *ctx, uint32_t period);
int frr_profile_reset(frr_ctx *ctx);
int frr_profile_get(frr_ctx *ctx, const char *kernel, float *total_ms, uint32_t *launches);

/* ---- host helpers mirroring matrix_util.rs:3-35 (pure host arithmetic, no device) --------- */
void frr_set_identity(float m[16]);
void frr_set_look_at(const float eye[3], const float at[3], const float up[3], float m[16]);
void frr_set_perspective(float fovy, float aspect, float zn, float zf, float m[16]);

/* ---- debug / parity hooks ----------------------------------------------------------------- */
/* device evaluation of the library's atan2f (bit-exact fdlibm port) on n (y,x) pairs */
int frr_debug_atan2f(frr_ctx *ctx, const float *y, const float *x, float *out, uint64_t n);
/* the same source compiled for the host, to pin the port against glibc without a GPU */
float frr_host_atan2f(float y, float x);
/* MVP x vertex contraction of a pos3/uv2/normal3 mesh with the current uniforms: clip xyzw per vertex
 * (ntris*3*4 floats) by the exact VALU form (use_mfma = 0, glam's association) or by
 * v_mfma_f32_16x16x4_f32 (use_mfma = 1, an fmaf chain: NOT bit-identical); *ms = kernel time. */
int frr_debug_mvp(frr_ctx *ctx, int mesh, int use_mfma, float *clip_out, float *ms);
/* PMC calibration: gathers 2^log2_records distinct 64-byte records (true bytes = 64 << log2_records) */
int frr_debug_gather_calib(frr_ctx *ctx, uint32_t log2_records);
/* device wave64 inclusive prefix sum (DPP) of 64 values, for tests */
int frr_debug_scan64(frr_ctx *ctx, const uint32_t *in, uint32_t *out);
/* the three-instruction reciprocal of the fragment loop (frr_exact.h: recip_exact) against the IEEE
 * division, and the shaders' 1/sqrt (rsqrt_exact) against 1.0f / sqrtf, for every f32 bit pattern in
 * [lo_bits, hi_bits): number of differing results and the smallest offending pattern (0xFFFFFFFF if none). */
int frr_debug_rcp_check(frr_ctx *ctx, uint32_t lo_bits, uint32_t hi_bits, uint64_t *mismatches, uint32_t *first_bad);

#ifdef __cplusplus
}
#endif
#endif
