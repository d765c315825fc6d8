// frr_device.h -- device-side data layout and the per-stage device functions of the gfx950
// rasterization path.  Stage by stage this follows /root/reference/f_renderer/src/renderer.rs
// (cited inline); the decomposition into kernels is new.
#pragma once
#include "../../include/frr.h"
#include "frr_exact.h"

namespace frr {

constexpr int TILE = 32;          // screen tile edge in pixels (one workgroup, 8 KiB of LDS keys)
constexpr int TILE_PX = TILE * TILE;
constexpr int GEOM_BLOCK = 256;   // input triangles per geometry workgroup

// ---- HBM layout ---------------------------------------------------------------------------
// Setup triangle = what renderer.rs:387-394 `Vertex<T>` x3 carries into rasterization, split in
// two arrays so the coverage/z loop touches exactly one 64-byte line per triangle:
//   RasterRec[n]           64 B : post-orientation-swap (renderer.rs:309-312) vertex order
//   varyings[n][3][K] f32       : same vertex order, read only by the resolve/shade step
// spi (renderer.rs:233-234) is not stored: it is `(spf + 0.5) as i32` of the stored spf, recomputed by whoever needs it
// (the brute-force sweep, frr_readback_setup).  Its 24 bytes hold the tile-independent half of the span set-up instead
// (edge_words below), computed once per triangle by the geometry kernel instead of once per (triangle, tile) pair.
struct alignas(64) RasterRec {
    uint32_t e[6];   // edges 01, 12, 20: {kd, c} -- see edge_words (meaningful for coordinates within +-SPAN_SAFE only)
    float s[6];      // spf: p0.x p0.y p1.x p1.y p2.x p2.y
    float rhw[3];
    uint32_t flags;  // bit0: v1/v2 were swapped; bits1..3: edge 01,12,20 is NOT top-left (bias 1); bits4..6: A > 0 on that
                     // edge; bits16..28: the triangle's emission offset within its 256-input geometry block (< 19 * 256)
};
constexpr int REC_POS_SHIFT = 4, REC_EOFF_SHIFT = 16;
constexpr uint32_t REC_EOFF_MASK = 0x1FFFu;
static_assert(sizeof(RasterRec) == 64, "one cache line per triangle");

#ifdef FRR_DEBUG_COUNTERS
constexpr int DBG_COPIES = 256;
#else
constexpr int DBG_COPIES = 1;
#endif

// Fan slots (the triangles clipped inputs expand to) are handed out by ONE returning device atomic per geometry block that
// clips anything.  Atomics on one address are served one after the other (~17 ns each), so a mesh in which most blocks
// clip something -- any big mesh that crosses the frustum -- queued for as long as the rest of the kernel takes (977
// blocks: 17 us).  The fan space is therefore FAN_REGIONS regions of fan_cap / FAN_REGIONS slots with a cursor each (on
// its own cache line); block b allocates in region b % FAN_REGIONS.  Slot numbers stay unique, which is all the order
// keys, the binning records and the read-back need; the binning walks the USED part of every region (FanMap).
constexpr int FAN_REGIONS = 8;
struct alignas(128) FanCursor { uint32_t v; uint32_t pad[31]; };

// Per-geometry-pass table, double-buffered by the parity of the pass: the tile kernel of pass n reads table n & 1 while the
// geometry kernel of pass n + 1 (on the ctx's second stream) already fills the other one.
struct GeomTab {
    uint32_t n_emit;        // triangles this pass emits (the reference's count; known once the block sums are scanned)
    uint32_t tri_base;      // emission index of this pass's first triangle within the frame
    uint32_t ntris_draw;    // input triangles of this pass (= its first fan slot)
    uint32_t need_fans;     // fan slots this pass needs (valid even on overflow)
    uint32_t frame_no;      // the frame (frr_clear count) this pass belongs to: statistics of older frames are not folded
    uint32_t clip_q;        // clipped inputs handed to the clip kernel's queue (GeomArgs::clipq)
    uint32_t clip_block_max; // the most clipped inputs any one 256-triangle block had (the host's hint for that queue)
    uint32_t pad0;
    // per-pass tables the tile kernel's resolve reads (set by the geometry kernel; kernel arguments would cost it registers)
    const uint32_t *tinfo;           // [ntris] fan size (bits 0-4) | emission offset within its 256-triangle block (bits 5..)
    const uint32_t *fanbase;         // [ntris] clipped inputs: first fan slot, relative to ntris_draw
    const uint32_t *fan_okey;        // [fan slots] order key (within the draw) of each fan triangle
    const uint32_t *block_prefix;    // [blocks] triangles emitted by the blocks before (exclusive scan of the block sums)
    unsigned long long frag_covered; // by the tile kernels that rasterize this pass
    unsigned long long frag_nan;
    FanCursor fan_cursor[FAN_REGIONS]; // fan slots handed out per region (zeroed by the pass BEFORE: it runs beside nobody who uses them)
};
// Per-binning table (segmented binning), same double buffering by the parity of the raster pass.
struct BinTab {
    unsigned long long seg_total; // entries reserved by this pass's chunks (zeroed and folded into the frame total by the NEXT pass)
    uint32_t ent_cursor;          // tile kernel: space handed out in the overflow arena of bins2 (zeroed by this pass's own binning)
    uint32_t frame_no;
};

// A command (geometry pass or raster pass) carries a sequence number.  A pass that finds a work list too small raises
// Counters::first_bad to its number (the smallest such number >= epoch wins; older values are stale); every kernel of a
// LATER command, and the tile kernel of the failed command itself, then does nothing, so that the frame targets and the
// tables hold exactly the state before the failed command.  The host sees the number at its next synchronisation point,
// grows the list and replays the commands from there (frr_api.hip: finish) -- the caller never sees the overflow.
constexpr uint32_t SEQ_NONE = 0xFFFFFFFFu;

// A LANE is the set of tables one frame's passes alternate between.  A ctx that keeps two frames in flight (own targets:
// consecutive frames run on two streams, frr_api.hip) gives each its own lane, so that the bookkeeping threads of two
// frames never touch the same words; within a lane the geometry + binning kernels of consecutive passes never overlap.
struct alignas(128) Lane {
    uint32_t totals_frame;  // the frame the totals below belong to
    uint32_t pad0;
    unsigned long long tot_frag_covered, tot_frag_nan, tot_bin_entries; // of the passes whose tables have been recycled
    unsigned long long bin_total;     // (triangle,tile) pairs of the latest CSR binning
    BinTab btab[2];
    GeomTab gtab[2];
};
struct Counters {
    uint32_t first_bad;     // see above
    uint32_t overflow;      // bit0 fan capacity, bit1 bin capacity (which list to grow)
    uint32_t *host_bad;     // a word of host-visible memory that receives the number of every failed command: the host looks at
                            // it (no synchronisation) whenever a new frame starts, and repairs before it goes on
    Lane lane[2];
    // FRR_DEBUG_COUNTERS builds only (tools/debug_counters.py): funnel counters and per-phase wave cycles of the tile
    // kernel, in DBG_COPIES copies (workgroup b adds to copy b % DBG_COPIES: thousands of device atomics on one cache
    // line would serialise at ~17 ns each and distort what they measure); the host adds the copies up
    unsigned long long dbg[DBG_COPIES][24];
};

// has a command with sequence number `seq` been cancelled by an earlier failure?  (`own`: also by its own)
// The hot kernels read the number first thing (seq_first_bad) and test it where they wait for their first loads anyway
// (seq_is_cancelled): a test right at the top would put one more memory round trip in front of every workgroup.
__device__ __forceinline__ uint32_t seq_first_bad(const Counters *cnt)
{
    return __hip_atomic_load(&cnt->first_bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool seq_is_cancelled(uint32_t b, uint32_t seq, uint32_t epoch, bool own)
{
    return b >= epoch && (own ? b <= seq : b < seq);
}
__device__ __forceinline__ bool seq_cancelled(const Counters *cnt, uint32_t seq, uint32_t epoch, bool own)
{
    return seq_is_cancelled(seq_first_bad(cnt), seq, epoch, own);
}
// a work list of command `seq` is too small (bits: which)
__device__ __forceinline__ void seq_fail(Counters *cnt, uint32_t seq, uint32_t epoch, uint32_t bits)
{
    atomicOr(&cnt->overflow, bits);
    uint32_t old = __hip_atomic_load(&cnt->first_bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (old < epoch || old > seq) {
        const uint32_t seen = atomicCAS(&cnt->first_bad, old, seq);
        if (seen == old) break;
        old = seen;
    }
    if (cnt->host_bad) __hip_atomic_store(cnt->host_bad, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- slots and order keys --------------------------------------------------------------------------------------
// A draw's setup triangles live at SLOTS: input triangle t that emits exactly one triangle (the common, unclipped
// case) uses slot t; the fan of a clipped input (2..19 triangles, renderer.rs:245-264) takes consecutive slots of
// the fan region behind the inputs (slot = ntris + fanbase[t] + q), handed out by one atomic per geometry block.
// Nothing has to be counted before it can be placed, so geometry is ONE pass.  Slots are not in emission order,
// but the z tie-break (renderer.rs:363: the later triangle wins) only needs a key that is: the ORDER KEY of a
// triangle is  32 * t  for an unclipped input and  32 * t + 1 + q  for fan triangle q -- monotone in the reference's
// emission order within the draw (2^27 input triangles per mesh).  Keys of different draws never meet: a draw starts
// from the depth buffer with id 0 = "what was there", which any fragment of equal depth beats.  The tile kernel keeps
// order key + 1  in the low half of its 64-bit pixel keys; the resolve maps a winner back to its slot and to the
// reference's emission index  tri_base + block_prefix[t / 256] + (tinfo[t] >> 5) + q.
constexpr int FAN_BITS = 5;

struct DevUniforms {
    float mvp[16];     // (proj*view)*model, phong.rs:119 hoisted (pure, so exact)
    float model[16];
    float view_pos[3];
    float light_pos[3];
    float light_color[3];
    float ambient_strength, specular_strength;
    float flat_color[4];
    const uint8_t *tex;                  // the texture in uniforms.texture_slot (PSUniform.place, phong.rs:34-38,147-151)
    uint32_t tex_w, tex_h;
    float user[FRR_MAX_USER_UNIFORMS];   // frr_set_user_uniforms: what a user shader's closure would have captured
    // every texture slot (PSUniform holds three textures, phong.rs:41-47, and a closure may sample any of them:
    // sample_2d_slot); null / 0 x 0 where nothing was uploaded
    const uint8_t *slot_tex[FRR_MAX_TEXTURES];
    uint32_t slot_w[FRR_MAX_TEXTURES], slot_h[FRR_MAX_TEXTURES];
};

struct GeomArgs {
    const float *in;        // [ntris][3][NF]
    uint32_t ntris;
    uint32_t width, height; // viewport of renderer.rs:107-108
    uint32_t fan_cap;       // capacity of the fan space (triangles): FAN_REGIONS regions of fan_cap / FAN_REGIONS (a multiple of FAN_REGIONS)
    uint32_t seq, epoch;    // sequence number of this geometry pass / failures before `epoch` are stale (Counters::first_bad)
    uint32_t frame_no;      // frr_clear count: statistics are folded per frame (GeomTab::frame_no)
    int32_t lane;           // which Counters::lane this pass's tables live in
    int32_t part_rank, part_world; // tile-row ownership filter (world == 1: none); only set by frr_draw,
    int32_t part_y0, part_y1;      // which knows the raster window's height range
    int32_t part_blocked, part_brow0, part_brow1; // blocked partition: the rank owns tile rows [brow0, brow1) (RasterArgs)
    int32_t gpar;           // parity of this pass: which Counters::gtab it fills
    uint32_t *block_sums;   // [nblocks] triangles emitted per 256-triangle block
    uint32_t *block_prefix; // [nblocks] their exclusive scan (geom_scan; a separate array: a replayed raster pass may scan again)
    uint32_t *tinfo;        // [ntris]   see Counters
    uint32_t *fanbase;      // [ntris]
    uint32_t *fan_okey;     // [fan_cap]
    RasterRec *recs;        // [ntris + fan_cap]
    float *vary;            // [ntris + fan_cap][3][K]
    uint2 *clipq;           // use_clipq: {input triangle, its first fan slot} of the clipped inputs a block does not expand itself
    int32_t use_clipq;      //   (k_geom_clip expands them, one wavefront each, balanced over the chip); 0: every block expands all of its own
    uint4 *pbox;            // the binning input, 16 bytes per entry: {minx|miny<<16, maxx|maxy<<16 (i16 pixel bbox of spi), zkey of an
                            // upper bound of |rhw|, slot}.  Inputs: block b's triangles that are set up here (unclipped, and on a
                            // partitioned ctx touching a tile row of the rank) are the first bcount[b] entries of [256 b, 256 b + 256),
                            // the rest of that range is zero (= an empty box); fan triangle at fan slot f: entry ntris + f
    uint32_t *bcount;       // [nblocks] see pbox
    Counters *cnt;
};

struct RasterArgs {
    int32_t x0, x1, y0, y1;           // width_range / height_range (renderer.rs:270-271)
    int32_t win_w, win_h;             // x1-x0, y1-y0
    int32_t cstride, dstride;         // colour row stride (fb.width), depth row stride (= x1, :362)
    int32_t tiles_x, tiles_y;
    uint32_t tiles_x_magic;           // 2^32 / tiles_x + 1: block index -> tile row by one multiplication (0: plain division)
    int32_t rank, world;              // tile-row ownership: ty % world == rank (interleaved), or ...
    int32_t blocked, brow0, brow1;    // ... blocked != 0: the rank owns the contiguous tile rows [brow0, brow1)
    const RasterRec *recs;
    const float *vary;
    const uint4 *pbox;                // see GeomArgs::pbox
    const uint32_t *bcount;           // see GeomArgs::bcount
    // the rasterized geometry pass's id tables (GeomArgs / GeomTab hold the same pointers; as kernel arguments the tile
    // kernel reads them from the constant kernarg segment whenever it needs them -- no dependent global load in front
    // of every triangle-id lookup, and nothing to keep in registers)
    const uint32_t *tinfo, *fanbase, *fan_okey, *block_prefix;
    uint32_t ntris_draw;              // input triangles of that pass (= its first fan slot)
    uint32_t *tile_counts;            // [ntiles]
    uint32_t *tile_offsets;           // [ntiles+1]
    uint32_t *tile_cursor;            // [ntiles]
    const uint32_t *seg;              // segmented binning: [nseg][ntiles+1] segment starts (+ end sentinel) into bins; nseg == 0: CSR (tile_offsets)
    uint32_t nseg;
    int32_t bpar;                     // parity of this raster pass: which Counters::btab it uses
    int32_t gpar;                     // parity of the geometry pass it rasterizes: which Counters::gtab it reads
    uint32_t seq, epoch, frame_no;    // this raster pass's sequence number etc. (Counters::first_bad, GeomTab::frame_no)
    int32_t lane;                     // which Counters::lane both tables live in
    uint32_t geom_seq;                // the geometry pass's number (its block-sum scan may ride in this pass's binning launch)
    int32_t fused_clear;              // this draw also performs the pending frr_clear for the tiles it owns:
    uint32_t clear_rgba;              //   keys start from clear_depth instead of the depth buffer and every pixel
    float clear_depth;                //   of the tile is written (full-window draws of the span kernel only)
    uint32_t ent_slot;                // segmented binning: bins2 = [ntiles][ent_slot] fixed slots + an overflow arena of bin_cap records
    uint4 *bins;                      // one 16-byte cull record {tri, zub, bbox.x, bbox.y} per (triangle, tile) pair, CSR by tile
    uint4 *bins2;                     // the same records in near-first order per tile (written by the tile kernel's pre-pass)
    uint32_t bin_cap;
    uint8_t *color;
    float *depth;
    uint32_t *tri_id;
    Counters *cnt;
#ifdef FRR_DEBUG_COUNTERS
    unsigned long long *dbg_tiles;    // [tiles][8] per-tile timeline of the tile kernel (tools/tile_timeline.py), or null
#endif
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

// does this rank rasterize tile row ty (window-local)?  interleaved rows, or the block [brow0, brow1)
struct RowOwner { int rank, world, blocked, brow0, brow1; };
__device__ __forceinline__ bool owns_tile_row(int ty, const RowOwner &o)
{
    return o.world <= 1 || (o.blocked ? (ty >= o.brow0 && ty < o.brow1) : ty % o.world == o.rank);
}

// Index of an OWNED tile row among the rank's rows (0, 1, 2 ...): the segmented binning and the tile kernel number a
// rank's tiles row-major over its own rows only, so that a rank of N keeps 1/N of the tile counters, segment-table
// columns and near-first slots.  (Rows the rank does not own may map anywhere: callers test ownership first.)
__device__ __forceinline__ int local_tile_row(int ty, const RowOwner &o)
{
    if (o.world <= 1) return ty;
    if (o.blocked) return ty - o.brow0;
    return (int)(((float)ty + 0.5f) * (1.0f / (float)o.world)); // ty / world: exact for tile rows (< 2^11)
}

// The slots of the current draw as ONE virtual index range [0, total): the inputs' own slots, then the used part of fan
// region 0, of region 1, ...  (what the binning kernels walk and split into chunks).
struct FanMap { uint32_t ntris, region, pre[FAN_REGIONS + 1]; };   // pre[k]: used fan slots of the regions before k
__device__ __forceinline__ FanMap fan_map(const GeomTab *gt, uint32_t fan_cap)
{
    FanMap m;
    m.ntris = gt->ntris_draw;
    m.region = fan_cap / FAN_REGIONS;
    m.pre[0] = 0u;
#pragma unroll
    for (int k = 0; k < FAN_REGIONS; ++k) m.pre[k + 1] = m.pre[k] + min(gt->fan_cursor[k].v, m.region);
    return m;
}
__device__ __forceinline__ uint32_t fan_map_total(const FanMap &m) { return m.ntris + m.pre[FAN_REGIONS]; }
__device__ __forceinline__ uint32_t fan_map_slot(const FanMap &m, uint32_t v)
{
    if (v < m.ntris) return v;
    const uint32_t u = v - m.ntris;
    uint32_t k = 0;
#pragma unroll
    for (int j = 1; j < FAN_REGIONS; ++j) k += u >= m.pre[j] ? 1u : 0u;
    uint32_t pk = 0;
#pragma unroll
    for (int j = 1; j < FAN_REGIONS; ++j) pk = k >= (uint32_t)j ? m.pre[j] : pk;
    return m.ntris + k * m.region + (u - pk);
}

// Frame statistics are kept per pass (GeomTab / BinTab) and folded into the totals when a table is recycled; a table
// or a total of an older frame (frr_clear count) is simply dropped -- frr_clear itself touches nothing on the device.
__device__ __forceinline__ void totals_for_frame(Counters *cnt, Lane &L, uint32_t frame_no)
{
    if (L.totals_frame != frame_no) {
        L.totals_frame = frame_no;
        L.tot_frag_covered = 0ull; L.tot_frag_nan = 0ull; L.tot_bin_entries = 0ull;
        for (int j = 0; j < DBG_COPIES; ++j) for (int k = 0; k < 24; ++k) cnt->dbg[j][k] = 0;
    }
}

// the binning launch's bookkeeping (one thread).  The OTHER table belongs to the previous raster pass, whose binning has
// drained (same stream): its total is folded and zeroed for the pass after this one.  (Its ent_cursor may still be in use
// by that pass's tile kernel on the other stream: every pass zeroes its OWN before its tile kernel starts.)
__device__ __forceinline__ void bin_bookkeeping(Counters *cnt, int lane, int bpar, uint32_t frame_no)
{
    Lane &L = cnt->lane[lane];
    totals_for_frame(cnt, L, frame_no);
    BinTab &o = L.btab[bpar ^ 1];
    if (o.frame_no == frame_no) L.tot_bin_entries += o.seg_total;
    o.seg_total = 0ull;
    o.frame_no = SEQ_NONE;
    L.btab[bpar].ent_cursor = 0u;
    L.btab[bpar].frame_no = frame_no;
}

// ---- glam pieces used by the shader table -------------------------------------
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz)
{
    return (ax * bx + ay * by) + az * bz;
}
__device__ __forceinline__ void normalize3(float &x, float &y, float &z)
{
    float r = rsqrt_exact(dot3(x, y, z, x, y, z)); // Vec3::normalize = self * length_recip(); == 1.0f / sqrtf, bit for bit
    x = x * r; y = y * r; z = z * r;
}
__device__ __forceinline__ void mat4_mul_vec4(const float *m, float x, float y, float z, float w, float o[4])
{
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = ((m[r] * x + m[4 + r] * y) + m[8 + r] * z) + m[12 + r] * w;
}

// ---- vertex shader table (contract renderer.rs:105,116) -----------------------------------
template <int VS> struct VSInfo;
template <> struct VSInfo<FRR_VS_CLIP> { static constexpr int NF = 4, K = 0; };
template <> struct VSInfo<FRR_VS_CLIP_COLOR> { static constexpr int NF = 7, K = 3; };
template <> struct VSInfo<FRR_VS_PHONG> { static constexpr int NF = 8, K = 8; };
template <> struct VSInfo<FRR_VS_GOURAUD> { static constexpr int NF = 8, K = 3; };

} // namespace frr
#ifdef FRR_USER_SHADER
// User shaders (frr_shader_register): the program text hiprtc compiles is  this header, the user's source, the kernels.
// The user's source defines these two functions -- the bodies of the reference's vertex_shader / pixel_shader closures
// (renderer.rs:105,110 / :273,283) -- with everything above (glam pieces, frr_exact.h) and sample_2d below to build on.
__device__ void frr_user_vs(const frr::DevUniforms &u, const float *in, float pos[4], float *ctx);
__device__ void frr_user_ps(const frr::DevUniforms &u, const float *ctx, float out[4], const float *u8lut);
#endif
namespace frr {
#ifdef FRR_USER_SHADER
template <> struct VSInfo<FRR_SHADER_USER_BASE> { static constexpr int NF = FRR_USER_NF, K = FRR_USER_K; };
#endif

template <int VS, bool WITH_CTX>
__device__ __forceinline__ void run_vs(const DevUniforms &u, const float *__restrict__ in, float pos[4], float *ctx)
{
#ifdef FRR_USER_SHADER
    if constexpr (VS >= FRR_SHADER_USER_BASE) {
        float c[VSInfo<VS>::K > 0 ? VSInfo<VS>::K : 1];
#pragma unroll
        for (int k = 0; k < VSInfo<VS>::K; ++k) c[k] = 0.0f;       // T::default() (renderer.rs:113)
        frr_user_vs(u, in, pos, c);
        if constexpr (WITH_CTX) {
#pragma unroll
            for (int k = 0; k < VSInfo<VS>::K; ++k) ctx[k] = c[k];
        }
    } else
#endif
    if constexpr (VS == FRR_VS_CLIP) {
        float4 v = *reinterpret_cast<const float4 *>(in);
        pos[0] = v.x; pos[1] = v.y; pos[2] = v.z; pos[3] = v.w;
    } else if constexpr (VS == FRR_VS_CLIP_COLOR) {
        pos[0] = in[0]; pos[1] = in[1]; pos[2] = in[2]; pos[3] = in[3];
        if constexpr (WITH_CTX) { ctx[0] = in[4]; ctx[1] = in[5]; ctx[2] = in[6]; }
    } else if constexpr (VS == FRR_VS_PHONG) {
        float4 a = *reinterpret_cast<const float4 *>(in);      // pos.xyz, uv.x
        float4 b = *reinterpret_cast<const float4 *>(in + 4);  // uv.y, normal.xyz
        if constexpr (WITH_CTX) {
            float w[4];
            ctx[0] = a.w; ctx[1] = b.x;                        // phong.rs:120
            ctx[2] = b.y; ctx[3] = b.z; ctx[4] = b.w;          // phong.rs:121-122
            mat4_mul_vec4(u.model, a.x, a.y, a.z, 1.0f, w);    // phong.rs:123-124
            ctx[5] = w[0]; ctx[6] = w[1]; ctx[7] = w[2];
        }
        mat4_mul_vec4(u.mvp, a.x, a.y, a.z, 1.0f, pos);        // phong.rs:125
    } else { // FRR_VS_GOURAUD: per-vertex Lambert (no reference arithmetic; mirrors the oracle)
        float4 a = *reinterpret_cast<const float4 *>(in);
        float4 b = *reinterpret_cast<const float4 *>(in + 4);
        if constexpr (WITH_CTX) {
            float w[4];
            mat4_mul_vec4(u.model, a.x, a.y, a.z, 1.0f, w);
            float nx = b.y, ny = b.z, nz = b.w;
            normalize3(nx, ny, nz);
            float lx = u.light_pos[0] - w[0], ly = u.light_pos[1] - w[1], lz = u.light_pos[2] - w[2];
            normalize3(lx, ly, lz);
            float diff = f32_max(dot3(nx, ny, nz, lx, ly, lz), 0.0f);
#pragma unroll
            for (int k = 0; k < 3; ++k)
                ctx[k] = u.light_color[k] * u.ambient_strength + diff * u.light_color[k];
        }
        mat4_mul_vec4(u.mvp, a.x, a.y, a.z, 1.0f, pos);
    }
}

// ---- clip classification (renderer.rs:46-73, 123-174) -------------------------------------
// plane order [X_LEFT, X_RIGHT, Y_UP, Y_DOWN, Z_NEAR, Z_FAR] (renderer.rs:123-131)
__device__ __forceinline__ uint32_t inside_bits(const float p[4])
{
    float w = p[3];
    uint32_t b = 0;
    b |= (p[0] >= -w) ? 1u : 0u;
    b |= (p[0] <= w) ? 2u : 0u;
    b |= (p[1] <= w) ? 4u : 0u;
    b |= (p[1] >= -w) ? 8u : 0u;
    b |= (p[2] >= 0.0f) ? 16u : 0u;
    b |= (p[2] <= w) ? 32u : 0u;
    return b;
}
__device__ __forceinline__ float intersect_ratio(int plane, const float a[4], const float b[4])
{
    float a_w = a[3], b_w = b[3];
    switch (plane) {
    case 0: return -(a[0] + a_w) / (b_w + b[0] - a[0] - a_w);   // X_LEFT  :65
    case 1: return (a_w - a[0]) / (a_w - b_w - a[0] + b[0]);    // X_RIGHT :66
    case 2: return (a_w - a[1]) / (a_w - b_w - a[1] + b[1]);    // Y_UP    :67
    case 3: return -(a[1] + a_w) / (b_w + b[1] - a_w - a[1]);   // Y_DOWN  :68
    case 4: return a_w / (a_w - b_w);                           // Z_NEAR  :70 (sic)
    default: return (a_w - a[2]) / (a_w - b_w - a[2] + b[2]);   // Z_FAR   :69
    }
}
constexpr float CLIP_EPSILON = 1.0e-5f; // renderer.rs:44

// Number of triangles geometry_processing returns for clip positions pos[3][4]:
// 0 (None, :117-119), 1 (all inside), else n-2 with n = 3 + kept intersections (:150-171).
// `clipped` tells the caller the slow path is needed (at least one intersection vertex was kept).
__device__ __forceinline__ uint32_t classify(const float pos[3][4], bool &clipped)
{
    clipped = false;
    if (pos[0][3] == 0.0f || pos[1][3] == 0.0f || pos[2][3] == 0.0f) return 0;
    uint32_t in0 = inside_bits(pos[0]), in1 = inside_bits(pos[1]), in2 = inside_bits(pos[2]);
    if ((in0 & in1 & in2) == 63u) return 1;
    clipped = true;
    uint32_t n = 3;
    // pairs (0,1),(0,2),(1,2) x planes in list order (:152-169); spelled out so that every index is
    // a compile-time constant (runtime-indexed arrays would go to scratch)
    auto pair = [&](const float *a, const float *b, uint32_t diff) {
#pragma unroll
        for (int p = 0; p < 6; ++p)
            if (diff & (1u << p)) {
                float t = intersect_ratio(p, a, b);
                float w = a[3] + t * (b[3] - a[3]);                 // :89, w lane
                if (fabsf(w) > CLIP_EPSILON) ++n;                   // :164
            }
    };
    pair(pos[0], pos[1], in0 ^ in1);
    pair(pos[0], pos[2], in0 ^ in2);
    pair(pos[1], pos[2], in1 ^ in2);
    // No intersection kept (e.g. all three vertices beyond the same plane): the reference's vertex list is
    // just the three originals, exactly the all-inside case -- one code path there, the fast path here.
    clipped = n != 3u;
    return n - 2;
}

// renderer.rs:220-235 for one vertex
struct ScreenVtx { float rhw, ndcx, ndcy, sx, sy; int32_t ix, iy; };
__device__ __forceinline__ ScreenVtx to_screen(const float pos[4], float fw, float fh)
{
    ScreenVtx v;
    v.rhw = recip_exact(pos[3]);                 // == 1.0f / w, bit for bit (frr_exact.h; checked for all 2^32 operands on the device)
    v.ndcx = pos[0] * v.rhw;
    v.ndcy = pos[1] * v.rhw;
    v.sx = (v.ndcx + 1.0f) * fw * 0.5f;
    v.sy = (1.0f - v.ndcy) * fh * 0.5f;
    v.ix = f32_as_i32(v.sx + 0.5f);
    v.iy = f32_as_i32(v.sy + 0.5f);
    return v;
}
// Multi-GPU: does a (not clipped) triangle with snapped corner rows iy0..iy2 touch a tile row this rank owns?
// (also asked, with the rows widened by one, for the fan of a clipped input: k_geom_single)
__device__ __forceinline__ bool tri_rows_owned(const GeomArgs &g, int iy0, int iy1, int iy2)
{
    if (g.part_world <= 1) return true;
    const int miny = clampi(min(iy0, min(iy1, iy2)), g.part_y0, g.part_y1);
    const int maxy = clampi(max(iy0, max(iy1, iy2)), g.part_y0, g.part_y1);
    if (maxy <= miny) return false;
    const int ty0 = (miny - g.part_y0) / TILE, ty1 = (maxy - 1 - g.part_y0) / TILE;
    if (g.part_blocked) return ty0 < g.part_brow1 && ty1 >= g.part_brow0;
    const int first = ty0 + ((g.part_rank - ty0 % g.part_world) + g.part_world) % g.part_world;
    return first <= ty1;
}
// renderer.rs:26-29
__device__ __forceinline__ bool is_top_left(int ax, int ay, int bx, int by)
{
    return ((ay == by) && (ax < bx)) || (ay > by);
}

// The tile-independent half of the span set-up of one edge a -> b (renderer.rs:314-320, 329-341):  A = -(b.y - a.y),
// B = b.x - a.x,  E(cx, cy) = A (cx - a.x) + B (cy - a.y), accepted where E >= bias (bias 0 for top-left edges, else 1).
// Row `row` of a bbox-in-tile with origin (bx0, by0) is covered where A dx >= bias - E(bx0, by0 + row); the tile kernel
// evaluates that as  M = m + k row,  q = floor(M / D)  with  m = c - D bx0 + k by0  (frr_raster.h: SpanTri):
//   A > 0:  dx >= floor(M / D)     k = -B   D = A    c = bias + A a.x + B a.y + A - 1
//   A < 0:  dx <= floor(M / D)     k = B    D = -A   c = -bias - A a.x - B a.y
//   A = 0:  all dx iff M > 0       k = 2B   D = 0    c = 1 - 2 bias - 2 B a.y     (M is odd: never 0)
// kd = k (low 16 bits, signed) | D << 16.  Exact for coordinates within +-8191 (every product below 2^28); outside that
// range the words mean nothing and nobody reads them (such triangles take the brute-force sweep).  Wrapping arithmetic.
struct EdgeWords { uint32_t kd, c, pos; };
__device__ __forceinline__ EdgeWords edge_words(int ax, int ay, int bx, int by, uint32_t bias)
{
    const uint32_t A = 0u - (uint32_t)(by - ay), B = (uint32_t)(bx - ax);
    const int Ai = (int)A;
    const uint32_t D = Ai > 0 ? A : 0u - A;
    const uint32_t axby = A * (uint32_t)ax + B * (uint32_t)ay;
    EdgeWords w;
    const uint32_t k = Ai > 0 ? 0u - B : (Ai < 0 ? B : 2u * B);
    w.c = Ai > 0 ? bias + axby + D - 1u : (Ai < 0 ? 0u - bias - axby : 1u - 2u * bias - 2u * (B * (uint32_t)ay));
    w.kd = (k & 0xFFFFu) | (D << 16);
    w.pos = Ai > 0 ? 1u : 0u;
    return w;
}

// Pixel bbox of a setup triangle for the binning passes: 4 x i16, saturated (the raster window is
// required to lie within the i16 range, so saturating first and clamping to the window later is the
// same as renderer.rs:285-298's clamp of the i32 values).
__device__ __forceinline__ uint2 pack_pbox(int x0, int y0, int x1, int y1, int x2, int y2)
{
    auto s16 = [](int v) { return (uint32_t)min(max(v, -32768), 32767) & 0xFFFFu; };
    const int mnx = min(x0, min(x1, x2)), mny = min(y0, min(y1, y2)), mxx = max(x0, max(x1, x2)), mxy = max(y0, max(y1, y2));
    return make_uint2(s16(mnx) | (s16(mny) << 16), s16(mxx) | (s16(mxy) << 16));
}

// Whole-triangle early-z bound: rhw = (r0*a + r1*b) + r2*c with a+b+c = 1 up to a few roundings, so
// |rhw| <= max|r_i| * (1 + 2^-18); the z key of that bound, or all-ones (never culled) for NaN vertices.
__device__ __forceinline__ uint32_t cull_zub(float r0, float r1, float r2)
{
    const float ar0 = fabsf(r0), ar1 = fabsf(r1), ar2 = fabsf(r2);
    const float ub = fmaxf(fmaxf(ar0, ar1), ar2) * 1.000003814697265625f;
    return (ar0 == ar0 && ar1 == ar1 && ar2 == ar2) ? zkey(ub) : 0xFFFFFFFFu;
}

// ---- fragment arithmetic (renderer.rs:343-360), shared by the coverage loop and the resolve --
struct Frag { float a, b, c, rhw; bool valid; };
__device__ __forceinline__ Frag frag_eval(float s0x_, float s0y_, float s1x_, float s1y_, float s2x_, float s2y_,
                                          float r0, float r1, float r2, int cx, int cy)
{
    Frag f;
    float pxx = (float)cx + 0.5f, pxy = (float)cy + 0.5f;          // :325
    float s0x = s0x_ - pxx, s0y = s0y_ - pxy;                      // :343-345
    float s1x = s1x_ - pxx, s1y = s1y_ - pxy;
    float s2x = s2x_ - pxx, s2y = s2y_ - pxy;
    float a = fabsf(s1x * s2y - s1y * s2x);                        // :347-349
    float b = fabsf(s2x * s0y - s2y * s0x);
    float c = fabsf(s0x * s1y - s0y * s1x);
    float s = a + b + c;                                           // :351
    f.valid = !(s == 0.0f);                                        // :352-354
    float inv = recip_exact(s);                                    // :356-358 (== 1.0f / s, bit for bit)
    f.a = a * inv; f.b = b * inv; f.c = c * inv;
    f.rhw = r0 * f.a + r1 * f.b + r2 * f.c;                        // :360
    return f;
}

// ---- pixel shader table (contract renderer.rs:283,380) ----------------------------------
// FrameBuffer::sample_2d renderer.rs:516-538 (+ get_pixel :505-514, u8_array_to_vec4 :16-24)
// u8lut: optional 256-entry table of (float)i / 255.0f (the tile kernel keeps one in LDS: the 16 IEEE
// divisions of a bilinear sample become 16 table reads of the very same quotients)
__device__ __forceinline__ void sample_2d_of(const uint8_t *tex, uint32_t tex_w, uint32_t tex_h, float uu, float vv, float out[4], const float *u8lut = nullptr)
{
    float x = uu * (float)tex_w;
    float y = vv * (float)tex_h;
    float a = x - truncf(x);
    float b = y - truncf(y);
    uint32_t wm1 = tex_w - 1u;
    uint32_t x1 = min(f32_as_u32(x), wm1);
    uint32_t y1 = min(f32_as_u32(y), wm1); // sic: width (:523)
    uint32_t x2 = min(x1 + 1u, wm1);
    uint32_t y2 = min(y1 + 1u, wm1);       // sic: width (:525)
    const uchar4 *t = reinterpret_cast<const uchar4 *>(tex);
    uchar4 q11 = t[y1 * tex_w + x1], q12 = t[y2 * tex_w + x1];
    uchar4 q21 = t[y1 * tex_w + x2], q22 = t[y2 * tex_w + x2];
    float oma = 1.0f - a, omb = 1.0f - b;
    const uint8_t *p11 = reinterpret_cast<const uint8_t *>(&q11), *p12 = reinterpret_cast<const uint8_t *>(&q12);
    const uint8_t *p21 = reinterpret_cast<const uint8_t *>(&q21), *p22 = reinterpret_cast<const uint8_t *>(&q22);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float t11, t12, t21, t22;
        if (u8lut) { // wave-uniform: one branch, not sixteen selects that would evaluate the divisions anyway
            t11 = u8lut[p11[k]]; t12 = u8lut[p12[k]]; t21 = u8lut[p21[k]]; t22 = u8lut[p22[k]];
        } else {
            t11 = (float)p11[k] / 255.0f; t12 = (float)p12[k] / 255.0f; t21 = (float)p21[k] / 255.0f; t22 = (float)p22[k] / 255.0f;
        }
        float c11 = t11 * oma * omb;
        float c12 = t12 * oma * b;
        float c21 = t21 * a * omb;
        float c22 = t22 * a * b;
        out[k] = c11 + c12 + c21 + c22;
    }
}

// the texture in uniforms.texture_slot ...
__device__ __forceinline__ void sample_2d(const DevUniforms &u, float uu, float vv, float out[4], const float *u8lut = nullptr)
{
    sample_2d_of(u.tex, u.tex_w, u.tex_h, uu, vv, out, u8lut);
}
// ... or any slot by number (user shaders: a closure over PSUniform may sample each of its textures, phong.rs:41-47,147-151);
// an empty slot samples as zero
__device__ __forceinline__ void sample_2d_slot(const DevUniforms &u, int slot, float uu, float vv, float out[4], const float *u8lut = nullptr)
{
    if (slot < 0 || slot >= FRR_MAX_TEXTURES || !u.slot_tex[slot]) { out[0] = out[1] = out[2] = out[3] = 0.0f; return; }
    sample_2d_of(u.slot_tex[slot], u.slot_w[slot], u.slot_h[slot], uu, vv, out, u8lut);
}

template <int PS>
__device__ __forceinline__ void run_ps(const DevUniforms &u, const float *ctx, float out[4], const float *u8lut = nullptr)
{
#ifdef FRR_USER_SHADER
    if constexpr (PS >= FRR_SHADER_USER_BASE) {
        frr_user_ps(u, ctx, out, u8lut);
    } else
#endif
    if constexpr (PS == FRR_PS_FLAT) {
        out[0] = u.flat_color[0]; out[1] = u.flat_color[1]; out[2] = u.flat_color[2]; out[3] = u.flat_color[3];
    } else if constexpr (PS == FRR_PS_COLOR) {
        out[0] = ctx[0]; out[1] = ctx[1]; out[2] = ctx[2]; out[3] = 1.0f;
    } else if constexpr (PS == FRR_PS_PHONG || PS == FRR_PS_BLINN) {
        float amb[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) amb[k] = u.light_color[k] * u.ambient_strength;       // phong.rs:134
        float nx = ctx[2], ny = ctx[3], nz = ctx[4];
        normalize3(nx, ny, nz);                                                           // :136
        float lx = u.light_pos[0] - ctx[5], ly = u.light_pos[1] - ctx[6], lz = u.light_pos[2] - ctx[7];
        normalize3(lx, ly, lz);                                                           // :137
        float diff = f32_max(dot3(nx, ny, nz, lx, ly, lz), 0.0f);                         // :138
        float vx = u.view_pos[0] - ctx[5], vy = u.view_pos[1] - ctx[6], vz = u.view_pos[2] - ctx[7];
        normalize3(vx, vy, vz);                                                           // :141
        float s;
        if constexpr (PS == FRR_PS_PHONG) {
            float Lx = -lx, Ly = -ly, Lz = -lz;                                           // :142
            float t = 2.0f * dot3(Lx, Ly, Lz, nx, ny, nz);                                // vector_util.rs:6
            float rx = t * nx - Lx, ry = t * ny - Ly, rz = t * nz - Lz;
            normalize3(rx, ry, rz);
            s = f32_max(dot3(vx, vy, vz, rx, ry, rz), 0.0f);                              // :143
        } else {
            float hx = lx + vx, hy = ly + vy, hz = lz + vz;
            normalize3(hx, hy, hz);
            s = f32_max(dot3(nx, ny, nz, hx, hy, hz), 0.0f);
        }
        s = s * s; s = s * s; s = s * s; s = s * s; s = s * s;                            // powi(32)
        float tex[4];
        sample_2d(u, ctx[0], ctx[1], tex, u8lut);                                         // :146-151
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float diffuse = diff * u.light_color[k];                                      // :139
            float spec = u.specular_strength * s * u.light_color[k];                      // :144
            out[k] = tex[k] * (amb[k] + diffuse + spec);                                  // :153
        }
        out[3] = tex[3] * 1.0f;
    } else {
        out[0] = out[1] = out[2] = out[3] = 0.0f;
    }
}

} // namespace frr
