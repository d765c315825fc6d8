// frr_api.hip -- host side of libfrr_hip.so: the C ABI of include/frr.h over the kernels in
// frr_kernels.h.  gfx950 only; no CPU fallback (every compute entry point needs the device).
#include "frr_kernels.h"

#include <math.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include <string>
#include <vector>

using namespace frr;

namespace {

enum KernelId { KID_CLEAR, KID_GEOM, KID_GEOM_SCAN, KID_BIN_COUNT,
                KID_TILE_SCAN, KID_BIN_FILL, KID_RASTER, KID_BIN_SEG, KID_COUNT };
const char *const kKernelNames[KID_COUNT] = {"k_clear", "k_geom", "k_geom_scan",
                                             "k_bin_count", "k_tile_scan", "k_bin_fill",
                                             "k_raster", "k_bin_seg"};

struct Mesh {
    const float *dev = nullptr;
    bool owned = false, used = false;
    uint64_t ntris = 0;
    int vs = 0;
};
struct Texture {
    uint8_t *dev = nullptr;
    uint32_t w = 0, h = 0;
};
struct ProfRec { int kid; hipEvent_t a, b; };

} // namespace

struct frr_ctx {
    int device = 0;
    uint32_t W = 0, H = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    uint8_t *color = nullptr, *own_color = nullptr;
    float *depth = nullptr, *own_depth = nullptr;
    uint32_t *tri_id = nullptr, *own_tri_id = nullptr;
    Counters *cnt = nullptr;
    // geometry workspace (slots and order keys: frr_device.h)
    uint32_t *block_sums = nullptr; size_t block_sums_cap = 0; // per 256-triangle block: triangles emitted; scanned in place
    uint32_t *tinfo = nullptr; size_t tinfo_cap = 0;           // per input: fan size | emission offset in its block
    uint32_t *fanbase = nullptr; size_t fanbase_cap = 0;       // per clipped input: first fan slot
    uint32_t *fan_okey = nullptr; size_t fan_okey_cap = 0;     // per fan slot: order key within the draw
    struct GeomFilter { bool active; int32_t y0, y1; int rank, world; bool blocked; };
    GeomFilter geom_filter = {false, 0, 0, 0, 1, false}; // tile-row ownership filter the latest setup list was built with (frr_draw on a partitioned ctx)
    int geom_slot = 0;         // Counters::fan_cursor slot of the latest draw (alternates per draw)
    uint32_t geom_fan_cap = 0; // fan capacity the latest draw was launched with
    uint32_t geom_nblocks = 0;
    bool scan_pending = false; // the latest draw's block sums are not scanned yet (geom_scan: by the binning launch, or k_geom_scan)
    size_t fan_hint = 0;       // fan capacity asked for by a draw that overflowed
    int bin_g = 0;             // option bin_chunks: override the number of binning chunks (dev)
    uint32_t ent_slot_override = 0; // option tile_slot_records: per-tile slot of bins2 in records (tests of the overflow arena)
    // frr_clear is deferred: the first full-window draw of the span kernel performs it inside the tile kernel
    // (keys start from the clear depth, every pixel of the tile is written); anything else that looks at the
    // targets or the counters first settles it with k_clear.  option clear_eager restores the immediate clear.
    bool part_blocked = false;     // frr_set_partition_layout: contiguous blocks of tile rows instead of interleaved rows
    bool clear_eager = false;
    bool clear_pending = false;    // targets not cleared yet
    bool counters_pending = false; // frame counters not reset yet (the next draw's bookkeeping thread does it)
    bool unowned_debt = false;     // partitioned ctx: the tile rows of other ranks missed a fused clear
    int debt_rpr = 0;              //   (RasterArgs::rpr of the draw that left the debt)
    uint32_t clear_rgba = 0; float clear_depth = 0.0f;
    int bin_slot = 0;          // Counters::seg_total / ent_cursor slot of the latest draw (alternates)
    bool bin_atomics = false;  // option bin_atomics: force the global-atomic binning fallback (tests)
    size_t bin_cap_init = 0;   // option bin_capacity: initial bin capacity in entries (tests of the overflow path)
    RasterRec *recs = nullptr; size_t setup_cap = 0;            // [input triangles + fan capacity]
    float *vary = nullptr; size_t vary_cap = 0; // floats
    uint4 *pbox = nullptr; size_t pbox_cap = 0;
    uint32_t *bcount = nullptr; size_t bcount_cap = 0;          // [geometry blocks] dense binning entries per block (GeomArgs::bcount)
    uint2 *clipq = nullptr; size_t clipq_cap = 0;               // [input triangles] the clip kernel's queue (GeomArgs::clipq)
    int clip_queue = -1;        // option clip_queue: 1 use the queue + k_geom_clip, 0 never, -1 when the latest counters read back
    bool clip_queue_auto = false; //   showed a block with more than CLIP_QUEUE_AT clipped inputs (results are the same either way)
    // binning workspace
    uint32_t *tile_counts = nullptr, *tile_offsets = nullptr, *tile_cursor = nullptr;
    uint32_t max_tiles = 0;
    uint4 *bins = nullptr; size_t bin_cap = 0;   // 16-byte cull records, one per (triangle, tile) pair
    uint4 *bins2 = nullptr; size_t bin2_cap = 0; // the same in near-first order per tile (tile kernel pre-pass)
    uint32_t *bin_matrix = nullptr; size_t bin_matrix_cap = 0; // [G][ntiles] per-chunk tile histograms
    bool lds_attr_set = false;
    std::vector<Mesh> meshes;
    Texture tex[FRR_MAX_TEXTURES];
    frr_uniforms uni;
    DevUniforms duni;
    int geom_vs = -1;         // VS of the last frr_geometry
    uint64_t geom_ntris = 0;
    int rank = 0, world = 1;
    bool count_frags = true;   // exact covered-fragment statistic (disables whole-triangle early-z)
    int raster_nw = 0;         // option raster_nw: force 3 / 4 / 6 / 8 / 16 waves per tile workgroup (dev)
    int raster_occ = 0;        // option raster_occ: force the 6- or 8-waves-per-SIMD build of the tile kernel (dev)
    bool raster_sweep = false; // option raster_sweep: brute-force tile kernel instead of the span kernel
#ifdef FRR_DEBUG_COUNTERS
    unsigned long long *dbg_tiles = nullptr; // FRR_DEBUG_TILES: per-tile timeline of the latest tile kernel
#endif
    hipEvent_t ev[16] = {};
    bool ev_set[16] = {};
    uint32_t prof_mask = 0;   // bit per KernelId
    std::vector<ProfRec> prof_pending;
    std::vector<hipEvent_t> ev_pool;
    uint32_t prof_period = 1;         // bracket only every prof_period-th launch of a kernel (frr_profile_set_period)
    uint32_t prof_seen[KID_COUNT] = {};
    double prof_ms[KID_COUNT] = {};
    uint32_t prof_n[KID_COUNT] = {};
    std::string err;
};

namespace {

int fail(frr_ctx *c, int code, const std::string &msg)
{
    if (c) c->err = msg;
    return code;
}
#define HIP_TRY(c, expr)                                                                                    \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess)                                                                               \
            return fail(c, FRR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                 \
    } while (0)

template <typename T> int ensure(frr_ctx *c, T *&p, size_t &cap, size_t need)
{
    if (need <= cap && p) return FRR_OK;
    if (p) { HIP_TRY(c, hipStreamSynchronize(c->stream)); HIP_TRY(c, hipFree(p)); p = nullptr; cap = 0; }
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, need * sizeof(T));
    if (e != hipSuccess) return fail(c, FRR_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    p = (T *)q;
    cap = need;
    return FRR_OK;
}

hipEvent_t get_event(frr_ctx *c)
{
    if (!c->ev_pool.empty()) { hipEvent_t e = c->ev_pool.back(); c->ev_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;   // (the caller drops the sample)
    return e;
}
struct ProfScope {
    frr_ctx *c; int kid; hipEvent_t a = nullptr;
    ProfScope(frr_ctx *c_, int kid_) : c(c_), kid(kid_)
    {
        if ((c->prof_mask & (1u << kid)) && (c->prof_seen[kid]++ % c->prof_period) == 0) {
            a = get_event(c);
            if (a && hipEventRecord(a, c->stream) != hipSuccess) { c->ev_pool.push_back(a); a = nullptr; }
        }
    }
    ~ProfScope()
    {
        // a sample whose events could not be created or recorded is dropped (frr_profile_get then reports fewer launches)
        if (!a) return;
        hipEvent_t b = get_event(c);
        if (b && hipEventRecord(b, c->stream) == hipSuccess) { c->prof_pending.push_back({kid, a, b}); return; }
        c->ev_pool.push_back(a);
        if (b) c->ev_pool.push_back(b);
    }
};
void prof_collect(frr_ctx *c)
{
    if (c->prof_pending.empty()) return;
    (void)hipStreamSynchronize(c->stream);
    for (auto &r : c->prof_pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { c->prof_ms[r.kid] += ms; c->prof_n[r.kid]++; }
        c->ev_pool.push_back(r.a); c->ev_pool.push_back(r.b);
    }
    c->prof_pending.clear();
}

// glam Mat4*Mat4 = columns (self * rhs.col_j), Mat4*Vec4 = ((c0*x + c1*y) + c2*z) + c3*w
void h_mat4_mul(const float *a, const float *b, float *out)
{
    float t[16];
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r)
            t[4 * c + r] = ((a[r] * b[4 * c] + a[4 + r] * b[4 * c + 1]) + a[8 + r] * b[4 * c + 2]) + a[12 + r] * b[4 * c + 3];
    memcpy(out, t, sizeof t);
}

void refresh_dev_uniforms(frr_ctx *c)
{
    DevUniforms &d = c->duni;
    float pv[16];
    h_mat4_mul(c->uni.proj, c->uni.view, pv);   // proj * view * model is left-associative (phong.rs:119)
    h_mat4_mul(pv, c->uni.model, d.mvp);
    memcpy(d.model, c->uni.model, sizeof d.model);
    memcpy(d.view_pos, c->uni.view_pos, sizeof d.view_pos);
    memcpy(d.light_pos, c->uni.light_pos, sizeof d.light_pos);
    memcpy(d.light_color, c->uni.light_color, sizeof d.light_color);
    d.ambient_strength = c->uni.ambient_strength;
    d.specular_strength = c->uni.specular_strength;
    memcpy(d.flat_color, c->uni.flat_color, sizeof d.flat_color);
    int s = c->uni.texture_slot;
    if (s >= 0 && s < FRR_MAX_TEXTURES) { d.tex = c->tex[s].dev; d.tex_w = c->tex[s].w; d.tex_h = c->tex[s].h; }
    else { d.tex = nullptr; d.tex_w = d.tex_h = 0; }
}

constexpr uint32_t kClipGrid = 2048;      // workgroups of k_geom_clip (four wavefronts each, one triangle per wavefront and step)
int check_frame_counters(frr_ctx *c, Counters *host)
{
    Counters h;
    HIP_TRY(c, hipMemcpyAsync(&h, c->cnt, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (host) *host = h;
    c->clip_queue_auto = std::max(h.clip_block_max[0], h.clip_block_max[1]) > (uint32_t)CLIP_QUEUE_AT;
    if (h.overflow) {
        // grow what overflowed so that re-issuing the frame succeeds
        if (h.overflow & 2u) {
            const uint64_t worst = std::max<uint64_t>(h.bin_total, std::max<uint64_t>(h.seg_total[0], h.seg_total[1]));
            size_t need = (size_t)(worst + worst / 4 + 1024);
            if (ensure(c, c->bins, c->bin_cap, need) != FRR_OK) return FRR_ERR_NOMEM;
            if (ensure(c, c->bins2, c->bin2_cap, need) != FRR_OK) return FRR_ERR_NOMEM;
        }
        if (h.overflow & 1u) c->fan_hint = (size_t)h.need_fans + h.need_fans / 8 + 1024;
        return fail(c, FRR_ERR_CAPACITY, "device work list overflowed; capacity grown, re-issue the frame");
    }
    return FRR_OK;
}

template <int VS> void launch_geometry(frr_ctx *c, GeomArgs &g, uint32_t nblocks)
{
    ProfScope p(c, KID_GEOM);
    hipLaunchKernelGGL(k_geom_single<VS>, dim3(nblocks), dim3(GEOM_BLOCK), 0, c->stream, g, c->duni);
    if (g.use_clipq) hipLaunchKernelGGL(k_geom_clip<VS>, dim3(std::min<uint32_t>(kClipGrid, nblocks * 4u)), dim3(GEOM_BLOCK), 0, c->stream, g, c->duni);
}

// the geometry kernel for the VS of the mesh
void launch_geometry_vs(frr_ctx *c, GeomArgs &g, uint32_t nblocks, int vs)
{
    switch (vs) {
    case FRR_VS_CLIP: launch_geometry<FRR_VS_CLIP>(c, g, nblocks); break;
    case FRR_VS_CLIP_COLOR: launch_geometry<FRR_VS_CLIP_COLOR>(c, g, nblocks); break;
    case FRR_VS_PHONG: launch_geometry<FRR_VS_PHONG>(c, g, nblocks); break;
    case FRR_VS_GOURAUD: launch_geometry<FRR_VS_GOURAUD>(c, g, nblocks); break;
    }
}
// the latest draw's block sums -> prefix (+ n_emit, the fan-capacity flag), if no binning launch has done it
int scan_now(frr_ctx *c)
{
    if (!c->scan_pending) return FRR_OK;
    { ProfScope p(c, KID_GEOM_SCAN); hipLaunchKernelGGL(k_geom_scan, dim3(1), dim3(1024), 0, c->stream, c->block_sums, c->geom_nblocks, c->cnt, c->geom_slot, c->geom_fan_cap); }
    HIP_TRY(c, hipGetLastError());
    c->scan_pending = false;
    return FRR_OK;
}

// Shape of the tile kernel's workgroups for `grid` tiles: NW waves per tile (and the waves per SIMD its registers are
// budgeted for).  Measured on the 1080p / 4096^2 / 4K frames (profiles/, tools/exp_shapes.py): a CU is issue-bound with
// six 4-wave workgroups, so budgeting registers for eight buys nothing; wider workgroups shorten a tile's chain and win
// when the tiles do not fill the chip (a partitioned rank, a small window); three waves per tile win when there are
// many lightly loaded tiles of a depth-only draw (about 120 records each on the 4096^2 frame: four waves would cull 30
// records apiece); shaded draws keep four (the resolve is most of their work: 55 vs 88 us on the 69k-triangle sphere).
struct SpanShape { int nw, occ; };
SpanShape span_shape(const frr_ctx *c, uint32_t grid, uint64_t ntris, int ps_id)
{
    static const SpanShape all[] = {{16, 4}, {8, 6}, {6, 6}, {4, 8}, {4, 6}, {LIGHT_NW, 6}};
    if (c->raster_nw || c->raster_occ)                   // options raster_nw / raster_occ (tests, tools)
        for (const SpanShape &k : all)
            if ((!c->raster_nw || k.nw == c->raster_nw) && (!c->raster_occ || k.occ == c->raster_occ)) return k;
    if (grid <= 256u) return {16, 4};
    if (grid <= 768u) return {8, 6};
    if (ps_id == FRR_PS_DEPTH && grid > 1536u && ntris * 2u <= 192ull * grid) return {LIGHT_NW, 6};
    return {4, 6};
}

template <int K, int PS> void launch_raster(frr_ctx *c, const RasterArgs &a, uint32_t grid, const SpanShape sh)
{
    ProfScope p(c, KID_RASTER);
    if (c->raster_sweep) {
        hipLaunchKernelGGL((k_raster<K, PS>), dim3(grid), dim3(256), 0, c->stream, a, c->duni);
    } else {
        // the span algebra needs every coordinate it touches within +-SPAN_SAFE (no i32 wrap)
        const int win_safe = a.x0 >= -SPAN_SAFE && a.y0 >= -SPAN_SAFE && a.x1 <= SPAN_SAFE && a.y1 <= SPAN_SAFE;
        auto go = [&](auto count_tag, auto nw_tag, auto occ_tag) {
            constexpr bool CNT = decltype(count_tag)::value;
            constexpr int NWV = decltype(nw_tag)::value, OCCV = decltype(occ_tag)::value;
            hipLaunchKernelGGL((k_raster_span<K, PS, CNT, NWV, OCCV>), dim3(grid), dim3(NWV * 64), 0, c->stream, a, c->duni, win_safe);
        };
        auto go_nw = [&](auto count_tag) {
            if (sh.nw == LIGHT_NW) go(count_tag, std::integral_constant<int, LIGHT_NW>{}, std::integral_constant<int, 6>{});
            else if (sh.nw == 4 && sh.occ == 8) go(count_tag, std::integral_constant<int, 4>{}, std::integral_constant<int, 8>{});
            else if (sh.nw == 4) go(count_tag, std::integral_constant<int, 4>{}, std::integral_constant<int, 6>{});
            else if (sh.nw == 6) go(count_tag, std::integral_constant<int, 6>{}, std::integral_constant<int, 6>{});
            else if (sh.nw == 8) go(count_tag, std::integral_constant<int, 8>{}, std::integral_constant<int, 6>{});
            else go(count_tag, std::integral_constant<int, 16>{}, std::integral_constant<int, 4>{});
        };
        if (c->count_frags) go_nw(std::true_type{}); else go_nw(std::false_type{});
    }
}

} // namespace

extern "C" {

int frr_abi_version(void) { return FRR_ABI_VERSION; }

int frr_vs_input_floats(int vs_id)
{
    switch (vs_id) {
    case FRR_VS_CLIP: return VSInfo<FRR_VS_CLIP>::NF;
    case FRR_VS_CLIP_COLOR: return VSInfo<FRR_VS_CLIP_COLOR>::NF;
    case FRR_VS_PHONG: return VSInfo<FRR_VS_PHONG>::NF;
    case FRR_VS_GOURAUD: return VSInfo<FRR_VS_GOURAUD>::NF;
    }
    return FRR_ERR_INVALID;
}
int frr_vs_num_varyings(int vs_id)
{
    switch (vs_id) {
    case FRR_VS_CLIP: return VSInfo<FRR_VS_CLIP>::K;
    case FRR_VS_CLIP_COLOR: return VSInfo<FRR_VS_CLIP_COLOR>::K;
    case FRR_VS_PHONG: return VSInfo<FRR_VS_PHONG>::K;
    case FRR_VS_GOURAUD: return VSInfo<FRR_VS_GOURAUD>::K;
    }
    return FRR_ERR_INVALID;
}

const char *frr_last_error(const frr_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int frr_create(int device, uint32_t width, uint32_t height, void *stream, frr_ctx **out)
{
    if (!out) return FRR_ERR_INVALID;
    *out = nullptr;
    if (width == 0 || height == 0 || (uint64_t)width * height > 0x3FFFFFFFull) return FRR_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return FRR_ERR_HIP;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return FRR_ERR_HIP;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return FRR_ERR_HIP; // gfx950 code objects only
    if (hipSetDevice(device) != hipSuccess) return FRR_ERR_HIP;
    frr_ctx *c = new frr_ctx();
    c->device = device; c->W = width; c->H = height;
    if (stream) c->stream = (hipStream_t)stream;
    else { if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return FRR_ERR_HIP; } c->own_stream = true; }
    const size_t npx = (size_t)width * height;
    bool ok = hipMalloc((void **)&c->own_color, npx * 4) == hipSuccess && hipMalloc((void **)&c->own_depth, npx * 4) == hipSuccess &&
              hipMalloc((void **)&c->own_tri_id, npx * 4) == hipSuccess && hipMalloc((void **)&c->cnt, sizeof(Counters)) == hipSuccess;
    c->max_tiles = ((width + TILE - 1) / TILE) * ((height + TILE - 1) / TILE);
    ok = ok && hipMalloc((void **)&c->tile_counts, (c->max_tiles + 1) * 4) == hipSuccess &&
         hipMalloc((void **)&c->tile_offsets, (c->max_tiles + 1) * 4) == hipSuccess &&
         hipMalloc((void **)&c->tile_cursor, (c->max_tiles + 1) * 4) == hipSuccess;
    if (!ok) { frr_destroy(c); return FRR_ERR_NOMEM; }
    c->color = c->own_color; c->depth = c->own_depth; c->tri_id = c->own_tri_id;
    (void)hipMemsetAsync(c->cnt, 0, sizeof(Counters), c->stream);
    (void)hipMemsetAsync(c->tile_counts, 0, (c->max_tiles + 1) * 4, c->stream);
    (void)hipMemsetAsync(c->own_color, 0, npx * 4, c->stream);      // FrameBuffer::new zero-fills (renderer.rs:423)
    (void)hipMemsetAsync(c->own_depth, 0, npx * 4, c->stream);
    (void)hipMemsetAsync(c->own_tri_id, 0xFF, npx * 4, c->stream);
    for (auto &e : c->ev) (void)hipEventCreate(&e);
    memset(&c->uni, 0, sizeof c->uni);
    frr_set_identity(c->uni.model); frr_set_identity(c->uni.view); frr_set_identity(c->uni.proj);
    c->uni.light_pos[0] = 1.2f; c->uni.light_pos[1] = 1.0f; c->uni.light_pos[2] = 2.0f;   // phong.rs:129
    c->uni.light_color[0] = c->uni.light_color[1] = c->uni.light_color[2] = 1.0f;         // phong.rs:128
    c->uni.ambient_strength = 0.1f; c->uni.specular_strength = 0.5f;                      // phong.rs:131-132
    c->uni.flat_color[0] = c->uni.flat_color[1] = c->uni.flat_color[2] = c->uni.flat_color[3] = 1.0f;
    refresh_dev_uniforms(c);
    if (hipStreamSynchronize(c->stream) != hipSuccess) { frr_destroy(c); return FRR_ERR_HIP; }
    *out = c;
    return FRR_OK;
}

void frr_destroy(frr_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    prof_collect(c);
    for (auto &m : c->meshes) if (m.used && m.owned) (void)hipFree((void *)m.dev);
    for (auto &t : c->tex) if (t.dev) (void)hipFree(t.dev);
    void *ptrs[] = {c->own_color, c->own_depth, c->own_tri_id, c->cnt, c->block_sums, c->tinfo, c->fanbase, c->fan_okey, c->recs, c->vary, c->pbox, c->bcount, c->clipq,
                    c->tile_counts, c->tile_offsets, c->tile_cursor, c->bins, c->bins2, c->bin_matrix};
    for (void *p : ptrs) if (p) (void)hipFree(p);
#ifdef FRR_DEBUG_COUNTERS
    if (c->dbg_tiles) (void)hipFree(c->dbg_tiles);
#endif
    for (auto &e : c->ev) if (e) (void)hipEventDestroy(e);
    for (auto &e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

static int settle(frr_ctx *c); // deferred frr_clear, below

int frr_set_option(frr_ctx *c, const char *name, int64_t v)
{
    if (!c || !name) return FRR_ERR_INVALID;
    { int rc = settle(c); if (rc != FRR_OK) return rc; }   // nothing half-done under the old setting
    const std::string n(name);
    if (n == "raster_sweep") c->raster_sweep = v != 0;
    else if (n == "raster_nw") { if (v != 0 && v != LIGHT_NW && v != 4 && v != 6 && v != 8 && v != 16) return fail(c, FRR_ERR_INVALID, "raster_nw: 0, 3, 4, 6, 8 or 16"); c->raster_nw = (int)v; }
    else if (n == "raster_occ") { if (v != 0 && v != 4 && v != 6 && v != 8) return fail(c, FRR_ERR_INVALID, "raster_occ: 0, 4, 6 or 8"); c->raster_occ = (int)v; }
    else if (n == "bin_chunks") { if (v < 0) return fail(c, FRR_ERR_INVALID, "bin_chunks >= 0"); c->bin_g = (int)std::min<int64_t>(v, BIN_MAX_G); }
    else if (n == "clear_eager") c->clear_eager = v != 0;
    else if (n == "clip_queue") { if (v < -1 || v > 1) return fail(c, FRR_ERR_INVALID, "clip_queue: -1, 0 or 1"); c->clip_queue = (int)v; }
    else if (n == "tile_slot_records") { if (v < 0 || v > 0x7FFFFFFF) return fail(c, FRR_ERR_INVALID, "tile_slot_records out of range"); c->ent_slot_override = (uint32_t)v; }
    else if (n == "bin_atomics") c->bin_atomics = v != 0;
    else if (n == "bin_capacity") { if (v < 0) return fail(c, FRR_ERR_INVALID, "bin_capacity >= 0"); c->bin_cap_init = (size_t)v; }
    else return fail(c, FRR_ERR_INVALID, "unknown option");
    return FRR_OK;
}

int frr_set_partition(frr_ctx *c, int rank, int world)
{
    if (!c || world < 1 || rank < 0 || rank >= world) return fail(c, FRR_ERR_INVALID, "bad partition");
    { int rc = settle(c); if (rc != FRR_OK) return rc; } // rows skipped by a fused clear are defined by the old partition
    c->rank = rank; c->world = world;
    return FRR_OK;
}
int frr_set_partition_layout(frr_ctx *c, int blocked)
{
    if (!c) return FRR_ERR_INVALID;
    { int rc = settle(c); if (rc != FRR_OK) return rc; }
    c->part_blocked = blocked != 0;
    return FRR_OK;
}
// owned tile rows of a window of `wh` pixel rows, by the same rule the kernels use (owns_tile_row)
static int owned_band(const frr_ctx *c, int64_t wh, int band, int32_t *row0, int32_t *row1)
{
    const int tiles_y = (int)((wh + TILE - 1) / TILE);
    if (c->world <= 1) {
        if (row0) { *row0 = 0; *row1 = (int32_t)wh; }
        return wh > 0 ? 1 : 0;
    }
    if (c->part_blocked) {
        const int rpr = std::max(1, (tiles_y + c->world - 1) / c->world);
        const int t0 = c->rank * rpr, t1 = std::min(tiles_y, (c->rank + 1) * rpr);
        if (t1 <= t0) return 0;
        if (row0) { *row0 = t0 * TILE; *row1 = (int32_t)std::min<int64_t>(wh, (int64_t)t1 * TILE); }
        return 1;
    }
    const int n = tiles_y > c->rank ? (tiles_y - c->rank + c->world - 1) / c->world : 0;
    if (row0 && band < n) {
        const int ty = c->rank + band * c->world;
        *row0 = ty * TILE; *row1 = (int32_t)std::min<int64_t>(wh, (int64_t)(ty + 1) * TILE);
    }
    return n;
}
int frr_owned_band_count(const frr_ctx *c, int32_t y0, int32_t y1)
{
    if (!c || y0 > y1) return FRR_ERR_INVALID;
    return owned_band(c, (int64_t)y1 - y0, 0, nullptr, nullptr);
}
int frr_owned_rows(const frr_ctx *c, int32_t y0, int32_t y1, int32_t band, int32_t *row0, int32_t *row1)
{
    if (!c || y0 > y1 || !row0 || !row1 || band < 0) return FRR_ERR_INVALID;
    int32_t a = 0, b = 0;
    const int n = owned_band(c, (int64_t)y1 - y0, band, &a, &b);
    if (band >= n) return FRR_ERR_INVALID;
    *row0 = a; *row1 = b;
    return FRR_OK;
}
int frr_set_count_fragments(frr_ctx *c, int enable)
{
    if (!c) return FRR_ERR_INVALID;
    c->count_frags = enable != 0;
    return FRR_OK;
}

int frr_bind_targets(frr_ctx *c, void *color, void *depth, void *tri_id)
{
    if (!c) return FRR_ERR_INVALID;
    { int rc = settle(c); if (rc != FRR_OK) return rc; } // a pending clear belongs to the targets bound when it was issued
    c->color = color ? (uint8_t *)color : c->own_color;
    c->depth = depth ? (float *)depth : c->own_depth;
    c->tri_id = tri_id ? (uint32_t *)tri_id : c->own_tri_id;
    return FRR_OK;
}
int frr_target_ptrs(frr_ctx *c, void **color, void **depth, void **tri_id)
{
    if (!c) return FRR_ERR_INVALID;
    { int rc = settle(c); if (rc != FRR_OK) return rc; } // the caller is about to look at them
    if (color) *color = c->color;
    if (depth) *depth = c->depth;
    if (tri_id) *tri_id = c->tri_id;
    return FRR_OK;
}

static int mesh_register(frr_ctx *c, const float *dev, bool owned, uint64_t ntris, int vs, int *mesh_out)
{
    Mesh m; m.dev = dev; m.owned = owned; m.used = true; m.ntris = ntris; m.vs = vs;
    for (size_t i = 0; i < c->meshes.size(); ++i)
        if (!c->meshes[i].used) { c->meshes[i] = m; *mesh_out = (int)i; return FRR_OK; }
    c->meshes.push_back(m);
    *mesh_out = (int)c->meshes.size() - 1;
    return FRR_OK;
}
int frr_mesh_upload(frr_ctx *c, const float *vs_inputs, uint64_t ntris, int vs_id, int *mesh_out)
{
    if (!c || !mesh_out || frr_vs_input_floats(vs_id) < 0 || (ntris && !vs_inputs)) return fail(c, FRR_ERR_INVALID, "bad mesh");
    if (ntris >= (1ull << 27)) return fail(c, FRR_ERR_UNSUPPORTED, "more than 2^27 triangles per mesh (order keys: 32 per input triangle)");
    HIP_TRY(c, hipSetDevice(c->device));
    size_t bytes = (size_t)ntris * 3 * frr_vs_input_floats(vs_id) * sizeof(float);
    void *d = nullptr;
    if (hipMalloc(&d, bytes ? bytes : 16) != hipSuccess) return fail(c, FRR_ERR_NOMEM, "hipMalloc mesh");
    if (bytes) {
        hipError_t e = hipMemcpyAsync(d, vs_inputs, bytes, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { (void)hipFree(d); return fail(c, FRR_ERR_HIP, hipGetErrorString(e)); }
    }
    return mesh_register(c, (const float *)d, true, ntris, vs_id, mesh_out);
}
int frr_mesh_bind_device(frr_ctx *c, const void *dev, uint64_t ntris, int vs_id, int *mesh_out)
{
    if (!c || !mesh_out || frr_vs_input_floats(vs_id) < 0 || (ntris && !dev)) return fail(c, FRR_ERR_INVALID, "bad mesh");
    if (ntris >= (1ull << 27)) return fail(c, FRR_ERR_UNSUPPORTED, "more than 2^27 triangles per mesh (order keys: 32 per input triangle)");
    if (((uintptr_t)dev & 15u) != 0) return fail(c, FRR_ERR_INVALID, "mesh pointer must be 16-byte aligned");
    return mesh_register(c, (const float *)dev, false, ntris, vs_id, mesh_out);
}
int frr_mesh_free(frr_ctx *c, int mesh)
{
    if (!c || mesh < 0 || mesh >= (int)c->meshes.size() || !c->meshes[mesh].used) return fail(c, FRR_ERR_INVALID, "bad mesh id");
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->meshes[mesh].owned) (void)hipFree((void *)c->meshes[mesh].dev);
    c->meshes[mesh] = Mesh();
    return FRR_OK;
}

int frr_texture_upload(frr_ctx *c, int slot, const uint8_t *rgba, uint32_t w, uint32_t h)
{
    if (!c || slot < 0 || slot >= FRR_MAX_TEXTURES || !rgba || w == 0 || h == 0) return fail(c, FRR_ERR_INVALID, "bad texture");
    if (h < w) return fail(c, FRR_ERR_UNSUPPORTED, "texture height < width: sample_2d clamps y with width (renderer.rs:523) and would index out of bounds");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    Texture &t = c->tex[slot];
    if (t.dev) { (void)hipFree(t.dev); t.dev = nullptr; }
    size_t bytes = (size_t)w * h * 4;
    if (hipMalloc((void **)&t.dev, bytes) != hipSuccess) return fail(c, FRR_ERR_NOMEM, "hipMalloc texture");
    HIP_TRY(c, hipMemcpyAsync(t.dev, rgba, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    t.w = w; t.h = h;
    refresh_dev_uniforms(c);
    return FRR_OK;
}

int frr_set_uniforms(frr_ctx *c, const frr_uniforms *u)
{
    if (!c || !u) return FRR_ERR_INVALID;
    c->uni = *u;
    refresh_dev_uniforms(c);
    return FRR_OK;
}

// the clear itself (k_clear); `counters`: also reset the frame counters
static int clear_now(frr_ctx *c, uint32_t packed, float depth, bool counters)
{
    const uint32_t n = c->W * c->H, n4 = n / 4;
    {
        ProfScope p(c, KID_CLEAR);
        uint32_t grid = std::min<uint32_t>((n4 + 255) / 256, 2048);
        hipLaunchKernelGGL(k_clear, dim3(grid ? grid : 1), dim3(256), 0, c->stream, (uint4 *)c->color, (uint4 *)c->depth,
                           (uint4 *)c->tri_id, n4, packed, depth, counters ? c->cnt : (Counters *)nullptr);
        if (n4 * 4 < n)
            hipLaunchKernelGGL(k_clear_tail, dim3(1), dim3(64), 0, c->stream, (uint32_t *)c->color, (uint32_t *)c->depth,
                               c->tri_id, n4 * 4, n, packed, depth);
    }
    HIP_TRY(c, hipGetLastError());
    return FRR_OK;
}

// bring targets and counters to the state the API promises (called by everything that looks at them)
static int settle(frr_ctx *c)
{
    HIP_TRY(c, hipSetDevice(c->device));
    { int rcs = scan_now(c); if (rcs != FRR_OK) return rcs; }   // n_emit of the latest draw (statistics, setup read-back)
    if (c->clear_pending) {
        int rc = clear_now(c, c->clear_rgba, c->clear_depth, c->counters_pending);
        if (rc != FRR_OK) return rc;
        c->clear_pending = c->counters_pending = c->unowned_debt = false;
    } else if (c->unowned_debt) {
        hipLaunchKernelGGL(k_clear_unowned_rows, dim3(c->H), dim3(256), 0, c->stream, (uint32_t *)c->color, (uint32_t *)c->depth,
                           c->tri_id, c->W, c->H, c->rank, c->world, c->debt_rpr, c->clear_rgba, c->clear_depth);
        HIP_TRY(c, hipGetLastError());
        c->unowned_debt = false;
    }
    return FRR_OK;
}

int frr_clear(frr_ctx *c, const uint8_t rgba[4], float depth)
{
    if (!c || !rgba) return FRR_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    uint32_t packed;
    memcpy(&packed, rgba, 4);
    c->geom_ntris = 0;          // the setup list of a preceding frr_geometry is gone (frr_raster then draws nothing)
    c->scan_pending = false;
    if (c->clear_eager) return clear_now(c, packed, depth, true);
    c->clear_rgba = packed; c->clear_depth = depth;
    c->clear_pending = c->counters_pending = true;
    c->unowned_debt = false; // superseded: the pending clear covers every row
    return FRR_OK;
}

static int geometry_impl(frr_ctx *c, int mesh, uint64_t *ntris_setup, bool filter, int32_t fy0, int32_t fy1)
{
    if (!c || mesh < 0 || mesh >= (int)c->meshes.size() || !c->meshes[mesh].used) return fail(c, FRR_ERR_INVALID, "bad mesh id");
    HIP_TRY(c, hipSetDevice(c->device));
    const Mesh &m = c->meshes[mesh];
    const int K = frr_vs_num_varyings(m.vs);
    const uint64_t nt = m.ntris;
    int rc;
    if ((rc = scan_now(c)) != FRR_OK) return rc;   // the previous draw's n_emit feeds this draw's tri_base
    // fan space: clipped inputs are the ones that straddle a frustum plane, usually few; room for as many fan triangles
    // as there are inputs (+ 4096) to start with, grown on demand (FRR_ERR_CAPACITY: the frame is re-issued) up to the
    // worst case of 19 per input (small meshes get their worst case outright: 2^20 slots are cheap) ...
    const uint32_t nblocks = (uint32_t)((nt + GEOM_BLOCK - 1) / GEOM_BLOCK);
    // ... in FAN_REGIONS regions (block b allocates in region b % FAN_REGIONS: frr_device.h); a region never needs more
    // than 19 slots for every input of the blocks that use it
    const uint64_t region_worst = (uint64_t)FRR_MAX_OUT_TRIS * GEOM_BLOCK * ((nblocks + FAN_REGIONS - 1) / FAN_REGIONS);
    uint64_t region = std::max<uint64_t>(std::max<uint64_t>((nt + 4096 + FAN_REGIONS - 1) / FAN_REGIONS, std::min<uint64_t>(region_worst, (1u << 20) / FAN_REGIONS)),
                                         (c->fan_hint + FAN_REGIONS - 1) / FAN_REGIONS);
    region = std::max<uint64_t>(std::min<uint64_t>(region, region_worst), 1);
    uint64_t fan_cap = region * FAN_REGIONS;
    if (nt + fan_cap > 0xFFFFFFF0ull) fan_cap = (0xFFFFFFF0ull - nt) / FAN_REGIONS * FAN_REGIONS;
    const size_t slots = (size_t)(nt + fan_cap);
    if ((rc = ensure(c, c->block_sums, c->block_sums_cap, (size_t)nblocks + 1)) != FRR_OK) return rc;
    if ((rc = ensure(c, c->tinfo, c->tinfo_cap, (size_t)std::max<uint64_t>(nt, 1))) != FRR_OK) return rc;
    if ((rc = ensure(c, c->fanbase, c->fanbase_cap, (size_t)std::max<uint64_t>(nt, 1))) != FRR_OK) return rc;
    if ((rc = ensure(c, c->fan_okey, c->fan_okey_cap, (size_t)std::max<uint64_t>(fan_cap, 1))) != FRR_OK) return rc;
    if ((rc = ensure(c, c->recs, c->setup_cap, std::max<size_t>(slots, 1024))) != FRR_OK) return rc;
    if ((rc = ensure(c, c->pbox, c->pbox_cap, c->setup_cap)) != FRR_OK) return rc;
    if ((rc = ensure(c, c->bcount, c->bcount_cap, (size_t)nblocks + 1)) != FRR_OK) return rc;
    if (K > 0 && (rc = ensure(c, c->vary, c->vary_cap, (size_t)c->setup_cap * 3 * 8 /* K <= 8 in the shader table */)) != FRR_OK) return rc;
    const bool use_clipq = nt > 0 && (c->clip_queue > 0 || (c->clip_queue < 0 && c->clip_queue_auto));
    if (use_clipq && (rc = ensure(c, c->clipq, c->clipq_cap, (size_t)nt)) != FRR_OK) return rc;
    GeomArgs g;
    g.in = m.dev; g.ntris = (uint32_t)nt; g.width = c->W; g.height = c->H;
    g.fan_cap = (uint32_t)fan_cap;
    g.reset_frame = c->counters_pending ? 1 : 0; // (committed with the launch, below)
    g.part_rank = c->rank; g.part_world = filter ? c->world : 1; g.part_y0 = fy0; g.part_y1 = fy1;
    g.part_rpr = 0;
    if (filter && c->part_blocked) {
        const int tiles_y = (int)(((int64_t)fy1 - fy0 + TILE - 1) / TILE);
        g.part_rpr = std::max(1, (tiles_y + c->world - 1) / c->world);
    }
    // the per-draw slot (fan cursor) alternates; everything is committed only now that nothing can fail any more
    // (a failed allocation above must not leave a slot toggled that nobody zeroed)
    const int slot = c->geom_slot ^ 1;
    g.fslot = slot;
    g.block_sums = c->block_sums; g.tinfo = c->tinfo; g.fanbase = c->fanbase; g.fan_okey = c->fan_okey;
    g.recs = c->recs; g.vary = c->vary; g.pbox = c->pbox; g.cnt = c->cnt;
    g.clipq = c->clipq; g.use_clipq = use_clipq ? 1 : 0;
    g.bcount = c->bcount;
    // what the setup list about to be built was filtered by (frr_raster / frr_readback_setup check it)
    c->geom_filter = frr_ctx::GeomFilter{filter, fy0, fy1, c->rank, c->world, c->part_blocked};
    c->geom_slot = slot;
    c->geom_fan_cap = (uint32_t)fan_cap;
    c->geom_nblocks = nblocks;
    c->counters_pending = false;
    c->geom_vs = m.vs; c->geom_ntris = nt;
    if (nt == 0) {
        hipLaunchKernelGGL(k_geom_empty, dim3(1), dim3(64), 0, c->stream, g);
    } else {
        launch_geometry_vs(c, g, nblocks, m.vs);
        c->scan_pending = true;
    }
    HIP_TRY(c, hipGetLastError());
    if (ntris_setup) {
        if ((rc = scan_now(c)) != FRR_OK) return rc;
        Counters h;
        if ((rc = check_frame_counters(c, &h)) < FRR_OK) return rc;
        *ntris_setup = h.n_emit;
    }
    return FRR_OK;
}

int frr_geometry(frr_ctx *c, int mesh, uint64_t *ntris_setup) { return geometry_impl(c, mesh, ntris_setup, false, 0, 0); }

int frr_raster(frr_ctx *c, int ps_id, int32_t x0, int32_t x1, int32_t y0, int32_t y1)
{
    if (!c || c->geom_vs < 0) return fail(c, FRR_ERR_INVALID, "frr_raster before frr_geometry");
    if (x0 > x1 || y0 > y1) return fail(c, FRR_ERR_INVALID, "range min > max (i32::clamp would panic, renderer.rs:285)");
    const int64_t ww = (int64_t)x1 - x0, wh = (int64_t)y1 - y0;
    if (ww > (int64_t)c->W || wh > (int64_t)c->H) return fail(c, FRR_ERR_INVALID, "window larger than the FrameBuffer");
    if (x0 < -32768 || y0 < -32768 || x1 > 32767 || y1 > 32767) return fail(c, FRR_ERR_UNSUPPORTED, "window coordinates outside the i16 range");
    if (ww > 0 && wh > 0 && (x1 <= 0 || (wh - 1) * (int64_t)x1 + ww > (int64_t)c->W * c->H))
        return fail(c, FRR_ERR_INVALID, "depth index (cy-y0)*x1+(cx-x0) would leave the depth buffer (renderer.rs:362)");
    const int K = frr_vs_num_varyings(c->geom_vs);
    if ((ps_id == FRR_PS_COLOR && K != 3) || ((ps_id == FRR_PS_PHONG || ps_id == FRR_PS_BLINN) && K != 8) || ps_id < 0 || ps_id > FRR_PS_BLINN)
        return fail(c, FRR_ERR_INVALID, "pixel shader does not match the vertex shader's varyings");
    if ((ps_id == FRR_PS_PHONG || ps_id == FRR_PS_BLINN) && !c->duni.tex) return fail(c, FRR_ERR_INVALID, "no texture bound to uniforms.texture_slot");
    {
        // frr_draw on a partitioned ctx keeps only the triangles that touch the rank's tile rows of ITS window; that
        // list serves no other window or partition (the reference may reuse one geometry for several ranges,
        // renderer.rs:269-271: use frr_geometry for that, it never filters)
        const frr_ctx::GeomFilter &gf = c->geom_filter;
        if (gf.active && (gf.y0 != y0 || gf.y1 != y1 || gf.rank != c->rank || gf.world != c->world || gf.blocked != c->part_blocked))
            return fail(c, FRR_ERR_INVALID, "the setup list was filtered by frr_draw for another window/partition; re-run frr_geometry (unfiltered) before frr_raster");
    }
    HIP_TRY(c, hipSetDevice(c->device));
    if (ww == 0 || wh == 0 || c->geom_ntris == 0) return FRR_OK;
    bool fuse = false;
    if (c->clear_pending) {
        const bool full = x0 == 0 && y0 == 0 && x1 == (int32_t)c->W && y1 == (int32_t)c->H;
        if (full && !c->counters_pending && !c->raster_sweep) fuse = true; // the tile kernel performs the clear
        else { int rcs = settle(c); if (rcs != FRR_OK) return rcs; }
    }
    RasterArgs a;
    a.fused_clear = fuse ? 1 : 0; a.clear_rgba = c->clear_rgba; a.clear_depth = c->clear_depth;
    a.x0 = x0; a.x1 = x1; a.y0 = y0; a.y1 = y1; a.win_w = (int)ww; a.win_h = (int)wh;
    a.cstride = (int)c->W; a.dstride = x1;
    a.tiles_x = (int)((ww + TILE - 1) / TILE); a.tiles_y = (int)((wh + TILE - 1) / TILE);
    a.tiles_x_magic = 0u; // set below once the grid is known (exact only for block indices and tile counts < 2^16)
    a.rank = c->rank; a.world = c->world;
    a.rpr = (c->part_blocked && c->world > 1) ? std::max(1, (a.tiles_y + c->world - 1) / c->world) : 0;
    a.recs = c->recs; a.vary = c->vary; a.pbox = c->pbox; a.bcount = c->bcount;
    a.tile_counts = c->tile_counts; a.tile_offsets = c->tile_offsets; a.tile_cursor = c->tile_cursor;
    const uint32_t ntiles = (uint32_t)a.tiles_x * a.tiles_y;
    int rc;
    if (!c->bins) {
        size_t want = std::max<size_t>((size_t)c->geom_ntris * 8 + 4 * (size_t)c->max_tiles, (size_t)1 << 22);
        if (c->bin_cap_init) want = c->bin_cap_init;
        if ((rc = ensure(c, c->bins, c->bin_cap, want)) != FRR_OK) return rc;
        if ((rc = ensure(c, c->bins2, c->bin2_cap, want)) != FRR_OK) return rc;
    }
    a.bins2 = c->bins2;
    a.bins = c->bins; a.bin_cap = (uint32_t)std::min<size_t>(c->bin_cap, 0xFFFFFFFFu);
    a.color = c->color; a.depth = c->depth; a.tri_id = c->tri_id; a.cnt = c->cnt;
#ifdef FRR_DEBUG_COUNTERS
    if (!c->dbg_tiles && getenv("FRR_DEBUG_TILES")) {
        if (hipMalloc((void **)&c->dbg_tiles, (size_t)c->max_tiles * 64) != hipSuccess) c->dbg_tiles = nullptr;
    }
    if (c->dbg_tiles) (void)hipMemsetAsync(c->dbg_tiles, 0, (size_t)c->max_tiles * 64, c->stream);
    a.dbg_tiles = c->dbg_tiles;
#endif
    a.seg = nullptr; a.nseg = 0; a.slot = 0;
    const int owned_rows = a.rpr > 0 ? std::max(0, std::min(a.tiles_y, (a.rank + 1) * a.rpr) - a.rank * a.rpr)
                                     : (a.tiles_y > a.rank ? (a.tiles_y - a.rank + a.world - 1) / a.world : 0);
    const uint32_t grid = (uint32_t)a.tiles_x * owned_rows;
    if (a.tiles_x >= 2 && a.tiles_x < 65536 && grid < 65536u) a.tiles_x_magic = (uint32_t)(0x100000000ull / (uint64_t)a.tiles_x + 1ull);
    const SpanShape sh = span_shape(c, grid, c->geom_ntris, ps_id);
    if (grid <= BIN_LDS_MAX_TILES && !c->bin_atomics && !c->raster_sweep) {
        const uint32_t ltiles = std::max<uint32_t>(grid, 1u);   // the binning numbers the rank's OWN tiles only (local_tile_row)
        // segmented LDS multi-split (one launch, no per-entry global atomics): G chunk workgroups, ~3K triangles each
        // (small meshes: one triangle per thread, so that the launch is not three workgroups doing all the work)
        uint32_t G = (uint32_t)std::min<uint64_t>(std::max<uint64_t>((c->geom_ntris + BIN_WG - 1) / BIN_WG, 1), BIN_MAX_G);
        if (c->bin_g > 0) G = (uint32_t)std::min(c->bin_g, BIN_MAX_G);
        G = std::min<uint32_t>(G, sh.nw == 3 ? 256u : (uint32_t)sh.nw * 64u); // the tile kernel reads one segment per thread (three waves: wave 0 reads a second one)
        // (+ one workgroup that scans the geometry kernel's block sums, unless an earlier launch has; a binning workgroup
        // fills a CU's LDS, so the launch stays within 256 workgroups: a 257th would wait for a whole one to finish)
        int do_scan = c->scan_pending ? 1 : 0;
        if (do_scan && G > 255u) G = 255u;
        if ((rc = ensure(c, c->bin_matrix, c->bin_matrix_cap, (size_t)BIN_MAX_G * ((size_t)c->max_tiles + 1))) != FRR_OK) return rc;
        // dynamic LDS: tile counters + as many staged 16-B records as fit (a chunk emits ~1.8 records per triangle)
        constexpr size_t kLdsBudget = 160 * 1024 - 1024; // the kernel's static LDS is < 1 KB
        const size_t hist_bytes = (((size_t)ltiles + 3) & ~(size_t)3) * sizeof(uint32_t);
        const uint32_t stage_cap = (uint32_t)std::min<size_t>((kLdsBudget - hist_bytes) / 16, 9216);
        const size_t lds = hist_bytes + (size_t)stage_cap * 16;
        if (!c->lds_attr_set) {
            HIP_TRY(c, hipFuncSetAttribute((const void *)k_bin_seg, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget));
            c->lds_attr_set = true;
        }
        c->bin_slot ^= 1;
        a.seg = c->bin_matrix; a.nseg = G; a.slot = c->bin_slot;
        // near-first copies (bins2): a fixed slot per tile, 8x the mean tile load, + an overflow arena of bin_cap records
        uint64_t S = std::max<uint64_t>(256, (c->geom_ntris * 16 + ntiles - 1) / ntiles);   // (per-tile load of the whole window: ownership does not change it)
        S = std::min<uint64_t>(S, ((uint64_t)1 << 30) / ltiles);
        if (c->ent_slot_override) S = c->ent_slot_override;
        a.ent_slot = (uint32_t)S;
        a.bin_cap = (uint32_t)std::min<size_t>(c->bin_cap, 0xBFFFFFFFu);
        if ((rc = ensure(c, c->bins2, c->bin2_cap, (size_t)ltiles * S + a.bin_cap)) != FRR_OK) return rc;
        a.bins2 = c->bins2;
        {
            ProfScope p(c, KID_BIN_SEG);
            hipLaunchKernelGGL(k_bin_seg, dim3(G + do_scan), dim3(BIN_WG), lds, c->stream, a, ltiles, c->bin_matrix, a.slot, stage_cap,
                               c->geom_slot, c->geom_fan_cap, c->block_sums, c->geom_nblocks, do_scan);
        }
        c->scan_pending = false;
    } else {
        // fallback for frames with more tiles than fit LDS counters: global atomics
        if ((rc = scan_now(c)) != FRR_OK) return rc;
        const uint32_t bin_grid = (uint32_t)std::min<uint64_t>((c->geom_ntris + c->geom_fan_cap + 255) / 256, 2048);
        { ProfScope p(c, KID_BIN_COUNT); hipLaunchKernelGGL(k_bin<false>, dim3(bin_grid), dim3(256), 0, c->stream, a, c->geom_slot, c->geom_fan_cap); }
        { ProfScope p(c, KID_TILE_SCAN); hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, c->stream, a, ntiles); }
        { ProfScope p(c, KID_BIN_FILL); hipLaunchKernelGGL(k_bin<true>, dim3(bin_grid), dim3(256), 0, c->stream, a, c->geom_slot, c->geom_fan_cap); }
    }
    if (grid) {
        switch (ps_id) {
        case FRR_PS_DEPTH: launch_raster<0, FRR_PS_DEPTH>(c, a, grid, sh); break;
        case FRR_PS_FLAT: launch_raster<0, FRR_PS_FLAT>(c, a, grid, sh); break;
        case FRR_PS_COLOR: launch_raster<3, FRR_PS_COLOR>(c, a, grid, sh); break;
        case FRR_PS_PHONG: launch_raster<8, FRR_PS_PHONG>(c, a, grid, sh); break;
        case FRR_PS_BLINN: launch_raster<8, FRR_PS_BLINN>(c, a, grid, sh); break;
        }
    }
    HIP_TRY(c, hipGetLastError());
    if (fuse) {
        c->clear_pending = false;
        // the tile rows of other ranks missed this clear: owed to the ctx's own targets (frr_readback shows the
        // whole image); caller-bound targets of a partitioned ctx only ever have their owned rows defined
        c->unowned_debt = c->world > 1 && c->color == c->own_color && c->depth == c->own_depth && c->tri_id == c->own_tri_id;
        c->debt_rpr = a.rpr;
    }
    return FRR_OK;
}

int frr_draw(frr_ctx *c, int mesh, int ps_id, int32_t x0, int32_t x1, int32_t y0, int32_t y1)
{
    // frr_draw knows the raster window, so a partitioned ctx can skip the setup records of triangles
    // that touch none of its tile rows (frr_geometry alone cannot: the window comes later)
    const bool filter = c && c->world > 1 && y0 <= y1;
    int rc = geometry_impl(c, mesh, nullptr, filter, y0, y1);
    if (rc != FRR_OK) return rc;
    return frr_raster(c, ps_id, x0, x1, y0, y1);
}

int frr_sync(frr_ctx *c)
{
    if (!c) return FRR_ERR_INVALID;
    { int rc = settle(c); if (rc != FRR_OK) return rc; }
    return check_frame_counters(c, nullptr);
}

int frr_readback(frr_ctx *c, uint8_t *rgba, float *depth, uint32_t *tri_id)
{
    if (!c) return FRR_ERR_INVALID;
    { int rc = settle(c); if (rc != FRR_OK) return rc; }
    const size_t bytes = (size_t)c->W * c->H * 4;
    if (rgba) HIP_TRY(c, hipMemcpyAsync(rgba, c->color, bytes, hipMemcpyDeviceToHost, c->stream));
    if (depth) HIP_TRY(c, hipMemcpyAsync(depth, c->depth, bytes, hipMemcpyDeviceToHost, c->stream));
    if (tri_id) HIP_TRY(c, hipMemcpyAsync(tri_id, c->tri_id, bytes, hipMemcpyDeviceToHost, c->stream));
    return check_frame_counters(c, nullptr);
}

int frr_readback_setup(frr_ctx *c, frr_setup_vertex *out, uint64_t cap_tris, uint64_t *ntris)
{
    if (!c || !ntris || c->geom_vs < 0) return fail(c, FRR_ERR_INVALID, "no geometry to read back");
    if (c->geom_filter.active)
        return fail(c, FRR_ERR_INVALID, "the setup list of a partitioned frr_draw holds only this rank's triangles; use frr_geometry to read back the full Vec<[Vertex;3]>");
    { int rcs = settle(c); if (rcs != FRR_OK) return rcs; }
    Counters h;
    const int rc_frame = check_frame_counters(c, &h);
    if (rc_frame < FRR_OK) return rc_frame;
    *ntris = h.n_emit;
    if (!out) return rc_frame;
    // the records live at slots (frr_device.h): input t's own slot, or its fan's slots behind the inputs; walking the
    // inputs in order and each fan in order is the reference's emission order
    const uint64_t nt = c->geom_ntris;
    const uint64_t slots = nt + c->geom_fan_cap;   // (fan slots are spread over the regions of the fan space)
    const int K = frr_vs_num_varyings(c->geom_vs);
    std::vector<uint32_t> tinfo(nt), fanbase(nt);
    std::vector<RasterRec> recs(slots);
    std::vector<float> vary((size_t)slots * 3 * K);
    if (nt) HIP_TRY(c, hipMemcpy(tinfo.data(), c->tinfo, nt * 4, hipMemcpyDeviceToHost));
    if (nt) HIP_TRY(c, hipMemcpy(fanbase.data(), c->fanbase, nt * 4, hipMemcpyDeviceToHost));
    // the inputs' own slots, then the used part of every fan region
    const uint64_t region = c->geom_fan_cap / FAN_REGIONS;
    for (int k = -1; k < FAN_REGIONS; ++k) {
        const uint64_t first = k < 0 ? 0 : nt + (uint64_t)k * region;
        const uint64_t count = k < 0 ? nt : std::min<uint64_t>(h.fan_cursor[c->geom_slot][k].v, region);
        if (!count) continue;
        HIP_TRY(c, hipMemcpy(recs.data() + first, c->recs + first, count * sizeof(RasterRec), hipMemcpyDeviceToHost));
        if (K) HIP_TRY(c, hipMemcpy(vary.data() + first * 3 * K, c->vary + first * 3 * K, count * 3 * K * sizeof(float), hipMemcpyDeviceToHost));
    }
    uint64_t i = 0;
    for (uint64_t t = 0; t < nt && i < cap_tris; ++t) {
        const uint32_t n = tinfo[t] & ((1u << FAN_BITS) - 1u);
        for (uint32_t q = 0; q < n && i < cap_tris; ++q, ++i) {
            const uint64_t slot = n == 1u ? t : nt + fanbase[t] + q;     // (a clipped input emits at least two triangles)
            if (slot >= slots) return fail(c, FRR_ERR_HIP, "setup tables are inconsistent");
            const RasterRec &r = recs[slot];
            const bool sw = r.flags & 1u;
            for (int v = 0; v < 3; ++v) {
                const int s = sw ? (v == 1 ? 2 : (v == 2 ? 1 : 0)) : v; // undo the orientation swap
                frr_setup_vertex &o = out[i * 3 + v];
                memset(&o, 0, sizeof o);
                o.spf[0] = r.s[2 * s]; o.spf[1] = r.s[2 * s + 1];
                o.spi[0] = r.p[2 * s]; o.spi[1] = r.p[2 * s + 1];
                o.rhw = r.rhw[s];
                for (int k = 0; k < K; ++k) o.ctx[k] = vary[(size_t)slot * 3 * K + (size_t)s * K + k];
            }
        }
    }
    return rc_frame;
}

int frr_get_stats(frr_ctx *c, frr_stats *out)
{
    if (!c || !out) return FRR_ERR_INVALID;
    { int rc = settle(c); if (rc != FRR_OK) return rc; }
    Counters h;
    HIP_TRY(c, hipMemcpyAsync(&h, c->cnt, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->clip_queue_auto = std::max(h.clip_block_max[0], h.clip_block_max[1]) > (uint32_t)CLIP_QUEUE_AT;
    out->tris_in = h.tris_in;
    out->tris_setup = (uint64_t)h.tri_base + h.n_emit;   // (settle() has scanned the latest draw's block sums)
    out->bin_entries = h.bin_entries_frame + h.seg_total[0] + h.seg_total[1];
    out->frag_covered = h.frag_covered;
    out->frag_nan = h.frag_nan;
    out->draws = h.draws;
    out->overflow = h.overflow;
#ifdef FRR_DEBUG_COUNTERS
    if (getenv("FRR_DEBUG_PRINT")) {
        fprintf(stderr, "frr dbg:");
        for (int k = 0; k < 24; ++k) {
            unsigned long long v = 0;
            for (int j = 0; j < DBG_COPIES; ++j) v += h.dbg[j][k];
            fprintf(stderr, " %llu", v);
        }
        fprintf(stderr, "\n");
    }
#endif
    return FRR_OK;
}

int frr_event_record(frr_ctx *c, int slot)
{
    if (!c || slot < 0 || slot >= 16) return FRR_ERR_INVALID;
    HIP_TRY(c, hipEventRecord(c->ev[slot], c->stream));
    c->ev_set[slot] = true;
    return FRR_OK;
}
int frr_event_elapsed_ms(frr_ctx *c, int a, int b, float *ms)
{
    if (!c || !ms || a < 0 || a >= 16 || b < 0 || b >= 16 || !c->ev_set[a] || !c->ev_set[b]) return FRR_ERR_INVALID;
    HIP_TRY(c, hipEventSynchronize(c->ev[b]));
    HIP_TRY(c, hipEventElapsedTime(ms, c->ev[a], c->ev[b]));
    return FRR_OK;
}
int frr_profile_enable(frr_ctx *c, int enable)
{
    if (!c) return FRR_ERR_INVALID;
    prof_collect(c);
    c->prof_mask = enable < 0 ? 0xFFFFFFFFu : (uint32_t)enable;
    return FRR_OK;
}
int frr_profile_set_period(frr_ctx *c, uint32_t period)
{
    if (!c || period == 0) return FRR_ERR_INVALID;
    c->prof_period = period;
    for (int i = 0; i < KID_COUNT; ++i) c->prof_seen[i] = 0;
    return FRR_OK;
}
int frr_profile_reset(frr_ctx *c)
{
    if (!c) return FRR_ERR_INVALID;
    prof_collect(c);
    for (int i = 0; i < KID_COUNT; ++i) { c->prof_ms[i] = 0; c->prof_n[i] = 0; }
    return FRR_OK;
}
int frr_profile_get(frr_ctx *c, const char *kernel, float *total_ms, uint32_t *launches)
{
    if (!c || !kernel) return FRR_ERR_INVALID;
    prof_collect(c);
    for (int i = 0; i < KID_COUNT; ++i)
        if (strcmp(kernel, kKernelNames[i]) == 0) {
            if (total_ms) *total_ms = (float)c->prof_ms[i];
            if (launches) *launches = c->prof_n[i];
            return FRR_OK;
        }
    return FRR_ERR_INVALID;
}

// ---- host helpers: matrix_util.rs:3-35 ------------------------------------------------------
void frr_set_identity(float m[16])
{
    for (int i = 0; i < 16; ++i) m[i] = (i % 5 == 0) ? 1.0f : 0.0f;
}
static inline float h_dot3(const float *a, const float *b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
static inline void h_norm3(float *v) { float r = 1.0f / sqrtf(h_dot3(v, v)); v[0] *= r; v[1] *= r; v[2] *= r; }
static inline void h_cross3(const float *a, const float *b, float *o)
{
    o[0] = a[1] * b[2] - b[1] * a[2]; o[1] = a[2] * b[0] - b[2] * a[0]; o[2] = a[0] * b[1] - b[0] * a[1];
}
void frr_set_look_at(const float eye[3], const float at[3], const float up[3], float m[16])
{
    float z[3] = {at[0] - eye[0], at[1] - eye[1], at[2] - eye[2]}, x[3], y[3];
    h_norm3(z);                 // z_axis = (at - eye).normalize()      :11
    h_cross3(up, z, x); h_norm3(x); // x_axis = up.cross(z_axis).normalize() :12
    h_cross3(z, x, y);          // y_axis = z_axis.cross(x_axis)         :13
    m[0] = x[0]; m[1] = y[0]; m[2] = z[0]; m[3] = 0.0f;
    m[4] = x[1]; m[5] = y[1]; m[6] = z[1]; m[7] = 0.0f;
    m[8] = x[2]; m[9] = y[2]; m[10] = z[2]; m[11] = 0.0f;
    m[12] = -h_dot3(eye, x); m[13] = -h_dot3(eye, y); m[14] = -h_dot3(eye, z); m[15] = 1.0f;
}
void frr_set_perspective(float fovy, float aspect, float zn, float zf, float m[16])
{
    const float fax = 1.0f / tanf(fovy * 0.5f);         // f32::tan(..).recip()  :26
    memset(m, 0, 16 * sizeof(float));
    m[0] = fax / aspect;                                // :28
    m[5] = fax;                                         // :29
    m[10] = zf / (zf - zn);                             // :30
    m[14] = -zn * zf / (zf - zn);                       // :31
    m[11] = 1.0f;                                       // :32
}

// ---- debug hooks ------------------------------------------------------------------------------
float frr_host_atan2f(float y, float x) { return fd_atan2f(y, x); }

int frr_debug_mvp(frr_ctx *c, int mesh, int use_mfma, float *clip_out, float *ms_out)
{
    if (!c || mesh < 0 || mesh >= (int)c->meshes.size() || !c->meshes[mesh].used || !clip_out) return fail(c, FRR_ERR_INVALID, "bad arguments");
    const Mesh &m = c->meshes[mesh];
    if (m.vs != FRR_VS_PHONG && m.vs != FRR_VS_GOURAUD) return fail(c, FRR_ERR_INVALID, "needs a pos3/uv2/normal3 mesh");
    HIP_TRY(c, hipSetDevice(c->device));
    const uint32_t nverts = (uint32_t)(m.ntris * 3);
    float4 *d = nullptr;
    if (hipMalloc((void **)&d, (size_t)nverts * 16 + 16) != hipSuccess) return fail(c, FRR_ERR_NOMEM, "hipMalloc");
    const dim3 grid((nverts + 63) / 64), block(64);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) { // last iteration is the timed one
        (void)hipEventRecord(e0, c->stream);
        if (use_mfma) hipLaunchKernelGGL(k_debug_mvp_mfma, grid, block, 0, c->stream, m.dev, nverts, 8, c->duni, d);
        else hipLaunchKernelGGL(k_debug_mvp_exact, grid, block, 0, c->stream, m.dev, nverts, 8, c->duni, d);
        (void)hipEventRecord(e1, c->stream);
    }
    hipError_t e = hipMemcpyAsync(clip_out, d, (size_t)nverts * 16, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms_out) *ms_out = ms;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(c, FRR_ERR_HIP, hipGetErrorString(e));
    return FRR_OK;
}

int frr_debug_gather_calib(frr_ctx *c, uint32_t log2_records)
{
    if (!c || log2_records < 10 || log2_records > 24) return FRR_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n = (size_t)1 << log2_records;
    void *d = nullptr;
    if (hipMalloc(&d, n * 64 + 64) != hipSuccess) return fail(c, FRR_ERR_NOMEM, "hipMalloc");
    hipError_t e = hipMemsetAsync(d, 1, n * 64 + 64, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_debug_gather, dim3((uint32_t)(n / 256)), dim3(256), 0, c->stream, (const uint4 *)d, (uint32_t)(n - 1), (uint32_t *)((char *)d + n * 64));
        e = hipStreamSynchronize(c->stream);
    }
    (void)hipFree(d);
    if (e != hipSuccess) return fail(c, FRR_ERR_HIP, hipGetErrorString(e));
    return FRR_OK;
}

int frr_debug_rcp_check(frr_ctx *c, uint32_t lo_bits, uint32_t hi_bits, uint64_t *mismatches, uint32_t *first_bad)
{
    if (!c || !mismatches || !first_bad) return FRR_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    unsigned long long *d = nullptr;
    if (hipMalloc((void **)&d, 16) != hipSuccess) return fail(c, FRR_ERR_NOMEM, "hipMalloc");
    const unsigned long long init[2] = {0ull, 0xFFFFFFFFull};
    hipError_t e = hipMemcpyAsync(d, init, 16, hipMemcpyHostToDevice, c->stream);
    unsigned long long out[2] = {0, 0};
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_debug_rcp, dim3(4096), dim3(256), 0, c->stream, lo_bits, hi_bits, d, (uint32_t *)(d + 1));
        e = hipMemcpyAsync(out, d, 16, hipMemcpyDeviceToHost, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(c, FRR_ERR_HIP, hipGetErrorString(e));
    *mismatches = out[0];
    *first_bad = (uint32_t)out[1];
    return FRR_OK;
}

#ifdef FRR_DEBUG_COUNTERS
// dev builds only (not part of include/frr.h): the per-tile timeline of the latest tile kernel, [max_tiles][8] u64
int frr_debug_tiles(frr_ctx *c, unsigned long long *out, uint32_t *ntiles)
{
    if (!c || !out || !ntiles || !c->dbg_tiles) return FRR_ERR_INVALID;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(out, c->dbg_tiles, (size_t)c->max_tiles * 64, hipMemcpyDeviceToHost));
    *ntiles = c->max_tiles;
    return FRR_OK;
}
#endif

int frr_debug_scan64(frr_ctx *c, const uint32_t *in, uint32_t *out)
{
    if (!c || !in || !out) return FRR_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    uint32_t *d = nullptr;
    if (hipMalloc((void **)&d, 128 * 4) != hipSuccess) return fail(c, FRR_ERR_NOMEM, "hipMalloc");
    hipError_t e = hipMemcpyAsync(d, in, 64 * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_debug_scan, dim3(1), dim3(64), 0, c->stream, d, d + 64);
        e = hipMemcpyAsync(out, d + 64, 64 * 4, hipMemcpyDeviceToHost, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(c, FRR_ERR_HIP, hipGetErrorString(e));
    return FRR_OK;
}

int frr_debug_atan2f(frr_ctx *c, const float *y, const float *x, float *out, uint64_t n)
{
    if (!c || !y || !x || !out) return FRR_ERR_INVALID;
    if (n == 0) return FRR_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    float *d = nullptr;
    if (hipMalloc((void **)&d, n * 12) != hipSuccess) return fail(c, FRR_ERR_NOMEM, "hipMalloc");
    hipError_t e = hipMemcpyAsync(d, y, n * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d + n, x, n * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_debug_atan2f, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, c->stream, d, d + n, d + 2 * n, n);
        e = hipMemcpyAsync(out, d + 2 * n, n * 4, hipMemcpyDeviceToHost, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(c, FRR_ERR_HIP, hipGetErrorString(e));
    return FRR_OK;
}

} // extern "C"
