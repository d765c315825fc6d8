// frr_api.hip -- host side of libfrr_hip.so: the C ABI of include/frr.h over the kernels in
// frr_kernels.h.  gfx950 only; no CPU fallback (every compute entry point needs the device).
#include "frr_kernels.h"

#include <math.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include <hip/hiprtc.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

// The text of the device headers, for frr_shader_register (hiprtc compiles user shaders into the library's own kernels):
// the assembler copies the files into .rodata of the host object.
#ifndef FRR_CSRC_DIR
#define FRR_CSRC_DIR "f_renderer_amd/csrc"   // (builds started at the repository root; f_renderer_amd/_native.py passes the absolute path)
#endif
#if !defined(__HIP_DEVICE_COMPILE__)
#define FRR_EMBED(sym, file) __asm__(".pushsection .rodata\n.global " #sym "\n" #sym ":\n.incbin \"" FRR_CSRC_DIR "/" file "\"\n.byte 0\n.popsection\n")
FRR_EMBED(frr_src_device_h, "frr_device.h");
FRR_EMBED(frr_src_exact_h, "frr_exact.h");
FRR_EMBED(frr_src_kernels_h, "frr_kernels.h");
FRR_EMBED(frr_src_raster_h, "frr_raster.h");
FRR_EMBED(frr_src_frr_h, "../../include/frr.h");
#endif
extern "C" const char frr_src_device_h[], frr_src_exact_h[], frr_src_kernels_h[], frr_src_raster_h[], frr_src_frr_h[];

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
    uint64_t gen = 0;   // which registration of the ctx this is (DrawSig: a later mesh may live at a freed one's address)
};
struct Texture {
    uint8_t *dev = nullptr;
    uint32_t w = 0, h = 0;
};
struct ProfRec { int kid; hipEvent_t a, b; };

// ---- user shaders (frr_shader_register): a per-process registry of code objects, loaded per ctx on first use ---------
constexpr int kSpanShapes[6][2] = {{16, 4}, {8, 6}, {6, 6}, {4, 8}, {4, 6}, {LIGHT_NW, 6}};   // the shapes launch_raster knows
struct UserShader {
    int nf = 0, K = 0;
    std::vector<char> code;
    std::string geom, clip, sweep, span[2][6];   // lowered kernel names ([count fragments][shape]; sweep: the brute-force tile kernel)
};
std::mutex g_shader_mu;
std::vector<UserShader *> g_shaders;      // id = FRR_SHADER_USER_BASE + index; never shrinks
const UserShader *user_shader(int id)
{
    std::lock_guard<std::mutex> lk(g_shader_mu);
    const int i = id - FRR_SHADER_USER_BASE;
    return (i >= 0 && i < (int)g_shaders.size()) ? g_shaders[(size_t)i] : nullptr;
}
// The library's private streams are recycled across contexts (per device): a process's HIP streams are mapped onto a handful
// of hardware queues in creation order, and a ctx created after others must not end up with its two frame streams -- or a frame
// stream and the caller's -- on one queue (two streams that share a queue run nothing beside each other).  Recycling keeps the
// queues a process's contexts use the same few from the first context to the last.
std::mutex g_stream_mu;
std::map<int, std::vector<hipStream_t>> g_stream_pool;
hipError_t acquire_stream(int device, hipStream_t *out)
{
    {
        std::lock_guard<std::mutex> lk(g_stream_mu);
        std::vector<hipStream_t> &v = g_stream_pool[device];
        if (!v.empty()) { *out = v.back(); v.pop_back(); return hipSuccess; }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}
void release_stream(int device, hipStream_t st)
{
    if (!st) return;
    (void)hipStreamSynchronize(st);
    std::lock_guard<std::mutex> lk(g_stream_mu);
    g_stream_pool[device].push_back(st);
}

struct UserModule {
    hipModule_t mod = nullptr;
    hipFunction_t geom = nullptr, clip = nullptr, sweep = nullptr, span[2][6] = {};
};

// Workspace of one geometry pass / one raster pass.  There are two of each, used alternately (parity of the pass), so
// that the geometry + binning kernels of pass n + 1 can run on the ctx's second stream while the tile kernel of pass n
// still reads what pass n left (frr_device.h: GeomTab / BinTab are the device-side halves of the same scheme).
struct GeomSet {
    uint32_t *block_sums = nullptr; size_t block_sums_cap = 0; // per 256-triangle block: triangles emitted
    uint32_t *block_prefix = nullptr; size_t block_prefix_cap = 0; // ... and their exclusive scan
    uint32_t *tinfo = nullptr; size_t tinfo_cap = 0;           // per input: fan size | emission offset in its block
    uint32_t *fanbase = nullptr; size_t fanbase_cap = 0;       // per clipped input: first fan slot
    uint32_t *fan_okey = nullptr; size_t fan_okey_cap = 0;     // per fan slot: order key within the draw
    RasterRec *recs = nullptr; size_t setup_cap = 0;            // [input triangles + fan capacity]
    float *vary = nullptr; size_t vary_cap = 0;                 // floats
    uint4 *pbox = nullptr; size_t pbox_cap = 0;
    uint32_t *bcount = nullptr; size_t bcount_cap = 0;          // [geometry blocks] dense binning entries per block (GeomArgs::bcount)
    uint2 *clipq = nullptr; size_t clipq_cap = 0;               // [input triangles] the clip kernel's queue (GeomArgs::clipq)
    hipEvent_t reader_ev = nullptr; bool reader_pending = false; // fires when the latest tile kernel that reads this set is done
    hipStream_t reader_stream = nullptr;                         // ... on this stream
    bool reader_recorded = false;                                // the event was recorded right behind that kernel
};
struct BinSet {
    uint4 *bins = nullptr; size_t bin_cap = 0;   // 16-byte cull records, one per (triangle, tile) pair
    uint4 *bins2 = nullptr; size_t bin2_cap = 0; // the same in near-first order per tile (tile kernel pre-pass)
    uint32_t *bin_matrix = nullptr; size_t bin_matrix_cap = 0; // [G][ntiles] per-chunk tile histograms
    hipEvent_t reader_ev = nullptr; bool reader_pending = false;
    hipStream_t reader_stream = nullptr;
    bool reader_recorded = false;
};

struct GeomFilter { bool active; int32_t y0, y1; int rank, world; bool blocked; };

// Everything a command reads and changes on the host.  Every logged command carries the state it started from, so that
// the commands from a failed one onwards can be replayed (finish()).
struct FrameState {
    uint8_t *color = nullptr; float *depth = nullptr; uint32_t *tri_id = nullptr; // the frame targets (own or caller-bound)
    int tset = 0;                                                                 // which of the ctx's own target sets is current
    int rank = 0, world = 1; bool part_blocked = false;                           // tile-row ownership (frr_set_partition*)
    // frr_clear is deferred: the first full-window draw of the span kernel performs it inside the tile kernel
    // (keys start from the clear depth, every pixel of the tile is written); anything else that looks at the
    // targets first settles it with k_clear.  option clear_eager restores the immediate clear.
    bool clear_pending = false;    // targets not cleared yet
    bool unowned_debt = false;     // partitioned ctx: the tile rows of other ranks missed a fused clear
    int debt_tiles_y = 0;          //   (tile rows of the window of the draw that left the debt)
    uint32_t clear_rgba = 0; float clear_depth = 0.0f;
    uint32_t frame_no = 1;         // frr_clear count (device statistics are tagged with it)
    uint64_t tris_in = 0; uint32_t draws = 0;   // statistics the host knows: inputs submitted / geometry passes since frr_clear
    int lane = 0;                  // which of the two sets of device tables this frame's passes use (frr_device.h: Lane)
    int gpars[2] = {0, 0}, bpars[2] = {0, 0};   // per lane: parity of the latest geometry / raster pass (GeomTab / BinTab)
    int gpar() const { return gpars[lane]; }
    int bpar() const { return bpars[lane]; }
    int gset = 0, bset = 0;        // workspace set of the latest geometry / raster pass (GeomSet / BinSet)
    bool on_g = false;             // the latest geometry pass runs on the second stream (its binning follows it there)
    GeomFilter geom_filter = {false, 0, 0, 0, 1, false}; // tile-row ownership filter the latest setup list was built with (frr_draw on a partitioned ctx)
    uint32_t geom_fan_cap = 0;     // fan capacity the latest geometry pass was launched with
    uint32_t geom_nblocks = 0;
    uint32_t geom_seq = 0;         // its sequence number
    bool scan_pending = false;     // its block sums are not scanned yet (geom_scan: by the binning launch, or k_geom_scan)
    int geom_vs = -1;              // VS of the latest frr_geometry
    uint64_t geom_ntris = 0;
    const float *geom_mesh = nullptr; uint64_t geom_mesh_gen = 0, geom_duni_hash = 0;   // ... its mesh and (a digest of) its uniforms: DrawSig
};

struct Cmd {
    enum Kind { GEOM, RASTER } kind;
    FrameState pre;            // host state before the command
    uint32_t seq = 0;          // sequence number of its latest execution
    DevUniforms duni;          // uniforms at the time of the call
    // GEOM
    int mesh = -1; bool filter = false; int32_t fy0 = 0, fy1 = 0;
    // RASTER
    int ps = 0; int32_t x0 = 0, x1 = 0, y0 = 0, y1 = 0; bool count_frags = true;
    int par = 0;               // the parity it ran with (finish(): which table a failed command left behind)
    int set = 0;               // RASTER: the BinSet it ran with
    int lane = 0;              // the lane of device tables it ran in
};

// What decides how much of the work lists (fan space, (triangle, tile) records) a raster pass and its geometry pass need.
struct DrawSig {
    const float *mesh; uint64_t mesh_gen; uint64_t ntris; uint64_t duni_hash; uint32_t join_epoch;
    int32_t vs, x0, x1, y0, y1, rank, world, blocked, filter, fy0, fy1, gset, bset;
};
inline bool same_sig(const DrawSig &a, const DrawSig &b) { return memcmp(&a, &b, sizeof a) == 0; }
inline uint64_t fnv1a(const void *p, size_t n)
{
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) h = (h ^ ((const unsigned char *)p)[i]) * 1099511628211ull;
    return h;
}

} // namespace

struct frr_ctx {
    int device = 0;
    uint32_t W = 0, H = 0;
    hipStream_t stream = nullptr;   // the caller's stream (or a private one): tile kernels, clears, copies -- everything that touches the targets
    hipStream_t gstream = nullptr;  // private: geometry + binning of the next pass, beside the tile kernel of the current one
    bool own_stream = false;
    int overlap = 2;                // option overlap: 0 never use gstream, 1 always, 2 (default) for passes with varyings (exec_geometry)
    bool g_used = false;            // some pass has run on gstream since the streams were last drained (cross-stream events are needed)
    // The ctx's own frame targets: TWO sets.  A frame that starts with frr_clear on own targets takes the other set and
    // its tile kernels the other tile stream (tstream2 for set 1), so that the tile kernel of frame n + 1 fills the drain
    // of frame n's (2,040 tiles on 1,536 workgroup slots end with a third of the chip idle).  Nobody can look at own targets
    // except through this library (frr_readback, frr_target_ptrs, which join the streams first), so the only visible
    // change is that frr_target_ptrs' pointers are those of the CURRENT frame.  Caller-bound targets: one set, one stream.
    uint8_t *own_color[2] = {}; float *own_depth[2] = {}; uint32_t *own_tri_id[2] = {};
    hipStream_t tstream2 = nullptr;  // everything of the frames that use own target set 1 (or, option bound_targets_in_flight, of every other frame)
    hipStream_t tstream1 = nullptr;  // option bound_targets_in_flight: the frames in between (the caller's stream then carries no frame work at all)
    hipEvent_t ev_t2 = nullptr, ev_t1 = nullptr;   // join them into a caller's stream
    bool t2_dirty = false, t1_dirty = false;       // they hold work `stream` has not waited for
    bool t2_xdirty = false, t1_xdirty = false;     // ... that the stream of the latest frr_frame_fence has not waited for
    hipStream_t xfence = nullptr;                  //     (that stream)
    bool bound_in_flight = false;    // option bound_targets_in_flight: frames on caller-bound targets alternate between the two private streams too;
                                     // the caller binds another target set for each of two consecutive frames and fences its reads (frr_frame_fence)
    int frames_in_flight = 2;        // option frames_in_flight (1: one target set, everything on the caller's stream)
    Counters *cnt = nullptr;
    FrameState fs;
    GeomSet gset[2];
    BinSet bset[2];
    // cross-stream ordering: every workspace set has an event that fires when the latest tile kernel reading it is done (the second
    // stream waits for it before it overwrites a workspace that kernel reads), ev_bin[k & 3] when binning k is done
    hipEvent_t ev_bin[4] = {}, ev_join = nullptr;
    uint64_t bin_serial = 0;
    // inputs the caller wrote on `stream` (a device-bound mesh) must be visible to the ctx's private streams: every bind
    // starts a new epoch, and a private stream waits for `stream` once per epoch before its next geometry pass
    uint32_t join_epoch = 1, joined_g = 0, joined_t1 = 0, joined_t2 = 0;
    // command log since the last synchronisation point / frr_clear (finish(): replay)
    std::vector<Cmd> log;
    uint32_t next_seq = 1, epoch = 1;
    uint32_t replays = 0;           // replays since frr_clear (frr_stats.replays)
    bool in_replay = false;
    Counters hc;                    // host copy of the device counters as of the latest finish()
    uint32_t *host_bad = nullptr;   // host-visible word the device writes a failed command's number to (Counters::host_bad)
    uint32_t seen_bad = SEQ_NONE;   // its value when the host last looked
    // A draw cannot fail, whoever consumes its targets and however (stream order, frr_frame_fence, frr_sync): a raster pass whose
    // need of the work lists is not known to fit -- no pass with the same DrawSig has completed on the current lists -- is
    // VERIFIED before frr_raster returns: the host waits for the pass's binning launch (not for its tile kernel), looks at
    // host_bad, and repairs (finish(): grow + replay) if the pass or its geometry overflowed.  A proven pass is not waited for.
    std::vector<DrawSig> proven;
    uint64_t mesh_gen = 0;          // meshes registered so far (Mesh::gen)
    hipEvent_t ev_verify = nullptr;
    bool verify_pending = false;
    DrawSig verify_sig;
    // own targets handed out (frr_target_ptrs / frr_frame_fence): the frame that next renders into that set waits for what
    // the stream they were handed to holds by then (the caller's reads), see frr_clear
    bool exported[2] = {false, false};
    hipStream_t export_stream[2] = {nullptr, nullptr};
    hipEvent_t ev_export = nullptr;
    // frr_frame_wait: the next kernel that writes the frame targets waits for this event
    hipEvent_t ev_wait = nullptr;
    bool wait_pending = false;
    size_t fan_hint = 0;       // fan capacity asked for by a draw that overflowed
    int bin_g = 0;             // option bin_chunks: override the number of binning chunks (dev)
    uint32_t ent_slot_override = 0; // option tile_slot_records: per-tile slot of bins2 in records (tests of the overflow arena)
    bool clear_eager = false;
    bool bin_atomics = false;  // option bin_atomics: force the global-atomic binning fallback (tests)
    size_t bin_cap_init = 0;   // option bin_capacity: initial capacity of the (triangle, tile) lists in records (overflow / replay tests)
    size_t fan_cap_init = 0;   // option fan_capacity: initial fan capacity (the same)
    int clip_queue = -1;        // option clip_queue: 1 use the queue + k_geom_clip, 0 never, -1 when the latest counters read back
    bool clip_queue_auto = false; //   showed a block with more than CLIP_QUEUE_AT clipped inputs (results are the same either way)
    // binning workspace of the CSR fallback (one set: that path does not overlap)
    uint32_t *tile_counts = nullptr, *tile_offsets = nullptr, *tile_cursor = nullptr;
    uint32_t max_tiles = 0;
    bool lds_attr_set = false;
    std::vector<Mesh> meshes;
    std::map<int, UserModule> user_modules;   // user shader id -> its code object loaded on this ctx's device
    Texture tex[FRR_MAX_TEXTURES];
    frr_uniforms uni;
    DevUniforms duni;
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


// the stream of everything that touches the current frame targets
bool own_targets(const frr_ctx *c)
{
    const FrameState &f = c->fs;
    return f.color == c->own_color[f.tset] && f.depth == c->own_depth[f.tset] && f.tri_id == c->own_tri_id[f.tset];
}
// do consecutive frames (frr_clear) alternate between two streams / target sets / workspace sets / lanes?
bool frames_alternate(const frr_ctx *c) { return c->frames_in_flight == 2 && (own_targets(c) || c->bound_in_flight); }
hipStream_t tstream_of(const frr_ctx *c)
{
    if (c->bound_in_flight && !own_targets(c) && c->tstream1 && c->tstream2) return c->fs.tset ? c->tstream2 : c->tstream1;
    return (c->tstream2 && c->fs.tset == 1 && own_targets(c)) ? c->tstream2 : c->stream;
}
// the stream of the latest geometry pass and of the binning that follows it: the second stream, or the targets' stream
hipStream_t gstream_of(const frr_ctx *c) { return c->fs.on_g ? c->gstream : tstream_of(c); }
// do passes ever run beside each other on this ctx (workspace sets then carry events)?
bool multi_stream(const frr_ctx *c) { return c->g_used || c->tstream2 != nullptr || c->tstream1 != nullptr; }

// the ctx's own target set t (the second one is allocated on first use)
int ensure_own_set(frr_ctx *c, int t)
{
    if (c->own_color[t] && c->own_depth[t] && c->own_tri_id[t]) return FRR_OK;
    const size_t npx = (size_t)c->W * c->H;
    const bool ok = (c->own_color[t] || hipMalloc((void **)&c->own_color[t], npx * 4) == hipSuccess) &&
                    (c->own_depth[t] || hipMalloc((void **)&c->own_depth[t], npx * 4) == hipSuccess) &&
                    (c->own_tri_id[t] || hipMalloc((void **)&c->own_tri_id[t], npx * 4) == hipSuccess);
    if (!ok) return fail(c, FRR_ERR_NOMEM, "second target set");
    return FRR_OK;
}

// all streams idle
int drain(frr_ctx *c)
{
    if (c->gstream) HIP_TRY(c, hipStreamSynchronize(c->gstream));
    if (c->tstream2) HIP_TRY(c, hipStreamSynchronize(c->tstream2));
    if (c->tstream1) HIP_TRY(c, hipStreamSynchronize(c->tstream1));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->t2_dirty = c->t1_dirty = c->t2_xdirty = c->t1_xdirty = false;
    c->g_used = false;
    for (GeomSet &S : c->gset) S.reader_pending = false;
    for (BinSet &B : c->bset) B.reader_pending = false;
    return FRR_OK;
}
// a caller's stream waits for the ctx's frame streams: what is enqueued on it next sees every frame issued so far
int fence_stream(frr_ctx *c, hipStream_t st)
{
    const bool own = st == c->stream;
    if (!own && st != c->xfence) { c->xfence = st; c->t1_xdirty = c->tstream1 != nullptr; c->t2_xdirty = c->tstream2 != nullptr; }
    if (c->tstream2 && (own ? c->t2_dirty : c->t2_xdirty)) {
        HIP_TRY(c, hipEventRecord(c->ev_t2, c->tstream2));
        HIP_TRY(c, hipStreamWaitEvent(st, c->ev_t2, 0));
    }
    if (c->tstream1 && (own ? c->t1_dirty : c->t1_xdirty)) {
        HIP_TRY(c, hipEventRecord(c->ev_t1, c->tstream1));
        HIP_TRY(c, hipStreamWaitEvent(st, c->ev_t1, 0));
    }
    if (own) c->t2_dirty = c->t1_dirty = false;
    else {
        c->t2_xdirty = c->t1_xdirty = false;
        if (!(c->bound_in_flight && c->tstream1)) {   // (frames may have run on the ctx's stream itself)
            HIP_TRY(c, hipEventRecord(c->ev_join, c->stream));
            HIP_TRY(c, hipStreamWaitEvent(st, c->ev_join, 0));
        }
    }
    return FRR_OK;
}
int join_tile_streams(frr_ctx *c) { return fence_stream(c, c->stream); }

template <typename T> int ensure(frr_ctx *c, T *&p, size_t &cap, size_t need)
{
    if (need <= cap && p) return FRR_OK;
    if (p) { int rc = drain(c); if (rc != FRR_OK) return rc; HIP_TRY(c, hipFree(p)); p = nullptr; cap = 0; }
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, need * sizeof(T));
    if (e != hipSuccess) return fail(c, FRR_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    p = (T *)q;
    cap = need;
    return FRR_OK;
}

// ---- cross-stream ordering (no-ops when everything runs on one stream) ----------------------------------------------
// the second stream waits until the latest tile kernel that reads a workspace set is done
template <class SET> int gstream_wait_readers(frr_ctx *c, SET &S)
{
    if (!S.reader_pending) return FRR_OK;
    S.reader_pending = false;
    if (S.reader_stream == gstream_of(c)) return FRR_OK;   // same stream: already in order
    // (the event may not have been recorded at the launch: then now, after whatever else the reader's stream has been given
    // since -- more than needed, and right for the rare change of pattern this serves)
    if (!S.reader_recorded) HIP_TRY(c, hipEventRecord(S.reader_ev, S.reader_stream));
    HIP_TRY(c, hipStreamWaitEvent(gstream_of(c), S.reader_ev, 0));
    return FRR_OK;
}
// the stream of the geometry pass about to be issued waits for everything the caller's stream holds so far (mesh data
// written by the caller before it bound the mesh), once per bind
int gstream_join(frr_ctx *c)
{
    hipStream_t st = gstream_of(c);
    if (st == c->stream) return FRR_OK;
    uint32_t &joined = st == c->gstream ? c->joined_g : (st == c->tstream1 ? c->joined_t1 : c->joined_t2);
    if (joined == c->join_epoch) return FRR_OK;
    HIP_TRY(c, hipEventRecord(c->ev_join, c->stream));
    HIP_TRY(c, hipStreamWaitEvent(st, c->ev_join, 0));
    joined = c->join_epoch;
    return FRR_OK;
}
// the tile stream waits for what the second stream holds so far (binning -> tile kernel)
int tstream_wait_gstream(frr_ctx *c)
{
    if (gstream_of(c) == tstream_of(c)) return FRR_OK;
    hipEvent_t e = c->ev_bin[++c->bin_serial & 3];
    HIP_TRY(c, hipEventRecord(e, gstream_of(c)));
    HIP_TRY(c, hipStreamWaitEvent(tstream_of(c), e, 0));
    return FRR_OK;
}
// a tile kernel has just been launched: the workspace sets it reads are busy until it is done
int tile_launched(frr_ctx *c, GeomSet &gs, BinSet &bs)
{
    hipStream_t ts = tstream_of(c);
    if (ts == c->tstream2) c->t2_dirty = c->t2_xdirty = true;
    if (ts == c->tstream1) c->t1_dirty = c->t1_xdirty = true;
    if (!multi_stream(c)) return FRR_OK;   // everything has run on one stream so far
    // Frames in flight give every frame stream its own workspace sets: the next writer of a set is on the reader's stream, and
    // no event is needed (should the pattern change, gstream_wait_readers records one late).  With geometry + binning on the
    // second stream the event is what lets that stream run beside the NEXT tile kernel: recorded right here.
    const bool eager = c->g_used;
    gs.reader_pending = bs.reader_pending = true; gs.reader_stream = bs.reader_stream = ts; gs.reader_recorded = bs.reader_recorded = eager;
    if (eager) { HIP_TRY(c, hipEventRecord(gs.reader_ev, ts)); HIP_TRY(c, hipEventRecord(bs.reader_ev, ts)); }
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
    frr_ctx *c; int kid; hipStream_t st; hipEvent_t a = nullptr;
    ProfScope(frr_ctx *c_, int kid_, hipStream_t st_) : c(c_), kid(kid_), st(st_)
    {
        if ((c->prof_mask & (1u << kid)) && (c->prof_seen[kid]++ % c->prof_period) == 0) {
            a = get_event(c);
            if (a && hipEventRecord(a, st) != hipSuccess) { c->ev_pool.push_back(a); a = nullptr; }
        }
    }
    ~ProfScope()
    {
        // a sample whose events could not be created or recorded is dropped (frr_profile_get then reports fewer launches)
        if (!a) return;
        hipEvent_t b = get_event(c);
        if (b && hipEventRecord(b, st) == hipSuccess) { c->prof_pending.push_back({kid, a, b}); return; }
        c->ev_pool.push_back(a);
        if (b) c->ev_pool.push_back(b);
    }
};
void prof_collect(frr_ctx *c)
{
    if (c->prof_pending.empty()) return;
    (void)drain(c);
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
    for (int k = 0; k < FRR_MAX_TEXTURES; ++k) { d.slot_tex[k] = c->tex[k].dev; d.slot_w[k] = c->tex[k].w; d.slot_h[k] = c->tex[k].h; }
}

// Tile rows [t0, t1) of rank `rank` in the blocked layout: the first tiles_y % world ranks own one row more than the
// others, so that no rank of world <= tiles_y is left without rows (34 rows over 8 ranks: 5,5,4,4,4,4,4,4)
void blocked_rows(int tiles_y, int rank, int world, int *t0, int *t1)
{
    const int q = tiles_y / world, r = tiles_y % world;
    *t0 = rank * q + std::min(rank, r);
    *t1 = *t0 + q + (rank < r ? 1 : 0);
}

// the code object of user shader `id` on this ctx's device
int user_module(frr_ctx *c, int id, const UserModule **out)
{
    auto it = c->user_modules.find(id);
    if (it != c->user_modules.end()) { *out = &it->second; return FRR_OK; }
    const UserShader *us = user_shader(id);
    if (!us) return fail(c, FRR_ERR_INVALID, "unknown shader id");
    UserModule m;
    HIP_TRY(c, hipModuleLoadData(&m.mod, us->code.data()));
    HIP_TRY(c, hipModuleGetFunction(&m.geom, m.mod, us->geom.c_str()));
    HIP_TRY(c, hipModuleGetFunction(&m.clip, m.mod, us->clip.c_str()));
    HIP_TRY(c, hipModuleGetFunction(&m.sweep, m.mod, us->sweep.c_str()));
    for (int cnt = 0; cnt < 2; ++cnt)
        for (int sh = 0; sh < 6; ++sh) HIP_TRY(c, hipModuleGetFunction(&m.span[cnt][sh], m.mod, us->span[cnt][sh].c_str()));
    *out = &(c->user_modules[id] = m);
    return FRR_OK;
}

constexpr uint32_t kClipGrid = 2048;      // workgroups of k_geom_clip (four wavefronts each, one triangle per wavefront and step)

template <int VS> void launch_geometry(frr_ctx *c, GeomArgs &g, uint32_t nblocks, const DevUniforms &du)
{
    hipStream_t st = gstream_of(c);
    ProfScope p(c, KID_GEOM, st);
    hipLaunchKernelGGL(k_geom_single<VS>, dim3(nblocks), dim3(GEOM_BLOCK), 0, st, g, du);
    if (g.use_clipq) hipLaunchKernelGGL(k_geom_clip<VS>, dim3(std::min<uint32_t>(kClipGrid, nblocks * 4u)), dim3(GEOM_BLOCK), 0, st, g, du);
}

// the geometry kernel for the VS of the mesh
void launch_geometry_vs(frr_ctx *c, GeomArgs &g, uint32_t nblocks, int vs, const DevUniforms &du, const UserModule *um)
{
    if (um) {   // a user shader: the same kernels, compiled with the user's functions (frr_shader_register)
        hipStream_t st = gstream_of(c);
        ProfScope p(c, KID_GEOM, st);
        DevUniforms d = du;
        void *args[] = {&g, &d};
        (void)hipModuleLaunchKernel(um->geom, nblocks, 1, 1, GEOM_BLOCK, 1, 1, 0, st, args, nullptr);
        if (g.use_clipq) (void)hipModuleLaunchKernel(um->clip, std::min<uint32_t>(kClipGrid, nblocks * 4u), 1, 1, GEOM_BLOCK, 1, 1, 0, st, args, nullptr);
        return;
    }
    switch (vs) {
    case FRR_VS_CLIP: launch_geometry<FRR_VS_CLIP>(c, g, nblocks, du); break;
    case FRR_VS_CLIP_COLOR: launch_geometry<FRR_VS_CLIP_COLOR>(c, g, nblocks, du); break;
    case FRR_VS_PHONG: launch_geometry<FRR_VS_PHONG>(c, g, nblocks, du); break;
    case FRR_VS_GOURAUD: launch_geometry<FRR_VS_GOURAUD>(c, g, nblocks, du); break;
    }
}
// the latest geometry pass's block sums -> prefix (+ n_emit, the fan-capacity flag), if no binning launch has done it
int scan_now(frr_ctx *c)
{
    FrameState &f = c->fs;
    if (!f.scan_pending) return FRR_OK;
    hipStream_t st = gstream_of(c);
    { ProfScope p(c, KID_GEOM_SCAN, st); hipLaunchKernelGGL(k_geom_scan, dim3(1), dim3(1024), 0, st, c->gset[f.gset].block_sums, c->gset[f.gset].block_prefix, f.geom_nblocks, c->cnt, f.lane, f.gpar(), f.geom_fan_cap, f.geom_seq, c->epoch); }
    HIP_TRY(c, hipGetLastError());
    f.scan_pending = false;
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

template <int K, int PS> void launch_raster(frr_ctx *c, const RasterArgs &a, uint32_t grid, const SpanShape sh, const DevUniforms &du, bool count_frags)
{
    hipStream_t ts = tstream_of(c);
    ProfScope p(c, KID_RASTER, ts);
    if (c->raster_sweep) {
        hipLaunchKernelGGL((k_raster<K, PS>), dim3(grid), dim3(256), 0, ts, a, du);
    } else {
        // the span algebra needs every coordinate it touches within +-SPAN_SAFE (no i32 wrap)
        const int win_safe = a.x0 >= -SPAN_SAFE && a.y0 >= -SPAN_SAFE && a.x1 <= SPAN_SAFE && a.y1 <= SPAN_SAFE;
        auto go = [&](auto count_tag, auto nw_tag, auto occ_tag) {
            constexpr bool CNT = decltype(count_tag)::value;
            constexpr int NWV = decltype(nw_tag)::value, OCCV = decltype(occ_tag)::value;
            hipLaunchKernelGGL((k_raster_span<K, PS, CNT, NWV, OCCV>), dim3(grid), dim3(NWV * 64), 0, ts, a, du, win_safe);
        };
        auto go_nw = [&](auto count_tag) {
            if (sh.nw == LIGHT_NW) go(count_tag, std::integral_constant<int, LIGHT_NW>{}, std::integral_constant<int, 6>{});
            else if (sh.nw == 4 && sh.occ == 8) go(count_tag, std::integral_constant<int, 4>{}, std::integral_constant<int, 8>{});
            else if (sh.nw == 4) go(count_tag, std::integral_constant<int, 4>{}, std::integral_constant<int, 6>{});
            else if (sh.nw == 6) go(count_tag, std::integral_constant<int, 6>{}, std::integral_constant<int, 6>{});
            else if (sh.nw == 8) go(count_tag, std::integral_constant<int, 8>{}, std::integral_constant<int, 6>{});
            else go(count_tag, std::integral_constant<int, 16>{}, std::integral_constant<int, 4>{});
        };
        if (count_frags) go_nw(std::true_type{}); else go_nw(std::false_type{});
    }
}

// the clear itself (k_clear), on the caller's stream
int clear_now(frr_ctx *c, uint32_t packed, float depth)
{
    const FrameState &f = c->fs;
    const uint32_t n = c->W * c->H, n4 = n / 4;
    hipStream_t ts = tstream_of(c);
    if (ts == c->tstream2) c->t2_dirty = c->t2_xdirty = true;
    if (ts == c->tstream1) c->t1_dirty = c->t1_xdirty = true;
    if (c->wait_pending) { HIP_TRY(c, hipStreamWaitEvent(ts, c->ev_wait, 0)); c->wait_pending = false; }   // frr_frame_wait
    {
        ProfScope p(c, KID_CLEAR, ts);
        uint32_t grid = std::min<uint32_t>((n4 + 255) / 256, 2048);
        hipLaunchKernelGGL(k_clear, dim3(grid ? grid : 1), dim3(256), 0, ts, (uint4 *)f.color, (uint4 *)f.depth,
                           (uint4 *)f.tri_id, n4, packed, depth);
        if (n4 * 4 < n)
            hipLaunchKernelGGL(k_clear_tail, dim3(1), dim3(64), 0, ts, (uint32_t *)f.color, (uint32_t *)f.depth,
                               f.tri_id, n4 * 4, n, packed, depth);
    }
    HIP_TRY(c, hipGetLastError());
    return FRR_OK;
}

// bring the targets to the state the API promises (a pending frr_clear; rows a fused clear of a partitioned ctx skipped)
int settle_targets(frr_ctx *c)
{
    FrameState &f = c->fs;
    if (f.clear_pending) {
        int rc = clear_now(c, f.clear_rgba, f.clear_depth);
        if (rc != FRR_OK) return rc;
        f.clear_pending = f.unowned_debt = false;
    } else if (f.unowned_debt) {
        RowOwner own = {f.rank, f.world, f.part_blocked ? 1 : 0, 0, 0};
        if (own.blocked) blocked_rows(f.debt_tiles_y, f.rank, f.world, &own.brow0, &own.brow1);
        if (tstream_of(c) == c->tstream2) c->t2_dirty = c->t2_xdirty = true;
        if (tstream_of(c) == c->tstream1) c->t1_dirty = c->t1_xdirty = true;
        hipLaunchKernelGGL(k_clear_unowned_rows, dim3(c->H), dim3(256), 0, tstream_of(c), (uint32_t *)f.color, (uint32_t *)f.depth,
                           f.tri_id, c->W, c->H, own, f.clear_rgba, f.clear_depth);
        HIP_TRY(c, hipGetLastError());
        f.unowned_debt = false;
    }
    return FRR_OK;
}
// ... and the tables of the latest geometry pass (called by everything that looks at either)
int settle(frr_ctx *c)
{
    HIP_TRY(c, hipSetDevice(c->device));
    { int rcs = scan_now(c); if (rcs != FRR_OK) return rcs; }   // n_emit of the latest pass (statistics, setup read-back)
    return settle_targets(c);
}

// ---- the two commands ------------------------------------------------------------------------------------------------
// Loop A (phong.rs:321-331) over the mesh: one geometry pass.  Nothing of the host state is committed before the last
// step that can fail.
int exec_geometry(frr_ctx *c, Cmd &cmd)
{
    FrameState &f = c->fs;
    const Mesh &m = c->meshes[cmd.mesh];
    const int K = frr_vs_num_varyings(m.vs);
    const uint64_t nt = m.ntris;
    const int par = f.gpar() ^ 1;
    // Beside the previous pass's tile kernel (second stream, the other workspace set) or after it (the targets' stream,
    // set 0: one set stays hot in the 256 MB Infinity Cache -- two sets of the 1M-triangle frame do not, which costs its
    // tile kernel 4 us).  Measured (profiles/r03_overlap_modes.txt): running beside pays for passes with varyings, whose
    // tile kernel spends long stretches shading (4K textured frame -6 %, a rank of 8 of it -10 %), not for depth-only ones.
    const bool fif2 = frames_alternate(c);   // consecutive frames already run beside each other, on two streams (frr_clear)
    const bool on_g = !fif2 && (c->overlap == 1 || (c->overlap == 2 && K > 0));
    const int si = on_g ? (f.gset ^ 1) : (fif2 ? f.tset : 0);
    GeomSet &S = c->gset[si];
    int rc;
    const UserModule *um = nullptr;
    if (m.vs >= FRR_SHADER_USER_BASE && (rc = user_module(c, m.vs, &um)) != FRR_OK) return rc;
    // fan space: clipped inputs are the ones that straddle a frustum plane, usually few; room for as many fan triangles
    // as there are inputs (+ 4096) to start with, grown on demand (the pass then fails on the device and finish() replays
    // it with more) up to the worst case of 19 per input (small meshes get their worst case outright: 2^20 slots are cheap) ...
    const uint32_t nblocks = (uint32_t)((nt + GEOM_BLOCK - 1) / GEOM_BLOCK);
    // ... in FAN_REGIONS regions (block b allocates in region b % FAN_REGIONS: frr_device.h); a region never needs more
    // than 19 slots for every input of the blocks that use it
    const uint64_t region_worst = (uint64_t)FRR_MAX_OUT_TRIS * GEOM_BLOCK * ((nblocks + FAN_REGIONS - 1) / FAN_REGIONS);
    uint64_t first_guess = std::max<uint64_t>((nt + 4096 + FAN_REGIONS - 1) / FAN_REGIONS, std::min<uint64_t>(region_worst, (1u << 20) / FAN_REGIONS));
    if (c->fan_cap_init) first_guess = (c->fan_cap_init + FAN_REGIONS - 1) / FAN_REGIONS;   // option fan_capacity (tests of the replay)
    uint64_t region = std::max<uint64_t>(first_guess, (c->fan_hint + FAN_REGIONS - 1) / FAN_REGIONS);
    region = std::max<uint64_t>(std::min<uint64_t>(region, region_worst), 1);
    uint64_t fan_cap = region * FAN_REGIONS;
    if (nt + fan_cap > 0xFFFFFFF0ull) fan_cap = (0xFFFFFFF0ull - nt) / FAN_REGIONS * FAN_REGIONS;
    const size_t slots = (size_t)(nt + fan_cap);
    if ((rc = ensure(c, S.block_sums, S.block_sums_cap, (size_t)nblocks + 1)) != FRR_OK) return rc;
    if ((rc = ensure(c, S.block_prefix, S.block_prefix_cap, (size_t)nblocks + 1)) != FRR_OK) return rc;
    if ((rc = ensure(c, S.tinfo, S.tinfo_cap, (size_t)std::max<uint64_t>(nt, 1))) != FRR_OK) return rc;
    if ((rc = ensure(c, S.fanbase, S.fanbase_cap, (size_t)std::max<uint64_t>(nt, 1))) != FRR_OK) return rc;
    if ((rc = ensure(c, S.fan_okey, S.fan_okey_cap, (size_t)std::max<uint64_t>(fan_cap, 1))) != FRR_OK) return rc;
    if ((rc = ensure(c, S.recs, S.setup_cap, std::max<size_t>(slots, 1024))) != FRR_OK) return rc;
    if ((rc = ensure(c, S.pbox, S.pbox_cap, S.setup_cap)) != FRR_OK) return rc;
    if ((rc = ensure(c, S.bcount, S.bcount_cap, (size_t)nblocks + 1)) != FRR_OK) return rc;
    if (K > 0 && (rc = ensure(c, S.vary, S.vary_cap, (size_t)S.setup_cap * 3 * std::max(K, 8) /* (K <= 8 in the shader table) */)) != FRR_OK) return rc;
    const bool use_clipq = nt > 0 && (c->clip_queue > 0 || (c->clip_queue < 0 && c->clip_queue_auto));
    if (use_clipq && (rc = ensure(c, S.clipq, S.clipq_cap, (size_t)nt)) != FRR_OK) return rc;
    if ((rc = scan_now(c)) != FRR_OK) return rc;   // the previous pass's n_emit feeds this pass's tri_base
    if (on_g && !c->gstream) {
        // created on first use: a process maps its HIP streams onto a handful of hardware queues, and two streams that
        // share one run nothing beside each other -- a ctx that never needs this stream does not take a queue for it
        HIP_TRY(c, acquire_stream(c->device, &c->gstream));
    }
    if (on_g != f.on_g) {
        // this pass changes streams: it follows the previous pass's geometry + binning (tri_base, fan cursors)
        hipEvent_t e = c->ev_bin[++c->bin_serial & 3];
        HIP_TRY(c, hipEventRecord(e, gstream_of(c)));
        HIP_TRY(c, hipStreamWaitEvent(on_g ? c->gstream : tstream_of(c), e, 0));
    }
    f.on_g = on_g;
    if (on_g) c->g_used = true;
    // second stream: after whatever the caller's stream holds that this pass may read (first use), and after the tile
    // kernel that last read this workspace
    if ((rc = gstream_join(c)) != FRR_OK) return rc;
    if ((rc = gstream_wait_readers(c, S)) != FRR_OK) return rc;
    GeomArgs g;
    g.in = m.dev; g.ntris = (uint32_t)nt; g.width = c->W; g.height = c->H;
    g.fan_cap = (uint32_t)fan_cap;
    g.seq = cmd.seq; g.epoch = c->epoch; g.frame_no = f.frame_no;
    g.part_rank = f.rank; g.part_world = cmd.filter ? f.world : 1; g.part_y0 = cmd.fy0; g.part_y1 = cmd.fy1;
    g.part_blocked = 0; g.part_brow0 = g.part_brow1 = 0;
    if (cmd.filter && f.part_blocked) {
        const int tiles_y = (int)(((int64_t)cmd.fy1 - cmd.fy0 + TILE - 1) / TILE);
        g.part_blocked = 1;
        blocked_rows(tiles_y, f.rank, f.world, &g.part_brow0, &g.part_brow1);
    }
    g.gpar = par; g.lane = f.lane;
    g.block_sums = S.block_sums; g.block_prefix = S.block_prefix; g.tinfo = S.tinfo; g.fanbase = S.fanbase; g.fan_okey = S.fan_okey;
    g.recs = S.recs; g.vary = S.vary; g.pbox = S.pbox; g.cnt = c->cnt;
    g.clipq = S.clipq; g.use_clipq = use_clipq ? 1 : 0;
    g.bcount = S.bcount;
    // what the setup list about to be built was filtered by (frr_raster / frr_readback_setup check it)
    f.geom_filter = GeomFilter{cmd.filter, cmd.fy0, cmd.fy1, f.rank, f.world, f.part_blocked};
    f.gpars[f.lane] = par; cmd.par = par; cmd.lane = f.lane; f.gset = si;
    f.geom_fan_cap = (uint32_t)fan_cap;
    f.geom_nblocks = nblocks;
    f.geom_seq = cmd.seq;
    f.geom_vs = m.vs; f.geom_ntris = nt;
    f.geom_mesh = m.dev; f.geom_mesh_gen = m.gen;   // (the uniforms' float fields: the struct has padding in front of its texture pointer)
    f.geom_duni_hash = fnv1a(&cmd.duni, offsetof(DevUniforms, flat_color) + sizeof cmd.duni.flat_color) ^ (fnv1a(cmd.duni.user, sizeof cmd.duni.user) * 31u);
    f.tris_in += nt; f.draws += 1;
    if (nt == 0) {
        hipLaunchKernelGGL(k_geom_empty, dim3(1), dim3(64), 0, gstream_of(c), g);
    } else {
        launch_geometry_vs(c, g, nblocks, m.vs, cmd.duni, um);
        f.scan_pending = true;
    }
    HIP_TRY(c, hipGetLastError());
    return FRR_OK;
}

// Loop B (phong.rs:361-381): binning + tile kernel over the latest geometry pass, window (x0,x1) x (y0,y1)
int exec_raster(frr_ctx *c, Cmd &cmd)
{
    FrameState &f = c->fs;
    const int32_t x0 = cmd.x0, x1 = cmd.x1, y0 = cmd.y0, y1 = cmd.y1;
    const int ps_id = cmd.ps;
    const int64_t ww = (int64_t)x1 - x0, wh = (int64_t)y1 - y0;
    const UserModule *um = nullptr;
    if (ps_id >= FRR_SHADER_USER_BASE) { const int rcu = user_module(c, ps_id, &um); if (rcu != FRR_OK) return rcu; }
    bool fuse = false;
    if (f.clear_pending) {
        const bool full = x0 == 0 && y0 == 0 && x1 == (int32_t)c->W && y1 == (int32_t)c->H;
        if (full && !c->raster_sweep) fuse = true; // the tile kernel performs the clear
        else { int rcs = settle_targets(c); if (rcs != FRR_OK) return rcs; }
    }
    GeomSet &S = c->gset[f.gset];
    RasterArgs a;
    a.fused_clear = fuse ? 1 : 0; a.clear_rgba = f.clear_rgba; a.clear_depth = f.clear_depth;
    a.x0 = x0; a.x1 = x1; a.y0 = y0; a.y1 = y1; a.win_w = (int)ww; a.win_h = (int)wh;
    a.cstride = (int)c->W; a.dstride = x1;
    a.tiles_x = (int)((ww + TILE - 1) / TILE); a.tiles_y = (int)((wh + TILE - 1) / TILE);
    a.tiles_x_magic = 0u; // set below once the grid is known (exact only for block indices and tile counts < 2^16)
    a.rank = f.rank; a.world = f.world;
    a.blocked = (f.part_blocked && f.world > 1) ? 1 : 0; a.brow0 = a.brow1 = 0;
    if (a.blocked) blocked_rows(a.tiles_y, a.rank, a.world, &a.brow0, &a.brow1);
    a.recs = S.recs; a.vary = S.vary; a.pbox = S.pbox; a.bcount = S.bcount;
    a.tinfo = S.tinfo; a.fanbase = S.fanbase; a.fan_okey = S.fan_okey; a.block_prefix = S.block_prefix; a.ntris_draw = (uint32_t)f.geom_ntris;
    a.tile_counts = c->tile_counts; a.tile_offsets = c->tile_offsets; a.tile_cursor = c->tile_cursor;
    a.gpar = f.gpar(); a.lane = f.lane; a.bpar = 0; a.seq = cmd.seq; a.epoch = c->epoch; a.frame_no = f.frame_no; a.geom_seq = f.geom_seq;
    const uint32_t ntiles = (uint32_t)a.tiles_x * a.tiles_y;
    a.color = f.color; a.depth = f.depth; a.tri_id = f.tri_id; a.cnt = c->cnt;
#ifdef FRR_DEBUG_COUNTERS
    if (!c->dbg_tiles && getenv("FRR_DEBUG_TILES")) {
        if (hipMalloc((void **)&c->dbg_tiles, (size_t)c->max_tiles * 64) != hipSuccess) c->dbg_tiles = nullptr;
    }
    if (c->dbg_tiles) (void)hipMemsetAsync(c->dbg_tiles, 0, (size_t)c->max_tiles * 64, tstream_of(c));
    a.dbg_tiles = c->dbg_tiles;
#endif
    a.seg = nullptr; a.nseg = 0;
    const int owned_rows = a.blocked ? a.brow1 - a.brow0 : (a.tiles_y > a.rank ? (a.tiles_y - a.rank + a.world - 1) / a.world : 0);
    const uint32_t grid = (uint32_t)a.tiles_x * owned_rows;
    if (a.tiles_x >= 2 && a.tiles_x < 65536 && grid < 65536u) a.tiles_x_magic = (uint32_t)(0x100000000ull / (uint64_t)a.tiles_x + 1ull);
    const SpanShape sh = span_shape(c, grid, f.geom_ntris, ps_id);
    const bool segmented = grid <= BIN_LDS_MAX_TILES && !c->bin_atomics && !c->raster_sweep;
    const int q = segmented ? (f.bpar() ^ 1) : 0;   // (the CSR fallback has one set of tile tables: it uses workspace 0 and overlaps nothing)
    const int bi = !segmented ? 0 : f.on_g ? (f.bset ^ 1) : (frames_alternate(c) ? f.tset : 0);
    BinSet &B = c->bset[bi];
    int rc;
    if (!B.bins) {
        size_t want = std::max<size_t>((size_t)f.geom_ntris * 8 + 4 * (size_t)c->max_tiles, (size_t)1 << 22);
        if (c->bin_cap_init) want = c->bin_cap_init;      // option bin_capacity (tests of the replay)
        want = std::max(want, c->bset[bi ^ 1].bin_cap);   // (what the other workspace has grown to)
        if ((rc = ensure(c, B.bins, B.bin_cap, want)) != FRR_OK) return rc;
        if ((rc = ensure(c, B.bins2, B.bin2_cap, want)) != FRR_OK) return rc;
    }
    a.bins2 = B.bins2;
    a.bins = B.bins; a.bin_cap = (uint32_t)std::min<size_t>(B.bin_cap, 0xFFFFFFFFu);
    hipStream_t gs = gstream_of(c);
    if (segmented) {
        const uint32_t ltiles = std::max<uint32_t>(grid, 1u);   // the binning numbers the rank's OWN tiles only (local_tile_row)
        // segmented LDS multi-split (one launch, no per-entry global atomics): G chunk workgroups, ~3K triangles each
        // (small meshes: one triangle per thread, so that the launch is not three workgroups doing all the work)
        uint32_t G = (uint32_t)std::min<uint64_t>(std::max<uint64_t>((f.geom_ntris + BIN_WG - 1) / BIN_WG, 1), BIN_MAX_G);
        if (c->bin_g > 0) G = (uint32_t)std::min(c->bin_g, BIN_MAX_G);
        G = std::min<uint32_t>(G, sh.nw == 3 ? 256u : (uint32_t)sh.nw * 64u); // the tile kernel reads one segment per thread (three waves: wave 0 reads a second one)
        // (+ one workgroup that scans the geometry kernel's block sums, unless an earlier launch has; a binning workgroup
        // fills a CU's LDS, so the launch stays within 256 workgroups: a 257th would wait for a whole one to finish)
        int do_scan = f.scan_pending ? 1 : 0;
        if (do_scan && G > 255u) G = 255u;
        if ((rc = ensure(c, B.bin_matrix, B.bin_matrix_cap, (size_t)BIN_MAX_G * ((size_t)c->max_tiles + 1))) != FRR_OK) return rc;
        // dynamic LDS: tile counters + as many staged 16-B records as fit (a chunk emits ~1.8 records per triangle)
        constexpr size_t kLdsBudget = 160 * 1024 - 1024; // the kernel's static LDS is < 1 KB
        const size_t hist_bytes = (((size_t)ltiles + 3) & ~(size_t)3) * sizeof(uint32_t);
        const uint32_t stage_cap = (uint32_t)std::min<size_t>((kLdsBudget - hist_bytes) / 16, 9216);
        const size_t lds = hist_bytes + (size_t)stage_cap * 16;
        if (!c->lds_attr_set) {
            HIP_TRY(c, hipFuncSetAttribute((const void *)k_bin_seg, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget));
            c->lds_attr_set = true;
        }
        a.seg = B.bin_matrix; a.nseg = G; a.bpar = q;
        // near-first copies (bins2): a fixed slot per tile, 8x the mean tile load, + an overflow arena of bin_cap records
        uint64_t SL = std::max<uint64_t>(256, (f.geom_ntris * 16 + ntiles - 1) / ntiles);   // (per-tile load of the whole window: ownership does not change it)
        SL = std::min<uint64_t>(SL, ((uint64_t)1 << 30) / ltiles);
        if (c->ent_slot_override) SL = c->ent_slot_override;
        a.ent_slot = (uint32_t)SL;
        a.bin_cap = (uint32_t)std::min<size_t>(B.bin_cap, 0xBFFFFFFFu);
        if ((rc = ensure(c, B.bins2, B.bin2_cap, (size_t)ltiles * SL + a.bin_cap)) != FRR_OK) return rc;
        a.bins2 = B.bins2;
        if ((rc = gstream_wait_readers(c, B)) != FRR_OK) return rc;   // the tile kernel that last read this workspace
        {
            ProfScope p(c, KID_BIN_SEG, gs);
            hipLaunchKernelGGL(k_bin_seg, dim3(G + do_scan), dim3(BIN_WG), lds, gs, a, ltiles, B.bin_matrix, stage_cap,
                               f.geom_fan_cap, S.block_sums, S.block_prefix, f.geom_nblocks, do_scan);
        }
        f.scan_pending = false;
        f.bpars[f.lane] = q; f.bset = bi;
    } else {
        // fallback for frames with more tiles than fit LDS counters: global atomics (one set of tile tables: after every
        // tile kernel so far)
        for (GeomSet &G2 : c->gset) if ((rc = gstream_wait_readers(c, G2)) != FRR_OK) return rc;
        for (BinSet &B2 : c->bset) if ((rc = gstream_wait_readers(c, B2)) != FRR_OK) return rc;
        if ((rc = scan_now(c)) != FRR_OK) return rc;
        const uint32_t bin_grid = (uint32_t)std::min<uint64_t>((f.geom_ntris + f.geom_fan_cap + 255) / 256, 2048);
        { ProfScope p(c, KID_BIN_COUNT, gs); hipLaunchKernelGGL(k_bin<false>, dim3(bin_grid), dim3(256), 0, gs, a, f.geom_fan_cap); }
        { ProfScope p(c, KID_TILE_SCAN, gs); hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, gs, a, ntiles); }
        { ProfScope p(c, KID_BIN_FILL, gs); hipLaunchKernelGGL(k_bin<true>, dim3(bin_grid), dim3(256), 0, gs, a, f.geom_fan_cap); }
    }
    cmd.par = q; cmd.set = bi; cmd.lane = f.lane;
    HIP_TRY(c, hipGetLastError());
    if (!c->in_replay) {
        // does this pass's need of the work lists fit for sure?  (exec_cmd waits for the event if not: frr_ctx::proven)
        DrawSig sig;
        memset(&sig, 0, sizeof sig);
        sig.mesh = f.geom_mesh; sig.mesh_gen = f.geom_mesh_gen; sig.ntris = f.geom_ntris; sig.duni_hash = f.geom_duni_hash; sig.join_epoch = c->join_epoch;
        sig.vs = f.geom_vs; sig.x0 = x0; sig.x1 = x1; sig.y0 = y0; sig.y1 = y1; sig.rank = f.rank; sig.world = f.world;
        sig.blocked = f.part_blocked ? 1 : 0; sig.filter = f.geom_filter.active ? 1 : 0; sig.fy0 = f.geom_filter.y0; sig.fy1 = f.geom_filter.y1;
        sig.gset = f.gset; sig.bset = bi;
        bool known = false;
        for (const DrawSig &p : c->proven) known = known || same_sig(p, sig);
        if (!known) {
            HIP_TRY(c, hipEventRecord(c->ev_verify, gs));
            c->verify_pending = true;
            c->verify_sig = sig;
        }
    }
    if ((rc = tstream_wait_gstream(c)) != FRR_OK) return rc;   // the tile kernel runs on the targets' stream, after the binning
    if (c->wait_pending) {   // frr_frame_wait: the targets' next writer follows what the caller's stream held
        HIP_TRY(c, hipStreamWaitEvent(tstream_of(c), c->ev_wait, 0));
        c->wait_pending = false;
    }
    if (grid && um) {
        hipStream_t ts = tstream_of(c);
        ProfScope p(c, KID_RASTER, ts);
        int shi = 4;
        for (int k = 0; k < 6; ++k) if (kSpanShapes[k][0] == sh.nw && kSpanShapes[k][1] == sh.occ) shi = k;
        const int win_safe = a.x0 >= -SPAN_SAFE && a.y0 >= -SPAN_SAFE && a.x1 <= SPAN_SAFE && a.y1 <= SPAN_SAFE;
        RasterArgs ra = a; DevUniforms d = cmd.duni; int ws = win_safe;
        void *args[] = {&ra, &d, &ws};
        if (c->raster_sweep) (void)hipModuleLaunchKernel(um->sweep, grid, 1, 1, 256, 1, 1, 0, ts, args, nullptr);   // k_raster(RasterArgs, DevUniforms)
        else (void)hipModuleLaunchKernel(um->span[cmd.count_frags ? 1 : 0][shi], grid, 1, 1, (unsigned)kSpanShapes[shi][0] * 64u, 1, 1, 0, ts, args, nullptr);
    } else if (grid) {
        switch (ps_id) {
        case FRR_PS_DEPTH: launch_raster<0, FRR_PS_DEPTH>(c, a, grid, sh, cmd.duni, cmd.count_frags); break;
        case FRR_PS_FLAT: launch_raster<0, FRR_PS_FLAT>(c, a, grid, sh, cmd.duni, cmd.count_frags); break;
        case FRR_PS_COLOR: launch_raster<3, FRR_PS_COLOR>(c, a, grid, sh, cmd.duni, cmd.count_frags); break;
        case FRR_PS_PHONG: launch_raster<8, FRR_PS_PHONG>(c, a, grid, sh, cmd.duni, cmd.count_frags); break;
        case FRR_PS_BLINN: launch_raster<8, FRR_PS_BLINN>(c, a, grid, sh, cmd.duni, cmd.count_frags); break;
        }
    }
    HIP_TRY(c, hipGetLastError());
    if ((rc = tile_launched(c, S, B)) != FRR_OK) return rc;
    if (fuse) {
        f.clear_pending = false;
        // the tile rows of other ranks missed this clear: owed to the ctx's own targets (frr_readback shows the
        // whole image); caller-bound targets of a partitioned ctx only ever have their owned rows defined
        f.unowned_debt = f.world > 1 && own_targets(c);
        f.debt_tiles_y = a.tiles_y;
    }
    return FRR_OK;
}

int finish(frr_ctx *c);

// run a command and remember it (finish() replays the commands from a failed one onwards)
int exec_cmd(frr_ctx *c, Cmd cmd)
{
    cmd.pre = c->fs;
    cmd.seq = c->next_seq++;
    c->verify_pending = false;
    const int rc = cmd.kind == Cmd::GEOM ? exec_geometry(c, cmd) : exec_raster(c, cmd);
    if (rc != FRR_OK) { c->fs = cmd.pre; c->verify_pending = false; return rc; }
    c->log.push_back(cmd);
    if (c->verify_pending) {
        // An unproven raster pass: wait for its binning launch (the tile kernel behind it is not waited for: it cancels itself
        // if the pass or its geometry found a list too small), then look at the word failed commands write to host memory.
        c->verify_pending = false;
        const DrawSig sig = c->verify_sig;
        bool failed = true;   // (no host-visible word: a full synchronisation point decides)
        if (c->host_bad) {
            HIP_TRY(c, hipEventSynchronize(c->ev_verify));
            failed = *(volatile uint32_t *)c->host_bad != c->seen_bad;
        }
        if (failed) { const int rf = finish(c); if (rf != FRR_OK) return rf; }   // grows the lists and replays, as often as it takes
        if (c->proven.size() >= 32) c->proven.erase(c->proven.begin());
        c->proven.push_back(sig);
        return FRR_OK;
    }
    if (c->log.size() >= 4096 && !c->in_replay) return finish(c);   // (a caller that never synchronises: bound the log)
    return FRR_OK;
}

// Synchronisation point: both streams drained, device counters on the host, and -- if a command found a work list too
// small (Counters::first_bad) -- the list grown and the commands from that one onwards replayed, as often as it takes.
// The reference's draw cannot fail (renderer.rs:269-384); neither can this one, short of running out of memory.
int finish(frr_ctx *c)
{
    if (c->in_replay) return FRR_OK;
    int rc = settle(c);
    if (rc != FRR_OK) return rc;
    for (int round = 0;; ++round) {
        if ((rc = drain(c)) != FRR_OK) return rc;
        HIP_TRY(c, hipMemcpy(&c->hc, c->cnt, sizeof(Counters), hipMemcpyDeviceToHost));
        Counters &h = c->hc;
        uint32_t cbm = 0, need_fans = 0; uint64_t worst_bins = 0;
        for (const Lane &L : h.lane) {
            cbm = std::max(cbm, std::max(L.gtab[0].clip_block_max, L.gtab[1].clip_block_max));
            need_fans = std::max(need_fans, std::max(L.gtab[0].need_fans, L.gtab[1].need_fans));
            worst_bins = std::max<uint64_t>(worst_bins, std::max<uint64_t>(L.bin_total, std::max<uint64_t>(L.btab[0].seg_total, L.btab[1].seg_total)));
        }
        c->clip_queue_auto = cbm > (uint32_t)CLIP_QUEUE_AT;
        const uint32_t bad = h.first_bad;
        if (c->host_bad) c->seen_bad = *(volatile uint32_t *)c->host_bad;   // (everything has drained: the word is final)
        if (bad == SEQ_NONE) break;     // (a failure that has been dealt with is reset below)
        size_t i = 0;
        while (i < c->log.size() && c->log[i].seq != bad) ++i;
        // the failed command may belong to a frame whose log is gone (frr_clear came before anybody synchronised): that
        // frame cannot be replayed any more -- and its targets have been cleared since -- but the lists are grown, so that
        // the frames from now on fit
        const bool gone = i == c->log.size();
        if (round == 8) return fail(c, FRR_ERR_CAPACITY, "device work lists still too small after eight replays");
        // grow what was too small
        if (h.overflow & 2u) {
            const uint64_t worst = worst_bins;
            const size_t need = (size_t)(worst + worst / 4 + 1024);
            for (BinSet &B : c->bset) {
                if (!B.bins && (gone || &B != &c->bset[c->log[i].set])) continue;   // (a workspace nobody has used yet is sized when it is)
                if ((rc = ensure(c, B.bins, B.bin_cap, std::max(need, B.bin_cap))) != FRR_OK) return rc;
                if ((rc = ensure(c, B.bins2, B.bin2_cap, std::max(need, B.bin2_cap))) != FRR_OK) return rc;
            }
        }
        if (h.overflow & 1u) {
            const uint64_t nf = need_fans;
            c->fan_hint = std::max<size_t>(c->fan_hint, (size_t)(nf + nf / 8 + 1024));
        }
        // the device tables as they were before the failed command: it has used its own parity's cursors (and, when its
        // block sums were scanned inside the next raster pass's binning launch, that pass has reserved bin space)
        h.first_bad = SEQ_NONE; h.overflow = 0u;
        for (size_t k = i; k < std::min(i + 2, c->log.size()); ++k) {
            const Cmd &m = c->log[k];
            if (m.kind == Cmd::GEOM && k == i) {
                GeomTab &gt = h.lane[m.lane].gtab[m.par];
                for (int r = 0; r < FAN_REGIONS; ++r) gt.fan_cursor[r].v = 0u;
                gt.clip_q = 0u; gt.clip_block_max = 0u; gt.n_emit = 0u; gt.need_fans = 0u;
            } else if (m.kind == Cmd::RASTER) {
                h.lane[m.lane].btab[m.par].seg_total = 0ull; h.lane[m.lane].btab[m.par].ent_cursor = 0u;
            }
        }
        HIP_TRY(c, hipMemcpy(c->cnt, &h, offsetof(Counters, dbg), hipMemcpyHostToDevice));
        if (gone) continue;
        // replay
        std::vector<Cmd> todo(c->log.begin() + (ptrdiff_t)i, c->log.end());
        c->log.resize(i);
        const FrameState now = c->fs;
        c->fs = todo[0].pre;
        c->epoch = c->next_seq;
        c->in_replay = true;
        ++c->replays;
        for (const Cmd &m : todo) {
            // what the caller set between the commands travels with them
            c->fs.color = m.pre.color; c->fs.depth = m.pre.depth; c->fs.tri_id = m.pre.tri_id; c->fs.tset = m.pre.tset;
            c->fs.rank = m.pre.rank; c->fs.world = m.pre.world; c->fs.part_blocked = m.pre.part_blocked;
            if ((rc = exec_cmd(c, m)) != FRR_OK) { c->in_replay = false; return rc; }
        }
        c->in_replay = false;
        c->fs.color = now.color; c->fs.depth = now.depth; c->fs.tri_id = now.tri_id; c->fs.tset = now.tset;
        c->fs.rank = now.rank; c->fs.world = now.world; c->fs.part_blocked = now.part_blocked;
        if ((rc = settle(c)) != FRR_OK) return rc;
    }
    c->log.clear();
    c->epoch = c->next_seq;
    if (c->next_seq > 0xF0000000u) {   // sequence numbers start over (nothing is in flight)
        const uint32_t none = SEQ_NONE;
        HIP_TRY(c, hipMemcpy(&c->cnt->first_bad, &none, sizeof none, hipMemcpyHostToDevice));
        c->next_seq = c->epoch = 1;
    }
    return FRR_OK;
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
    if (const UserShader *us = user_shader(vs_id)) return us->nf;
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
    if (const UserShader *us = user_shader(vs_id)) return us->K;
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
    else { if (acquire_stream(device, &c->stream) != hipSuccess) { delete c; return FRR_ERR_HIP; } c->own_stream = true; }
    const size_t npx = (size_t)width * height;
    bool ok = hipMalloc((void **)&c->own_color[0], npx * 4) == hipSuccess && hipMalloc((void **)&c->own_depth[0], npx * 4) == hipSuccess &&
              hipMalloc((void **)&c->own_tri_id[0], npx * 4) == hipSuccess && hipMalloc((void **)&c->cnt, sizeof(Counters)) == hipSuccess;
    c->max_tiles = ((width + TILE - 1) / TILE) * ((height + TILE - 1) / TILE);
    ok = ok && hipMalloc((void **)&c->tile_counts, (c->max_tiles + 1) * 4) == hipSuccess &&
         hipMalloc((void **)&c->tile_offsets, (c->max_tiles + 1) * 4) == hipSuccess &&
         hipMalloc((void **)&c->tile_cursor, (c->max_tiles + 1) * 4) == hipSuccess;
    for (GeomSet &S : c->gset) ok = ok && hipEventCreateWithFlags(&S.reader_ev, hipEventDisableTiming) == hipSuccess;
    for (BinSet &B : c->bset) ok = ok && hipEventCreateWithFlags(&B.reader_ev, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->ev_t2, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&c->ev_t1, hipEventDisableTiming) == hipSuccess;
    for (auto &e : c->ev_bin) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->ev_verify, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&c->ev_export, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&c->ev_wait, hipEventDisableTiming) == hipSuccess;
    if (!ok) { frr_destroy(c); return FRR_ERR_NOMEM; }
    c->fs.color = c->own_color[0]; c->fs.depth = c->own_depth[0]; c->fs.tri_id = c->own_tri_id[0];
    {
        // tables of frame 0 (no frame has that number), no failed command
        memset(&c->hc, 0, sizeof c->hc);
        c->hc.first_bad = SEQ_NONE;
        if (hipHostMalloc((void **)&c->host_bad, 64, hipHostMallocMapped) == hipSuccess) {
            *c->host_bad = SEQ_NONE;
            void *dp = nullptr;
            if (hipHostGetDevicePointer(&dp, c->host_bad, 0) == hipSuccess) c->hc.host_bad = (uint32_t *)dp;
        }
        if (hipMemcpy(c->cnt, &c->hc, sizeof(Counters), hipMemcpyHostToDevice) != hipSuccess) { frr_destroy(c); return FRR_ERR_HIP; }
    }
    (void)hipMemsetAsync(c->tile_counts, 0, (c->max_tiles + 1) * 4, c->stream);
    (void)hipMemsetAsync(c->own_color[0], 0, npx * 4, c->stream);      // FrameBuffer::new zero-fills (renderer.rs:423)
    (void)hipMemsetAsync(c->own_depth[0], 0, npx * 4, c->stream);
    (void)hipMemsetAsync(c->own_tri_id[0], 0xFF, npx * 4, c->stream);
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
    if (c->gstream) (void)hipStreamSynchronize(c->gstream);
    if (c->tstream2) (void)hipStreamSynchronize(c->tstream2);
    if (c->tstream1) (void)hipStreamSynchronize(c->tstream1);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    prof_collect(c);
    for (auto &m : c->meshes) if (m.used && m.owned) (void)hipFree((void *)m.dev);
    for (auto &t : c->tex) if (t.dev) (void)hipFree(t.dev);
    if (c->host_bad) (void)hipHostFree(c->host_bad);
    for (auto &um : c->user_modules) if (um.second.mod) (void)hipModuleUnload(um.second.mod);
    std::vector<void *> ptrs = {c->own_color[0], c->own_depth[0], c->own_tri_id[0], c->own_color[1], c->own_depth[1], c->own_tri_id[1],
                                c->cnt, c->tile_counts, c->tile_offsets, c->tile_cursor};
    for (GeomSet &S : c->gset) for (void *p : {(void *)S.block_sums, (void *)S.block_prefix, (void *)S.tinfo, (void *)S.fanbase, (void *)S.fan_okey, (void *)S.recs,
                                               (void *)S.vary, (void *)S.pbox, (void *)S.bcount, (void *)S.clipq}) ptrs.push_back(p);
    for (BinSet &B : c->bset) for (void *p : {(void *)B.bins, (void *)B.bins2, (void *)B.bin_matrix}) ptrs.push_back(p);
    for (void *p : ptrs) if (p) (void)hipFree(p);
#ifdef FRR_DEBUG_COUNTERS
    if (c->dbg_tiles) (void)hipFree(c->dbg_tiles);
#endif
    for (auto &e : c->ev) if (e) (void)hipEventDestroy(e);
    for (GeomSet &S : c->gset) if (S.reader_ev) (void)hipEventDestroy(S.reader_ev);
    for (BinSet &B : c->bset) if (B.reader_ev) (void)hipEventDestroy(B.reader_ev);
    if (c->ev_t2) (void)hipEventDestroy(c->ev_t2);
    if (c->ev_t1) (void)hipEventDestroy(c->ev_t1);
    for (auto &e : c->ev_bin) if (e) (void)hipEventDestroy(e);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    for (hipEvent_t e : {c->ev_verify, c->ev_export, c->ev_wait}) if (e) (void)hipEventDestroy(e);
    for (auto &e : c->ev_pool) (void)hipEventDestroy(e);
    // (back to the pool in the reverse order of their typical acquisition, so that the next ctx gets them in the same roles)
    release_stream(c->device, c->gstream);
    release_stream(c->device, c->tstream1);
    release_stream(c->device, c->tstream2);
    if (c->own_stream) release_stream(c->device, c->stream);
    delete c;
}

int frr_set_option(frr_ctx *c, const char *name, int64_t v)
{
    if (!c || !name) return FRR_ERR_INVALID;
    { int rc = finish(c); if (rc != FRR_OK) return rc; }   // nothing in flight, nothing to replay under the old setting
    c->proven.clear();                                     // (options change what a pass needs of the lists, or the lists)
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
    else if (n == "fan_capacity") { if (v < 0) return fail(c, FRR_ERR_INVALID, "fan_capacity >= 0"); c->fan_cap_init = (size_t)v; }
    else if (n == "overlap") { if (v < 0 || v > 2) return fail(c, FRR_ERR_INVALID, "overlap: 0, 1 or 2"); c->overlap = (int)v; }
    else if (n == "bound_targets_in_flight") c->bound_in_flight = v != 0;
    else if (n == "frames_in_flight") { if (v != 1 && v != 2) return fail(c, FRR_ERR_INVALID, "frames_in_flight: 1 or 2"); c->frames_in_flight = (int)v; }
    else return fail(c, FRR_ERR_INVALID, "unknown option");
    return FRR_OK;
}

int frr_set_partition(frr_ctx *c, int rank, int world)
{
    if (!c || world < 1 || rank < 0 || rank >= world) return fail(c, FRR_ERR_INVALID, "bad partition");
    { int rc = settle(c); if (rc != FRR_OK) return rc; } // rows skipped by a fused clear are defined by the old partition
    c->fs.rank = rank; c->fs.world = world;
    return FRR_OK;
}
int frr_set_partition_layout(frr_ctx *c, int blocked)
{
    if (!c) return FRR_ERR_INVALID;
    { int rc = settle(c); if (rc != FRR_OK) return rc; }
    c->fs.part_blocked = blocked != 0;
    return FRR_OK;
}
// owned tile rows of a window of `wh` pixel rows, by the same rule the kernels use (owns_tile_row)
static int owned_band_of(int rank, int world, bool blocked, int64_t wh, int band, int32_t *row0, int32_t *row1)
{
    const int tiles_y = (int)((wh + TILE - 1) / TILE);
    if (world <= 1) {
        if (row0 && band == 0) { *row0 = 0; *row1 = (int32_t)wh; }
        return wh > 0 ? 1 : 0;
    }
    if (blocked) {
        int t0, t1;
        blocked_rows(tiles_y, rank, world, &t0, &t1);
        if (t1 <= t0) return 0;
        if (row0 && band == 0) { *row0 = t0 * TILE; *row1 = (int32_t)std::min<int64_t>(wh, (int64_t)t1 * TILE); }
        return 1;
    }
    const int n = tiles_y > rank ? (tiles_y - rank + world - 1) / world : 0;
    if (row0 && band < n) {
        const int ty = rank + band * world;
        *row0 = ty * TILE; *row1 = (int32_t)std::min<int64_t>(wh, (int64_t)(ty + 1) * TILE);
    }
    return n;
}
static int owned_band(const frr_ctx *c, int64_t wh, int band, int32_t *row0, int32_t *row1)
{
    return owned_band_of(c->fs.rank, c->fs.world, c->fs.part_blocked, wh, band, row0, row1);
}
int frr_partition_rows(int32_t y0, int32_t y1, int rank, int world, int blocked, int32_t band, int32_t *row0, int32_t *row1)
{
    if (y0 > y1 || world < 1 || rank < 0 || rank >= world || band < 0 || (row0 == nullptr) != (row1 == nullptr)) return FRR_ERR_INVALID;
    return owned_band_of(rank, world, blocked != 0, (int64_t)y1 - y0, band, row0, row1);
}
int frr_exchange_plan(int32_t y0, int32_t y1, uint32_t row_elems, int rank, int world, int blocked, int root, frr_xfer *ops, int cap)
{
    if (y0 > y1 || world < 1 || rank < 0 || rank >= world || root < 0 || root >= world || cap < 0 || (cap && !ops)) return FRR_ERR_INVALID;
    const int64_t wh = (int64_t)y1 - y0;
    int n = 0;
    auto put = [&](int kind, int peer, int32_t r0, int32_t r1) {
        if (n < cap) ops[n] = frr_xfer{kind, peer, (uint64_t)r0 * row_elems, (uint64_t)(r1 - r0) * row_elems};
        ++n;
    };
    auto bands_of = [&](int p, int kind, int peer) {
        const int nb = owned_band_of(p, world, blocked != 0, wh, 0, nullptr, nullptr);
        for (int b = 0; b < nb; ++b) {
            int32_t r0 = 0, r1 = 0;
            (void)owned_band_of(p, world, blocked != 0, wh, b, &r0, &r1);
            if (r1 > r0) put(kind, peer, r0, r1);
        }
    };
    if (rank != root) { bands_of(rank, FRR_XFER_SEND, root); return n; }
    for (int p = 0; p < world; ++p) if (p != root) bands_of(p, FRR_XFER_RECV, p);
    bands_of(root, FRR_XFER_COPY, root);
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
    if (c->bound_in_flight && (!color || !depth || !tri_id) && (color || depth || tri_id))
        return fail(c, FRR_ERR_INVALID, "option bound_targets_in_flight: bind all three targets, or none (back to the ctx's own)");
    if (!(c->bound_in_flight && !own_targets(c))) { int rc = join_tile_streams(c); if (rc != FRR_OK) return rc; }   // (frames in flight on bound targets: the caller fences)
    if (!color || !depth || !tri_id) {
        // (part of) the ctx's own set: frames on bound targets may have toggled the set index without it ever being allocated
        if (c->bound_in_flight) { int rc = finish(c); if (rc != FRR_OK) return rc; }   // back to own targets: nothing of the bound frames in flight
        int rc = ensure_own_set(c, c->fs.tset);
        if (rc != FRR_OK) return rc;
    }
    c->fs.color = color ? (uint8_t *)color : c->own_color[c->fs.tset];
    c->fs.depth = depth ? (float *)depth : c->own_depth[c->fs.tset];
    c->fs.tri_id = tri_id ? (uint32_t *)tri_id : c->own_tri_id[c->fs.tset];
    return FRR_OK;
}
int frr_target_ptrs(frr_ctx *c, void **color, void **depth, void **tri_id)
{
    if (!c) return FRR_ERR_INVALID;
    { int rc = settle(c); if (rc != FRR_OK) return rc; } // the caller is about to look at them ...
    { int rc = join_tile_streams(c); if (rc != FRR_OK) return rc; } // ... from its stream
    if (own_targets(c)) { c->exported[c->fs.tset] = true; c->export_stream[c->fs.tset] = c->stream; }   // (frr_clear orders the set's next frame behind those reads)
    if (color) *color = c->fs.color;
    if (depth) *depth = c->fs.depth;
    if (tri_id) *tri_id = c->fs.tri_id;
    return FRR_OK;
}

static int mesh_register(frr_ctx *c, const float *dev, bool owned, uint64_t ntris, int vs, int *mesh_out)
{
    Mesh m; m.dev = dev; m.owned = owned; m.used = true; m.ntris = ntris; m.vs = vs; m.gen = ++c->mesh_gen;
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
    c->join_epoch += 1;    // the ctx's private streams have to see what the caller's stream wrote into that memory up to now
    return mesh_register(c, (const float *)dev, false, ntris, vs_id, mesh_out);
}
int frr_mesh_free(frr_ctx *c, int mesh)
{
    if (!c || mesh < 0 || mesh >= (int)c->meshes.size() || !c->meshes[mesh].used) return fail(c, FRR_ERR_INVALID, "bad mesh id");
    { int rc = finish(c); if (rc != FRR_OK) return rc; }   // nothing reads it any more, nothing will replay a draw of it
    if (c->meshes[mesh].owned) (void)hipFree((void *)c->meshes[mesh].dev);
    c->meshes[mesh] = Mesh();
    return FRR_OK;
}

int frr_texture_upload(frr_ctx *c, int slot, const uint8_t *rgba, uint32_t w, uint32_t h)
{
    if (!c || slot < 0 || slot >= FRR_MAX_TEXTURES || !rgba || w == 0 || h == 0) return fail(c, FRR_ERR_INVALID, "bad texture");
    if (h < w) return fail(c, FRR_ERR_UNSUPPORTED, "texture height < width: sample_2d clamps y with width (renderer.rs:523) and would index out of bounds");
    HIP_TRY(c, hipSetDevice(c->device));
    { int rc = finish(c); if (rc != FRR_OK) return rc; }   // the draws issued so far sample the old texture
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

int frr_set_user_uniforms(frr_ctx *c, const float *values, int n)
{
    if (!c || n < 0 || n > FRR_MAX_USER_UNIFORMS || (n && !values)) return fail(c, FRR_ERR_INVALID, "user uniforms: at most FRR_MAX_USER_UNIFORMS floats");
    memset(c->duni.user, 0, sizeof c->duni.user);
    if (n) memcpy(c->duni.user, values, (size_t)n * sizeof(float));
    return FRR_OK;
}

int frr_shader_register(frr_ctx *c, const char *hip_source, int vs_input_floats, int num_varyings, int *shader_id)
{
    if (!hip_source || !shader_id || vs_input_floats < 1 || vs_input_floats > 64 || num_varyings < 0 || num_varyings > FRR_MAX_VARYINGS)
        return fail(c, FRR_ERR_INVALID, "bad shader description (1..64 input floats, 0..FRR_MAX_VARYINGS varyings)");
    // the program: fixed-width types (hiprtc keeps them in a namespace), this shader's shape, the device header, the
    // user's functions, the kernels -- the very headers libfrr_hip.so itself was built from
    std::string src = "using __hip_internal::int8_t; using __hip_internal::uint8_t; using __hip_internal::int16_t; using __hip_internal::uint16_t;\n"
                      "using __hip_internal::int32_t; using __hip_internal::uint32_t; using __hip_internal::int64_t; using __hip_internal::uint64_t;\n"
                      "typedef unsigned long uintptr_t;\n#define FRR_USER_SHADER 1\n";
    src += "#define FRR_USER_NF " + std::to_string(vs_input_floats) + "\n#define FRR_USER_K " + std::to_string(num_varyings) + "\n";
    src += "#include \"frr_device.h\"\n#line 1 \"user_shader\"\n";
    src += hip_source;
    src += "\n#include \"frr_kernels.h\"\n";
    const char *hdr_txt[] = {frr_src_device_h, frr_src_exact_h, frr_src_kernels_h, frr_src_raster_h, frr_src_frr_h};
    const char *hdr_name[] = {"frr_device.h", "frr_exact.h", "frr_kernels.h", "frr_raster.h", "../../include/frr.h"};
    hiprtcProgram prog = nullptr;
    if (hiprtcCreateProgram(&prog, src.c_str(), "frr_user_program.hip", 5, hdr_txt, hdr_name) != HIPRTC_SUCCESS) return fail(c, FRR_ERR_HIP, "hiprtcCreateProgram");
    UserShader *us = new UserShader();
    us->nf = vs_input_floats; us->K = num_varyings;
    const std::string U = std::to_string(FRR_SHADER_USER_BASE), Ks = std::to_string(num_varyings);
    std::vector<std::string> exprs = {"frr::k_geom_single<" + U + ">", "frr::k_geom_clip<" + U + ">"};
    const std::string sweep_expr = "frr::k_raster<" + Ks + ", " + U + ">";   // the brute-force tile kernel (option raster_sweep)
    (void)hiprtcAddNameExpression(prog, sweep_expr.c_str());
    for (int cnt = 0; cnt < 2; ++cnt)
        for (int sh = 0; sh < 6; ++sh)
            exprs.push_back("frr::k_raster_span<" + Ks + ", " + U + ", " + (cnt ? "true" : "false") + ", " + std::to_string(kSpanShapes[sh][0]) + ", " + std::to_string(kSpanShapes[sh][1]) + ">");
    for (const std::string &e : exprs) (void)hiprtcAddNameExpression(prog, e.c_str());
    const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-Wno-unused-function"};
    const hiprtcResult res = hiprtcCompileProgram(prog, 7, opts);
    if (res != HIPRTC_SUCCESS) {
        size_t n = 0;
        (void)hiprtcGetProgramLogSize(prog, &n);
        std::string log(n + 1, '\0');
        if (n) (void)hiprtcGetProgramLog(prog, &log[0]);
        (void)hiprtcDestroyProgram(&prog);
        delete us;
        return fail(c, FRR_ERR_UNSUPPORTED, std::string("user shader does not compile:\n") + log.c_str());
    }
    bool ok = true;
    auto lowered = [&](const std::string &e) { const char *n = nullptr; ok = ok && hiprtcGetLoweredName(prog, e.c_str(), &n) == HIPRTC_SUCCESS && n; return std::string(n ? n : ""); };
    us->geom = lowered(exprs[0]); us->clip = lowered(exprs[1]); us->sweep = lowered(sweep_expr);
    for (int cnt = 0; cnt < 2; ++cnt)
        for (int sh = 0; sh < 6; ++sh) us->span[cnt][sh] = lowered(exprs[2 + (size_t)cnt * 6 + sh]);
    size_t cs = 0;
    ok = ok && hiprtcGetCodeSize(prog, &cs) == HIPRTC_SUCCESS && cs;
    if (ok) { us->code.resize(cs); ok = hiprtcGetCode(prog, us->code.data()) == HIPRTC_SUCCESS; }
    (void)hiprtcDestroyProgram(&prog);
    if (!ok) { delete us; return fail(c, FRR_ERR_HIP, "hiprtc: no code object"); }
    std::lock_guard<std::mutex> lk(g_shader_mu);
    g_shaders.push_back(us);
    *shader_id = FRR_SHADER_USER_BASE + (int)g_shaders.size() - 1;
    return FRR_OK;
}

int frr_clear(frr_ctx *c, const uint8_t rgba[4], float depth)
{
    if (!c || !rgba) return FRR_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    uint32_t packed;
    memcpy(&packed, rgba, 4);
    FrameState &f = c->fs;
    // Has a command failed since the host last looked (a word the device writes into host memory: no synchronisation)?
    // Then repair now -- replay the frame if its log is still here, grow the lists either way -- before this frame goes on.
    if (c->host_bad && *(volatile uint32_t *)c->host_bad != c->seen_bad) { const int rc = finish(c); if (rc != FRR_OK) return rc; }
    // A new frame: the commands logged so far are history.  (One of them may have failed unseen: it and everything after
    // it left the targets untouched, and they are overwritten now.  Device statistics are tagged with the frame number.)
    c->log.clear();
    c->epoch = c->next_seq;
    c->replays = 0;
    f.frame_no += 1;
    f.tris_in = 0; f.draws = 0;
    f.geom_ntris = 0;          // the setup list of a preceding frr_geometry is gone (frr_raster then draws nothing)
    f.scan_pending = false;
    f.unowned_debt = false;    // superseded: the clear covers every row
    if (c->frames_in_flight == 2 && own_targets(c)) {
        // own targets: this frame takes the other set (and its tile kernels the other stream), so that it need not wait
        // for the previous frame's tile kernel to drain; the old set's content is dead (the clear overwrites everything)
        const int t = f.tset ^ 1;
        { const int rc = ensure_own_set(c, t); if (rc != FRR_OK) return rc; }
        if (!c->tstream2 && acquire_stream(c->device, &c->tstream2) != hipSuccess) return fail(c, FRR_ERR_HIP, "frame stream");
        f.tset = t;
        f.color = c->own_color[t]; f.depth = c->own_depth[t]; f.tri_id = c->own_tri_id[t];
        f.lane = t;    // ... and its own device tables: nothing the two frames' bookkeeping threads write is shared
    } else if (frames_alternate(c)) {
        // caller-bound targets, option bound_targets_in_flight: the caller has bound another target set for this frame;
        // the frame takes the other private stream, workspace set and lane
        bool ok = (c->tstream2 || acquire_stream(c->device, &c->tstream2) == hipSuccess) &&
                  (c->tstream1 || acquire_stream(c->device, &c->tstream1) == hipSuccess);
        if (!ok) return fail(c, FRR_ERR_HIP, "frame streams");
        f.tset ^= 1;
        f.lane = f.tset;
    } else {
        f.lane = 0;
    }
    if (own_targets(c) && c->exported[f.tset]) {
        // The pointers of this own target set were handed out (frr_target_ptrs / frr_frame_fence) when it last held a frame:
        // what the caller has queued on that stream since -- its reads of that frame -- comes before this frame's writes.
        // (Same stream: in order anyway.  Only callers that take the pointers pay for the event.)
        c->exported[f.tset] = false;
        hipStream_t ts = tstream_of(c);
        if (c->export_stream[f.tset] != ts) {
            HIP_TRY(c, hipEventRecord(c->ev_export, c->export_stream[f.tset]));
            HIP_TRY(c, hipStreamWaitEvent(ts, c->ev_export, 0));
        }
    }
    if (c->clear_eager) { f.clear_pending = false; return clear_now(c, packed, depth); }
    f.clear_rgba = packed; f.clear_depth = depth;
    f.clear_pending = true;
    return FRR_OK;
}

int frr_geometry(frr_ctx *c, int mesh, uint64_t *ntris_setup)
{
    if (!c || mesh < 0 || mesh >= (int)c->meshes.size() || !c->meshes[mesh].used) return fail(c, FRR_ERR_INVALID, "bad mesh id");
    HIP_TRY(c, hipSetDevice(c->device));
    Cmd cmd;
    cmd.kind = Cmd::GEOM; cmd.mesh = mesh; cmd.duni = c->duni;
    int rc = exec_cmd(c, cmd);
    if (rc != FRR_OK) return rc;
    if (ntris_setup) {
        if ((rc = finish(c)) != FRR_OK) return rc;
        *ntris_setup = c->hc.lane[c->fs.lane].gtab[c->fs.gpar()].n_emit;
    }
    return FRR_OK;
}

// argument checks of frr_raster; *nothing: the call is valid and draws nothing
static int raster_check(frr_ctx *c, int ps_id, int32_t x0, int32_t x1, int32_t y0, int32_t y1, bool *nothing)
{
    const FrameState &f = c->fs;
    if (f.geom_vs < 0) return fail(c, FRR_ERR_INVALID, "frr_raster before frr_geometry");
    if (x0 > x1 || y0 > y1) return fail(c, FRR_ERR_INVALID, "range min > max (i32::clamp would panic, renderer.rs:285)");
    const int64_t ww = (int64_t)x1 - x0, wh = (int64_t)y1 - y0;
    if (ww > (int64_t)c->W || wh > (int64_t)c->H) return fail(c, FRR_ERR_INVALID, "window larger than the FrameBuffer");
    if (x0 < -32768 || y0 < -32768 || x1 > 32767 || y1 > 32767) return fail(c, FRR_ERR_UNSUPPORTED, "window coordinates outside the i16 range");
    if (ww > 0 && wh > 0 && (x1 <= 0 || (wh - 1) * (int64_t)x1 + ww > (int64_t)c->W * c->H))
        return fail(c, FRR_ERR_INVALID, "depth index (cy-y0)*x1+(cx-x0) would leave the depth buffer (renderer.rs:362)");
    const int K = frr_vs_num_varyings(f.geom_vs);
    if (ps_id >= FRR_SHADER_USER_BASE) {
        // a user pixel shader: the mesh's vertex shader -- the same user shader, another one, or a built-in -- has to hand it
        // the varyings it was registered with (the reference's two closures share one ShaderContext type, renderer.rs:97-110)
        const UserShader *us = user_shader(ps_id);
        if (!us) return fail(c, FRR_ERR_INVALID, "unknown shader id");
        if (us->K != K) return fail(c, FRR_ERR_INVALID, "the user pixel shader's varyings do not match the vertex shader's");
    } else if ((ps_id == FRR_PS_COLOR && K != 3) || ((ps_id == FRR_PS_PHONG || ps_id == FRR_PS_BLINN) && K != 8) || ps_id < 0 || ps_id > FRR_PS_BLINN)
        return fail(c, FRR_ERR_INVALID, "pixel shader does not match the vertex shader's varyings");
    if ((ps_id == FRR_PS_PHONG || ps_id == FRR_PS_BLINN) && !c->duni.tex) return fail(c, FRR_ERR_INVALID, "no texture bound to uniforms.texture_slot");
    {
        // frr_draw on a partitioned ctx keeps only the triangles that touch the rank's tile rows of ITS window; that
        // list serves no other window or partition (the reference may reuse one geometry for several ranges,
        // renderer.rs:269-271: use frr_geometry for that, it never filters)
        const GeomFilter &gf = f.geom_filter;
        if (gf.active && (gf.y0 != y0 || gf.y1 != y1 || gf.rank != f.rank || gf.world != f.world || gf.blocked != f.part_blocked))
            return fail(c, FRR_ERR_INVALID, "the setup list was filtered by frr_draw for another window/partition; re-run frr_geometry (unfiltered) before frr_raster");
    }
    *nothing = ww == 0 || wh == 0 || f.geom_ntris == 0;
    return FRR_OK;
}

int frr_raster(frr_ctx *c, int ps_id, int32_t x0, int32_t x1, int32_t y0, int32_t y1)
{
    if (!c) return FRR_ERR_INVALID;
    bool nothing = false;
    const int rc = raster_check(c, ps_id, x0, x1, y0, y1, &nothing);
    if (rc != FRR_OK || nothing) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    Cmd cmd;
    cmd.kind = Cmd::RASTER; cmd.ps = ps_id; cmd.x0 = x0; cmd.x1 = x1; cmd.y0 = y0; cmd.y1 = y1;
    cmd.count_frags = c->count_frags; cmd.duni = c->duni;
    return exec_cmd(c, cmd);
}

int frr_draw(frr_ctx *c, int mesh, int ps_id, int32_t x0, int32_t x1, int32_t y0, int32_t y1)
{
    if (!c || mesh < 0 || mesh >= (int)c->meshes.size() || !c->meshes[mesh].used) return fail(c, FRR_ERR_INVALID, "bad mesh id");
    HIP_TRY(c, hipSetDevice(c->device));
    // frr_draw knows the raster window, so a partitioned ctx can skip the setup records of triangles
    // that touch none of its tile rows (frr_geometry alone cannot: the window comes later)
    Cmd cmd;
    cmd.kind = Cmd::GEOM; cmd.mesh = mesh; cmd.duni = c->duni;
    cmd.filter = c->fs.world > 1 && y0 <= y1; cmd.fy0 = y0; cmd.fy1 = y1;
    const int rc = exec_cmd(c, cmd);
    if (rc != FRR_OK) return rc;
    return frr_raster(c, ps_id, x0, x1, y0, y1);
}

int frr_frame_fence(frr_ctx *c, void *stream)
{
    if (!c) return FRR_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    { int rc = settle(c); if (rc != FRR_OK) return rc; }
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    if (own_targets(c)) { c->exported[c->fs.tset] = true; c->export_stream[c->fs.tset] = st; }
    return fence_stream(c, st);
}

int frr_frame_wait(frr_ctx *c, void *stream)
{
    if (!c) return FRR_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipEventRecord(c->ev_wait, stream ? (hipStream_t)stream : c->stream));
    c->wait_pending = true;
    return FRR_OK;
}

int frr_sync(frr_ctx *c)
{
    if (!c) return FRR_ERR_INVALID;
    return finish(c);
}

int frr_readback(frr_ctx *c, uint8_t *rgba, float *depth, uint32_t *tri_id)
{
    if (!c) return FRR_ERR_INVALID;
    { int rc = finish(c); if (rc != FRR_OK) return rc; }
    const size_t bytes = (size_t)c->W * c->H * 4;
    if (rgba) HIP_TRY(c, hipMemcpyAsync(rgba, c->fs.color, bytes, hipMemcpyDeviceToHost, c->stream));
    if (depth) HIP_TRY(c, hipMemcpyAsync(depth, c->fs.depth, bytes, hipMemcpyDeviceToHost, c->stream));
    if (tri_id) HIP_TRY(c, hipMemcpyAsync(tri_id, c->fs.tri_id, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return FRR_OK;
}

int frr_readback_setup(frr_ctx *c, frr_setup_vertex *out, uint64_t cap_tris, uint64_t *ntris)
{
    if (!c || !ntris || c->fs.geom_vs < 0) return fail(c, FRR_ERR_INVALID, "no geometry to read back");
    if (c->fs.geom_filter.active)
        return fail(c, FRR_ERR_INVALID, "the setup list of a partitioned frr_draw holds only this rank's triangles; use frr_geometry to read back the full Vec<[Vertex;3]>");
    { int rcs = finish(c); if (rcs != FRR_OK) return rcs; }
    const FrameState &f = c->fs;
    const GeomTab &gt = c->hc.lane[f.lane].gtab[f.gpar()];
    const GeomSet &S = c->gset[f.gset];
    const uint64_t nt = f.geom_ntris;
    *ntris = nt ? gt.n_emit : 0;
    if (!out || !nt) return FRR_OK;
    // the records live at slots (frr_device.h): input t's own slot, or its fan's slots behind the inputs; walking the
    // inputs in order and each fan in order is the reference's emission order
    const uint64_t slots = nt + f.geom_fan_cap;   // (fan slots are spread over the regions of the fan space)
    const int K = frr_vs_num_varyings(f.geom_vs);
    std::vector<uint32_t> tinfo(nt), fanbase(nt);
    std::vector<RasterRec> recs(slots);
    std::vector<float> vary((size_t)slots * 3 * K);
    HIP_TRY(c, hipMemcpy(tinfo.data(), S.tinfo, nt * 4, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(fanbase.data(), S.fanbase, nt * 4, hipMemcpyDeviceToHost));
    // the inputs' own slots, then the used part of every fan region
    const uint64_t region = f.geom_fan_cap / FAN_REGIONS;
    for (int k = -1; k < FAN_REGIONS; ++k) {
        const uint64_t first = k < 0 ? 0 : nt + (uint64_t)k * region;
        const uint64_t count = k < 0 ? nt : std::min<uint64_t>(gt.fan_cursor[k].v, region);
        if (!count) continue;
        HIP_TRY(c, hipMemcpy(recs.data() + first, S.recs + first, count * sizeof(RasterRec), hipMemcpyDeviceToHost));
        if (K) HIP_TRY(c, hipMemcpy(vary.data() + first * 3 * K, S.vary + first * 3 * K, count * 3 * K * sizeof(float), hipMemcpyDeviceToHost));
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
                o.spi[0] = f32_as_i32(r.s[2 * s] + 0.5f); o.spi[1] = f32_as_i32(r.s[2 * s + 1] + 0.5f);   // renderer.rs:233-234 (the record keeps spf only)
                o.rhw = r.rhw[s];
                for (int k = 0; k < K; ++k) o.ctx[k] = vary[(size_t)slot * 3 * K + (size_t)s * K + k];
            }
        }
    }
    return FRR_OK;
}

int frr_get_stats(frr_ctx *c, frr_stats *out)
{
    if (!c || !out) return FRR_ERR_INVALID;
    { int rc = finish(c); if (rc != FRR_OK) return rc; }
    const FrameState &f = c->fs;
    const Lane &h = c->hc.lane[f.lane];
    const bool tot = h.totals_frame == f.frame_no;
    memset(out, 0, sizeof *out);
    out->tris_in = f.tris_in;
    out->draws = f.draws;
    out->replays = c->replays;
    out->frag_covered = tot ? h.tot_frag_covered : 0;
    out->frag_nan = tot ? h.tot_frag_nan : 0;
    out->bin_entries = tot ? h.tot_bin_entries : 0;
    for (int p = 0; p < 2; ++p) {
        if (h.gtab[p].frame_no == f.frame_no) { out->frag_covered += h.gtab[p].frag_covered; out->frag_nan += h.gtab[p].frag_nan; }
        if (h.btab[p].frame_no == f.frame_no) out->bin_entries += h.btab[p].seg_total;
    }
    if (f.draws && h.gtab[f.gpar()].frame_no == f.frame_no) out->tris_setup = (uint64_t)h.gtab[f.gpar()].tri_base + h.gtab[f.gpar()].n_emit;
#ifdef FRR_DEBUG_COUNTERS
    if (getenv("FRR_DEBUG_PRINT")) {
        fprintf(stderr, "frr dbg:");
        for (int k = 0; k < 24; ++k) {
            unsigned long long v = 0;
            for (int j = 0; j < DBG_COPIES; ++j) v += c->hc.dbg[j][k];
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
    // (a measuring call: the caller's stream first waits for the ctx's other streams, so that the event brackets whole frames)
    if (c->g_used) { HIP_TRY(c, hipEventRecord(c->ev_join, c->gstream)); HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_join, 0)); }
    { int rcj = join_tile_streams(c); if (rcj != FRR_OK) return rcj; }
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
    { int rc = drain(c); if (rc != FRR_OK) return rc; }
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
