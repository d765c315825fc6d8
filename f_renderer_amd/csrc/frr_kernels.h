// frr_kernels.h -- the gfx950 kernels of the rasterization path (wave64, LDS-tiled).
//
// Frame = clear -> per draw { geometry: count, emit (clipping inside, wave-cooperative) ;
// binning: one segmented LDS multi-split launch ; raster+resolve per 32x32 tile }.  See DESIGN.md for the roofline of each kernel.
#pragma once
#include "frr_device.h"
#ifndef __HIPCC_RTC__
#include <type_traits>
#endif

namespace frr {

// ---------------------------------------------------------------------------------------------
// K0 clear: FrameBuffer::fill (renderer.rs:485-494) + depth_buffer.fill (phong.rs:317) + ids.
// 16 B per lane streaming stores.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_clear(uint4 *__restrict__ color, uint4 *__restrict__ depth,
                                               uint4 *__restrict__ ids, uint32_t n4, uint32_t rgba, float d)
{
    const uint32_t db = f2u(d);
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n4; i += gridDim.x * 256u) {
        color[i] = make_uint4(rgba, rgba, rgba, rgba);
        depth[i] = make_uint4(db, db, db, db);
        ids[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
    }
}
// the pixel rows of the tile rows a partitioned ctx does NOT own (they are skipped by a fused clear and
// brought up to date only when somebody looks: frr_readback, frr_target_ptrs)
__global__ __launch_bounds__(256) void k_clear_unowned_rows(uint32_t *__restrict__ color, uint32_t *__restrict__ depth,
                                                            uint32_t *__restrict__ ids, uint32_t W, uint32_t H, RowOwner own,
                                                            uint32_t rgba, float d)
{
    const uint32_t y = blockIdx.x;
    if (y >= H || owns_tile_row((int)(y / TILE), own)) return;
    for (uint32_t x = threadIdx.x; x < W; x += 256u) {
        const size_t i = (size_t)y * W + x;
        color[i] = rgba; depth[i] = f2u(d); ids[i] = ~0u;
    }
}
// tail elements when W*H is not a multiple of 4
__global__ void k_clear_tail(uint32_t *color, uint32_t *depth, uint32_t *ids, uint32_t from, uint32_t n,
                             uint32_t rgba, float d)
{
    uint32_t i = from + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { color[i] = rgba; depth[i] = f2u(d); ids[i] = ~0u; }
}

// ---- block-level helpers -------------------------------------------------------------------
// wave64 inclusive prefix sum on the DPP network (row_shr within 16-lane rows, then row broadcasts)
__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t x)
{
    uint32_t v = x;
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false); // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false); // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); // row_bcast:15 -> rows 1,3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); // row_bcast:31 -> rows 2,3
    return v;
}

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) { return wave_incl_scan_dpp(v); }
// exclusive scan of one value per thread over a 256-thread block; returns block total via `total`
__device__ __forceinline__ uint32_t block_excl_scan256(uint32_t v, uint32_t *s_w /*[4]*/, uint32_t &total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = wave_incl_scan(v);
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    uint32_t w0 = s_w[0], w1 = s_w[1], w2 = s_w[2], w3 = s_w[3];
    uint32_t base = (w > 0 ? w0 : 0u) + (w > 1 ? w1 : 0u) + (w > 2 ? w2 : 0u);
    total = w0 + w1 + w2 + w3;
    return base + inc - v;
}

// ---- binning helpers (used by K2 and by the fused geometry + binning kernel) ------------------------------------
constexpr int BIN_WG = 1024;
constexpr int BIN_MAX_G = 256;
constexpr uint32_t BIN_LDS_MAX_TILES = 36864; // 144 KiB of u32 counters

struct TileRange { int tx0, tx1, ty0, ty1; }; // inclusive-exclusive tile ranges (window-local)


__device__ __forceinline__ TileRange tiles_of_pbox(const RasterArgs &a, const uint4 b)
{
    const int mnx = (int)(short)(b.x & 0xFFFFu), mny = (int)(short)(b.x >> 16);
    const int mxx = (int)(short)(b.y & 0xFFFFu), mxy = (int)(short)(b.y >> 16);
    const int minx = clampi(mnx, a.x0, a.x1), maxx = clampi(mxx, a.x0, a.x1);
    const int miny = clampi(mny, a.y0, a.y1), maxy = clampi(mxy, a.y0, a.y1);
    TileRange t;
    if (maxx <= minx || maxy <= miny) { t.tx0 = t.tx1 = t.ty0 = t.ty1 = 0; return t; }
    t.tx0 = (minx - a.x0) / TILE; t.tx1 = (maxx - 1 - a.x0) / TILE + 1;
    t.ty0 = (miny - a.y0) / TILE; t.ty1 = (maxy - 1 - a.y0) / TILE + 1;
    return t;
}

// Where a record goes.  Positions are RELATIVE to the workgroup's region of `bins` (the LDS cursors count from 0): the
// first stage_cap records of the region are collected in LDS in their final order and leave as coalesced full-line
// stores (a scattered 16-B store is a partial-line write: WRITE_SIZE showed 2.3x the bytes).  PutStaged: the whole
// region fits the staging, so nobody needs to know where the region starts until the walk is over (the global atomic
// that reserves it is in flight meanwhile); PutMixed: what does not fit goes straight to memory.  pos = ~0u: nothing.
struct PutNone { __device__ __forceinline__ void operator()(uint32_t, const uint4 &) const {} };
struct PutStaged {
    uint4 *s_stage;
    __device__ __forceinline__ void operator()(uint32_t pos, const uint4 &ent) const { if (pos != ~0u) s_stage[pos] = ent; }
};
struct PutMixed {
    uint4 *s_stage, *bins;
    uint32_t region, stage_cap, bin_cap;
    __device__ __forceinline__ void operator()(uint32_t pos, const uint4 &ent) const
    {
        if (pos != ~0u) {
            if (pos < stage_cap) s_stage[pos] = ent;
            else if (region + pos < bin_cap) bins[region + pos] = ent;   // (region == bin_cap: the frame overflowed, nothing is stored)
        }
    }
};

// One binning record per lane (pb = a pbox entry, all zero = nothing): count it (SCATTER = false) or place it (true) in every owned tile its clamped bbox touches.
// Called by whole waves: large footprints are spread over the lanes.
template <bool SCATTER, class PUT>
__device__ __forceinline__ void bin_one(const RasterArgs &a, uint32_t *s_hist, const uint4 pb, int lane, const PUT &put)
{
    const uint32_t i = pb.w;    // the slot the entry stands for
    const RowOwner own = {a.rank, a.world, a.blocked, a.brow0, a.brow1};
    const TileRange t = tiles_of_pbox(a, pb);
    const int ntx = t.tx1 - t.tx0, nty = t.ty1 - t.ty0;
    const int nt = ntx * nty;
    auto visit = [&](const uint4 &ent, int tx, int ty) {
        if (!owns_tile_row(ty, own)) return;
        const int tile = local_tile_row(ty, own) * a.tiles_x + tx;
        if constexpr (SCATTER) {
            put(atomicAdd(&s_hist[tile], 1u), ent);
        } else {
            atomicAdd(&s_hist[tile], 1u);
        }
    };
    const uint4 mine = make_uint4(i, pb.z, pb.x, pb.y);
    // footprints of at most 2x2 tiles (the common case) as straight-line code: up to four
    // independent LDS atomics in flight instead of a loop of dependent atomic -> store steps.  (A second straight-line
    // form for 3x3 footprints cost more than it saved: nearly every wave has ONE such lane and then runs all nine
    // predicated positions; 25.9 -> 23.0 us per launch on the 1080p frame without it.)
    const bool small = nt > 0 && ntx <= 2 && nty <= 2;
    if (small) {
        const bool own0 = owns_tile_row(t.ty0, own);
        const bool own1 = nty == 2 && owns_tile_row(t.ty0 + 1, own);
        const int t00 = local_tile_row(t.ty0, own) * a.tiles_x + t.tx0;
        const int t10 = local_tile_row(t.ty0 + 1, own) * a.tiles_x + t.tx0;
        const bool v0 = own0, v1 = own0 && ntx == 2, v2 = own1, v3 = own1 && ntx == 2;
        if constexpr (SCATTER) {
            uint32_t p0 = ~0u, p1 = ~0u, p2 = ~0u, p3 = ~0u;
            if (v0) p0 = atomicAdd(&s_hist[t00], 1u);
            if (v1) p1 = atomicAdd(&s_hist[t00 + 1], 1u);
            if (v2) p2 = atomicAdd(&s_hist[t10], 1u);
            if (v3) p3 = atomicAdd(&s_hist[t10 + 1], 1u);
            put(p0, mine); put(p1, mine); put(p2, mine); put(p3, mine);
        } else {
            if (v0) atomicAdd(&s_hist[t00], 1u);
            if (v1) atomicAdd(&s_hist[t00 + 1], 1u);
            if (v2) atomicAdd(&s_hist[t10], 1u);
            if (v3) atomicAdd(&s_hist[t10 + 1], 1u);
        }
    }
    // larger footprints: four at a time, each spread over a quarter of the wave (16 lanes)
    unsigned long long big = __ballot(nt > 0 && !small);
    while (big) {
        int src = -1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int sj = big ? __builtin_ctzll(big) : -1;
            big &= big - 1;                    // (0 stays 0)
            if ((lane >> 4) == j) src = sj;
        }
        const int sl = src < 0 ? 0 : src;
        // (cross-lane reads are unconditional: they must not sit under `src >= 0`, inactive source lanes read as 0)
        const uint32_t o0 = __shfl((uint32_t)t.tx0 | ((uint32_t)t.ty0 << 16), sl);   // tile coordinates are below 1024
        const uint32_t o1 = __shfl((uint32_t)ntx | ((uint32_t)nty << 16), sl);
        const int bx0 = (int)(o0 & 0xFFFFu), by0 = (int)(o0 >> 16), bnx = (int)(o1 & 0xFFFFu);
        const int bnt = src < 0 ? 0 : bnx * (int)(o1 >> 16);
        const uint4 ent = make_uint4(__shfl(i, sl), __shfl(pb.z, sl), __shfl(pb.x, sl), __shfl(pb.y, sl));
        // q -> (q % bnx, q / bnx) through a 1-ulp reciprocal: (q + 0.5) / bnx is at least 0.5 / bnx away from an integer
        const float inv = __builtin_amdgcn_rcpf((float)bnx);
        for (int q = lane & 15; q < bnt; q += 16) {
            const int qy = (int)(((float)q + 0.5f) * inv);
            visit(ent, bx0 + (q - qy * bnx), by0 + qy);
        }
    }
}

// The fan entries with VIRTUAL indices [lo, hi) (FanMap: the used parts of the fan regions laid end to end; index ntris
// + ... ), by the waves of a BIN_WG-thread workgroup: a wave takes 64 consecutive ones per round, BIN_PF rounds at a time --
// their bboxes are fetched up front (the loop is latency-bound otherwise).
#ifndef FRR_BIN_PF
#define FRR_BIN_PF 4
#endif
constexpr int BIN_PF = FRR_BIN_PF;
template <bool SCATTER, class PUT>
__device__ __forceinline__ void bin_walk_fans(const RasterArgs &a, uint32_t *s_hist, const FanMap &fm, const uint4 *__restrict__ pbox,
                                              uint32_t lo, uint32_t hi, int lane, uint32_t wave, const PUT &put)
{
    for (uint32_t base0 = lo + wave * 64u; base0 < hi; base0 += BIN_PF * BIN_WG) {
        uint4 pb[BIN_PF];
#pragma unroll
        for (int k = 0; k < BIN_PF; ++k) {
            const uint32_t v = base0 + k * BIN_WG + lane;
            pb[k] = v < hi ? pbox[fan_map_slot(fm, v)] : make_uint4(0u, 0u, 0u, 0u); // (0,0)-(0,0) is an empty box
        }
#pragma unroll
        for (int k = 0; k < BIN_PF; ++k) {
            if (base0 + k * BIN_WG >= hi) break;
            bin_one<SCATTER>(a, s_hist, pb[k], lane, put);
        }
    }
}
// The inputs of the geometry blocks [b_lo, b_hi).  A block's entries are dense from the start of its 256-entry range
// (GeomArgs::pbox), so only the 64-entry quarters below its count exist -- a rank of an 8-way partition walks an eighth of
// what the whole window walks.  The existing quarters of 64 blocks at a time are numbered by a prefix sum over the blocks'
// counts (every wave computes it: one load per lane and a DPP scan) and dealt to the waves round-robin, BIN_PF per wave
// and step, their bboxes fetched together.
template <bool SCATTER, class PUT>
__device__ __forceinline__ void bin_walk_blocks(const RasterArgs &a, uint32_t *s_hist, const uint4 *__restrict__ pbox,
                                                const uint32_t *__restrict__ bcount, uint32_t ntris, uint32_t b_lo, uint32_t b_hi,
                                                int lane, uint32_t wave, const PUT &put)
{
    if (a.world <= 1) {
        // the whole window: nearly every quarter exists, so they are taken as they come, without looking at the counts
        // (entries past a block's count are zero: an empty box)
        const uint32_t nq = (b_hi - b_lo) * (GEOM_BLOCK / 64);
        for (uint32_t q0 = wave; q0 < nq; q0 += BIN_PF * (BIN_WG / 64)) {
            uint4 pb[BIN_PF];
#pragma unroll
            for (int k = 0; k < BIN_PF; ++k) {
                const uint32_t q = q0 + k * (BIN_WG / 64);
                const uint32_t e = b_lo * GEOM_BLOCK + q * 64u + (uint32_t)lane;   // (the last block ends at ntris: fan entries follow)
                pb[k] = q < nq && e < ntris ? pbox[e] : make_uint4(0u, 0u, 0u, 0u);
            }
#pragma unroll
            for (int k = 0; k < BIN_PF; ++k)
                if (q0 + k * (BIN_WG / 64) < nq) bin_one<SCATTER>(a, s_hist, pb[k], lane, put);
        }
        return;
    }
    for (uint32_t bb = b_lo; bb < b_hi; bb += 64u) {
        const uint32_t c_l = bb + (uint32_t)lane < b_hi ? bcount[bb + lane] : 0u;
        const uint32_t incl = wave_incl_scan_dpp((c_l + 63u) >> 6);           // quarters of the blocks up to and including this lane's
        const uint32_t U = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        for (uint32_t u0 = wave; u0 < U; u0 += BIN_PF * (BIN_WG / 64)) {
            uint4 pb[BIN_PF];
#pragma unroll
            for (int k = 0; k < BIN_PF; ++k) {
                const uint32_t u = u0 + k * (BIN_WG / 64);                    // wave-uniform
                pb[k] = make_uint4(0u, 0u, 0u, 0u);
                if (u < U) {
                    const int blk = __popcll(__ballot(incl <= u));            // the block quarter u belongs to (lanes past the range hold U)
                    const uint32_t before = blk ? (uint32_t)__builtin_amdgcn_readlane((int)incl, blk - 1) : 0u;
                    const uint32_t cb = (uint32_t)__builtin_amdgcn_readlane((int)c_l, blk);
                    const uint32_t e = (u - before) * 64u + (uint32_t)lane;   // entry within the block
                    if (e < cb) pb[k] = pbox[(size_t)(bb + blk) * GEOM_BLOCK + e];
                }
            }
#pragma unroll
            for (int k = 0; k < BIN_PF; ++k)
                if (u0 + k * (BIN_WG / 64) < U) bin_one<SCATTER>(a, s_hist, pb[k], lane, put);
        }
    }
}

// Exclusive scan of the workgroup's tile histogram in place (counts -> where each tile's records start, relative to the
// workgroup's region); returns the region's size.  4096 tiles per step: every thread takes FOUR CONSECUTIVE counters as
// one 16-byte LDS access (a thread owning a longer slice of consecutive tiles reads LDS at a stride of the slice length:
// with 16 tiles per thread, the 4096^2 frame, that was a 32-way bank conflict and 14 us per workgroup); the wave totals
// are combined by a second DPP scan in every wave.  The histogram is padded with zeros to a multiple of four counters.
// sw: two alternating arrays of wave totals, so that one barrier per step is enough.
__device__ __forceinline__ uint32_t bin_scan_relative(uint32_t *s_hist, uint32_t ntiles, uint32_t (*sw)[BIN_WG / 64], int lane, uint32_t wave)
{
    static_assert(BIN_WG / 64 <= 16, "the wave totals are scanned inside one 16-lane DPP row");
    uint32_t carry = 0;
    int it = 0;
    for (uint32_t base = 0; base < ntiles; base += 4 * BIN_WG, it ^= 1) {
        const uint32_t i = base + 4u * threadIdx.x;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (i < ntiles) v = *reinterpret_cast<const uint4 *>(s_hist + i);
        const uint32_t s = (v.x + v.y) + (v.z + v.w);
        const uint32_t inc = wave_incl_scan_dpp(s);
        if (lane == 63) sw[it][wave] = inc;
        __syncthreads();
        const uint32_t wt = wave_incl_scan_dpp(lane < BIN_WG / 64 ? sw[it][lane] : 0u);
        const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)wt, BIN_WG / 64 - 1);
        const uint32_t wbase = wave ? (uint32_t)__builtin_amdgcn_readlane((int)wt, (int)wave - 1) : 0u;
        if (i < ntiles) {
            const uint32_t e0 = carry + wbase + inc - s;
            *reinterpret_cast<uint4 *>(s_hist + i) = make_uint4(e0, e0 + v.x, e0 + v.x + v.y, e0 + v.x + v.y + v.z);
        }
        carry += tot;
    }
    __syncthreads(); // the cursors start moving
    return carry;
}
// the workgroup's region of `bins`: ONE global atomic (by one thread; bin_cap when the frame overflowed, which flags it)
__device__ __forceinline__ uint32_t bin_region_of(const RasterArgs &a, unsigned long long before, uint32_t total)
{
    if (before + total > (unsigned long long)a.bin_cap) { seq_fail(a.cnt, a.seq, a.epoch, 2u); return a.bin_cap; }
    return (uint32_t)before;
}

// ---------------------------------------------------------------------------------------------
// K1 geometry in ONE pass (slots and order keys: frr_device.h).  Thread = input triangle:
// renderer.rs:113-148 (VS, reject, classify), then for the common unclipped case :180-218 (centroid +
// stable angle sort of 3), :220-235 (divide, viewport, snap), :237-243 and the per-triangle prologue of
// rasterization (:300-320), so that the record at slot t is ready to scan.  Clipped inputs reserve their
// fan slots with ONE atomic per block and are then expanded by the block's four waves, one wave per
// triangle (clip_triangle_wave).  Per block the kernel leaves the number of triangles it emits
// (block_sums, scanned later: geom_scan) and per input its fan size and emission offset (tinfo).
// HBM: reads the inputs once, writes 64 + 16 (+ 12K) bytes per emitted triangle.
// ---------------------------------------------------------------------------------------------
// Clipped triangles: the reference's quirky clipper (renderer.rs:150-171: one intersection per
// (edge, plane) with differing in/out flags, outside vertices kept), centroid + stable angle sort of
// up to 21 vertices (:180-218), fan emission (:245-266) -- done by ONE WAVE per triangle:
// lanes 0..17 = the 3 pairs x 6 planes in the reference's loop order, lanes 18..20 = the originals,
// so "list order" is simply lane order among the kept lanes.
constexpr int CLIP_MAXV = 21;
constexpr int CLIP_QUEUE_AT = 16;   // clipped inputs in one block from which on the block reports itself (the host then switches the clip queue on)
constexpr int CLIP_INBLOCK = GEOM_BLOCK / 64;   // clipped inputs a geometry block expands itself when the clip queue is in use

// Returns the lane's binning record (all zero for lanes that emit no fan triangle).  eoff: the input's emission offset
// within its geometry block (tinfo[t] >> FAN_BITS).
template <int VS>
__device__ __forceinline__ uint4 clip_triangle_wave(const GeomArgs &g, const DevUniforms &u, uint32_t t, uint32_t fbase, uint32_t eoff, int lane,
                                                   float (*s_xy)[2], int32_t *s_key, float (*s_v)[7 + (VSInfo<VS>::K > 0 ? VSInfo<VS>::K : 1)])
{
    constexpr int NF = VSInfo<VS>::NF, K = VSInfo<VS>::K, KS = K > 0 ? K : 1;
    float pos[3][4], ctx[3][KS];
    const float *in = g.in + (size_t)t * (3 * NF);
#pragma unroll
    for (int v = 0; v < 3; ++v) run_vs<VS, true>(u, in + v * NF, pos[v], ctx[v]);
    // this lane's vertex
    float p[4], c[KS];
    bool keep;
    if (lane < 18) {
        const int pi = lane / 6, plane = lane - pi * 6;      // pairs (0,1),(0,2),(1,2)
        float a[4], b[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { a[k] = pi == 2 ? pos[1][k] : pos[0][k]; b[k] = pi == 0 ? pos[1][k] : pos[2][k]; }
        const bool differ = ((inside_bits(a) ^ inside_bits(b)) >> plane) & 1u;
        const float r = intersect_ratio(plane, a, b);
#pragma unroll
        for (int k = 0; k < 4; ++k) p[k] = a[k] + r * (b[k] - a[k]);                       // :89
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float ca = pi == 2 ? ctx[1][k] : ctx[0][k], cb = pi == 0 ? ctx[1][k] : ctx[2][k];
            c[k] = ca + (cb - ca) * r;                                                     // :91
        }
        keep = differ && fabsf(p[3]) > CLIP_EPSILON;                                       // :161-166
    } else {
        const int v = lane - 18;
#pragma unroll
        for (int k = 0; k < 4; ++k) p[k] = v == 0 ? pos[0][k] : (v == 1 ? pos[1][k] : pos[2][k]);
#pragma unroll
        for (int k = 0; k < K; ++k) c[k] = v == 0 ? ctx[0][k] : (v == 1 ? ctx[1][k] : ctx[2][k]);
        keep = lane < CLIP_MAXV;                                                           // :171
    }
    const unsigned long long km = __ballot(keep);
    const int n = __popcll(km);
    const int li = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0u));
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (keep) { s_xy[li][0] = p[0]; s_xy[li][1] = p[1]; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    float cx = 0.0f, cy = 0.0f;                                                            // :180-187, list order
    for (int k = 0; k < n; ++k) { cx += s_xy[k][0]; cy += s_xy[k][1]; }
    const float inv_n = 1.0f / (float)n;
    cx *= inv_n; cy *= inv_n;
    const int32_t key = total_order_key(sort_angle(p[1] - cy, p[0] - cx));
    if (keep) s_key[li] = key;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    int rank = 0;                                                                          // stable sort position (:205-218)
    for (int k = 0; k < n; ++k) { const int32_t ok = s_key[k]; rank += (ok < key || (ok == key && k < li)) ? 1 : 0; }
    const ScreenVtx sv = to_screen(p, (float)g.width, (float)g.height);                    // :220-235
    if (keep) {
        float *o = s_v[rank];
        o[0] = sv.rhw; o[1] = sv.ndcx; o[2] = sv.ndcy; o[3] = sv.sx; o[4] = sv.sy; o[5] = __int_as_float(sv.ix); o[6] = __int_as_float(sv.iy);
#pragma unroll
        for (int k = 0; k < K; ++k) o[7 + k] = c[k];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // fan (:237-266): n == 3 -> [0,1,2]; else [0,last-1,last] for last = n-1 .. 4, then [0,2,3], [0,1,2]
    uint4 mine = make_uint4(0u, 0u, 0u, 0u);
    if (lane < n - 2) {
        const int q = lane;
        int i1, i2;
        if (q < n - 4) { i2 = n - 1 - q; i1 = i2 - 1; } else if (q == n - 4) { i1 = 2; i2 = 3; } else { i1 = 1; i2 = 2; }
        const float *v0 = s_v[0];
        const float nz = (s_v[i1][1] - v0[1]) * (s_v[i2][2] - v0[2]) - (s_v[i2][1] - v0[1]) * (s_v[i1][2] - v0[2]); // :300-305
        const bool swap = nz > 0.0f;
        const float *v1 = s_v[swap ? i2 : i1], *v2 = s_v[swap ? i1 : i2];                  // :309-312
        const int p0x = __float_as_int(v0[5]), p0y = __float_as_int(v0[6]), p1x = __float_as_int(v1[5]), p1y = __float_as_int(v1[6]);
        const int p2x = __float_as_int(v2[5]), p2y = __float_as_int(v2[6]);
        uint32_t flags = swap ? 1u : 0u;
        flags |= is_top_left(p0x, p0y, p1x, p1y) ? 0u : 2u;                                // :318-320
        flags |= is_top_left(p1x, p1y, p2x, p2y) ? 0u : 4u;
        flags |= is_top_left(p2x, p2y, p0x, p0y) ? 0u : 8u;
        const uint32_t fslot = fbase + (uint32_t)q;             // fan triangle q of input t: emission order is q order
        const uint32_t idx = g.ntris + fslot;
        g.fan_okey[fslot] = (t << FAN_BITS) + 1u + (uint32_t)q;
        { const uint2 pb = pack_pbox(p0x, p0y, p1x, p1y, p2x, p2y); mine = make_uint4(pb.x, pb.y, cull_zub(v0[0], v1[0], v2[0]), idx); g.pbox[idx] = mine; }
        const EdgeWords e01 = edge_words(p0x, p0y, p1x, p1y, (flags >> 1) & 1u);
        const EdgeWords e12 = edge_words(p1x, p1y, p2x, p2y, (flags >> 2) & 1u);
        const EdgeWords e20 = edge_words(p2x, p2y, p0x, p0y, (flags >> 3) & 1u);
        flags |= (e01.pos | (e12.pos << 1) | (e20.pos << 2)) << REC_POS_SHIFT;
        flags |= (eoff + (uint32_t)q) << REC_EOFF_SHIFT;     // emission offset within the input's geometry block
        uint4 *dst = reinterpret_cast<uint4 *>(g.recs + idx);
        dst[0] = make_uint4(e01.kd, e01.c, e12.kd, e12.c);
        dst[1] = make_uint4(e20.kd, e20.c, f2u(v0[3]), f2u(v0[4]));
        dst[2] = make_uint4(f2u(v1[3]), f2u(v1[4]), f2u(v2[3]), f2u(v2[4]));
        dst[3] = make_uint4(f2u(v0[0]), f2u(v1[0]), f2u(v2[0]), flags);
        if constexpr (K > 0) {
            float *o = g.vary + (size_t)idx * (3 * K);
#pragma unroll
            for (int k = 0; k < K; ++k) { o[k] = v0[7 + k]; o[K + k] = v1[7 + k]; o[2 * K + k] = v2[7 + k]; }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // staging is reused by the wave's next triangle
    return mine;
}

// The unclipped case of geometry_processing after the vertex shader, for one triangle: renderer.rs:180-218 (centroid +
// stable angle sort of 3), :220-235 (divide, viewport, snap -- done by the caller: s0..s2), :237-243, and the
// per-triangle prologue of rasterization (:300-320: orientation swap, top-left flags).  Returns the 64-byte record
// (q0..q3), the binning record, and where input vertices 0 and 1 ended up (d0, d1: for the varyings).
struct SetupOut { uint4 q0, q1, q2, q3, pbox; int d0, d1; };
__device__ __forceinline__ SetupOut setup_unclipped(const float pos[3][4], const ScreenVtx &s0, const ScreenVtx &s1, const ScreenVtx &s2, uint32_t eoff)
{
    SetupOut o;
    // centroid (:180-187), n == 3
    float cx = 0.0f, cy = 0.0f;
#pragma unroll
    for (int v = 0; v < 3; ++v) { cx += pos[v][0]; cy += pos[v][1]; }
    const float inv_n = 1.0f / 3.0f;
    cx *= inv_n; cy *= inv_n;
    // The sort of :205-218 only needs the ORDER of the three angles.  With d_v = vertex - centroid
    // (the very f32 values atan2 would see), angle(a) < angle(b) in [0, 2pi) is decided by the
    // half plane (sign of dy) and the sign of cross(a, b) -- exact whenever the true angles are
    // farther apart, and farther from the 0 / pi axis, than SORT_EPS radians, which dwarfs the
    // < 1e-6 error of the f32 atan2 + 2pi the reference compares.  Anything closer (or non-finite)
    // takes the exact atan2f keys.
    float dx[3], dy[3], d2[3];
#pragma unroll
    for (int v = 0; v < 3; ++v) { dx[v] = pos[v][0] - cx; dy[v] = pos[v][1] - cy; d2[v] = dx[v] * dx[v] + dy[v] * dy[v]; }
    constexpr float SORT_EPS2 = 1.0e-8f; // (1e-4 rad)^2
    bool lt10, lt20, lt21, le01;
    {
        bool safe = true;
#pragma unroll
        for (int v = 0; v < 3; ++v) safe = safe && (dy[v] * dy[v] > SORT_EPS2 * d2[v]) && (d2[v] < 1.0e30f);
        auto less = [&](int a, int b) { // angle(a) < angle(b)
            const float cr = dx[a] * dy[b] - dy[a] * dx[b];
            const bool ha = dy[a] < 0.0f, hb = dy[b] < 0.0f;
            safe = safe && (ha != hb || cr * cr > SORT_EPS2 * (d2[a] * d2[b]));
            return ha != hb ? hb : cr > 0.0f;
        };
        lt10 = less(1, 0); lt20 = less(2, 0); lt21 = less(2, 1); le01 = !lt10;
        if (!safe) {
            int32_t key[3];
#pragma unroll
            for (int v = 0; v < 3; ++v) key[v] = total_order_key(sort_angle(dy[v], dx[v]));
            lt10 = key[1] < key[0]; lt20 = key[2] < key[0]; lt21 = key[2] < key[1]; le01 = key[0] <= key[1];
        }
    }
    // stable rank of each vertex (:205-218)
    int r0 = (int)lt10 + (int)lt20;
    int r1 = (int)le01 + (int)lt21;
    // (the third rank is implied by the other two)
    // Everything below is a permutation of the three vertices: sorted order, then the orientation
    // swap of renderer.rs:300-312.  It is written as scalar selects on the destination slot of each
    // input vertex (struct-valued selects end up as runtime-indexed scratch).
    auto by_rank = [&](int r, float x0, float x1, float x2) { return r0 == r ? x0 : (r1 == r ? x1 : x2); };
    const float ax = by_rank(0, s0.ndcx, s1.ndcx, s2.ndcx), ay = by_rank(0, s0.ndcy, s1.ndcy, s2.ndcy);
    const float bx = by_rank(1, s0.ndcx, s1.ndcx, s2.ndcx), by = by_rank(1, s0.ndcy, s1.ndcy, s2.ndcy);
    const float cx2 = by_rank(2, s0.ndcx, s1.ndcx, s2.ndcx), cy2 = by_rank(2, s0.ndcy, s1.ndcy, s2.ndcy);
    const float v01x = bx - ax, v01y = by - ay, v02x = cx2 - ax, v02y = cy2 - ay;
    const bool swap = (v01x * v02y - v02x * v01y) > 0.0f;                 // :300-309
    auto slot_of = [&](int r) { return r == 0 ? 0 : ((r == 1) != swap ? 1 : 2); };
    const int d0 = slot_of(r0), d1 = slot_of(r1);                        // destination slot of input vertex 0, 1 (2: the other)
    auto in_slot_f = [&](int s, float x0, float x1, float x2) { return d0 == s ? x0 : (d1 == s ? x1 : x2); };
    auto in_slot_i = [&](int s, int x0, int x1, int x2) { return d0 == s ? x0 : (d1 == s ? x1 : x2); };
    int px[3], py[3];
    float sx[3], sy[3], rw[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        px[s] = in_slot_i(s, s0.ix, s1.ix, s2.ix); py[s] = in_slot_i(s, s0.iy, s1.iy, s2.iy);
        sx[s] = in_slot_f(s, s0.sx, s1.sx, s2.sx); sy[s] = in_slot_f(s, s0.sy, s1.sy, s2.sy);
        rw[s] = in_slot_f(s, s0.rhw, s1.rhw, s2.rhw);
    }
    uint32_t flags = swap ? 1u : 0u;
    flags |= is_top_left(px[0], py[0], px[1], py[1]) ? 0u : 2u;           // :318-320
    flags |= is_top_left(px[1], py[1], px[2], py[2]) ? 0u : 4u;
    flags |= is_top_left(px[2], py[2], px[0], py[0]) ? 0u : 8u;
    // the tile-independent half of the span set-up (frr_device.h: edge_words), in place of spi
    const EdgeWords e01 = edge_words(px[0], py[0], px[1], py[1], (flags >> 1) & 1u);
    const EdgeWords e12 = edge_words(px[1], py[1], px[2], py[2], (flags >> 2) & 1u);
    const EdgeWords e20 = edge_words(px[2], py[2], px[0], py[0], (flags >> 3) & 1u);
    flags |= (e01.pos | (e12.pos << 1) | (e20.pos << 2)) << REC_POS_SHIFT;
    flags |= eoff << REC_EOFF_SHIFT;
    { const uint2 pb = pack_pbox(px[0], py[0], px[1], py[1], px[2], py[2]); o.pbox = make_uint4(pb.x, pb.y, cull_zub(rw[0], rw[1], rw[2]), 0u); }
    o.q0 = make_uint4(e01.kd, e01.c, e12.kd, e12.c);
    o.q1 = make_uint4(e20.kd, e20.c, f2u(sx[0]), f2u(sy[0]));
    o.q2 = make_uint4(f2u(sx[1]), f2u(sy[1]), f2u(sx[2]), f2u(sy[2]));
    o.q3 = make_uint4(f2u(rw[0]), f2u(rw[1]), f2u(rw[2]), flags);
    o.d0 = d0; o.d1 = d1;
    return o;
}

// per-pass bookkeeping, by ONE thread of the pass's geometry kernel.  It runs beside the kernel's other blocks, so it
// touches nothing they touch: the fan cursors of THIS pass's table were zeroed by the pass before (every geometry path
// zeroes the other table's -- the tables alternate per pass), and flags are raised only by later kernels (geom_scan,
// binning).  The table being recycled belongs to the pass two back, whose tile kernels have finished (the host orders
// that with an event when the passes run on two streams): its fragment counts are folded into the frame totals.
__device__ __forceinline__ void geom_bookkeeping(const GeomArgs &g)
{
    Counters *cnt = g.cnt;
    Lane &L = cnt->lane[g.lane];
    totals_for_frame(cnt, L, g.frame_no);
    GeomTab &me = L.gtab[g.gpar], &prev = L.gtab[g.gpar ^ 1];
    if (me.frame_no == g.frame_no) { L.tot_frag_covered += me.frag_covered; L.tot_frag_nan += me.frag_nan; }
    me.frag_covered = 0ull; me.frag_nan = 0ull;
    me.tri_base = prev.frame_no == g.frame_no ? prev.tri_base + prev.n_emit : 0u; // the previous pass's triangles precede this pass's (its block sums are scanned by now)
    me.n_emit = 0u; me.need_fans = 0u;
    me.frame_no = g.frame_no;
#pragma unroll
    for (int k = 0; k < FAN_REGIONS; ++k) prev.fan_cursor[k].v = 0u;
    prev.clip_q = 0u; prev.clip_block_max = 0u;
    me.ntris_draw = g.ntris;
    me.tinfo = g.tinfo; me.fanbase = g.fanbase; me.fan_okey = g.fan_okey; me.block_prefix = g.block_prefix;
}

template <int VS>
__global__ __launch_bounds__(GEOM_BLOCK) void k_geom_single(GeomArgs g, DevUniforms u)
{
    __shared__ uint32_t s_w[4];
    __shared__ uint4 s_stage[GEOM_BLOCK / 64][64 * 5];    // per wave: 64 records at an 80-byte stride (see the record stores)
    __shared__ float s_cxy[GEOM_BLOCK / 64][CLIP_MAXV][2]; // per wave: clip x,y in list order
    __shared__ int32_t s_ckey[GEOM_BLOCK / 64][CLIP_MAXV];
    __shared__ float s_cv[GEOM_BLOCK / 64][CLIP_MAXV][7 + (VSInfo<VS>::K > 0 ? VSInfo<VS>::K : 1)];
    __shared__ uint32_t s_cl[GEOM_BLOCK];                 // the block's clipped inputs: thread | fan offset within the block << 8
    __shared__ uint16_t s_ce[GEOM_BLOCK];                 // per thread: its input's emission offset within the block (clipped inputs)
    __shared__ uint32_t s_slot[GEOM_BLOCK / 64][64];      // per wave: the slot of the k-th lane that stores a record (when they are not a run)
    __shared__ uint32_t s_ncl, s_fbase;
    __shared__ uint32_t s_wown[GEOM_BLOCK / 64];          // per wave: triangles it sets up (the block's dense binning entries)
    constexpr int NF = VSInfo<VS>::NF, K = VSInfo<VS>::K;
    const uint32_t bid = blockIdx.x;
    const uint32_t t = bid * GEOM_BLOCK + threadIdx.x;
    const uint32_t first_bad = seq_first_bad(g.cnt);   // (tested below, where the vertex loads are waited for anyway)
    GeomTab *const gt = &g.cnt->lane[g.lane].gtab[g.gpar];
    if (threadIdx.x == 0) s_ncl = 0; // ordered before its use by the barriers inside block_excl_scan256
    float pos[3][4];
    float ctx[3][K > 0 ? K : 1];
    uint32_t n = 0;   // triangles this input emits (the reference's count)
    bool clipped = false;
    if (t < g.ntris) {
        const float *in = g.in + (size_t)t * (3 * NF);
#pragma unroll
        for (int v = 0; v < 3; ++v) run_vs<VS, true>(u, in + v * NF, pos[v], ctx[v]);
        n = classify(pos, clipped);
    }
    if (seq_is_cancelled(first_bad, g.seq, g.epoch, false)) return;   // an earlier command failed: the host replays from there (nothing written yet)
    if (bid == 0 && threadIdx.x == 0) geom_bookkeeping(g);
    // Multi-GPU: a clipped input is expanded (and gets fan slots) only where its fan can reach a tile row of this rank.
    // The fan's vertices are the three originals and intersections a + r(b - a); with all w > 0 and no vertex beyond the
    // near plane (whose ratio is the reference's a_w / (a_w - b_w), not a point of the edge: renderer.rs:70) every r is in
    // [0, 1] up to rounding, so the fan's screen rows lie within the originals' rows -- widened by one for that rounding.
    // Its triangle COUNT (n: ids of everything after it) is kept either way.
    bool fan_here = clipped;
    if (clipped && g.part_world > 1) {
        const uint32_t in_all = inside_bits(pos[0]) & inside_bits(pos[1]) & inside_bits(pos[2]);
        if ((in_all & 16u) && pos[0][3] > 0.0f && pos[1][3] > 0.0f && pos[2][3] > 0.0f) {
            const float fw = (float)g.width, fh = (float)g.height;
            const int y0 = clampi(to_screen(pos[0], fw, fh).iy, -(1 << 30), 1 << 30), y1 = clampi(to_screen(pos[1], fw, fh).iy, -(1 << 30), 1 << 30);
            const int y2 = clampi(to_screen(pos[2], fw, fh).iy, -(1 << 30), 1 << 30);
            fan_here = tri_rows_owned(g, min(y0, min(y1, y2)) - 1, max(y0, max(y1, y2)) + 1, y0);
        }
    }
    // ONE block scan for two prefix sums: emission offset within the block (low half) and fan slots of the clipped inputs
    // before this one (high half); an input emits at most 19 triangles, so a block's sums stay below 2^13
    uint32_t ptotal;
    const uint32_t poff = block_excl_scan256(n | ((fan_here ? n : 0u) << 16), s_w, ptotal);
    const uint32_t eoff = poff & 0xFFFFu, foff = poff >> 16, total = ptotal & 0xFFFFu, ftotal = ptotal >> 16;
    // ONE returning atomic per block that clips anything.  Device atomics on one address are served one after the other
    // (~17 ns each): when most blocks of a mesh clip something -- a mesh that crosses the frustum -- that is many
    // microseconds of queueing, so the result is not waited for here: the block sets up and stores its unclipped
    // triangles first (the 250,000-triangle sheets of the 4K frame, with the eight cursors: 41 -> 22 us)
    uint32_t fbase_pending = 0u;
    if (threadIdx.x == 0) {
        g.block_sums[bid] = total;
        if (ftotal) fbase_pending = atomicAdd(&gt->fan_cursor[bid % FAN_REGIONS].v, ftotal);   // (within the block's region)
    }
    if (t < g.ntris) g.tinfo[t] = n | (eoff << FAN_BITS);
    if (fan_here) { s_cl[atomicAdd(&s_ncl, 1u)] = threadIdx.x | (foff << 8); s_ce[threadIdx.x] = (uint16_t)eoff; }
    // Multi-GPU: a rank that owns none of the tile rows an (unclipped) triangle's bbox touches never reads its record
    // (geometry is replicated, so this is what keeps the replicated part small)
    ScreenVtx s0 = {}, s1 = {}, s2 = {};
    bool emit = false;
    if (n == 1u && !clipped) {
        const float fw = (float)g.width, fh = (float)g.height;
        s0 = to_screen(pos[0], fw, fh); s1 = to_screen(pos[1], fw, fh); s2 = to_screen(pos[2], fw, fh);
        emit = tri_rows_owned(g, s0.iy, s1.iy, s2.iy);
    }
    // the block's binning entries: dense from the start of its range, zero behind them (GeomArgs::pbox)
    const unsigned long long own_m = __ballot(emit);
    if ((threadIdx.x & 63) == 0) s_wown[threadIdx.x >> 6] = (uint32_t)__popcll(own_m);
    __syncthreads();
    uint32_t own_base = 0, own_total = 0;
#pragma unroll
    for (int k = 0; k < GEOM_BLOCK / 64; ++k) { const uint32_t x = s_wown[k]; if (k < (int)(threadIdx.x >> 6)) own_base += x; own_total += x; }
    if (threadIdx.x == 0) g.bcount[bid] = own_total;
    if (threadIdx.x >= own_total && t < g.ntris) g.pbox[t] = make_uint4(0u, 0u, 0u, 0u);
    if (emit) {
    const SetupOut so = setup_unclipped(pos, s0, s1, s2, eoff);
    {
        const uint32_t dense = own_base + __builtin_amdgcn_mbcnt_hi((uint32_t)(own_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)own_m, 0u));
        g.pbox[bid * GEOM_BLOCK + dense] = make_uint4(so.pbox.x, so.pbox.y, so.pbox.z, t);
    }
    const uint4 q0 = so.q0, q1 = so.q1, q2 = so.q2, q3 = so.q3;
    auto in_slot_f = [&](int sl, float x0, float x1, float x2) { return so.d0 == sl ? x0 : (so.d1 == sl ? x1 : x2); };
    const unsigned long long am = __ballot(true);
    const int rk = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(am >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)am, 0u));
    const int np = __popcll(am);
    const uint32_t t0 = __builtin_amdgcn_readfirstlane(t);
    const bool run = __ballot(t == t0 + (uint32_t)rk) == am;   // the lanes here hold consecutive slots t0, t0+1, ...
    // The lanes that are here write the record of their own slot t.  A lane-per-record store writes 16 B at a 64-B stride
    // (64 partial-line writes per instruction, write-through); staged through the wave's LDS (80-B record stride:
    // conflict-free b128 writes) and read back four lanes per record, the same bytes leave as whole records -- as one
    // contiguous range when the lanes hold a run of consecutive slots (the usual case: every lane), else record by record
    // at the slots listed in s_slot (meshes with culled or clipped triangles in between: most waves of the 4K frame).
    uint4 *const st = s_stage[threadIdx.x >> 6];
    uint32_t *const slots = s_slot[threadIdx.x >> 6];
    if (!run) slots[rk] = t;
    auto slot_at = [&](int r) { return run ? t0 + (uint32_t)r : slots[r]; };
    {
        st[rk * 5 + 0] = q0; st[rk * 5 + 1] = q1; st[rk * 5 + 2] = q2; st[rk * 5 + 3] = q3;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = rk + j * np, r = c >> 2;
            reinterpret_cast<uint4 *>(g.recs + slot_at(r))[c & 3] = st[r * 5 + (c & 3)];
        }
    }
    if constexpr (K > 0) {
        float vv[3 * K];
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int k = 0; k < K; ++k) vv[s * K + k] = in_slot_f(s, ctx[0][k], ctx[1][k], ctx[2][k]);
        if constexpr (K == 8) {
            // 96 bytes of varyings per triangle: stored lane by lane that is 24 four-byte writes at a 96-byte stride
            // (partial lines); staged the same way, 32 triangles at a time at a 7 x 16-byte stride, they leave six lanes
            // per triangle as 16-byte stores
            for (int lo = 0; lo < np; lo += 32) {
                const int nh = min(np - lo, 32);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // (the staging is in use again)
                __builtin_amdgcn_wave_barrier();
                if (rk >= lo && rk < lo + 32) {
#pragma unroll
                    for (int j = 0; j < 6; ++j)
                        st[(rk - lo) * 7 + j] = make_uint4(f2u(vv[4 * j]), f2u(vv[4 * j + 1]), f2u(vv[4 * j + 2]), f2u(vv[4 * j + 3]));
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                for (int c = rk; c < nh * 6; c += np) {
                    const int tri = (c * 171) >> 10, part = c - tri * 6;   // c / 6, c % 6 for c < 192
                    reinterpret_cast<uint4 *>(g.vary + (size_t)slot_at(lo + tri) * (3 * K))[part] = st[tri * 7 + part];
                }
            }
        } else {
            float *o = g.vary + (size_t)t * (3 * K);
#pragma unroll
            for (int q = 0; q < 3 * K; ++q) o[q] = vv[q];
        }
    }
    } // emit
    // the block's clipped inputs, one wave per triangle (lanes = candidate vertices)
    if (threadIdx.x == 0) s_fbase = fbase_pending;
    __syncthreads();
    const uint32_t fregion = g.fan_cap / FAN_REGIONS;
    const bool fans_ok = s_fbase + ftotal <= fregion;  // else the frame is flagged invalid by geom_scan (a cursor > its region) and re-issued
    const uint32_t fbase = (bid % FAN_REGIONS) * fregion + s_fbase;   // first fan slot of the block, relative to ntris
    if (fan_here) g.fanbase[t] = fbase + foff;
    const uint32_t ncl = fans_ok ? s_ncl : 0u;
    if (ncl) {
        // Meshes clip along lines that follow their index order (a grid sheet crossing a frustum plane: whole runs of
        // consecutive triangles), so some blocks have hundreds of clipped inputs and most have none: expanded where they
        // are, those blocks are the kernel's tail.  With the queue in use a block expands CLIP_INBLOCK of its own (one
        // per wave) and hands the rest to k_geom_clip.
        uint32_t nin = ncl;
        if (threadIdx.x == 0 && ncl > (uint32_t)CLIP_QUEUE_AT) atomicMax(&gt->clip_block_max, ncl);   // (few blocks: see above)
        if (g.use_clipq && ncl > (uint32_t)CLIP_INBLOCK) {
            nin = CLIP_INBLOCK;
            if (threadIdx.x == 0) s_fbase = atomicAdd(&gt->clip_q, ncl - nin);   // (s_fbase has been read by everyone)
            __syncthreads();
            const uint32_t qb = s_fbase;
            for (uint32_t i = threadIdx.x; i < ncl - nin; i += GEOM_BLOCK) {
                const uint32_t en = s_cl[nin + i];
                g.clipq[qb + i] = make_uint2(bid * GEOM_BLOCK + (en & 255u), fbase + (en >> 8));
            }
        }
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        for (uint32_t e = (uint32_t)w; e < nin; e += GEOM_BLOCK / 64) {
            const uint32_t en = s_cl[e];
            (void)clip_triangle_wave<VS>(g, u, bid * GEOM_BLOCK + (en & 255u), fbase + (en >> 8), s_ce[en & 255u], lane, s_cxy[w], s_ckey[w], s_cv[w]);
        }
    }
}

// the clipped inputs the geometry blocks queued (GeomArgs::use_clipq), one wavefront per triangle over the whole chip
template <int VS>
__global__ __launch_bounds__(GEOM_BLOCK) void k_geom_clip(GeomArgs g, DevUniforms u)
{
    __shared__ float s_cxy[GEOM_BLOCK / 64][CLIP_MAXV][2];
    __shared__ int32_t s_ckey[GEOM_BLOCK / 64][CLIP_MAXV];
    __shared__ float s_cv[GEOM_BLOCK / 64][CLIP_MAXV][7 + (VSInfo<VS>::K > 0 ? VSInfo<VS>::K : 1)];
    if (seq_cancelled(g.cnt, g.seq, g.epoch, false)) return;
    const uint32_t n = g.cnt->lane[g.lane].gtab[g.gpar].clip_q;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t stride = gridDim.x * (GEOM_BLOCK / 64);
    for (uint32_t e = blockIdx.x * (GEOM_BLOCK / 64) + (uint32_t)w; e < n; e += stride) {
        const uint2 q = g.clipq[e];
        (void)clip_triangle_wave<VS>(g, u, q.x, q.y, g.tinfo[q.x] >> FAN_BITS, lane, s_cxy[w], s_ckey[w], s_cv[w]);
    }
}

// An empty mesh launches no geometry kernel proper: the per-draw bookkeeping alone.
__global__ void k_geom_empty(GeomArgs g)
{
    if (seq_cancelled(g.cnt, g.seq, g.epoch, false)) return;
    if (threadIdx.x == 0 && blockIdx.x == 0) geom_bookkeeping(g);
}

// Exclusive scan of the draw's block sums (in place: they become Counters::block_prefix), by one workgroup of 1024
// threads (the extra workgroup of k_bin_seg, or k_geom_scan on its own); publishes n_emit and raises the capacity flag
// when the draw asked for more fan slots than there are.
__device__ __forceinline__ void geom_scan(const uint32_t *sums, uint32_t *prefix, uint32_t nblocks, Counters *cnt, int tab_lane, int gpar, uint32_t fan_cap, uint32_t seq, uint32_t epoch)
{
    __shared__ uint32_t s_sw[16];
    __shared__ uint32_t s_carry;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nblocks; base += 1024) {
        uint32_t i = base + threadIdx.x;
        uint32_t v = i < nblocks ? sums[i] : 0u;
        uint32_t inc = wave_incl_scan(v);
        if (lane == 63) s_sw[w] = inc;
        __syncthreads();
        uint32_t wbase = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { uint32_t x = s_sw[k]; if (k < w) wbase += x; tot += x; }
        uint32_t carry = s_carry;
        if (i < nblocks) prefix[i] = carry + wbase + inc - v;
        __syncthreads();
        if (threadIdx.x == 0) s_carry = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        GeomTab &gt = cnt->lane[tab_lane].gtab[gpar];
        gt.n_emit = s_carry;
        uint32_t fans = 0;   // the fullest region decides: every region has fan_cap / FAN_REGIONS slots
#pragma unroll
        for (int k = 0; k < FAN_REGIONS; ++k)
            fans = max(fans, __hip_atomic_load(&gt.fan_cursor[k].v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        gt.need_fans = fans * FAN_REGIONS;
        if (fans > fan_cap / FAN_REGIONS) seq_fail(cnt, seq, epoch, 1u);
    }
}
__global__ __launch_bounds__(1024) void k_geom_scan(const uint32_t *sums, uint32_t *prefix, uint32_t nblocks, Counters *cnt, int lane, int gpar, uint32_t fan_cap, uint32_t seq, uint32_t epoch)
{
    if (seq_cancelled(cnt, seq, epoch, false)) return;
    geom_scan(sums, prefix, nblocks, cnt, lane, gpar, fan_cap, seq, epoch);
}

// ---------------------------------------------------------------------------------------------
// K2 binning.  A setup triangle goes into every OWNED 32x32 tile its clamped bbox
// (renderer.rs:285-298) touches.  Small footprints are handled per lane; a triangle touching
// more than BIN_COOP tiles is spread over the 64 lanes of its wave.
// ---------------------------------------------------------------------------------------------
constexpr int BIN_COOP = 6;

template <bool FILL>
__global__ __launch_bounds__(256) void k_bin(RasterArgs a, uint32_t fan_cap)
{
    if (seq_cancelled(a.cnt, a.seq, a.epoch, false)) return;
    const FanMap fm = fan_map(&a.cnt->lane[a.lane].gtab[a.gpar], fan_cap);
    const uint32_t n = fan_map_total(fm);
    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane((blockIdx.x * 256u + threadIdx.x) >> 6);
    const uint32_t nwaves = gridDim.x * 4u;
    for (uint32_t base = wave * 64u; base < n; base += nwaves * 64u) {
        // entry index: the inputs' entries (dense per block, zero = nothing, GeomArgs::pbox), then the used fan entries
        const uint4 cu = base + lane < n ? a.pbox[fan_map_slot(fm, base + lane)] : make_uint4(0u, 0u, 0u, 0u); // (0,0)-(0,0) is an empty box
        const uint32_t i = cu.w;   // the slot the entry stands for
        const RowOwner own = {a.rank, a.world, a.blocked, a.brow0, a.brow1};
        const TileRange t = tiles_of_pbox(a, cu);
        const int ntx = t.tx1 - t.tx0, nty = t.ty1 - t.ty0;
        const int nt = ntx * nty;
        auto visit = [&](const uint4 &ent, int tx, int ty) {
            if (!owns_tile_row(ty, own)) return;
            const int tile = ty * a.tiles_x + tx;
            if constexpr (FILL) {
                uint32_t pos = a.tile_offsets[tile] + atomicAdd(&a.tile_cursor[tile], 1u);
                if (pos < a.bin_cap) a.bins[pos] = ent;
            } else {
                atomicAdd(&a.tile_counts[tile], 1u);
            }
        };
        const uint4 mine = make_uint4(i, cu.z, cu.x, cu.y);
        if (nt > 0 && nt <= BIN_COOP)
            for (int ty = t.ty0; ty < t.ty1; ++ty)
                for (int tx = t.tx0; tx < t.tx1; ++tx) visit(mine, tx, ty);
        unsigned long long big = __ballot(nt > BIN_COOP);
        while (big) {
            const int src = __builtin_ctzll(big);
            big &= big - 1;
            const int bx0 = __shfl(t.tx0, src), by0 = __shfl(t.ty0, src), bnx = __shfl(ntx, src), bnt = __shfl(nt, src);
            const uint4 ent = make_uint4(__shfl(i, src), __shfl(cu.z, src), __shfl(cu.x, src), __shfl(cu.y, src));
            for (int k = lane; k < bnt; k += 64) visit(ent, bx0 + k % bnx, by0 + k / bnx);
        }
    }
}

// exclusive scan of the tile counts (single workgroup); zeroes the counts for the next draw and
// the fill cursors; publishes bin_total and the bin-capacity overflow flag.
__global__ __launch_bounds__(1024) void k_tile_scan(RasterArgs a, uint32_t ntiles)
{
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_carry;
    if (seq_cancelled(a.cnt, a.seq, a.epoch, false)) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < ntiles; base += 1024) {
        uint32_t i = base + threadIdx.x;
        uint32_t v = i < ntiles ? a.tile_counts[i] : 0u;
        uint32_t inc = wave_incl_scan(v);
        if (lane == 63) s_w[w] = inc;
        __syncthreads();
        uint32_t wbase = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { uint32_t x = s_w[k]; if (k < w) wbase += x; tot += x; }
        uint32_t carry = s_carry;
        if (i < ntiles) { a.tile_offsets[i] = carry + wbase + inc - v; a.tile_counts[i] = 0; a.tile_cursor[i] = 0; }
        __syncthreads();
        if (threadIdx.x == 0) s_carry = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        uint32_t total = s_carry;
        a.tile_offsets[ntiles] = total;
        Lane &L = a.cnt->lane[a.lane];
        L.bin_total = total;
        totals_for_frame(a.cnt, L, a.frame_no);
        if (total > a.bin_cap) seq_fail(a.cnt, a.seq, a.epoch, 2u);
        else L.tot_bin_entries += total;
    }
}

// ---------------------------------------------------------------------------------------------
// K2' binning in ONE launch, without per-entry global atomics (default path): a segmented LDS
// multi-split.  Workgroup g of G owns a contiguous chunk of the setup triangles and
//   1. histograms the chunk over the tiles in LDS (ds_add);
//   2. scans the histogram (exclusive, over tiles) and reserves its T_g entries of `bins` with ONE
//      global atomic (base_g): its records for tile t then live at base_g + prefix_g[t] .. -- a
//      private, contiguous, tile-sorted region;
//   3. publishes the row seg[g][0..ntiles] = those starts (+ the end sentinel);
//   4. walks the chunk again (L2-hot) and scatters the 16-byte cull records with returning LDS atomics.
// Tiles are numbered over the rank's OWN tile rows (local_tile_row), so `ntiles` is what the rank owns.
// The tile kernel reads column t of `seg` (G (start,end) pairs) and walks the segments; no column
// scan, no CSR scan and no second launch are needed.  The global-atomic CSR path (k_bin above) remains
// as the fallback when the tile count does not fit LDS.
// ---------------------------------------------------------------------------------------------

__global__ __launch_bounds__(BIN_WG) void k_bin_seg(RasterArgs a, uint32_t ntiles, uint32_t *__restrict__ seg,
                                                    uint32_t stage_cap, uint32_t fan_cap, const uint32_t *block_sums, uint32_t *block_prefix, uint32_t nblocks, int do_scan)
{
    const uint32_t first_bad = seq_first_bad(a.cnt);   // (tested once the histogram is zeroed: nothing is written before)
    extern __shared__ __attribute__((aligned(16))) uint32_t s_hist[]; // [ntiles], then the staging records
    uint4 *s_stage = reinterpret_cast<uint4 *>(s_hist + ((ntiles + 3u) & ~3u));
    __shared__ uint32_t s_w[2][BIN_WG / 64];
    __shared__ uint32_t s_base;
    const uint32_t g = blockIdx.x, G = gridDim.x - (uint32_t)do_scan;
    const bool scan_wg = do_scan && blockIdx.x == gridDim.x - 1;
    if (!scan_wg) for (uint32_t t = threadIdx.x; t < ((ntiles + 3u) & ~3u); t += BIN_WG) s_hist[t] = 0u;   // (padded to four: bin_scan_relative)
    if (seq_is_cancelled(first_bad, a.seq, a.epoch, false)) return;
    // do_scan: the launch's LAST workgroup scans the geometry kernel's block sums instead (the tile kernel's resolve
    // needs the prefix for triangle ids; here it costs no launch and sits on nobody's critical path)
    if (scan_wg) { geom_scan(block_sums, block_prefix, nblocks, a.cnt, a.lane, a.gpar, fan_cap, a.geom_seq, a.epoch); return; }
    if (g == 0 && threadIdx.x == 0) bin_bookkeeping(a.cnt, a.lane, a.bpar, a.frame_no);
    __syncthreads();
    // a workgroup's chunk: a range of geometry blocks (their dense entries) and a range of the used fan entries
    const FanMap fm = fan_map(&a.cnt->lane[a.lane].gtab[a.gpar], fan_cap);
    const uint32_t bpw = (nblocks + G - 1) / G;
    const uint32_t b_lo = min(nblocks, g * bpw), b_hi = min(nblocks, b_lo + bpw);
    const uint32_t nfan = fm.pre[FAN_REGIONS];
    const uint32_t fchunk = ((nfan + G - 1) / G + 63u) & ~63u;
    const uint32_t lo = fm.ntris + min(nfan, g * fchunk), hi = fm.ntris + min(nfan, g * fchunk + fchunk);
    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef FRR_DEBUG_COUNTERS
    const unsigned long long d_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    // (keeping the chunk's bboxes in registers from the counting walk to the placing walk saved nothing: the second read
    // comes from L2 and the 16 extra registers cost the 4096^2 frame 2 us)
    bin_walk_blocks<false>(a, s_hist, a.pbox, a.bcount, fm.ntris, b_lo, b_hi, lane, wave, PutNone{});
    bin_walk_fans<false>(a, s_hist, fm, a.pbox, lo, hi, lane, wave, PutNone{});
    __syncthreads();
#ifdef FRR_DEBUG_COUNTERS
    const unsigned long long d_t1 = __builtin_amdgcn_s_memrealtime();
#endif
    const uint32_t total = bin_scan_relative(s_hist, ntiles, s_w, lane, wave);
#ifdef FRR_DEBUG_COUNTERS
    const unsigned long long d_t2 = __builtin_amdgcn_s_memrealtime();
#endif
    // the region is reserved now, but only a workgroup whose records do not all fit the LDS staging has to know where it
    // starts before it places them: the others walk while the atomic is in flight
    unsigned long long before = 0;
    if (threadIdx.x == 0) before = atomicAdd(&a.cnt->lane[a.lane].btab[a.bpar].seg_total, (unsigned long long)total);
    uint32_t base = 0;
    if (total > stage_cap) {
        if (threadIdx.x == 0) s_base = bin_region_of(a, before, total);
        __syncthreads();
        base = s_base;
        const PutMixed put{s_stage, a.bins, base, stage_cap, a.bin_cap};
        bin_walk_blocks<true>(a, s_hist, a.pbox, a.bcount, fm.ntris, b_lo, b_hi, lane, wave, put);
        bin_walk_fans<true>(a, s_hist, fm, a.pbox, lo, hi, lane, wave, put);
        __syncthreads();
    } else {
        const PutStaged put{s_stage};
        bin_walk_blocks<true>(a, s_hist, a.pbox, a.bcount, fm.ntris, b_lo, b_hi, lane, wave, put);
        bin_walk_fans<true>(a, s_hist, fm, a.pbox, lo, hi, lane, wave, put);
        if (threadIdx.x == 0) s_base = bin_region_of(a, before, total);
        __syncthreads();
        base = s_base;
    }
#ifdef FRR_DEBUG_COUNTERS
    const unsigned long long d_t3 = __builtin_amdgcn_s_memrealtime();
#endif
    // the row of the segment table: tile t starts where tile t-1 ended (every cursor has run to the end of its tile)
    uint32_t *__restrict__ row = seg + (size_t)g * (ntiles + 1);
    for (uint32_t t = threadIdx.x; t <= ntiles; t += BIN_WG) row[t] = min(base + (t ? s_hist[t - 1] : 0u), a.bin_cap);
    const uint32_t nstaged = min(total, stage_cap);
    for (uint32_t j = threadIdx.x; j < nstaged; j += BIN_WG)
        if (base + j < a.bin_cap) a.bins[base + j] = s_stage[j];
#ifdef FRR_DEBUG_COUNTERS
    if (threadIdx.x == 0) { // phase times of the binning workgroups, 10 ns units (tools/tile_timeline.py prints the sums)
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long d_t4 = __builtin_amdgcn_s_memrealtime();
        unsigned long long *d = a.cnt->dbg[blockIdx.x % DBG_COPIES];
        atomicAdd(d + 21, d_t1 - d_t0); atomicAdd(d + 22, d_t2 - d_t1); atomicAdd(d + 23, ((d_t3 - d_t2) << 32) | (d_t4 - d_t3));
    }
#endif
}

} // namespace frr
#include "frr_raster.h"
namespace frr {

// ---- debug --------------------------------------------------------------------------------
// recip_exact against the IEEE division, rsqrt_exact against 1.0f / sqrtf, and the single-instruction casts / max of
// frr_exact.h against their spelled-out forms, over a range of bit patterns: number of differing results, first offender
__global__ __launch_bounds__(256) void k_debug_rcp(uint32_t lo, uint32_t hi, unsigned long long *bad, uint32_t *first)
{
    unsigned long long n = 0;
    for (unsigned long long b = (unsigned long long)lo + blockIdx.x * 256ull + threadIdx.x; b < (unsigned long long)hi; b += gridDim.x * 256ull) {
        const float s = u2f((uint32_t)b);
        const float q = 1.0f / s, r = recip_exact(s);
        if (f2u(q) != f2u(r) && !(q != q && r != r)) { ++n; atomicMin(first, (uint32_t)b); }
        const float q2 = 1.0f / sqrtf(s), r2 = rsqrt_exact(s);
        if (f2u(q2) != f2u(r2) && !(q2 != q2 && r2 != r2)) { ++n; atomicMin(first, (uint32_t)b); }
        const float m = f32_max(s, 0.0f), mr = f32_max_ref(s, 0.0f);
        if (f32_as_i32(s) != f32_as_i32_ref(s) || f32_as_u32(s) != f32_as_u32_ref(s) || quantize_u8(s) != quantize_u8_ref(s) ||
            f2u(m) != f2u(mr)) { ++n; atomicMin(first, (uint32_t)b); }
    }
    if (n) atomicAdd(bad, n);
}

// MFMA experiment (north_star: "MFMA used only for the batched 4x4 MVP x vertex-block contraction").
// clip[4][16 vertices] = MVP[4x4] * P[4x16] with v_mfma_f32_16x16x4_f32: A = MVP in rows 0..3 of a 16x4
// tile, B = (x,y,z,1) of 16 vertices; lane j < 16 receives vertex j's clip xyzw in its 4 accumulators.
// The MFMA is an fmaf chain over k (one rounding per step); glam's Mat4*Vec4 rounds every product and
// every sum, so the results differ in the last bits -- measured by tests/test_gpu_mfma.py, which is
// why the parity path keeps the VALU form (k_debug_mvp_exact == run_vs<FRR_VS_PHONG>).
typedef float frr_f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void k_debug_mvp_mfma(const float *__restrict__ in, uint32_t nverts, int nf, DevUniforms u, float4 *out)
{
    const int lane = threadIdx.x;
    const uint32_t v = blockIdx.x * 64u + lane;
    float px = 0.0f, py = 0.0f, pz = 0.0f;
    if (v < nverts) { px = in[(size_t)v * nf]; py = in[(size_t)v * nf + 1]; pz = in[(size_t)v * nf + 2]; }
    const int r = lane & 15, k = lane >> 4;
    const float a = r < 4 ? u.mvp[k * 4 + r] : 0.0f;                 // A[row r][k] = MVP(r, k), column-major storage
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int src = 16 * g + r;
        const float bx = __shfl(px, src), by = __shfl(py, src), bz = __shfl(pz, src);
        const float b = k == 0 ? bx : (k == 1 ? by : (k == 2 ? bz : 1.0f)); // B[k][col r]
        frr_f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
        const uint32_t vo = blockIdx.x * 64u + 16u * g + lane;
        if (lane < 16 && vo < nverts) out[vo] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
}
__global__ __launch_bounds__(64) void k_debug_mvp_exact(const float *__restrict__ in, uint32_t nverts, int nf, DevUniforms u, float4 *out)
{
    const uint32_t v = blockIdx.x * 64u + threadIdx.x;
    if (v >= nverts) return;
    float o[4];
    mat4_mul_vec4(u.mvp, in[(size_t)v * nf], in[(size_t)v * nf + 1], in[(size_t)v * nf + 2], 1.0f, o);
    out[v] = make_float4(o[0], o[1], o[2], o[3]);
}
// PMC calibration: every lane gathers one distinct 64-byte record (4 x dwordx4, the tile kernel's
// phase-1 access pattern) from a table larger than the Infinity Cache; true bytes = n * 64.
__global__ __launch_bounds__(256) void k_debug_gather(const uint4 *__restrict__ table, uint32_t n_mask, uint32_t *out)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t j = (i * 0x9E3779B1u) & n_mask; // odd multiplier: a permutation of [0, n)
    const uint4 *r = table + (size_t)j * 4;
    const uint4 a = r[0], b = r[1], c = r[2], d = r[3];
    const uint32_t x = a.x ^ a.w ^ b.y ^ c.z ^ d.w ^ d.x;
    if (x == 0x12345678u) out[0] = i; // keeps the loads alive
}
__global__ void k_debug_atan2f(const float *y, const float *x, float *out, uint64_t n)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fd_atan2f(y[i], x[i]);
}

} // namespace frr
