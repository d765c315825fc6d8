// frr_raster.h -- K3: tile raster + resolve kernels (wave64, LDS keys).
//
// One workgroup (4 waves) per 32x32 screen tile.  Per-pixel 64-bit keys
//   zkey(rhw) << 32 | (triangle index + 1)
// live in LDS and are resolved with ds_max_u64: argmax over (rhw, emission index) is exactly the
// reference's sequential rule "`rhw < depth` rejects, ties go to the later triangle"
// (renderer.rs:363-366), so waves may race and bins may be unordered.  After the last
// triangle each pixel's winner is re-evaluated with the same arithmetic, shaded (:368-381) and
// stored row-contiguously.
//
// Two coverage strategies produce the same fragment set {pixels passing renderer.rs:329-341}:
//   k_raster       one triangle per wavefront, brute-force sweep of bbox-in-tile pixels (64/step);
//   k_raster_span  cull records -> survivors -> a batch of triangles per wavefront: exact per-row spans from the integer edge
//                  functions, then fragments packed 64 per step (no lane idles on uncovered bbox
//                  pixels).  Triangles whose coordinates could overflow i32 in the span algebra
//                  (|spi| > 8192, only the clipper's far-away vertices) take the brute-force sweep,
//                  which is exact under wrapping arithmetic.
#pragma once
#include "frr_device.h"
#ifndef __HIPCC_RTC__
#include <type_traits>
#endif

namespace frr {

__device__ __forceinline__ int xcd_remap(int bid, int nwg)
{
    // blocks are dealt round-robin to the 8 XCDs; give each XCD a contiguous run of tiles so
    // neighbouring tiles (which share triangle records) share an L2.  Bijective for any nwg.
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
}

// ---- order keys <-> slots <-> emission indices of the current draw (frr_device.h: slots and order keys) -------------
// (the tables come from the kernel arguments, RasterArgs::tinfo ...; only tri_base lives in the device-side GeomTab)
// low half of a pixel key for the triangle at `slot`: order key + 1 (0 = "what was there before this draw")
__device__ __forceinline__ uint32_t order_id(const RasterArgs &a, uint32_t slot)
{
    const uint32_t nt = a.ntris_draw;
    return 1u + (slot < nt ? slot << FAN_BITS : a.fan_okey[slot - nt]);
}
// the slot a winning id of THIS draw names, and the reference's emission index of that triangle within the frame
__device__ __forceinline__ uint32_t id_slot(const RasterArgs &a, uint32_t id)
{
    const uint32_t okl = id - 1u, t = okl >> FAN_BITS, f = okl & ((1u << FAN_BITS) - 1u);
    return f ? a.ntris_draw + a.fanbase[t] + (f - 1u) : t;
}
__device__ __forceinline__ uint32_t id_emission(const RasterArgs &a, uint32_t tri_base, uint32_t id)
{
    const uint32_t okl = id - 1u, t = okl >> FAN_BITS, f = okl & ((1u << FAN_BITS) - 1u);
    return tri_base + a.block_prefix[t / GEOM_BLOCK] + (a.tinfo[t] >> FAN_BITS) + (f ? f - 1u : 0u);
}

// Depth-only draws keep  emission index (within the draw) + 1  in their pixel keys instead of the order key: it is
// just as monotone, the lookups below are then paid once per surviving triangle of a tile instead of once per pixel,
// and the resolve adds tri_base.  (Shaded draws need the winner's SLOT per pixel, which only the order key gives.)
__device__ __forceinline__ uint32_t emission_id(const RasterArgs &a, uint32_t slot)
{
    const uint32_t nt = a.ntris_draw;
    uint32_t t = slot, q = 0u;
    if (slot >= nt) { const uint32_t ok = a.fan_okey[slot - nt]; t = ok >> FAN_BITS; q = (ok & ((1u << FAN_BITS) - 1u)) - 1u; }
    return 1u + a.block_prefix[t / GEOM_BLOCK] + (a.tinfo[t] >> FAN_BITS) + q;
}
// the same from the flags word of the triangle's record (its emission offset within its geometry block travels there):
// one table lookup instead of two dependent ones
__device__ __forceinline__ uint32_t emission_id_rec(const RasterArgs &a, uint32_t slot, uint32_t flags)
{
    const uint32_t nt = a.ntris_draw;
    uint32_t t = slot;
    if (slot >= nt) t = a.fan_okey[slot - nt] >> FAN_BITS;
    return 1u + a.block_prefix[t / GEOM_BLOCK] + ((flags >> REC_EOFF_SHIFT) & REC_EOFF_MASK);
}
// the slot of the triangle with emission index e (within the draw): two binary searches; only the depth-only resolve's
// rare re-evaluation (-0.0 / NaN depths) needs it
__device__ __forceinline__ uint32_t slot_of_emission(const RasterArgs &cnt, uint32_t e)
{
    const uint32_t nt = cnt.ntris_draw, nb = (nt + GEOM_BLOCK - 1) / GEOM_BLOCK;
    uint32_t lo = 0, hi = nb;                       // last block whose prefix <= e
    while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (cnt.block_prefix[mid] <= e) lo = mid; else hi = mid; }
    const uint32_t r = e - cnt.block_prefix[lo];
    uint32_t a = lo * GEOM_BLOCK, b = min(nt, a + GEOM_BLOCK); // last input of the block whose offset <= r and that emits anything
    while (b - a > 1u) { const uint32_t mid = (a + b) >> 1; if ((cnt.tinfo[mid] >> FAN_BITS) <= r) a = mid; else b = mid; }
    // (inputs that emit nothing share their successor's offset: step back to the one that covers r)
    while ((cnt.tinfo[a] & ((1u << FAN_BITS) - 1u)) == 0u && a > lo * GEOM_BLOCK) --a;
    const uint32_t ti = cnt.tinfo[a], n = ti & ((1u << FAN_BITS) - 1u), q = r - (ti >> FAN_BITS);
    return n == 1u ? a : nt + cnt.fanbase[a] + q;
}

__device__ __forceinline__ const GeomTab *gtab_of(const RasterArgs &a) { return &a.cnt->lane[a.lane].gtab[a.gpar]; }

struct TileCtx {
    int tile, ltile, lx0, ly0, tw, th, ax0, ay0; // tile index in the window / among the rank's own tiles; window-local origin, extent, absolute pixel origin
    uint32_t beg, end;
};

__device__ __forceinline__ TileCtx tile_ctx(const RasterArgs &a)
{
    TileCtx c;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    // bid / tiles_x by multiplication (scalar unit): exact for bid, tiles_x < 2^16 (tiles_x_magic = 2^32 / tiles_x + 1; 0: divide)
    const int trow = a.tiles_x_magic ? (int)__builtin_amdgcn_readfirstlane((int)__umulhi((uint32_t)bid, a.tiles_x_magic)) : bid / a.tiles_x;
    const int tx = bid - trow * a.tiles_x;
    const int ty = a.blocked ? a.brow0 + trow : a.rank + trow * a.world;
    c.tile = ty * a.tiles_x + tx;
    c.ltile = bid;                                      // its index among the rank's own tiles (segmented binning)
    c.lx0 = tx * TILE; c.ly0 = ty * TILE;
    c.tw = min(TILE, a.win_w - c.lx0); c.th = min(TILE, a.win_h - c.ly0);
    c.ax0 = a.x0 + c.lx0; c.ay0 = a.y0 + c.ly0;
    c.beg = c.end = 0;
    if (a.nseg == 0) { // CSR binning; with segmented binning the tile kernel derives its range itself
        c.beg = a.tile_offsets[c.tile];
        c.end = min(a.tile_offsets[c.tile + 1], a.bin_cap);
    }
    return c;
}

// keys <- current depth buffer (index 0 = "what is already there").  Pixels of a partial tile that
// lie outside the window are never touched again; `oob` is what they hold (all-ones keeps them
// from dragging the hierarchical-z minima down).
__device__ __forceinline__ void tile_load_keys(const RasterArgs &a, const TileCtx &c, unsigned long long *s_key,
                                               unsigned long long oob = 0ull)
{
    if (a.fused_clear && c.tw == TILE && c.th == TILE) {
        // a full tile of a draw that carries the frame's clear: one constant, 32 bytes per lane and store
        const uint32_t k = zkey_depth(a.clear_depth);
        uint4 *p = reinterpret_cast<uint4 *>(s_key);
        for (int i = threadIdx.x; i < TILE_PX / 2; i += (int)blockDim.x) p[i] = make_uint4(0u, k, 0u, k);
        return;
    }
    for (int i = threadIdx.x; i < TILE_PX; i += (int)blockDim.x) {
        const int x = i & (TILE - 1), y = i >> 5;
        unsigned long long k = oob;
        if (x < c.tw && y < c.th)
            k = (unsigned long long)zkey_depth(a.fused_clear ? a.clear_depth : a.depth[(size_t)(c.ly0 + y) * a.dstride + (c.lx0 + x)]) << 32;
        s_key[i] = k;
    }
}

// a tile with nothing binned into it still owes the pending frr_clear (fused_clear draws)
__device__ __forceinline__ void tile_fill_clear(const RasterArgs &a, const TileCtx &c)
{
    if (!a.fused_clear) return;
    for (int i = threadIdx.x; i < TILE_PX; i += (int)blockDim.x) {
        const int x = i & (TILE - 1), y = i >> 5;
        if (x >= c.tw || y >= c.th) continue;
        const size_t pi = (size_t)(c.ly0 + y) * a.cstride + (c.lx0 + x);
        reinterpret_cast<uint32_t *>(a.color)[pi] = a.clear_rgba;
        a.depth[pi] = a.clear_depth;
        a.tri_id[pi] = ~0u;
    }
}

// resolve: the owner of each pixel is re-evaluated with the same arithmetic and written out
template <int K, int PS>
__device__ __forceinline__ void tile_resolve(const RasterArgs &a, const DevUniforms &u, const TileCtx &c,
                                             const unsigned long long *s_key, const float *u8lut = nullptr)
{
    const uint32_t tri_base = gtab_of(a)->tri_base;
    for (int i = threadIdx.x; i < TILE_PX; i += (int)blockDim.x) {
        const int x = i & (TILE - 1), y = i >> 5;
        if (x >= c.tw || y >= c.th) continue;
        const uint32_t id = (uint32_t)s_key[i];
        if (a.fused_clear) { // the pending frr_clear: pixels nobody won get the clear values; colour first, a shader may overwrite it
            const size_t pi = (size_t)(c.ly0 + y) * a.cstride + (c.lx0 + x); // full-window draw: one index for all three targets
            if (id == 0u || PS == FRR_PS_DEPTH) reinterpret_cast<uint32_t *>(a.color)[pi] = a.clear_rgba;
            if (id == 0u) { a.depth[pi] = a.clear_depth; a.tri_id[pi] = ~0u; }
        }
        if (id == 0u) continue; // existing depth won (or nothing covered this pixel)
        const uint32_t t = id_slot(a, id);
        if constexpr (PS == FRR_PS_DEPTH) {
            // depth only: the z key is an invertible image of rhw except that it merges -0.0 with +0.0
            // (and NaNs are not ordered): those two cases are re-evaluated below, everything else is decoded
            const float dz = zkey_decode((uint32_t)(s_key[i] >> 32));
            if (dz != 0.0f && dz == dz) {
                const size_t di = (size_t)(c.ly0 + y) * a.dstride + (c.lx0 + x);
                a.depth[di] = dz;                                                   // :366
                a.tri_id[di] = id_emission(a, tri_base, id);
                continue;
            }
        }
        const uint4 *rp = reinterpret_cast<const uint4 *>(a.recs + t);
        const uint4 q1 = rp[1], q2 = rp[2], q3 = rp[3];
        const int cx = c.ax0 + x, cy = c.ay0 + y;
        const float r0 = u2f(q3.x), r1 = u2f(q3.y), r2 = u2f(q3.z);
        Frag f = frag_eval(u2f(q1.z), u2f(q1.w), u2f(q2.x), u2f(q2.y), u2f(q2.z), u2f(q2.w), r0, r1, r2, cx, cy);
        const size_t di = (size_t)(c.ly0 + y) * a.dstride + (c.lx0 + x);
        a.depth[di] = f.rhw;                                                    // :366
        // (the record is here anyway, and its flags carry the triangle's emission offset within its geometry block: one table
        // lookup instead of id_emission's two dependent ones)
        a.tri_id[di] = tri_base + a.block_prefix[((id - 1u) >> FAN_BITS) / GEOM_BLOCK] + ((q3.w >> REC_EOFF_SHIFT) & REC_EOFF_MASK);
        if constexpr (PS != FRR_PS_DEPTH) {
            const float w = recip_exact(f.rhw != 0.0f ? f.rhw : 1.0f);          // :368 (== 1.0f / x, bit for bit)
            const float c0 = r0 * f.a * w, c1 = r1 * f.b * w, c2 = r2 * f.c * w; // :370-372
            float in[K > 0 ? K : 1];
            if constexpr (K > 0) {
                const float *v = a.vary + (size_t)t * (3 * K);
#pragma unroll
                for (int k = 0; k < K; ++k) in[k] = v[k] * c0 + v[K + k] * c1 + v[2 * K + k] * c2; // :374-378
            }
            float col[4];
            run_ps<PS>(u, in, col, u8lut);                                      // :380
            const uint32_t q = quantize_u8(col[0]) | (quantize_u8(col[1]) << 8) | (quantize_u8(col[2]) << 16) |
                               (quantize_u8(col[3]) << 24);                     // :7-14
            reinterpret_cast<uint32_t *>(a.color)[(size_t)(c.ly0 + y) * a.cstride + (c.lx0 + x)] = q; // :381,:496-503
        }
    }
}

// Depth-only resolve, four pixels of a row per lane: 32 bytes of keys in, 16-byte stores of depth / ids (/ the clear
// colour) out -- every row of the tile leaves as one full 128-byte line per target.  Rows whose addresses are not
// 16-byte aligned (odd strides, caller-bound targets), partial tiles and -- when the draw does not carry the clear --
// groups with a pixel nobody won take scalar stores; keys that merge -0.0 / NaN (see tile_resolve) are re-evaluated.
__device__ __forceinline__ void tile_resolve_depth4(const RasterArgs &a, const TileCtx &c, const unsigned long long *s_key)
{
    const bool fused = a.fused_clear != 0;
    const uint32_t tri_base = gtab_of(a)->tri_base;      // (the keys of a depth-only draw hold emission indices: emission_id)
    for (int g = threadIdx.x; g < TILE_PX / 4; g += (int)blockDim.x) {
        const int y = g >> 3, x = (g & 7) * 4;
        if (y >= c.th || x >= c.tw) continue;
        const uint4 k01 = reinterpret_cast<const uint4 *>(s_key)[2 * g], k23 = reinterpret_cast<const uint4 *>(s_key)[2 * g + 1];
        const uint32_t id[4] = {k01.x, k01.z, k23.x, k23.z}, zk[4] = {k01.y, k01.w, k23.y, k23.w};
        uint32_t dv[4], iv[4];
        bool won[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            won[i] = id[i] != 0u && x + i < c.tw;      // (pixels of a partial tile beyond the window hold all-ones keys)
            float dz = zkey_decode(zk[i]);
            if (won[i] && !(dz != 0.0f && dz == dz)) { // -0.0 merged with +0.0, or NaN: the reference arithmetic decides
                const uint4 *rp = reinterpret_cast<const uint4 *>(a.recs + slot_of_emission(a, id[i] - 1u));
                const uint4 q1 = rp[1], q2 = rp[2], q3 = rp[3];
                dz = frag_eval(u2f(q1.z), u2f(q1.w), u2f(q2.x), u2f(q2.y), u2f(q2.z), u2f(q2.w), u2f(q3.x), u2f(q3.y), u2f(q3.z),
                               c.ax0 + x + i, c.ay0 + y).rhw;
            }
            dv[i] = won[i] ? f2u(dz) : f2u(a.clear_depth);                                         // :366
            iv[i] = won[i] ? tri_base + (id[i] - 1u) : ~0u;
        }
        const size_t di = (size_t)(c.ly0 + y) * a.dstride + (c.lx0 + x);
        const size_t ci = (size_t)(c.ly0 + y) * a.cstride + (c.lx0 + x);
        const bool all_in = x + 4 <= c.tw;
        const bool all_write = fused || (won[0] && won[1] && won[2] && won[3]);
        const bool al = (((uintptr_t)(a.depth + di) | (uintptr_t)(a.tri_id + di)) & 15u) == 0u;
        if (all_in && all_write && al) {
            *reinterpret_cast<uint4 *>(a.depth + di) = make_uint4(dv[0], dv[1], dv[2], dv[3]);
            *reinterpret_cast<uint4 *>(a.tri_id + di) = make_uint4(iv[0], iv[1], iv[2], iv[3]);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (x + i < c.tw && (fused || won[i])) { reinterpret_cast<uint32_t *>(a.depth)[di + i] = dv[i]; a.tri_id[di + i] = iv[i]; }
        }
        if (fused) { // depth-only draws leave the colour target at the clear colour
            uint32_t *cp = reinterpret_cast<uint32_t *>(a.color) + ci;
            if (all_in && (((uintptr_t)cp) & 15u) == 0u) *reinterpret_cast<uint4 *>(cp) = make_uint4(a.clear_rgba, a.clear_rgba, a.clear_rgba, a.clear_rgba);
            else
#pragma unroll
                for (int i = 0; i < 4; ++i) if (x + i < c.tw) cp[i] = a.clear_rgba;
        }
    }
}

// Brute-force sweep of ONE triangle (wave-uniform index t) by the whole wave: every pixel of
// bbox-in-tile is tested with the wrapping-i32 edge functions of renderer.rs:329-341.
// NANPASS = false: the main pass; a NaN fragment takes the key ZKEY_NAN_FRAG and raises *nanflag.
// NANPASS = true: the second pass of a tile that saw NaN fragments (tile_nan_begin): only pixels with nanL != 0 take
// part, and there only the non-NaN fragments submitted after the pixel's last NaN fragment (id > nanL).
template <bool NANPASS = false>
__device__ __forceinline__ void sweep_triangle(const RasterArgs &a, const TileCtx &c, uint32_t t, uint32_t key_id, int lane,
                                               unsigned long long *s_key, uint32_t &n_cov, uint32_t &n_nan, uint32_t *nanflag,
                                               const uint32_t *nanL = nullptr)
{
    const RasterRec *__restrict__ r = a.recs + t;
    const float s0x = r->s[0], s0y = r->s[1], s1x = r->s[2], s1y = r->s[3], s2x = r->s[4], s2y = r->s[5];
    // spi = (spf + 0.5) as i32 (renderer.rs:233-234; the record keeps spf only)
    const int p0x = f32_as_i32(s0x + 0.5f), p0y = f32_as_i32(s0y + 0.5f), p1x = f32_as_i32(s1x + 0.5f), p1y = f32_as_i32(s1y + 0.5f);
    const int p2x = f32_as_i32(s2x + 0.5f), p2y = f32_as_i32(s2y + 0.5f);
    // clamped bbox (renderer.rs:285-298; clamp is monotone so it commutes with min/max)
    int bx0 = clampi(min(p0x, min(p1x, p2x)), a.x0, a.x1), bx1 = clampi(max(p0x, max(p1x, p2x)), a.x0, a.x1);
    int by0 = clampi(min(p0y, min(p1y, p2y)), a.y0, a.y1), by1 = clampi(max(p0y, max(p1y, p2y)), a.y0, a.y1);
    bx0 = max(bx0, c.ax0); bx1 = min(bx1, c.ax0 + c.tw);        // ... intersected with this tile
    by0 = max(by0, c.ay0); by1 = min(by1, c.ay0 + c.th);
    const int bw = bx1 - bx0, bh = by1 - by0;
    if (bw <= 0 || bh <= 0) return;
    const int npx = bw * bh;
    // edge functions E = A*(cx - px) + B*(cy - py) in wrapping i32 (renderer.rs:329-331)
    const uint32_t A01 = 0u - (uint32_t)(p1y - p0y), B01 = (uint32_t)(p1x - p0x);
    const uint32_t A12 = 0u - (uint32_t)(p2y - p1y), B12 = (uint32_t)(p2x - p1x);
    const uint32_t A20 = 0u - (uint32_t)(p0y - p2y), B20 = (uint32_t)(p0x - p2x);
    const uint32_t E01o = A01 * (uint32_t)(bx0 - p0x) + B01 * (uint32_t)(by0 - p0y);
    const uint32_t E12o = A12 * (uint32_t)(bx0 - p1x) + B12 * (uint32_t)(by0 - p1y);
    const uint32_t E20o = A20 * (uint32_t)(bx0 - p2x) + B20 * (uint32_t)(by0 - p2y);
    const uint32_t fl = r->flags;
    // reject E < bias  <=>  accept E > bias-1   (bias 0 for top-left edges, else 1; :333-341)
    const int thr01 = (fl & 2u) ? 0 : -1, thr12 = (fl & 4u) ? 0 : -1, thr20 = (fl & 8u) ? 0 : -1;
    const float r0 = r->rhw[0], r1 = r->rhw[1], r2 = r->rhw[2];
    // p -> (dx, dy): dy = floor((p + 0.5) / bw) via a 1-ulp reciprocal, exact for p < 1024, bw <= 32
    const float inv_bw = __builtin_amdgcn_rcpf((float)bw);
    const unsigned long long idlow = (unsigned long long)key_id;
    for (int p = lane; p < npx; p += 64) {
        const int dy = (int)(((float)p + 0.5f) * inv_bw);
        const int dx = p - __mul24(dy, bw);
        const int E01 = (int)(E01o + A01 * (uint32_t)dx + B01 * (uint32_t)dy);
        const int E12 = (int)(E12o + A12 * (uint32_t)dx + B12 * (uint32_t)dy);
        const int E20 = (int)(E20o + A20 * (uint32_t)dx + B20 * (uint32_t)dy);
        bool covered = (E01 > thr01) & (E12 > thr12) & (E20 > thr20);
        const int cx = bx0 + dx, cy = by0 + dy;
        const int pix = (cy - c.ay0) * TILE + (cx - c.ax0);
        if (NANPASS) {
            const uint32_t last_nan = nanL[pix];
            covered = covered && last_nan != 0u && key_id > last_nan;
        } else {
            n_cov += (uint32_t)__popcll(__ballot(covered));
        }
        if (covered) {
            Frag f = frag_eval(s0x, s0y, s1x, s1y, s2x, s2y, r0, r1, r2, cx, cy);
            if (f.valid) {
                const bool isnan = f.rhw != f.rhw;
                if (NANPASS) {
                    if (!isnan) atomicMax(&s_key[pix], ((unsigned long long)zkey(f.rhw) << 32) | idlow);
                } else {
                    if (isnan) { ++n_nan; *nanflag = 1u; }
                    atomicMax(&s_key[pix], ((unsigned long long)zkey_frag(f.rhw) << 32) | idlow);
                }
            }
        }
    }
}

// First half of the NaN pass of a tile whose main pass saw NaN fragments: a pixel whose key carries ZKEY_NAN_FRAG holds
// the id of its LAST NaN fragment.  That id moves to nanL[pixel] (0 elsewhere) and the key becomes {ZKEY_NAN_BELOW, id}:
// the NaN fragment owns the pixel unless a later-submitted fragment covers it, and the first such fragment passes
// whatever its depth (renderer.rs:363-366).  The caller then sweeps all records of the tile again with NANPASS = true.
// Pixels that saw no NaN fragment already hold their final key.
__device__ __forceinline__ void tile_nan_begin(const TileCtx &c, unsigned long long *s_key, uint32_t *nanL)
{
    for (int i = threadIdx.x; i < TILE_PX; i += (int)blockDim.x) {
        const int x = i & (TILE - 1), y = i >> 5;
        const unsigned long long k = s_key[i];
        const bool nanpix = x < c.tw && y < c.th && (uint32_t)(k >> 32) == ZKEY_NAN_FRAG;
        nanL[i] = nanpix ? (uint32_t)k : 0u;
        if (nanpix) s_key[i] = ((unsigned long long)ZKEY_NAN_BELOW << 32) | (unsigned long long)(uint32_t)k;
    }
}

// ---------------------------------------------------------------------------------------------
// k_raster: one triangle per wavefront (brute force).  Kept as the reference form of the tile
// kernel (FRR_RASTER=sweep) and as the exact path for wrapping coordinates.
// ---------------------------------------------------------------------------------------------
template <int K, int PS>
__global__ __launch_bounds__(256) void k_raster(RasterArgs a, DevUniforms u)
{
    __shared__ unsigned long long s_key[TILE_PX];
    __shared__ uint32_t s_nanL[TILE_PX];
    __shared__ uint32_t s_nanflag;
    if (seq_cancelled(a.cnt, a.seq, a.epoch, true)) return; // this or an earlier command failed: the targets stay as they are, the host replays
    const TileCtx c = tile_ctx(a);
    if (c.beg >= c.end) return; // nothing binned here: colour, depth and ids stay as they are
    tile_load_keys(a, c, s_key);
    if (threadIdx.x == 0) s_nanflag = 0u;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t n_cov = 0, n_nan = 0;
    for (uint32_t e = c.beg + wave; e < c.end; e += 4) {
        const uint32_t t = __builtin_amdgcn_readfirstlane(a.bins[e].x);
        sweep_triangle(a, c, t, order_id(a, t), lane, s_key, n_cov, n_nan, &s_nanflag);
    }
    if (lane == 0 && n_cov) atomicAdd(&a.cnt->lane[a.lane].gtab[a.gpar].frag_covered, (unsigned long long)n_cov);
    if (n_nan) atomicAdd(&a.cnt->lane[a.lane].gtab[a.gpar].frag_nan, (unsigned long long)n_nan);
    __syncthreads();
    if (s_nanflag != 0u) { // (uniform: read after the barrier)
        tile_nan_begin(c, s_key, s_nanL);
        __syncthreads();
        for (uint32_t e = c.beg + wave; e < c.end; e += 4) {
            const uint32_t t = __builtin_amdgcn_readfirstlane(a.bins[e].x);
            sweep_triangle<true>(a, c, t, order_id(a, t), lane, s_key, n_cov, n_nan, nullptr, s_nanL);
        }
        __syncthreads();
    }
    tile_resolve<K, PS>(a, u, c, s_key);
}

// ---------------------------------------------------------------------------------------------
// k_raster_span
// ---------------------------------------------------------------------------------------------
constexpr int SPAN_SAFE = 8191; // |spi| and window coordinates up to this keep every edge value < 2^30 (and twice an edge delta in 16 bits)
constexpr int SPAN_BATCH = 32;  // triangles per wavefront batch (staging sized for 8 workgroups per CU)
#ifndef FRR_SPAN_CULL
#define FRR_SPAN_CULL 32
#endif
constexpr int SPAN_CULL = FRR_SPAN_CULL; // bin entries culled per step (<= 64): smaller = fresher z minima, more steps
#ifndef FRR_SPAN_AQ_MIN
#define FRR_SPAN_AQ_MIN 1
#endif
#ifndef FRR_LIGHT_NW
#define FRR_LIGHT_NW 3
#endif
#ifndef FRR_LIGHT_B
#define FRR_LIGHT_B 32
#endif
constexpr int LIGHT_NW = FRR_LIGHT_NW; // waves per tile of the shape for lightly loaded tiles, and the triangles a wave stages there
constexpr int LIGHT_B = FRR_LIGHT_B;
constexpr int SPAN_AQ_MIN = FRR_SPAN_AQ_MIN; // survivors a wave collects before it rasterizes them (1: after every cull step)

// A wave's LDS instructions execute in issue order, so lanes of ONE wave may exchange data through
// LDS without waiting; this fence only keeps the compiler from reordering the accesses.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// A 64-bit "heads" mask marks where segments start inside a window of 64 consecutive items: the segment of lane L is
// the (number of heads at positions <= L)-th one.  seg_rank: its index, given `base` = (heads in earlier windows) - 1.
// (Where a segment STARTS travels with the segment's own data -- SpanTri::misc, the span descriptors -- so nobody has
// to look for its head, also not across windows.)
__device__ __forceinline__ int seg_rank(uint32_t h_lo, uint32_t h_hi, uint32_t le_lo, uint32_t le_hi, int base)
{
    return __popc(h_hi & le_hi) + (__popc(h_lo & le_lo) + base);
}
// Per-triangle data of the span phase (48 bytes).  Row `row` of the triangle's bbox-in-tile is covered by the dx in
// [lo, hi) that satisfy A*dx >= N(row) on all three edges, N(row) = (thr + 1 - E at the bbox origin) - B*row, thr = -1
// for top-left edges else 0 (renderer.rs:329-341).  Each edge is stored in the form the row lanes evaluate without
// branches: M(row) = m + k*row, q = floor(M / D), and
//   A > 0:  dx >= ceil(N/A) = floor((N + A - 1)/A)      m = n + A - 1,  k = -B,  D = A,   q bounds lo
//   A < 0:  dx <= floor(-N/|A|)                         m = -n,         k = B,   D = |A|, q + 1 bounds hi
//   A = 0:  all dx if N <= 0, none otherwise            m = 1 - 2n,     k = 2B,  D = 0,   "reciprocal" 64: M is odd, so
//           the quotient is >= 41 (hi unchanged: a row has at most 32 pixels) or <= -3 (hi <= 0)
// k, D and the tile-independent part of m come from the geometry kernel (frr_device.h: edge_words, RasterRec::e).
struct alignas(16) SpanTri {
    int32_t m01, m12, m20;     // M at row 0
    uint32_t zub;              // zkey of an upper bound of rhw over the triangle
    uint32_t kd01, kd12, kd20; // k (low 16, signed) | D (high 16)
    uint32_t misc;             // bx0l:5 | by0l:5 <<5 | bw:6 <<10 | (A > 0) per edge <<16 | position of the triangle's first row among the step's rows <<19
    float r01, r12, r20, pad;  // v_rcp_f32 of D per edge
};
// narrows [lo, hi) by one edge; pmask = -1 for A > 0 else 0.  Only quotients in [0, 32] matter: the float quotient is
// clamped to [-2, 40] and ONE remainder step makes every in-range quotient exact (out-of-range ones stay out of range).
__device__ __forceinline__ void span_edge_bound(int m, uint32_t kd, float rD, int pmask, int row, int &lo, int &hi)
{
    const int k = (int)(kd << 16) >> 16, D = (int)(kd >> 16);
    const int M = m + __mul24(k, row);
    int q = (int)__builtin_amdgcn_fmed3f((float)M * rD, -2.0f, 40.0f);
    const int r = M - __mul24(q, D);
    q += (r >> 31) + (int)(r >= D);
    lo = max(lo, q & pmask);                   // (lo >= 0 throughout, so 0 is neutral)
    hi = min(hi, max(q + 1, pmask & 64));      // (negative bounds become 0: the span is empty either way)
}

// Hierarchical z: s_hiz[row*4 + seg] = min over the 8 pixels of (row, seg) of the depth part of the
// current keys, s_hiz8[by*4 + seg] = min over the 8x8 block.  Depth keys only grow, so a value read
// late, or written by another wave a moment ago, is still a valid lower bound: no synchronisation.
// Layout of the array hz[HZ_SIZE]: [HZ_SEG + row*4 + seg] 8-pixel row segments, [HZ_BLK + by*4 + bx]
// 8x8 blocks, [HZ_QUAD + qy*2 + qx] 16x16 quads, [HZ_C4 + row*8 + cell] 4-pixel cells of a row.
constexpr int DIRECT_MAX = 256;  // records of a tile that are culled straight from registers (k_raster_span) ...
#ifndef FRR_DIRECT_STEPS
#define FRR_DIRECT_STEPS 2
#endif
constexpr int DIRECT_STEPS = FRR_DIRECT_STEPS;  // ... in this many steps per wave (at least DIRECT_MIN lanes each)
constexpr int DIRECT_MIN = 4;
constexpr int HZ_SEG = 0, HZ_BLK = 128, HZ_QUAD = 144, HZ_C4 = 148, HZ_SIZE = 404;
// near-first order: buckets of the z bound's key -- its 4 low exponent bits and the top (BKT_LOG2 - 4) mantissa bits
#ifndef FRR_BKT_LOG2
#define FRR_BKT_LOG2 6
#endif
constexpr int BKT_LOG2 = FRR_BKT_LOG2, BKT_N = 1 << BKT_LOG2, BKT_SHIFT = 27 - BKT_LOG2, BKT_PER = BKT_N / 64;
__device__ __forceinline__ uint32_t z_bucket(uint32_t zub_key) { return (uint32_t)(BKT_N - 1) - ((zub_key >> BKT_SHIFT) & (uint32_t)(BKT_N - 1)); }
// exclusive scan of the bucket counters in place (one wave; BKT_PER consecutive buckets per lane), + base
__device__ __forceinline__ void bucket_scan(uint32_t *s_bkt, int lane, uint32_t base)
{
    uint32_t v[BKT_PER], sum = 0u;
#pragma unroll
    for (int j = 0; j < BKT_PER; ++j) { v[j] = s_bkt[lane * BKT_PER + j]; sum += v[j]; }
    uint32_t run = base + wave_incl_scan_dpp(sum) - sum;
#pragma unroll
    for (int j = 0; j < BKT_PER; ++j) { s_bkt[lane * BKT_PER + j] = run; run += v[j]; }
}
// lane ^ 1 and lane ^ 2 inside each quad of lanes, on the DPP network (no LDS round trip)
__device__ __forceinline__ uint32_t dpp_xor1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false); } // quad_perm [1,0,3,2]
__device__ __forceinline__ uint32_t dpp_xor2(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false); } // quad_perm [2,3,0,1]

__device__ __forceinline__ void hiz_rebuild(const unsigned long long *s_key, uint32_t *hz, int lane)
{
    // consecutive lanes read consecutive 16-byte chunks (2 keys): conflict-free; 4 lanes = one segment.
    // All eight reads are issued before anything is written (the compiler must not assume hz and the keys are disjoint)
    const uint4 *p = reinterpret_cast<const uint4 *>(s_key);
    uint32_t m[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { const uint4 v = p[64 * k + lane]; m[k] = min(v.y, v.w); }
    uint32_t c4[8], sg[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        c4[k] = min(m[k], dpp_xor1(m[k]));      // 4-pixel cell: lanes 2i, 2i+1
        sg[k] = min(c4[k], dpp_xor2(c4[k]));    // 8-pixel segment: lanes 4i .. 4i+3
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if ((lane & 1) == 0) hz[HZ_C4 + 32 * k + (lane >> 1)] = c4[k] >> 1;
        if ((lane & 3) == 0) hz[HZ_SEG + 16 * k + (lane >> 2)] = sg[k];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (lane < 16) {                       // 8x8 blocks: 8 rows of one segment column
        const int by = lane >> 2, bx = lane & 3;
        uint32_t mm = 0xFFFFFFFFu;
#pragma unroll
        for (int r = 0; r < 8; ++r) mm = min(mm, hz[HZ_SEG + (by * 8 + r) * 4 + bx]);
        hz[HZ_BLK + lane] = mm;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (lane < 4) {                        // 16x16 quads: 2x2 blocks
        const int qy = lane >> 1, qx = lane & 1;
        const uint32_t *b = hz + HZ_BLK + (qy * 2) * 4 + qx * 2;
        hz[HZ_QUAD + lane] = min(min(b[0], b[1]), min(b[4], b[5]));
    }
}

// NW = waves per tile workgroup, OCC = waves per SIMD the registers are budgeted for.  The host picks the pair that
// wastes the fewest workgroup slots of the chip for the number of tiles at hand (launch_raster): e.g. <4, 8> keeps
// 8 workgroups per CU resident (2048 tiles in ONE round), <6, 6> four workgroups of six waves (1024 tiles a round).
//
// LDS is carved by hand from one buffer so that what only the pre-pass needs (segment tables, the exchange buffer
// of the direct path) can share its bytes with what only the main loop needs (per-wave staging and queues):
//   [keys 8 KiB][union { main-loop staging | pre-pass tables }][hi-z][buckets][scalars][u8 table (textured only)]
// With B = 16 staged triangles per wave and NW = 4 that is 18.0 KiB (19.0 textured): eight workgroups per CU;
// with B = 32 it is 23.9 KiB: six.
template <int NW, int B> struct SpanLds {
    static constexpr int KEY = 0;
    static constexpr int U0 = KEY + TILE_PX * 8;
    // main loop, per wave
    static constexpr int TRI = U0;                      // SpanTri[NW][B]
    static constexpr int FA = TRI + NW * B * 48;        // float4[NW][B]  s0x s0y s1x s1y
    static constexpr int FB = FA + NW * B * 16;         // float4[NW][B]  s2x s2y rhw0 rhw1
    static constexpr int FC = FB + NW * B * 16;         // float2[NW][B]  rhw2, bit pattern of (triangle index + 1)
    static constexpr int HROW = FC + NW * B * 8;        // u64[NW][B/2]   heads of (triangle -> rows): B*32 bits
    static constexpr int HFRAG = HROW + NW * (B / 2) * 8; // u64[NW][32]  heads of (span -> fragments): 64*32 bits
    static constexpr int Q = HFRAG + NW * 32 * 8;       // u32[NW][64]    compacted span descriptors
    static constexpr int M_END = Q + NW * 64 * 4;
    // pre-pass (aliases the main-loop region; a barrier separates the two uses)
    static constexpr int SEGPRE = U0;                   // u32[BIN_MAX_G + 1] entries of this tile before segment g
    static constexpr int SEGSRC = SEGPRE + ((BIN_MAX_G + 1 + 3) & ~3) * 4; // u32[BIN_MAX_G] where segment g starts in a.bins
    static constexpr int EXCH = SEGSRC + BIN_MAX_G * 4; // uint4[DIRECT_MAX] exchange buffer of the direct path
    static constexpr int P_END = EXCH + DIRECT_MAX * 16;
    static constexpr int U1 = M_END > P_END ? M_END : P_END;
    static constexpr int HZ = U1;                       // u32[HZ_SIZE] hierarchical z (see hiz_rebuild)
    static constexpr int BKT = HZ + ((HZ_SIZE + 3) & ~3) * 4; // u32[BKT_N]
    static constexpr int SCAL = BKT + BKT_N * 4;        // u32[8]: next, dirty, ebase, -, w4[4]
    static constexpr int U8 = SCAL + 8 * 4;             // float[256] (textured shaders)
    static constexpr int bytes(bool textured) { return U8 + (textured ? 256 * 4 : 0); }
};

template <int K, int PS, bool COUNT, int NW, int OCC>
__global__ __launch_bounds__(NW * 64, OCC) void k_raster_span(RasterArgs a, DevUniforms u, int win_safe)
{
    constexpr int B = NW <= 3 ? LIGHT_B : (OCC >= 8 || NW >= 16) ? 16 : SPAN_BATCH; // staged triangles per wave (LDS budget: 8 workgroups per CU; 64 KiB of static LDS at NW = 16)
    constexpr bool TEXTURED = PS == FRR_PS_PHONG || PS == FRR_PS_BLINN || PS >= FRR_SHADER_USER_BASE;   // (the u8 -> float table of sample_2d)
    static_assert(B <= 32, "staging slots are five bits of a span descriptor, first rows ten bits of SpanTri::misc");
    using L = SpanLds<NW, B>;
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[L::bytes(TEXTURED)];
    unsigned long long *const s_key = reinterpret_cast<unsigned long long *>(s_raw + L::KEY);
    SpanTri (*const s_tri)[B] = reinterpret_cast<SpanTri (*)[B]>(s_raw + L::TRI);
    float4 (*const s_fa)[B] = reinterpret_cast<float4 (*)[B]>(s_raw + L::FA);
    float4 (*const s_fb)[B] = reinterpret_cast<float4 (*)[B]>(s_raw + L::FB);
    float2 (*const s_fc)[B] = reinterpret_cast<float2 (*)[B]>(s_raw + L::FC);
    unsigned long long (*const s_hrow)[B / 2] = reinterpret_cast<unsigned long long (*)[B / 2]>(s_raw + L::HROW);
    unsigned long long (*const s_hfrag)[32] = reinterpret_cast<unsigned long long (*)[32]>(s_raw + L::HFRAG);
    uint32_t (*const s_q)[64] = reinterpret_cast<uint32_t (*)[64]>(s_raw + L::Q);
    uint32_t *const s_segpre = reinterpret_cast<uint32_t *>(s_raw + L::SEGPRE);
    uint32_t *const s_segsrc = reinterpret_cast<uint32_t *>(s_raw + L::SEGSRC);
    uint4 *const s_exch = reinterpret_cast<uint4 *>(s_raw + L::EXCH);
    uint32_t *const s_hz = reinterpret_cast<uint32_t *>(s_raw + L::HZ);
    uint32_t *const s_bkt = reinterpret_cast<uint32_t *>(s_raw + L::BKT);
    uint32_t &s_next = *reinterpret_cast<uint32_t *>(s_raw + L::SCAL);
    uint32_t &s_dirty = *reinterpret_cast<uint32_t *>(s_raw + L::SCAL + 4); // keys changed since the hierarchical z was last rebuilt
    uint32_t &s_ebase = *reinterpret_cast<uint32_t *>(s_raw + L::SCAL + 8);
    uint32_t &s_nanflag = *reinterpret_cast<uint32_t *>(s_raw + L::SCAL + 12); // some fragment of this tile had a NaN rhw: second pass (tile_nan_begin)
    uint32_t *const s_nanL = reinterpret_cast<uint32_t *>(s_raw + L::U0);      // that pass's per-pixel ids; over the staging, after the main loop
    static_assert(L::U1 - L::U0 >= TILE_PX * 4, "the NaN pass keeps one u32 per pixel in the staging region");
    uint32_t *const s_w4 = reinterpret_cast<uint32_t *>(s_raw + L::SCAL + 16);
    float *const s_u8 = reinterpret_cast<float *>(s_raw + L::U8);  // (float)i / 255.0f for the texture taps of the resolve
    if (TEXTURED) for (int i = threadIdx.x; i < 256; i += NW * 64) s_u8[i] = (float)i / 255.0f; // ordered by the barriers below
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef FRR_DEBUG_COUNTERS
    // wave-cycles per phase (s_memtime): 0 pre-pass, 1 phase 1a, 2 hi-z rebuild, 3 phase 1b, 4 phase 2, 5 phase 3,
    // 6 wait at the final barrier, 7 resolve
    unsigned long long d_tlast = __builtin_amdgcn_s_memtime();
    const unsigned long long d_rt0 = __builtin_amdgcn_s_memrealtime(); // 100 MHz, the same counter on every XCD
    unsigned long long d_rt1 = 0, d_rt2 = 0;
    uint32_t d_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define FRR_T(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); d_t[i] += (uint32_t)(now_ - d_tlast); d_tlast = now_; } while (0)
#else
#define FRR_T(i) do { } while (0)
#endif
    // the pre-pass is a chain of dependent loads and barriers: its waves issue ahead of the waves that sit in the row /
    // fragment loops of other tiles on the same SIMD (the sooner its loads are out, the more of their latency those cover)
    __builtin_amdgcn_s_setprio(3);
    const uint32_t first_bad = seq_first_bad(a.cnt);   // (tested below, where the first loads are waited for anyway)
    TileCtx c = tile_ctx(a);
    const bool segmented = a.nseg != 0;
    if (!segmented && seq_is_cancelled(first_bad, a.seq, a.epoch, true)) return;
    if (segmented) {
        // column c.ltile of the segment table: (start, end) of this tile's records in each chunk's region
        uint32_t s0 = 0, cn = 0;
        if (threadIdx.x < a.nseg) {
            const uint32_t *r = a.seg + (size_t)threadIdx.x * ((size_t)gridDim.x + 1) + c.ltile;   // (grid = the rank's tiles)
            s0 = r[0];
            cn = r[1] - s0;
        }
        // this or an earlier command failed: the targets stay as they are, the host replays (frr_device.h: Counters::first_bad)
        if (seq_is_cancelled(first_bad, a.seq, a.epoch, true)) return;
        const uint32_t inc = wave_incl_scan_dpp(cn);
        if (lane == 63 && w < 4) s_w4[w] = inc;
        // three waves: segments 192..255 are a second column read of wave 0 (the host keeps nseg <= 256 then, else <= NW * 64)
        uint32_t s0x = 0, cnx = 0, incx = 0;
        if (NW == 3 && w == 0) {
            if (192u + (uint32_t)lane < a.nseg) {
                const uint32_t *r = a.seg + (size_t)(192 + lane) * ((size_t)gridDim.x + 1) + c.ltile;
                s0x = r[0];
                cnx = r[1] - s0x;
            }
            incx = wave_incl_scan_dpp(cnx);
            if (lane == 63) s_w4[3] = incx;
        }
        if (NW < 3 && threadIdx.x == 0) { s_w4[2] = 0u; s_w4[3] = 0u; }
        __syncthreads();
        const uint32_t w0 = s_w4[0], w1 = s_w4[1], w2 = s_w4[2], w3 = s_w4[3];
        const uint32_t total = __builtin_amdgcn_readfirstlane((w0 + w1) + (w2 + w3)); // wave-uniform, and the compiler should know
        if (total == 0u) { tile_fill_clear(a, c); return; }
        if (threadIdx.x < BIN_MAX_G) {
            s_segpre[threadIdx.x] = (w > 0 ? w0 : 0u) + (w > 1 ? w1 : 0u) + (w > 2 ? w2 : 0u) + inc - cn;
            s_segsrc[threadIdx.x] = s0;
        }
        if (NW == 3 && w == 0) { s_segpre[192 + lane] = (w0 + w1) + w2 + incx - cnx; s_segsrc[192 + lane] = s0x; }
        if (NW < 3) for (int i = threadIdx.x + NW * 64; i < BIN_MAX_G; i += NW * 64) { s_segpre[i] = total; s_segsrc[i] = 0u; }
        if (threadIdx.x == 0) {
            s_segpre[BIN_MAX_G] = total;
            // where this tile's near-first copy goes: its own fixed slot of bins2, or -- a tile hotter than
            // the slot -- space from the shared overflow arena behind the slots.  (One atomic per tile on a
            // single address costs ~17 ns each, serialised across the whole launch; hence the slots.)
            const uint32_t ntiles = gridDim.x;
            s_ebase = total <= a.ent_slot ? (uint32_t)c.ltile * a.ent_slot
                                          : ntiles * a.ent_slot + atomicAdd(&a.cnt->lane[a.lane].btab[a.bpar].ent_cursor, total);
        }
        __syncthreads();
        c.beg = __builtin_amdgcn_readfirstlane(s_ebase); // this tile's range of the near-first copy (bins2)
        c.end = c.beg + total;
    } else if (c.beg >= c.end) {
        tile_fill_clear(a, c);
        return;
    }
    // k-th record of this tile, k in [0, c.end - c.beg)
    auto source = [&](uint32_t k) -> uint4 {
        if (!segmented) return a.bins[c.beg + k];
        uint32_t g = 0;
#pragma unroll
        for (int st = BIN_MAX_G / 2; st >= 1; st >>= 1) if (s_segpre[g + st] <= k) g += st;
        return a.bins[s_segsrc[g] + (k - s_segpre[g])];
    };
    tile_load_keys(a, c, s_key, ~0ull);
    // A draw that carries the frame's clear starts from ONE depth: the minima are that constant (pixels of a partial tile
    // beyond the window would only raise them), and nothing has to be rebuilt before the first fragments land.  Else:
    // "nothing can be culled" until the first rebuild.
    const uint32_t hz0 = a.fused_clear ? zkey_depth(a.clear_depth) : 0u;
    if (threadIdx.x == 0) { s_next = 0; s_dirty = a.fused_clear ? 0u : 1u; s_nanflag = 0u; }
    for (int i = threadIdx.x; i < BKT_N; i += NW * 64) s_bkt[i] = 0;
    for (int i = threadIdx.x; i < HZ_SIZE; i += NW * 64) s_hz[i] = i >= HZ_C4 ? hz0 >> 1 : hz0;
    __syncthreads();

    // ---- pre-pass: this tile's 16-byte cull records {triangle, zkey of an upper bound of its rhw
    // (cull_zub), pixel bbox} are copied in NEAR-FIRST order (a 64-bucket sort on 4 exponent + 2
    // mantissa bits of the bound) into the tile's own range of `ents`.  The order cannot change any
    // output (the z resolution is order independent); it only makes the hierarchical early-z below
    // reject more.
    uint4 *__restrict__ ents = a.bins2;
    // A tile with few records (<= DIRECT_MAX, at most one per thread) never goes through bins2: the records
    // are ordered near-first through LDS (the bucket sort below on an exchange buffer that shares its bytes
    // with the main loop's staging), wave w keeps the records at sorted positions w, w+NW, ... in registers (nearest in
    // lane 0) and culls them in DIRECT_STEPS steps, so that later steps already see the depths the
    // nearer triangles left behind.
    const uint32_t nent = c.end - c.beg;
    const bool direct = nent <= (uint32_t)(DIRECT_MAX < NW * 64 ? DIRECT_MAX : NW * 64);
    uint4 dent = make_uint4(0, 0, 0, 0);
    bool dvalid = false;
    int dcount = 0, dpos = 0, dstep = 64; // lanes of this wave that hold a record; next lane to cull; lanes per step
    if (direct) {
        const bool have = threadIdx.x < nent;
        const uint4 rec = have ? source(threadIdx.x) : make_uint4(0, 0, 0, 0);
        const uint32_t bkt = COUNT ? 0u : z_bucket(rec.y);
        if (have) atomicAdd(&s_bkt[bkt], 1u);
        __syncthreads();
        if (w == 0) bucket_scan(s_bkt, lane, 0u);
        __syncthreads();
        uint4 *const exch = s_exch;
        if (have) exch[atomicAdd(&s_bkt[bkt], 1u)] = rec;
        __syncthreads();
        const uint32_t q = (uint32_t)lane * NW + (uint32_t)w;
        dvalid = q < nent;
        if (dvalid) dent = exch[q];
        dcount = __popcll(__ballot(dvalid));
        dstep = max(DIRECT_MIN, (dcount + DIRECT_STEPS - 1) / DIRECT_STEPS);
        __syncthreads(); // the staging is written again by phase 1b
    } else {
        const bool sorted = !COUNT && nent > 2u * B;
        auto bucket_of = [&](const uint4 &e) { return sorted ? z_bucket(e.y) : 0u; };
        uint4 ce[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t e = threadIdx.x + (uint32_t)(NW * 64) * k;
            ce[k] = make_uint4(0, 0, 0, 0);
            if (e < nent) { ce[k] = source(e); atomicAdd(&s_bkt[bucket_of(ce[k])], 1u); }
        }
        for (uint32_t e = threadIdx.x + 4u * NW * 64; e < nent; e += NW * 64) atomicAdd(&s_bkt[bucket_of(source(e))], 1u);
        __syncthreads();
        if (w == 0) bucket_scan(s_bkt, lane, c.beg);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t e = threadIdx.x + (uint32_t)(NW * 64) * k;
            if (e < nent) ents[atomicAdd(&s_bkt[bucket_of(ce[k])], 1u)] = ce[k];
        }
        for (uint32_t e = threadIdx.x + 4u * NW * 64; e < nent; e += NW * 64) {
            const uint4 en = source(e);
            ents[atomicAdd(&s_bkt[bucket_of(en)], 1u)] = en;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }

    __builtin_amdgcn_s_setprio(0);
    const uint32_t le_lo = lane < 32 ? (2u << lane) - 1u : 0xFFFFFFFFu;
    const uint32_t le_hi = lane < 32 ? 0u : (2u << (lane - 32)) - 1u;
    uint32_t n_cov = 0, n_nan = 0;
#ifdef FRR_DEBUG_COUNTERS
    uint32_t d_tri = 0, d_alive = 0, d_rows = 0, d_spans = 0, d_spans_live = 0, d_frags = 0, d_fwin = 0, d_rwin = 0;
    uint32_t d_pre = 0, d_win = 0, d_rebuild = 0; // fragments a per-pixel zub test would skip; fragments that took the pixel; hi-z rebuilds
#endif

    // bin entries culled per step: every survivor of a step gets a staging slot (<= B per wave), and its fragments are
    // resolved before the next step culls (fresh z minima: the early-z of the later, farther records lives on them)
    constexpr int CULL = SPAN_CULL < B ? SPAN_CULL : B;
    if (direct) { // (B can be smaller than half a wave)
        const int nsteps = max(DIRECT_STEPS, (dcount + B - 1) / B);
        dstep = max(min(DIRECT_MIN, B), (dcount + nsteps - 1) / nsteps);
    }
    FRR_T(0);
#ifdef FRR_DEBUG_COUNTERS
    d_rt1 = __builtin_amdgcn_s_memrealtime();
#endif
    for (;;) {
        // ---- phase 1: lane = bin entry, CULL per step.  bbox-in-tile + whole-triangle early-z on the 16-byte cull
        // record; a survivor then fetches its 64-byte setup record and stages its edge data at its RANK among the
        // survivors (lanes and ranks ascend together, so "the k-th head" below is staging slot k) ----
        uint32_t e0 = 0;
        int dlo = 0;
        if (direct) {
            dlo = dpos;
            dpos += dstep;
            if (dlo >= dcount) break;
        } else {
            uint32_t b = 0;
            if (lane == 0) b = atomicAdd(&s_next, 1u);
            b = __builtin_amdgcn_readfirstlane(b);
            e0 = c.beg + b * (uint32_t)CULL;
            if (e0 >= c.end) break;
        }
        // the minima are rebuilt only if some wave has resolved fragments since the last rebuild
        // (a stale minimum is a lower one: still conservative)
        if (__builtin_amdgcn_readfirstlane(s_dirty) != 0u) {
            FRR_T(1);
            if (lane == 0) s_dirty = 0u;
            hiz_rebuild(s_key, s_hz, lane);
#ifdef FRR_DEBUG_COUNTERS
            ++d_rebuild;
#endif
            FRR_T(2);
        }
        wave_lds_fence();
        const bool valid = direct ? (dvalid && lane >= dlo && lane < dlo + dstep) : lane < (int)min((uint32_t)CULL, c.end - e0);
        const uint4 en = direct ? dent : (valid ? ents[e0 + lane] : make_uint4(0, 0, 0, 0));
        const int mnx = (int)(short)(en.z & 0xFFFFu), mny = (int)(short)(en.z >> 16);
        const int mxx = (int)(short)(en.w & 0xFFFFu), mxy = (int)(short)(en.w >> 16);
        // clamped bbox (renderer.rs:285-298) in this tile: the i16 box is the saturated min/max of spi and the window
        // lies inside the i16 range, so clamping it equals clamping the i32 values
        const int bx0 = max(clampi(mnx, a.x0, a.x1), c.ax0), bx1 = min(clampi(mxx, a.x0, a.x1), c.ax0 + c.tw);
        const int by0 = max(clampi(mny, a.y0, a.y1), c.ay0), by1 = min(clampi(mxy, a.y0, a.y1), c.ay0 + c.th);
        const bool nonempty = valid && bx1 > bx0 && by1 > by0;
        // (the bbox saturates at +-32767, so out-of-range vertices show up here too)
        const int amax = max(max(abs(mnx), abs(mxx)), max(abs(mny), abs(mxy)));
        const bool safe = nonempty && win_safe && amax <= SPAN_SAFE;
        bool alive = safe;
        if (!COUNT && safe) {
            // against the 8x8 block minima when the bbox touches <= 2x2 blocks, else against the
            // 16x16 quad minima (a tile has 2x2 quads, so this always applies)
            const int gx0 = (bx0 - c.ax0) >> 3, gx1 = (bx1 - 1 - c.ax0) >> 3, gy0 = (by0 - c.ay0) >> 3, gy1 = (by1 - 1 - c.ay0) >> 3;
            const bool small = gx1 - gx0 <= 1 && gy1 - gy0 <= 1;
            const uint32_t *lv = s_hz + (small ? HZ_BLK : HZ_QUAD);
            const int sh = small ? 0 : 1, st = small ? 4 : 2;
            const int ix0 = gx0 >> sh, ix1 = gx1 >> sh, iy0 = gy0 >> sh, iy1 = gy1 >> sh;
            const uint32_t hm = min(min(lv[iy0 * st + ix0], lv[iy0 * st + ix1]), min(lv[iy1 * st + ix0], lv[iy1 * st + ix1]));
            alive = !(en.y < hm);
        }
#ifdef FRR_DEBUG_COUNTERS
        d_tri += __popcll(__ballot(nonempty)); d_alive += __popcll(__ballot(alive));
#endif
        // triangles outside the span algebra's safe range: exact brute-force sweep, right away
        unsigned long long um = __ballot(nonempty && !safe);
        while (um) {
            const int src = __builtin_ctzll(um);
            um &= um - 1;
            const uint32_t tu = (uint32_t)__builtin_amdgcn_readlane((int)en.x, src);
            uint32_t ncv = 0;
            sweep_triangle(a, c, tu, PS == FRR_PS_DEPTH ? emission_id(a, tu) : order_id(a, tu), lane, s_key, ncv, n_nan, &s_nanflag);
            n_cov += ncv;
        }
        const unsigned long long am = __ballot(alive);
        FRR_T(1);
        if (am == 0ull) continue;
        const int trank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(am >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)am, 0u));
        const uint32_t rows = alive ? (uint32_t)(by1 - by0) : 0u;
        // rows of all survivors laid end to end (at most B * 32 = 1024 of them): heads mark where each triangle's rows start
        const uint32_t rincl = wave_incl_scan_dpp(rows);
        if (alive) {
            const uint4 *rp = reinterpret_cast<const uint4 *>(a.recs + en.x);
            const uint4 q0 = rp[0], q1 = rp[1], q2 = rp[2], q3 = rp[3];
            // the record carries each edge as {kd, c} (frr_device.h: edge_words); what depends on the tile is M at the
            // origin of the bbox-in-tile, m = c - D bx0 + k by0 -- every factor of a `safe` triangle is below 2^15, the
            // 24-bit multiplies are exact -- and the 1-ulp reciprocal of D (D = 0, an A = 0 edge: 64, see SpanTri)
            auto m_of = [&](uint32_t kd, uint32_t cc) { return (int)cc + __mul24((int)(kd << 16) >> 16, by0) - __mul24((int)(kd >> 16), bx0); };
            auto r_of = [&](uint32_t kd) { return fminf(__builtin_amdgcn_rcpf((float)(kd >> 16)), 64.0f); };
            SpanTri t;
            t.m01 = m_of(q0.x, q0.y); t.m12 = m_of(q0.z, q0.w); t.m20 = m_of(q1.x, q1.y);
            t.zub = en.y;                        // zkey of an upper bound of rhw over the triangle (cull_zub)
            t.kd01 = q0.x; t.kd12 = q0.z; t.kd20 = q1.x;
            t.misc = (uint32_t)(bx0 - c.ax0) | ((uint32_t)(by0 - c.ay0) << 5) | ((uint32_t)(bx1 - bx0) << 10) |
                     (((q3.w >> REC_POS_SHIFT) & 7u) << 16) | ((rincl - rows) << 19);
            t.r01 = r_of(q0.x); t.r12 = r_of(q0.z); t.r20 = r_of(q1.x);
            t.pad = 0.0f;
            s_tri[w][trank] = t;
            s_fa[w][trank] = make_float4(u2f(q1.z), u2f(q1.w), u2f(q2.x), u2f(q2.y));
            s_fb[w][trank] = make_float4(u2f(q2.z), u2f(q2.w), u2f(q3.x), u2f(q3.y));
            s_fc[w][trank] = make_float2(u2f(q3.z), u2f(PS == FRR_PS_DEPTH ? emission_id_rec(a, en.x, q3.w) : order_id(a, en.x)));
        }
        const int R = (int)__builtin_amdgcn_readlane((int)rincl, 63);
        if (lane < B / 2) s_hrow[w][lane] = 0ull;
        wave_lds_fence();
        if (rows) {
            const uint32_t st = rincl - rows;
            atomicOr(reinterpret_cast<uint32_t *>(&s_hrow[w][0]) + (st >> 5), 1u << (st & 31));
        }
        wave_lds_fence();
        const unsigned long long hrow_mine = s_hrow[w][lane & (B / 2 - 1)]; // lane i keeps the heads of row window i
        wave_lds_fence();

        int jbase = 0;             // heads seen in earlier row windows
        FRR_T(3);
        for (int r0 = 0; r0 < R; r0 += 64) {
            // ---- phase 2: lane = (triangle, row).  Exact covered span of that row. ----
            const uint32_t h_lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)hrow_mine, r0 >> 6);
            const uint32_t h_hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(hrow_mine >> 32), r0 >> 6);
            const bool ractive = r0 + lane < R;
            const int j = seg_rank(h_lo, h_hi, le_lo, le_hi, jbase - 1);       // staging slot = rank of the triangle
            int len = 0, xl = 0, yl = 0;
            uint32_t zu = 0xFFFFFFFFu;
            if (ractive) {
                const SpanTri ti = s_tri[w][j];
                const int row = r0 + lane - (int)(ti.misc >> 19);
                const int bwj = (int)((ti.misc >> 10) & 63u);
                int lo = 0, hi = bwj;
                span_edge_bound(ti.m01, ti.kd01, ti.r01, (int)(ti.misc << 15) >> 31, row, lo, hi);
                span_edge_bound(ti.m12, ti.kd12, ti.r12, (int)(ti.misc << 14) >> 31, row, lo, hi);
                span_edge_bound(ti.m20, ti.kd20, ti.r20, (int)(ti.misc << 13) >> 31, row, lo, hi);
                len = max(hi - lo, 0);
                xl = (int)(ti.misc & 31u) + lo;
                yl = (int)((ti.misc >> 5) & 31u) + row;
                zu = ti.zub;
            }
#ifdef FRR_DEBUG_COUNTERS
            d_rwin++; d_rows += __popcll(__ballot(ractive)); d_spans += __popcll(__ballot(len > 0));
#endif
            jbase += __popc(h_lo) + __popc(h_hi);
            if (COUNT) n_cov += (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_dpp((uint32_t)len), 63);
            // span-level early-z against the minima of the row's eight 4-pixel cells: the span shrinks to
            // the hull of the cells in which the triangle's depth bound could still win
            if (len > 0) {
                const uint4 ca = *reinterpret_cast<const uint4 *>(&s_hz[HZ_C4 + yl * 8]);
                const uint4 cb = *reinterpret_cast<const uint4 *>(&s_hz[HZ_C4 + yl * 8 + 4]);
                // the cells hold (minimum >> 1): "zu >> 1 below it" implies zu < minimum (conservative in the last bit), and
                // with both operands below 2^31 the sign of the difference is the comparison -- a subtraction and a funnel
                // shift per cell instead of a compare and a select.  Bit i of cm: the bound cannot win in cell i.
                const uint32_t zh = zu >> 1;
                uint32_t cm = 0u;
                cm = __builtin_amdgcn_alignbit(cm, zh - cb.w, 31); cm = __builtin_amdgcn_alignbit(cm, zh - cb.z, 31);
                cm = __builtin_amdgcn_alignbit(cm, zh - cb.y, 31); cm = __builtin_amdgcn_alignbit(cm, zh - cb.x, 31);
                cm = __builtin_amdgcn_alignbit(cm, zh - ca.w, 31); cm = __builtin_amdgcn_alignbit(cm, zh - ca.z, 31);
                cm = __builtin_amdgcn_alignbit(cm, zh - ca.y, 31); cm = __builtin_amdgcn_alignbit(cm, zh - ca.x, 31);
                const int xr = xl + len - 1;
                const int c0 = xl >> 2, c1 = xr >> 2;
                const uint32_t pm = ~cm & ((2u << c1) - (1u << c0));
                if (pm == 0u) {
                    len = 0;
                } else {
                    const int f = __builtin_ctz(pm), l = 31 - __builtin_clz(pm);
                    const int nxl = max(xl, f << 2), nxr = min(xr, (l << 2) + 3);
                    xl = nxl;
                    len = nxr - nxl + 1;
                }
            }
            // ---- spans of this window laid end to end: heads mark where each span's fragments start ----
            const unsigned long long nz = __ballot(len > 0);
#ifdef FRR_DEBUG_COUNTERS
            d_spans_live += __popcll(nz);
#endif
            if (nz == 0ull) { FRR_T(4); continue; }
            const int srank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(nz >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nz, 0u));
            const uint32_t fincl = wave_incl_scan_dpp((uint32_t)len);
            const int F = (int)__builtin_amdgcn_readlane((int)fincl, 63);
            if (lane < 32) s_hfrag[w][lane] = 0ull;
            wave_lds_fence();
            if (len > 0) {
                const uint32_t st = fincl - (uint32_t)len;
                atomicOr(reinterpret_cast<uint32_t *>(&s_hfrag[w][0]) + (st >> 5), 1u << (st & 31));
                // span descriptor: staging slot, row, and x of the span's first pixel MINUS its first fragment's position (+ 2048:
                // a window holds at most 64 * 32 fragments), so that fragment f of the window is pixel x = field + f - 2048
                s_q[w][srank] = (uint32_t)j | ((uint32_t)yl << 5) | (((uint32_t)xl + 2048u - st) << 10);
            }
            wave_lds_fence();
            const unsigned long long hfrag_mine = s_hfrag[w][lane & 31]; // lane i keeps the heads of fragment window i

            // ---- phase 3: lane = fragment.  Barycentrics, rhw, z key, LDS atomic max ----
#ifdef FRR_DEBUG_COUNTERS
            d_frags += F; d_fwin += (F + 63) / 64;
#endif
            int qbase = 0;
            FRR_T(4);
            for (int f0 = 0; f0 < F; f0 += 64) {
                const uint32_t g_lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)hfrag_mine, f0 >> 6);
                const uint32_t g_hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(hfrag_mine >> 32), f0 >> 6);
                if (f0 + lane < F) {
                    const uint32_t d = s_q[w][seg_rank(g_lo, g_hi, le_lo, le_hi, qbase - 1)];
                    const int sj = (int)(d & 31u), y = (int)((d >> 5) & 31u), x = (int)(d >> 10) + (f0 - 2048) + lane;
                    const float4 fa = s_fa[w][sj], fb = s_fb[w][sj];
                    const float2 fc = s_fc[w][sj];
                    Frag f = frag_eval(fa.x, fa.y, fa.z, fa.w, fb.x, fb.y, fb.z, fb.w, fc.x, c.ax0 + x, c.ay0 + y);
                    if (f.valid) {
                        if (f.rhw != f.rhw) { ++n_nan; s_nanflag = 1u; }
                        const unsigned long long key = ((unsigned long long)zkey_frag(f.rhw) << 32) | (unsigned long long)f2u(fc.y);
#ifdef FRR_DEBUG_COUNTERS
                        const unsigned long long old = atomicMax(&s_key[y * TILE + x], key);
                        d_pre += __popcll(__ballot(s_tri[w][sj].zub < (uint32_t)(old >> 32)));
                        d_win += __popcll(__ballot(key > old));
#else
                        atomicMax(&s_key[y * TILE + x], key);
#endif
                    }
                }
                qbase += __popc(g_lo) + __popc(g_hi);
            }
            if (lane == 0) s_dirty = 1u;
            wave_lds_fence(); // s_q / s_hfrag are rewritten by the next row window
            FRR_T(5);
        }

        wave_lds_fence(); // staging is rewritten by the next step
    }
    FRR_T(1);
    if (lane == 0 && n_cov) atomicAdd(&a.cnt->lane[a.lane].gtab[a.gpar].frag_covered, (unsigned long long)n_cov);
#ifdef FRR_DEBUG_COUNTERS
    d_rt2 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
        unsigned long long *d = a.cnt->dbg[blockIdx.x % DBG_COPIES];
        atomicAdd(d + 0, (unsigned long long)d_tri); atomicAdd(d + 1, (unsigned long long)d_alive); atomicAdd(d + 2, (unsigned long long)d_rows);
        atomicAdd(d + 3, (unsigned long long)d_spans); atomicAdd(d + 4, (unsigned long long)d_spans_live); atomicAdd(d + 5, (unsigned long long)d_frags);
        atomicAdd(d + 6, (unsigned long long)d_fwin); atomicAdd(d + 7, (unsigned long long)d_rwin);
        atomicAdd(d + 8, (unsigned long long)d_pre); atomicAdd(d + 9, (unsigned long long)d_win); atomicAdd(d + 10, (unsigned long long)d_rebuild);
    }
#endif
    if (n_nan) atomicAdd(&a.cnt->lane[a.lane].gtab[a.gpar].frag_nan, (unsigned long long)n_nan);
    __syncthreads();
    if (s_nanflag != 0u) {
        // NaN fragments (renderer.rs:363-366): the pixels they cover are decided by a second, unculled sweep of ALL the
        // tile's records -- only what was submitted after a pixel's last NaN fragment counts there (tile_nan_begin)
        tile_nan_begin(c, s_key, s_nanL);
        __syncthreads();
        uint32_t dummy_cov = 0, dummy_nan = 0;
        if (direct) {
            unsigned long long m = __ballot(dvalid);
            while (m) {
                const int src = __builtin_ctzll(m);
                m &= m - 1;
                const uint32_t tu = (uint32_t)__builtin_amdgcn_readlane((int)dent.x, src);
                sweep_triangle<true>(a, c, tu, PS == FRR_PS_DEPTH ? emission_id(a, tu) : order_id(a, tu), lane, s_key, dummy_cov, dummy_nan, nullptr, s_nanL);
            }
        } else {
            for (uint32_t e = c.beg + (uint32_t)w; e < c.end; e += NW) {
                const uint32_t tu = __builtin_amdgcn_readfirstlane(ents[e].x);
                sweep_triangle<true>(a, c, tu, PS == FRR_PS_DEPTH ? emission_id(a, tu) : order_id(a, tu), lane, s_key, dummy_cov, dummy_nan, nullptr, s_nanL);
            }
        }
        __syncthreads();
    }
    FRR_T(6);
    if constexpr (PS == FRR_PS_DEPTH) tile_resolve_depth4(a, c, s_key);
    else tile_resolve<K, PS>(a, u, c, s_key, TEXTURED ? s_u8 : nullptr);
#ifdef FRR_DEBUG_COUNTERS
    FRR_T(7);
    if (lane == 0) {
        unsigned long long *d = a.cnt->dbg[blockIdx.x % DBG_COPIES];
#pragma unroll
        for (int i = 0; i < 8; ++i) atomicAdd(d + 12 + i, (unsigned long long)d_t[i]);
        atomicAdd(d + 20, 1ull); // waves that ran the main loop
        if (a.dbg_tiles && w == 0) {
            // timeline of this workgroup: start / main loop / end on the 100 MHz clock, where it ran, how much it had to do
            unsigned long long *tl = a.dbg_tiles + (size_t)blockIdx.x * 8;
            tl[0] = d_rt0; tl[1] = d_rt1; tl[2] = d_rt2; tl[3] = __builtin_amdgcn_s_memrealtime();
            tl[4] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32); // HW_ID, XCC_ID
            tl[5] = (unsigned long long)(c.end - c.beg) | ((unsigned long long)(uint32_t)c.tile << 32);
            tl[6] = d_t[1] + d_t[3] + d_t[4] + d_t[5]; tl[7] = d_t[0];
        }
    }
#endif
#undef FRR_T
}

// debug: the DPP scan against a serial sum (tests)
__global__ void k_debug_scan(const uint32_t *in, uint32_t *out)
{
    out[threadIdx.x] = wave_incl_scan_dpp(in[threadIdx.x]);
}

} // namespace frr
