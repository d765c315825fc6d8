// frr_exact.h -- scalar helpers that reproduce Rust / glibc fp32 semantics bit for bit on gfx950.
//
// The whole library is compiled with -ffp-contract=off (no FMA contraction), default IEEE
// division/sqrt (-fhip-fp32-correctly-rounded-divide-sqrt, hipcc default) and fp32 denormals
// preserved (hipcc default), so `a*b+c` below means two roundings exactly as in the reference
// (/root/reference/f_renderer/src/renderer.rs; Rust never contracts or reassociates).
#pragma once
#ifndef __HIPCC_RTC__      // (hiprtc -- user shaders, frr_shader_register -- brings the HIP device headers and the fixed-width types itself)
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif
#ifndef INT32_MAX
#define INT32_MAX 2147483647
#define INT32_MIN (-2147483647 - 1)
#endif

#define FRR_HD __host__ __device__ __forceinline__

namespace frr {

FRR_HD uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
FRR_HD float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

// Rust's saturating casts and f32::max, two forms each: `*_ref` spells the semantics out (host code, and what the device forms
// are checked against for all 2^32 operands: frr_debug_rcp_check); the device forms are the single gfx950 instruction that
// has exactly those semantics -- a compare-and-select chain costs 8+ cycles per link there (tools/valu_rates.hip).
// Rust `f32 as i32`: truncate, saturate, NaN -> 0  (used at renderer.rs:233-234)
FRR_HD int32_t f32_as_i32_ref(float f)
{
    if (f != f) return 0;
    if (f >= 2147483648.0f) return INT32_MAX;
    if (f <= -2147483648.0f) return INT32_MIN;
    return (int32_t)f;
}
// Rust `f32 as u32`  (renderer.rs:522-523)
FRR_HD uint32_t f32_as_u32_ref(float f)
{
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}
// Rust `f32 as u8` after f32::clamp(0,255) (renderer.rs:9-12): NaN -> 0
FRR_HD uint32_t quantize_u8_ref(float v)
{
    float x = v * 255.0f;
    if (x < 0.0f) x = 0.0f;
    if (x > 255.0f) x = 255.0f;
    if (!(x > 0.0f)) return 0u;
    return (uint32_t)x; // x in (0,255]
}
// f32::max: NaN operand yields the other
FRR_HD float f32_max_ref(float a, float b)
{
    if (a != a) return b;
    if (b != b) return a;
    return a > b ? a : b;
}
FRR_HD int32_t f32_as_i32(float f)
{
#if defined(__HIP_DEVICE_COMPILE__)
    int32_t r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(f));   // truncates, saturates, NaN -> 0
    return r;
#else
    return f32_as_i32_ref(f);
#endif
}
FRR_HD uint32_t f32_as_u32(float f)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(f));   // truncates, saturates (negative -> 0), NaN -> 0
    return r;
#else
    return f32_as_u32_ref(f);
#endif
}
FRR_HD uint32_t quantize_u8(float v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t u = f32_as_u32(v * 255.0f);        // negative and NaN -> 0 by the conversion itself
    return u < 255u ? u : 255u;
#else
    return quantize_u8_ref(v);
#endif
}
FRR_HD float f32_max(float a, float b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // the shaders' max(x, 0.0): v_max_f32 is IEEE maxNum (a NaN operand yields the other) and gives +0 for max(-0, +0), as
    // the comparison form does when b is +0
    if (__builtin_constant_p(b) && f2u(b) == 0u) return __builtin_fmaxf(a, 0.0f);
#endif
    return f32_max_ref(a, b);
}
// f32::total_cmp as a signed sortable key (renderer.rs:217)
FRR_HD int32_t total_order_key(float f)
{
    int32_t i = (int32_t)f2u(f);
    i ^= (int32_t)(((uint32_t)(i >> 31)) >> 1);
    return i;
}
// Monotone map f32 -> u32 for the `rhw < depth` test (renderer.rs:363): a < b  <=>  zkey(a) < zkey(b)
// for all non-NaN a,b, with -0.0 and +0.0 mapped to the same key (they compare equal).
FRR_HD uint32_t zkey(float f)
{
    uint32_t u = f2u(f + 0.0f); // -0.0 + 0.0 == +0.0 (round-to-nearest); every other value unchanged
    return u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u);
}
// NaN depths (renderer.rs:363-366: `rhw < depth` is false when either side is NaN, so a NaN fragment always passes and
// the next fragment on a NaN pixel always passes).  In pixel keys: a NaN already IN the depth buffer is the lowest key,
// everything beats it; a NaN FRAGMENT takes the highest key, so that after the main pass the pixel holds the id of its
// LAST NaN fragment, and the tile kernels' second pass (tile_nan_*) then keeps only the fragments submitted after it.
// Neither value is the image of a non-NaN float (those lie in [0x007FFFFF, 0xFF800000]).
constexpr uint32_t ZKEY_NAN_BELOW = 0u, ZKEY_NAN_FRAG = 0xFFFFFFFFu;
FRR_HD uint32_t zkey_depth(float d) { return d == d ? zkey(d) : ZKEY_NAN_BELOW; }
FRR_HD uint32_t zkey_frag(float rhw) { return rhw == rhw ? zkey(rhw) : ZKEY_NAN_FRAG; }
FRR_HD float zkey_decode(uint32_t k)
{
    return u2f((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// 1.0f / s, correctly rounded, in three instructions for the operands the fragment loop sees.
// v_rcp_f32 is a 1-ulp estimate y; one Newton step written with exact residuals, y + y*(1 - s*y),
// lands on the IEEE quotient for every s in [2^-64, 2^64] -- verified EXHAUSTIVELY against the
// compiler's IEEE division on the device (frr_debug_rcp_check, tests/test_gpu_parity.py); operands
// outside that range (and NaN/inf/0) take the IEEE division.  The three instructions run unconditionally and the
// division only if SOME lane of the wave needs it (one scalar branch that is almost never taken, instead of a
// divergent one around each path).
FRR_HD bool recip_fast_range(float s)
{
    const uint32_t e = (f2u(s) >> 23) & 0x1FFu; // sign | exponent
    return e - 63u < 129u;                      // positive, 2^-64 <= s < 2^65
}
FRR_HD float recip_exact(float s)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const bool in = recip_fast_range(s);
    const float y = __builtin_amdgcn_rcpf(s);
    float r = __builtin_fmaf(y, __builtin_fmaf(-s, y, 1.0f), y);
    if (__builtin_amdgcn_ballot_w64(!in) != 0ull) r = in ? r : 1.0f / s;
    return r;
#else
    return 1.0f / s;
#endif
}

// 1.0f / sqrtf(d) as the reference rounds it (two correctly rounded operations: Vec3::length_recip), for the
// normalisations of the shaders.  For d in [2^-64, 2^65) the square root needs none of the scaling the compiler's IEEE
// sqrtf carries for tiny operands: v_sqrt_f32 is a 1-ulp estimate, and the two neighbours are tried with exact
// residuals exactly as the compiler's own expansion does; the root then lies in recip_exact's verified range.
// Everything else takes the IEEE operations (again only if some lane of the wave needs them).
FRR_HD float rsqrt_exact(float d)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const bool in = recip_fast_range(d);
    float s = __builtin_amdgcn_sqrtf(d);
    const float sd = u2f(f2u(s) - 1u), su = u2f(f2u(s) + 1u);
    const float rd = __builtin_fmaf(-sd, s, d), ru = __builtin_fmaf(-su, s, d);
    s = rd <= 0.0f ? sd : s;
    s = ru > 0.0f ? su : s;
    const float y = __builtin_amdgcn_rcpf(s);
    float r = __builtin_fmaf(y, __builtin_fmaf(-s, y, 1.0f), y);
    if (__builtin_amdgcn_ballot_w64(!in) != 0ull) r = in ? r : 1.0f / sqrtf(d);
    return r;
#else
    return 1.0f / sqrtf(d);
#endif
}

// ---------------------------------------------------------------------------------------------
// atan2f: renderer.rs:208-209 calls f32::atan2, i.e. the platform libm.  This is a port of the
// fdlibm-derived float algorithm glibc 2.35 ships (sysdeps/ieee754/flt-32/e_atan2f.c, s_atanf.c:
// argument reduction to 4 breakpoints + an 11-term odd/even split polynomial), written from the
// published algorithm; it uses only fp32 + - * / so it rounds identically on host and device.
// tests/test_atan2f.py pins it against the container's glibc atan2f (exhaustive atanf sweep +
// random/structured atan2f pairs).
// ---------------------------------------------------------------------------------------------
FRR_HD float fd_atanf(float x)
{
    const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
    const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f,
                aT3 = -1.1111110449e-01f, aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f,
                aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f, aT8 = 4.9768779427e-02f,
                aT9 = -3.6531571299e-02f, aT10 = 1.6285819933e-02f;
    int32_t hx = (int32_t)f2u(x);
    int32_t ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c000000) { // |x| >= 2^25
        if (ix > 0x7f800000) return x + x; // NaN
        float r = atanhi[3] + atanlo[3];
        return hx > 0 ? r : -r;
    }
    if (ix < 0x3ee00000) {       // |x| < 0.4375
        if (ix < 0x31000000) return x; // |x| < 2^-29
        id = -1;
    } else {
        x = u2f((uint32_t)ix); // fabsf
        if (ix < 0x3f980000) {   // |x| < 1.1875
            if (ix < 0x3f300000) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); }
            else                 { id = 1; x = (x - 1.0f) / (x + 1.0f); }
        } else {
            if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); }
            else                 { id = 3; x = -1.0f / x; }
        }
    }
    float z = x * x;
    float w = z * z;
    float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (id < 0) return x - x * (s1 + s2);
    float hi = id == 0 ? atanhi[0] : id == 1 ? atanhi[1] : id == 2 ? atanhi[2] : atanhi[3];
    float lo = id == 0 ? atanlo[0] : id == 1 ? atanlo[1] : id == 2 ? atanlo[2] : atanlo[3];
    z = hi - ((x * (s1 + s2) - lo) - x);
    return hx < 0 ? -z : z;
}

FRR_HD float fd_atan2f(float y, float x)
{
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f,
                pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    int32_t hx = (int32_t)f2u(x), hy = (int32_t)f2u(y);
    int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y; // NaN
    if (hx == 0x3f800000) return fd_atanf(y);             // x == 1.0
    int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);          // 2*sign(x) + sign(y)
    if (iy == 0) {
        switch (m) {
        case 0: case 1: return y;
        case 2: return pi + tiny;
        default: return -pi - tiny;
        }
    }
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) {
            switch (m) {
            case 0: return pi_o_4 + tiny;
            case 1: return -pi_o_4 - tiny;
            case 2: return 3.0f * pi_o_4 + tiny;
            default: return -3.0f * pi_o_4 - tiny;
            }
        } else {
            switch (m) {
            case 0: return 0.0f;
            case 1: return -0.0f;
            case 2: return pi + tiny;
            default: return -pi - tiny;
            }
        }
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    int32_t k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = fd_atanf(u2f(f2u(y / x) & 0x7fffffffu));
    switch (m) {
    case 0: return z;
    case 1: return u2f(f2u(z) ^ 0x80000000u);
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
    }
}

// Branch-free forms of the two functions above for the GPU (the three angle evaluations of a
// triangle then interleave instead of serialising on divergent branches).  Every lane performs the
// operations of its own fdlibm path on the selected operands, so the results are bit-identical to
// fd_atanf / fd_atan2f (and to glibc): tools/atan2f_check.cpp sweeps all 2^32 atanf inputs and
// 2*10^9 atan2f pairs for both forms.
FRR_HD float fd_atanf_bf(float x)
{
    const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f,
                aT3 = -1.1111110449e-01f, aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f,
                aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f, aT8 = 4.9768779427e-02f,
                aT9 = -3.6531571299e-02f, aT10 = 1.6285819933e-02f;
    const int32_t hx = (int32_t)f2u(x);
    const int32_t ix = hx & 0x7fffffff;
    const float ax = u2f((uint32_t)ix);
    const int id = ix < 0x3ee00000 ? -1 : (ix < 0x3f300000 ? 0 : (ix < 0x3f980000 ? 1 : (ix < 0x401c0000 ? 2 : 3)));
    const float num = id == 0 ? 2.0f * ax - 1.0f : (id == 1 ? ax - 1.0f : (id == 2 ? ax - 1.5f : -1.0f));
    const float den = id == 0 ? 2.0f + ax : (id == 1 ? ax + 1.0f : (id == 2 ? 1.0f + 1.5f * ax : ax));
    const float q = num / (id < 0 ? 1.0f : den);
    const float xr = id < 0 ? x : q;
    const float z = xr * xr;
    const float w = z * z;
    const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    const float hi = id == 0 ? 4.6364760399e-01f : (id == 1 ? 7.8539812565e-01f : (id == 2 ? 9.8279368877e-01f : 1.5707962513e+00f));
    const float lo = id == 0 ? 5.0121582440e-09f : (id == 1 ? 3.7748947079e-08f : (id == 2 ? 3.4473217170e-08f : 7.5497894159e-08f));
    const float r_small = xr - xr * (s1 + s2);
    const float r_big0 = hi - ((xr * (s1 + s2) - lo) - xr);
    const float r_big = hx < 0 ? -r_big0 : r_big0;
    float res = id < 0 ? r_small : r_big;
    res = ix < 0x31000000 ? x : res;                                   // |x| < 2^-29
    const float pio2 = 1.5707962513e+00f + 7.5497894159e-08f;
    res = ix >= 0x4c000000 ? (ix > 0x7f800000 ? x + x : (hx > 0 ? pio2 : -pio2)) : res; // |x| >= 2^25, NaN
    return res;
}

FRR_HD float fd_atan2f_bf(float y, float x)
{
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f,
                pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const int32_t hx = (int32_t)f2u(x), hy = (int32_t)f2u(y);
    const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    const int32_t k = (iy - ix) >> 23;
    const float za = fd_atanf_bf(u2f(f2u(y / x) & 0x7fffffffu));
    const float z = k > 60 ? pi_o_2 + 0.5f * pi_lo : ((hx < 0 && k < -60) ? 0.0f : za);
    float res = m == 0 ? z : (m == 1 ? u2f(f2u(z) ^ 0x80000000u) : (m == 2 ? pi - (z - pi_lo) : (z - pi_lo) - pi));
    // the early returns of e_atan2f.c, lowest priority first (the x == 1.0 shortcut is value-neutral)
    const float hpi = hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    res = iy == 0x7f800000 ? hpi : res;
    const float xinf_yinf = m == 0 ? pi_o_4 + tiny : (m == 1 ? -pi_o_4 - tiny : (m == 2 ? 3.0f * pi_o_4 + tiny : -3.0f * pi_o_4 - tiny));
    const float xinf_yfin = m == 0 ? 0.0f : (m == 1 ? -0.0f : (m == 2 ? pi + tiny : -pi - tiny));
    res = ix == 0x7f800000 ? (iy == 0x7f800000 ? xinf_yinf : xinf_yfin) : res;
    res = ix == 0 ? hpi : res;
    res = iy == 0 ? (m < 2 ? y : (m == 2 ? pi + tiny : -pi - tiny)) : res;
    res = (ix > 0x7f800000 || iy > 0x7f800000) ? x + y : res;
    return res;
}

// sort angle of renderer.rs:205-216: atan2 mapped to [0, 2pi)
FRR_HD float sort_angle(float fy, float fx)
{
#if defined(__HIP_DEVICE_COMPILE__)
    float a = fd_atan2f_bf(fy, fx);
#else
    float a = fd_atan2f(fy, fx);
#endif
    if (a < 0.0f) a += 3.14159274101257324f * 2.0f;
    return a;
}

} // namespace frr
