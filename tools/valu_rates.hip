// Dev tool: issue rate of the instruction kinds the tile kernel is made of, per SIMD, at 1..8 waves per SIMD.
// Each wave runs REP x 64 copies of one instruction on 8 independent registers; cycles by s_memtime (shader clock).
//   hipcc --offload-arch=gfx950 -O2 -o tools/valu_rates tools/valu_rates.hip && tools/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP 512
#define OP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define BODY(S) OP8(S) OP8(S) OP8(S) OP8(S) OP8(S) OP8(S) OP8(S) OP8(S)

#define KERNEL(NAME, ASM)                                                                         \
    __global__ __launch_bounds__(1024) void NAME(unsigned long long *out, unsigned *sink, unsigned seed) \
    {                                                                                             \
        unsigned r[8];                                                                            \
        for (int i = 0; i < 8; ++i) r[i] = seed * (threadIdx.x + 1u) + i * 0x3f800000u;           \
        unsigned b = seed | 1u, c = seed + 3u; unsigned sb = __builtin_amdgcn_readfirstlane(seed), sc = sb + 1u; \
        __syncthreads();                                                                          \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                               \
        for (int k = 0; k < REP; ++k) {                                                           \
            BODY(ASM)                                                                             \
        }                                                                                         \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                               \
        unsigned s = 0;                                                                           \
        for (int i = 0; i < 8; ++i) s ^= r[i];                                                    \
        if (s == 0x12345u + sb + sc) sink[0] = s;                                                           \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;         \
    }

#define KERNEL2(NAME, ASM, PRE)                                                                         \
    __global__ __launch_bounds__(1024) void NAME(unsigned long long *out, unsigned *sink, unsigned seed) \
    {                                                                                             \
        unsigned r[8];                                                                            \
        for (int i = 0; i < 8; ++i) r[i] = seed * (threadIdx.x + 1u) + i * 0x3f800000u;           \
        unsigned b = seed | 1u, c = seed + 3u; unsigned sb = __builtin_amdgcn_readfirstlane(seed), sc = sb + 1u; \
        __syncthreads(); PRE                                                                      \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                               \
        for (int k = 0; k < REP; ++k) {                                                           \
            BODY(ASM)                                                                             \
        }                                                                                         \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                               \
        unsigned s = 0;                                                                           \
        for (int i = 0; i < 8; ++i) s ^= r[i];                                                    \
        if (s == 0x12345u + sb + sc) sink[0] = s;                                                           \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;         \
    }


#define A_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(b), "v"(c));
#define A_ADDF(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
#define A_MULF(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
#define A_ADDU(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
#define A_AND(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
#define A_SHL(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(r[i]));
#define A_CNDM(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(b) : );
#define A_CMP(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(r[i]), "v"(b) : "vcc");
#define A_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
#define A_MUL24(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r[i]) : "v"(b));
#define A_MAD24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r[i]) : "v"(b), "v"(c));
#define A_CVTFI(i) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(r[i]));
#define A_CVTIF(i) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(r[i]));
#define A_RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
#define A_SQRT(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(r[i]));
#define A_MINU(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
#define A_MAX3(i) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(b), "v"(c));
#define A_MED3(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(b), "v"(c));
#define A_BFE(i) asm volatile("v_bfe_u32 %0, %0, 3, 7" : "+v"(r[i]));
#define A_DPP(i) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r[i]));
#define A_ADDDPP(i) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r[i]));
#define A_BCNT(i) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
#define A_FFBH(i) asm volatile("v_ffbh_u32 %0, %0" : "+v"(r[i]));
#define A_LSHL64(i) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(*(unsigned long long *)&r[i & ~1]));
#define A_SUBF(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
#define A_XOR3(i) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(b), "v"(c));
#define A_LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(r[i]) : "v"(b));
#define A_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(unsigned long long *)&r[i & ~1]) : "v"(*(unsigned long long *)&r[(i & ~1) ^ 2]));
#define A_SALU(i) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sb));
#define A_MIX(i) asm volatile("v_add_u32 %0, %0, %2\n s_add_u32 %1, %1, 1" : "+v"(r[i]), "+s"(sc) : "v"(b));
#define A_READLANE(i) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sc) : "v"(r[i]));

KERNEL(k_fma, A_FMA) KERNEL(k_addf, A_ADDF) KERNEL(k_mulf, A_MULF) KERNEL(k_subf, A_SUBF) KERNEL(k_addu, A_ADDU) KERNEL(k_and, A_AND)
KERNEL(k_shl, A_SHL) KERNEL(k_cndm, A_CNDM) KERNEL(k_cmp, A_CMP) KERNEL(k_mullo, A_MULLO) KERNEL(k_mul24, A_MUL24)
KERNEL(k_mad24, A_MAD24) KERNEL(k_cvtfi, A_CVTFI) KERNEL(k_cvtif, A_CVTIF) KERNEL(k_rcp, A_RCP) KERNEL(k_sqrt, A_SQRT)
KERNEL(k_minu, A_MINU) KERNEL(k_max3, A_MAX3) KERNEL(k_med3, A_MED3) KERNEL(k_bfe, A_BFE) KERNEL(k_dpp, A_DPP)
KERNEL(k_adddpp, A_ADDDPP) KERNEL(k_bcnt, A_BCNT) KERNEL(k_ffbh, A_FFBH) KERNEL(k_lshl64, A_LSHL64) KERNEL(k_xad, A_XOR3)
KERNEL(k_lshladd, A_LSHLADD) KERNEL(k_pkfma, A_PKFMA)

#define X_cndm_vcc1(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(b) : );
#define X_cndm_e64(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(r[i]) : "v"(b) : "s20", "s21");
#define X_cndm_nodep(i) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(r[i]) : "v"(b), "v"(c) : );
#define X_cmp_e64(i) asm volatile("v_cmp_lt_u32_e64 s[20:21], %0, %1" : : "v"(r[i]), "v"(b) : "s20", "s21");
#define X_cmp_f(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(r[i]), "v"(b) : "vcc");
#define X_cmpcnd(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(r[i]) : "v"(b), "v"(c) : "vcc", "scc");
#define X_maxf(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
#define X_maxi(i) asm volatile("v_max_i32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
#define X_or(i) asm volatile("v_or_b32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
#define X_xor(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
#define X_subu(i) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
#define X_lshr(i) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(r[i]));
#define X_ashr(i) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(r[i]));
#define X_andor(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(b), "v"(c));
#define X_add3(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(b), "v"(c));
#define X_mulhi(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
#define X_muli24(i) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(r[i]) : "v"(b));
#define X_cvtfu(i) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(r[i]));
#define X_trunc(i) asm volatile("v_trunc_f32 %0, %0" : "+v"(r[i]));
#define X_mov(i) asm volatile("v_mov_b32 %0, %1" : "=v"(r[i]) : "v"(b));
#define X_bfi(i) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(b), "v"(c));
#define X_perm(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(b), "v"(c));
#define X_fmac(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(r[i]) : "v"(b), "v"(c));
#define X_addco(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(r[i]) : "v"(b) : "vcc");
#define X_addc(i) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(r[i]) : "v"(b) : "vcc");
#define X_fabs(i) asm volatile("v_and_b32 %0, 0x7fffffff, %0" : "+v"(r[i]));
#define X_mulabs(i) asm volatile("v_mul_f32_e64 %0, |%0|, %1" : "+v"(r[i]) : "v"(b));
#define X_subabs(i) asm volatile("v_sub_f32_e64 %0, |%0|, |%1|" : "+v"(r[i]) : "v"(b));
#define X_mbcnt(i) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(r[i]) : "v"(b));
#define X_minf(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
#define X_maxu(i) asm volatile("v_max_u32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
#define X_lshlor(i) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(r[i]) : "v"(b));
#define X_addlit(i) asm volatile("v_add_u32 %0, 0x12345, %0" : "+v"(r[i]));
#define X_addsgpr(i) asm volatile("v_add_u32 %0, s20, %0" : "+v"(r[i]) :: "s20");
#define X_addf_e64(i) asm volatile("v_add_f32_e64 %0, %0, %1" : "+v"(r[i]) : "v"(b));
#define X_sad(i) asm volatile("v_subrev_u32 %0, %1, %0" : "+v"(r[i]) : "v"(b));
#define X_readlane(i) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(r[i]) : "s20");
#define X_readfirst(i) asm volatile("v_readfirstlane_b32 s20, %0" : : "v"(r[i]) : "s20");
KERNEL2(kx_cndm_vcc1, X_cndm_vcc1, asm volatile("s_mov_b64 vcc, -1" ::: "vcc");)
KERNEL2(kx_cndm_e64, X_cndm_e64, asm volatile("s_mov_b64 s[20:21], 0x5555" ::: "s20", "s21");)
KERNEL2(kx_cndm_nodep, X_cndm_nodep, asm volatile("s_mov_b64 vcc, 0x5555" ::: "vcc");)
KERNEL2(kx_cmp_e64, X_cmp_e64, (void)0;)
KERNEL2(kx_cmp_f, X_cmp_f, (void)0;)
KERNEL2(kx_cmpcnd, X_cmpcnd, (void)0;)
KERNEL2(kx_maxf, X_maxf, (void)0;)
KERNEL2(kx_maxi, X_maxi, (void)0;)
KERNEL2(kx_or, X_or, (void)0;)
KERNEL2(kx_xor, X_xor, (void)0;)
KERNEL2(kx_subu, X_subu, (void)0;)
KERNEL2(kx_lshr, X_lshr, (void)0;)
KERNEL2(kx_ashr, X_ashr, (void)0;)
KERNEL2(kx_andor, X_andor, (void)0;)
KERNEL2(kx_add3, X_add3, (void)0;)
KERNEL2(kx_mulhi, X_mulhi, (void)0;)
KERNEL2(kx_muli24, X_muli24, (void)0;)
KERNEL2(kx_cvtfu, X_cvtfu, (void)0;)
KERNEL2(kx_trunc, X_trunc, (void)0;)
KERNEL2(kx_mov, X_mov, (void)0;)
KERNEL2(kx_bfi, X_bfi, (void)0;)
KERNEL2(kx_perm, X_perm, (void)0;)
KERNEL2(kx_fmac, X_fmac, (void)0;)
KERNEL2(kx_addco, X_addco, (void)0;)
KERNEL2(kx_addc, X_addc, (void)0;)
KERNEL2(kx_fabs, X_fabs, (void)0;)
KERNEL2(kx_mulabs, X_mulabs, (void)0;)
KERNEL2(kx_subabs, X_subabs, (void)0;)
KERNEL2(kx_mbcnt, X_mbcnt, (void)0;)
KERNEL2(kx_minf, X_minf, (void)0;)
KERNEL2(kx_maxu, X_maxu, (void)0;)
KERNEL2(kx_lshlor, X_lshlor, (void)0;)
KERNEL2(kx_addlit, X_addlit, (void)0;)
KERNEL2(kx_addsgpr, X_addsgpr, (void)0;)
KERNEL2(kx_addf_e64, X_addf_e64, (void)0;)
KERNEL2(kx_sad, X_sad, (void)0;)
KERNEL2(kx_readlane, X_readlane, (void)0;)
KERNEL2(kx_readfirst, X_readfirst, (void)0;)
#define XTRA_TAB {"v_cndmask_b32 vcc=-1 set", kx_cndm_vcc1}, {"v_cndmask_b32_e64 sgpr", kx_cndm_e64}, {"v_cndmask dst!=src", kx_cndm_nodep}, {"v_cmp_lt_u32_e64 sgpr", kx_cmp_e64}, {"v_cmp_lt_f32 vcc", kx_cmp_f}, {"v_cmp+v_cndmask (per pair)", kx_cmpcnd}, {"v_max_f32", kx_maxf}, {"v_max_i32", kx_maxi}, {"v_or_b32", kx_or}, {"v_xor_b32", kx_xor}, {"v_sub_u32", kx_subu}, {"v_lshrrev_b32", kx_lshr}, {"v_ashrrev_i32", kx_ashr}, {"v_and_or_b32", kx_andor}, {"v_add3_u32", kx_add3}, {"v_mul_hi_u32", kx_mulhi}, {"v_mul_i32_i24", kx_muli24}, {"v_cvt_f32_u32", kx_cvtfu}, {"v_trunc_f32", kx_trunc}, {"v_mov_b32", kx_mov}, {"v_bfi_b32", kx_bfi}, {"v_perm_b32", kx_perm}, {"v_fmac_f32", kx_fmac}, {"v_add_co_u32 vcc", kx_addco}, {"v_addc_co_u32", kx_addc}, {"v_and_b32 lit (fabs)", kx_fabs}, {"v_mul_f32 |src|", kx_mulabs}, {"v_sub_f32 |dst| via e64", kx_subabs}, {"v_mbcnt_lo_u32_b32", kx_mbcnt}, {"v_min_f32", kx_minf}, {"v_max_u32", kx_maxu}, {"v_lshl_or_b32", kx_lshlor}, {"v_add_u32 literal", kx_addlit}, {"v_add_u32 sgpr", kx_addsgpr}, {"v_add_f32_e64", kx_addf_e64}, {"v_sub_u32 sdwa?", kx_sad}, {"v_readlane_b32", kx_readlane}, {"v_readfirstlane_b32", kx_readfirst}
#define P_p_cmp1(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(r[i]) : "v"(b), "v"(c), "v"(*(unsigned long long *)&r[i & ~1]) : "vcc", "scc", "s20", "s21", "s22", "s23");
#define P_p_cmp2(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(b), "v"(c), "v"(*(unsigned long long *)&r[i & ~1]) : "vcc", "scc", "s20", "s21", "s22", "s23");
#define P_p_cmp4(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(b), "v"(c), "v"(*(unsigned long long *)&r[i & ~1]) : "vcc", "scc", "s20", "s21", "s22", "s23");
#define P_p_sand(i) asm volatile("s_and_b64 vcc, exec, s[20:21]\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(r[i]) : "v"(b), "v"(c), "v"(*(unsigned long long *)&r[i & ~1]) : "vcc", "scc", "s20", "s21", "s22", "s23");
#define P_p_e64(i) asm volatile("v_cmp_lt_u32_e64 s[20:21], %0, %1\n v_cndmask_b32_e64 %0, %0, %2, s[20:21]" : "+v"(r[i]) : "v"(b), "v"(c), "v"(*(unsigned long long *)&r[i & ~1]) : "vcc", "scc", "s20", "s21", "s22", "s23");
#define P_p_cmpand(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n s_and_b64 vcc, vcc, s[20:21]\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(r[i]) : "v"(b), "v"(c), "v"(*(unsigned long long *)&r[i & ~1]) : "vcc", "scc", "s20", "s21", "s22", "s23");
#define P_p_d1(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_add_u32 %0, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(r[i]) : "v"(b), "v"(c), "v"(*(unsigned long long *)&r[i & ~1]) : "vcc", "scc", "s20", "s21", "s22", "s23");
#define P_p_d4(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(r[i]) : "v"(b), "v"(c), "v"(*(unsigned long long *)&r[i & ~1]) : "vcc", "scc", "s20", "s21", "s22", "s23");
#define P_p_d16(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(r[i]) : "v"(b), "v"(c), "v"(*(unsigned long long *)&r[i & ~1]) : "vcc", "scc", "s20", "s21", "s22", "s23");
#define P_p_add16(i) asm volatile(" v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(b), "v"(c), "v"(*(unsigned long long *)&r[i & ~1]) : "vcc", "scc", "s20", "s21", "s22", "s23");
#define P_p_saveexec(i) asm volatile("s_and_saveexec_b64 s[20:21], vcc\n s_or_b64 exec, exec, s[20:21]" : "+v"(r[i]) : "v"(b), "v"(c), "v"(*(unsigned long long *)&r[i & ~1]) : "vcc", "scc", "s20", "s21", "s22", "s23");
#define P_p_cmpsave(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n s_and_saveexec_b64 s[20:21], vcc\n v_add_u32 %0, %0, %1\n s_or_b64 exec, exec, s[20:21]" : "+v"(r[i]) : "v"(b), "v"(c), "v"(*(unsigned long long *)&r[i & ~1]) : "vcc", "scc", "s20", "s21", "s22", "s23");
#define P_p_pkadd(i) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(*(unsigned long long *)&r[i & ~1]) : "v"(b), "v"(c) : "vcc", "scc");
#define P_p_pkmul(i) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(*(unsigned long long *)&r[i & ~1]) : "v"(b), "v"(c) : "vcc", "scc");
#define P_p_pkmov(i) asm volatile("v_pk_mov_b32 %0, %0, %0 op_sel:[1,0]" : "+v"(*(unsigned long long *)&r[i & ~1]) : "v"(b), "v"(c) : "vcc", "scc");
#define P_p_bitop3(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x36" : "+v"(r[i]) : "v"(b), "v"(c), "v"(*(unsigned long long *)&r[i & ~1]) : "vcc", "scc", "s20", "s21", "s22", "s23");
#define P_p_lshladd64(i) asm volatile("v_lshl_add_u64 %0, %0, 2, %0" : "+v"(*(unsigned long long *)&r[i & ~1]) : "v"(b), "v"(c) : "vcc", "scc");
#define P_p_divscale(i) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(r[i]) : "v"(b), "v"(c), "v"(*(unsigned long long *)&r[i & ~1]) : "vcc", "scc", "s20", "s21", "s22", "s23");
#define P_p_divfixup(i) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(b), "v"(c), "v"(*(unsigned long long *)&r[i & ~1]) : "vcc", "scc", "s20", "s21", "s22", "s23");
#define P_p_mulf_s(i) asm volatile("v_mul_f32 %0, s20, %0" : "+v"(r[i]) : "v"(b), "v"(c), "v"(*(unsigned long long *)&r[i & ~1]) : "vcc", "scc", "s20", "s21", "s22", "s23");
#define P_p_fma_s(i) asm volatile("v_fma_f32 %0, %0, s20, %1" : "+v"(r[i]) : "v"(b), "v"(c), "v"(*(unsigned long long *)&r[i & ~1]) : "vcc", "scc", "s20", "s21", "s22", "s23");
#define P_p_fma_c(i) asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(r[i]) : "v"(b), "v"(c), "v"(*(unsigned long long *)&r[i & ~1]) : "vcc", "scc", "s20", "s21", "s22", "s23");
#define P_p_fmaak(i) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f8ccccd" : "+v"(r[i]) : "v"(b), "v"(c), "v"(*(unsigned long long *)&r[i & ~1]) : "vcc", "scc", "s20", "s21", "s22", "s23");
#define P_p_mul_c(i) asm volatile("v_mul_f32 %0, 0.5, %0" : "+v"(r[i]) : "v"(b), "v"(c), "v"(*(unsigned long long *)&r[i & ~1]) : "vcc", "scc", "s20", "s21", "s22", "s23");
#define P_p_addu_c(i) asm volatile("v_add_u32 %0, 7, %0" : "+v"(r[i]) : "v"(b), "v"(c), "v"(*(unsigned long long *)&r[i & ~1]) : "vcc", "scc", "s20", "s21", "s22", "s23");
#define P_p_cmp_s(i) asm volatile("v_cmp_lt_u32 vcc, s20, %0" : "+v"(r[i]) : "v"(b), "v"(c), "v"(*(unsigned long long *)&r[i & ~1]) : "vcc", "scc", "s20", "s21", "s22", "s23");
#define P_p_cmp_c(i) asm volatile("v_cmp_lt_u32 vcc, 7, %0" : "+v"(r[i]) : "v"(b), "v"(c), "v"(*(unsigned long long *)&r[i & ~1]) : "vcc", "scc", "s20", "s21", "s22", "s23");
#define P_p_mov_s(i) asm volatile("v_mov_b32 %0, s20" : "+v"(r[i]) : "v"(b), "v"(c), "v"(*(unsigned long long *)&r[i & ~1]) : "vcc", "scc", "s20", "s21", "s22", "s23");
#define P_p_cvtub(i) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(r[i]) : "v"(b), "v"(c), "v"(*(unsigned long long *)&r[i & ~1]) : "vcc", "scc", "s20", "s21", "s22", "s23");
#define P_p_sdwa(i) asm volatile("v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(r[i]) : "v"(b), "v"(c), "v"(*(unsigned long long *)&r[i & ~1]) : "vcc", "scc", "s20", "s21", "s22", "s23");
#define P_p_class(i) asm volatile("v_cmp_class_f32 vcc, %0, %1" : "+v"(r[i]) : "v"(b), "v"(c), "v"(*(unsigned long long *)&r[i & ~1]) : "vcc", "scc", "s20", "s21", "s22", "s23");
#define P_p_cmpx(i) asm volatile("v_cmpx_le_u32 exec, 0, %0" : "+v"(r[i]) : "v"(b), "v"(c), "v"(*(unsigned long long *)&r[i & ~1]) : "vcc", "scc", "s20", "s21", "s22", "s23");
#define P_p_readlane_nop(i) asm volatile("v_readlane_b32 s20, %0, 3\n s_nop 0\n v_and_b32 %0, s20, %0" : "+v"(r[i]) : "v"(b), "v"(c), "v"(*(unsigned long long *)&r[i & ~1]) : "vcc", "scc", "s20", "s21", "s22", "s23");
#define P_p_ballot(i) asm volatile("v_cmp_lt_u32_e64 s[20:21], %0, %1\n s_bcnt1_i32_b64 s22, s[20:21]" : "+v"(r[i]) : "v"(b), "v"(c), "v"(*(unsigned long long *)&r[i & ~1]) : "vcc", "scc", "s20", "s21", "s22", "s23");
#define P_p_salu(i) asm volatile("s_add_u32 s20, s20, 1" : "+v"(r[i]) : "v"(b), "v"(c), "v"(*(unsigned long long *)&r[i & ~1]) : "vcc", "scc", "s20", "s21", "s22", "s23");
#define P_p_salu4(i) asm volatile("s_add_u32 s20, s20, 1\n s_and_b32 s21, s21, s20\n s_lshl_b32 s22, s20, 1\n s_bcnt1_i32_b32 s23, s22" : "+v"(r[i]) : "v"(b), "v"(c), "v"(*(unsigned long long *)&r[i & ~1]) : "vcc", "scc", "s20", "s21", "s22", "s23");
#define P_p_branch_nt(i) asm volatile("s_cmp_eq_u32 s20, 0x7fffffff\n s_cbranch_scc1 1f\n 1:" : "+v"(r[i]) : "v"(b), "v"(c), "v"(*(unsigned long long *)&r[i & ~1]) : "vcc", "scc", "s20", "s21", "s22", "s23");
#define P_p_mix(i) asm volatile("v_add_u32 %0, %0, %1\n s_add_u32 s20, s20, 1" : "+v"(r[i]) : "v"(b), "v"(c), "v"(*(unsigned long long *)&r[i & ~1]) : "vcc", "scc", "s20", "s21", "s22", "s23");
KERNEL2(kp_p_cmp1, P_p_cmp1, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
KERNEL2(kp_p_cmp2, P_p_cmp2, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
KERNEL2(kp_p_cmp4, P_p_cmp4, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
KERNEL2(kp_p_sand, P_p_sand, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
KERNEL2(kp_p_e64, P_p_e64, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
KERNEL2(kp_p_cmpand, P_p_cmpand, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
KERNEL2(kp_p_d1, P_p_d1, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
KERNEL2(kp_p_d4, P_p_d4, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
KERNEL2(kp_p_d16, P_p_d16, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
KERNEL2(kp_p_add16, P_p_add16, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
KERNEL2(kp_p_saveexec, P_p_saveexec, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
KERNEL2(kp_p_cmpsave, P_p_cmpsave, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
KERNEL2(kp_p_pkadd, P_p_pkadd, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
KERNEL2(kp_p_pkmul, P_p_pkmul, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
KERNEL2(kp_p_pkmov, P_p_pkmov, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
KERNEL2(kp_p_bitop3, P_p_bitop3, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
KERNEL2(kp_p_lshladd64, P_p_lshladd64, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
KERNEL2(kp_p_divscale, P_p_divscale, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
KERNEL2(kp_p_divfixup, P_p_divfixup, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
KERNEL2(kp_p_mulf_s, P_p_mulf_s, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
KERNEL2(kp_p_fma_s, P_p_fma_s, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
KERNEL2(kp_p_fma_c, P_p_fma_c, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
KERNEL2(kp_p_fmaak, P_p_fmaak, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
KERNEL2(kp_p_mul_c, P_p_mul_c, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
KERNEL2(kp_p_addu_c, P_p_addu_c, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
KERNEL2(kp_p_cmp_s, P_p_cmp_s, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
KERNEL2(kp_p_cmp_c, P_p_cmp_c, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
KERNEL2(kp_p_mov_s, P_p_mov_s, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
KERNEL2(kp_p_cvtub, P_p_cvtub, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
KERNEL2(kp_p_sdwa, P_p_sdwa, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
KERNEL2(kp_p_class, P_p_class, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
KERNEL2(kp_p_cmpx, P_p_cmpx, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
KERNEL2(kp_p_readlane_nop, P_p_readlane_nop, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
KERNEL2(kp_p_ballot, P_p_ballot, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
KERNEL2(kp_p_salu, P_p_salu, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
KERNEL2(kp_p_salu4, P_p_salu4, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
KERNEL2(kp_p_branch_nt, P_p_branch_nt, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
KERNEL2(kp_p_mix, P_p_mix, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
#define P_q_cmpnop1(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(r[i]) : "v"(b), "v"(c) : "vcc", "scc", "s20", "s21", "s22", "s23", "v100", "v101", "v102", "v103");
#define P_q_cmpnop4i(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 v100, %0, %2, vcc\n v_cndmask_b32 v101, %1, %2, vcc\n v_cndmask_b32 v102, %2, %1, vcc\n v_cndmask_b32 v103, %1, %0, vcc" : "+v"(r[i]) : "v"(b), "v"(c) : "vcc", "scc", "s20", "s21", "s22", "s23", "v100", "v101", "v102", "v103");
#define P_q_e64_4i(i) asm volatile("v_cmp_lt_u32_e64 s[20:21], %0, %1\n s_nop 1\n v_cndmask_b32_e64 v100, %0, %2, s[20:21]\n v_cndmask_b32_e64 v101, %1, %2, s[20:21]\n v_cndmask_b32_e64 v102, %2, %1, s[20:21]\n v_cndmask_b32_e64 v103, %1, %0, s[20:21]" : "+v"(r[i]) : "v"(b), "v"(c) : "vcc", "scc", "s20", "s21", "s22", "s23", "v100", "v101", "v102", "v103");
#define P_q_cmp4sep(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 v100, %0, %2, vcc\n v_cmp_lt_u32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 v101, %1, %2, vcc\n v_cmp_lt_u32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 v102, %2, %1, vcc\n v_cmp_lt_u32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 v103, %1, %0, vcc" : "+v"(r[i]) : "v"(b), "v"(c) : "vcc", "scc", "s20", "s21", "s22", "s23", "v100", "v101", "v102", "v103");
#define P_q_cnd_vccz(i) asm volatile("v_cndmask_b32 v100, %0, %2, vcc\n v_add_u32 %0, %0, %1\n v_cndmask_b32 v101, %1, %2, vcc\n v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(b), "v"(c) : "vcc", "scc", "s20", "s21", "s22", "s23", "v100", "v101", "v102", "v103");
#define P_q_addc(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 v100, vcc, %2, %1, vcc" : "+v"(r[i]) : "v"(b), "v"(c) : "vcc", "scc", "s20", "s21", "s22", "s23", "v100", "v101", "v102", "v103");
#define P_q_max_via_sub(i) asm volatile("v_sub_u32 v100, %1, %0\n v_ashrrev_i32 v101, 31, v100\n v_and_b32 v100, v100, v101\n v_add_u32 %0, %0, v100" : "+v"(r[i]) : "v"(b), "v"(c) : "vcc", "scc", "s20", "s21", "s22", "s23", "v100", "v101", "v102", "v103");
KERNEL2(kp_q_cmpnop1, P_q_cmpnop1, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
KERNEL2(kp_q_cmpnop4i, P_q_cmpnop4i, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
KERNEL2(kp_q_e64_4i, P_q_e64_4i, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
KERNEL2(kp_q_cmp4sep, P_q_cmp4sep, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
KERNEL2(kp_q_cnd_vccz, P_q_cnd_vccz, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
KERNEL2(kp_q_addc, P_q_addc, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
KERNEL2(kp_q_max_via_sub, P_q_max_via_sub, asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0" ::: "s20", "s21", "s22", "s23");)
#define PAT_TAB {"v_cmp; s_nop 1; 1 cndmask(vcc)", kp_q_cmpnop1}, {"v_cmp; s_nop 1; 4 cndmask(vcc) indep", kp_q_cmpnop4i}, {"v_cmp_e64 s; 4 cndmask_e64 s indep", kp_q_e64_4i}, {"4x (v_cmp; s_nop 1; cndmask) indep", kp_q_cmp4sep}, {"cndmask(vcc); v_add; cndmask(vcc); v_add", kp_q_cnd_vccz}, {"v_add_co vcc; v_addc vcc (64-bit add)", kp_q_addc}, {"max via sub/ashr/and/add (4 fast)", kp_q_max_via_sub}, {"v_cmp + 1 cndmask(vcc)", kp_p_cmp1}, {"v_cmp + 2 cndmask(vcc)", kp_p_cmp2}, {"v_cmp + 4 cndmask(vcc)", kp_p_cmp4}, {"s_and_b64 vcc + cndmask(vcc)", kp_p_sand}, {"v_cmp_e64 s + cndmask_e64 s", kp_p_e64}, {"v_cmp; s_and vcc; cndmask(vcc)", kp_p_cmpand}, {"v_cmp; 1 v_add; cndmask(vcc)", kp_p_d1}, {"v_cmp; 4 v_add; cndmask(vcc)", kp_p_d4}, {"v_cmp; 16 v_add; cndmask(vcc)", kp_p_d16}, {"16 v_add (reference)", kp_p_add16}, {"s_and_saveexec + s_or exec", kp_p_saveexec}, {"v_cmp; saveexec; v_add; s_or exec", kp_p_cmpsave}, {"v_pk_add_f32", kp_p_pkadd}, {"v_pk_mul_f32", kp_p_pkmul}, {"v_pk_mov_b32", kp_p_pkmov}, {"v_bitop3_b32", kp_p_bitop3}, {"v_lshl_add_u64", kp_p_lshladd64}, {"v_div_scale_f32", kp_p_divscale}, {"v_div_fixup_f32", kp_p_divfixup}, {"v_mul_f32 sgpr src", kp_p_mulf_s}, {"v_fma_f32 sgpr src", kp_p_fma_s}, {"v_fma_f32 inline const", kp_p_fma_c}, {"v_fmaak_f32 literal", kp_p_fmaak}, {"v_mul_f32 inline 0.5", kp_p_mul_c}, {"v_add_u32 inline 7", kp_p_addu_c}, {"v_cmp_lt_u32 sgpr src", kp_p_cmp_s}, {"v_cmp_lt_u32 inline", kp_p_cmp_c}, {"v_mov_b32 sgpr", kp_p_mov_s}, {"v_cvt_f32_ubyte1", kp_p_cvtub}, {"v_and_b32 sdwa byte", kp_p_sdwa}, {"v_cmp_class_f32", kp_p_class}, {"v_cmpx (writes exec)", kp_p_cmpx}, {"readlane; s_nop; v_and sgpr", kp_p_readlane_nop}, {"v_cmp_e64 + s_bcnt1", kp_p_ballot}, {"s_add_u32 x1", kp_p_salu}, {"s_add/s_and/s_lshl/s_bcnt x4", kp_p_salu4}, {"s_cbranch_scc1 not taken", kp_p_branch_nt}, {"v_add_u32 + s_add_u32", kp_p_mix}
// LDS: 64-bit atomic max (what the fragment loop does), b128 reads, b32 reads; addresses conflict-free or random
__global__ __launch_bounds__(1024) void k_lds(unsigned long long *out, unsigned *sink, unsigned seed, int mode)
{
    __shared__ unsigned long long s[1024 * 2];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) s[i] = i;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned idx = mode == 1 ? (unsigned)((threadIdx.x * 2654435761u) >> 22) : (unsigned)((lane + wv * 64) & 1023);
    unsigned long long acc = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int k = 0; k < REP * 8; ++k) {
        if (mode <= 1) atomicMax(&s[idx], (unsigned long long)k << 32 | threadIdx.x);
        else if (mode == 2) { const uint4 v = reinterpret_cast<const uint4 *>(s)[(idx + k) & 1023]; acc += v.x + v.w; }
        else { acc += reinterpret_cast<const unsigned *>(s)[(idx + k) & 4095]; }
        if (mode == 1) idx = (idx * 5u + 1u) & 1023u;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (acc == 0x12345u) sink[0] = (unsigned)acc;
    if (lane == 0) out[blockIdx.x * 16 + wv] = t1 - t0;
}

typedef void (*kern_t)(unsigned long long *, unsigned *, unsigned);
int main(int argc, char **argv)
{
    const bool only_pat = argc > 1;
    struct E { const char *name; kern_t k; } tab[] = {
        {"v_fma_f32", k_fma}, {"v_add_f32", k_addf}, {"v_sub_f32", k_subf}, {"v_mul_f32", k_mulf}, {"v_pk_fma_f32", k_pkfma}, {"v_add_u32", k_addu}, {"v_and_b32", k_and},
        {"v_lshlrev_b32", k_shl}, {"v_lshl_add_u32", k_lshladd}, {"v_xad_u32", k_xad}, {"v_cndmask_b32", k_cndm}, {"v_cmp_lt_u32", k_cmp}, {"v_mul_lo_u32", k_mullo},
        {"v_mul_u32_u24", k_mul24}, {"v_mad_u32_u24", k_mad24}, {"v_cvt_f32_i32", k_cvtfi}, {"v_cvt_i32_f32", k_cvtif},
        {"v_rcp_f32", k_rcp}, {"v_sqrt_f32", k_sqrt}, {"v_min_u32", k_minu}, {"v_max3_i32", k_max3}, {"v_med3_f32", k_med3},
        {"v_bfe_u32", k_bfe}, {"v_mov_b32_dpp", k_dpp}, {"v_add_u32_dpp", k_adddpp}, {"v_bcnt_u32_b32", k_bcnt}, {"v_ffbh_u32", k_ffbh},
        {"v_lshlrev_b64", k_lshl64}, XTRA_TAB, PAT_TAB};
    setvbuf(stdout, nullptr, _IONBF, 0);
    unsigned long long *out; unsigned *sink;
    hipMalloc(&out, 512 * 16 * 8); hipMalloc(&sink, 4);
    std::vector<unsigned long long> h(512 * 16);
    const int wps[] = {1, 2, 4};   // (one workgroup per CU: more than 4 waves per SIMD would need two, and nothing pins those to one CU)
    bool seen_pat = false;
    printf("cycles per wave64 instruction (or per listed GROUP of instructions) per SIMD, %d per wave, at W waves per SIMD on every CU\n%-38s", REP * 64, "instruction");
    for (int w : wps) printf("  W=%d   ", w);
    printf("\n");
    for (const E &e : tab) {
        if (only_pat && !(e.k == kp_q_cmpnop1 || seen_pat)) continue;
        seen_pat = true;
        printf("%-38s", e.name);
        for (int w : wps) {
            const int nb = w > 4 ? 2 : 1, nt = w * 256 / nb;
            hipLaunchKernelGGL(e.k, dim3(256 * nb), dim3(nt), 0, 0, out, sink, 7u);
            hipLaunchKernelGGL(e.k, dim3(256 * nb), dim3(nt), 0, 0, out, sink, 7u);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
            double mx = 0; for (int b = 0; b < 256 * nb; ++b) for (int i = 0; i < nt / 64; ++i) mx = h[b * 16 + i] > mx ? (double)h[b * 16 + i] : mx;
            const double n = (double)REP * 64 * 1;
            printf(" %7.2f", mx / (n * w));
        }
        printf("\n");
    }
    const char *mn[] = {"ds_max_rtn_u64 own", "ds_max_rtn_u64 rand", "ds_read_b128", "ds_read_b32"};
    for (int mode = 0; mode < 4; ++mode) {
        printf("%-38s", mn[mode]);
        for (int w : wps) {
            const int nb = w > 4 ? 2 : 1, nt = w * 256 / nb;
            hipLaunchKernelGGL(k_lds, dim3(256 * nb), dim3(nt), 0, 0, out, sink, 7u, mode);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
            double mx = 0; for (int b = 0; b < 256 * nb; ++b) for (int i = 0; i < nt / 64; ++i) mx = h[b * 16 + i] > mx ? (double)h[b * 16 + i] : mx;
            printf(" %7.2f", mx / ((double)REP * 8 * w * 4));   // per CU (LDS is shared by the 4 SIMDs): cycles per wave-instruction per CU
        }
        printf("   (per CU)\n");
    }
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("clockRate attribute: %d kHz\n", clk);
    return 0;
}
