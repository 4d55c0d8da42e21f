// Issue-rate microbenchmark for the vector instructions the path tracer leans on (gfx950).
//
// Every kernel times, with the shader clock (s_memtime) inside the wave, a loop whose body is one
// inline-asm block of 32 independent copies of ONE instruction over 16 registers, and runs it with
// 1, 2, 4 and 8 waves per SIMD (256-thread blocks = one wave per SIMD; W blocks per CU).  Reported:
// shader cycles per wave-instruction per SIMD = median wave time / (instructions per wave x W), which
// does not depend on the clock the chip holds under load, plus the wall-clock rate for reference.
// A SIMD of this chip is 32 lanes wide: a wave64 instruction occupies it for 2 cycles, and one wave
// alone issues every 4, so the floor is 4 at one wave per SIMD and 2 from two waves up.
//
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip      Run: ./valu_rates
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));

#define R16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define REGS16(a) "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), \
                  "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])

// one line per register i; %16 / %17 are the two extra inputs of the block
#define I_FMA_VVV(i)   "v_fma_f32 %" #i ", %" #i ", %16, %17\n"
#define I_FMA_VSV(i)   "v_fma_f32 %" #i ", %" #i ", %16, %17\n"      /* %16 bound to an SGPR */
#define I_FMA_VCC(i)   "v_fma_f32 %" #i ", %" #i ", 0.5, 1.0\n"      /* inline constants */
#define I_FMA_SELF(i)  "v_fma_f32 %" #i ", %" #i ", %" #i ", %" #i "\n"
#define I_FMAC(i)      "v_fmac_f32 %" #i ", %16, %17\n"              /* VOP2 */
#define I_MUL_S(i)     "v_mul_f32 %" #i ", %16, %" #i "\n"           /* VOP2, SGPR source */
#define I_MUL_V(i)     "v_mul_f32 %" #i ", %16, %" #i "\n"
#define I_ADD_V(i)     "v_add_f32 %" #i ", %16, %" #i "\n"
#define I_MIN_V(i)     "v_min_f32 %" #i ", %16, %" #i "\n"
#define I_MAX_S(i)     "v_max_f32 %" #i ", %16, %" #i "\n"
#define I_MOV(i)       "v_mov_b32 %" #i ", %16\n"
#define I_CNDMASK(i)   "v_cndmask_b32 %" #i ", %" #i ", %16, vcc\n"
#define I_CNDMASK_S(i) "v_cndmask_b32 %" #i ", %" #i ", %16, %17\n"  /* VOP3, mask in an SGPR pair */
#define I_AND_OR(i)    "v_and_or_b32 %" #i ", %" #i ", %16, %17\n"
#define I_MED3U(i)     "v_med3_u32 %" #i ", %" #i ", %16, %17\n"
#define I_MINU(i)      "v_min_u32 %" #i ", %16, %" #i "\n"
#define I_CMP_VCC(i)   "v_cmp_lt_f32 vcc, %16, %" #i "\n"
#define I_CMP_SGPR(i)  "v_cmp_lt_f32 s[20:21], %16, %" #i "\n"
#define I_RCP(i)       "v_rcp_f32 %" #i ", %" #i "\n"
#define I_RSQ(i)       "v_rsq_f32 %" #i ", %" #i "\n"
#define I_SQRT(i)      "v_sqrt_f32 %" #i ", %" #i "\n"
#define I_MUL_LO(i)    "v_mul_lo_u32 %" #i ", %" #i ", %16\n"
#define I_MUL_HI(i)    "v_mul_hi_u32 %" #i ", %" #i ", %16\n"
#define I_MAD_U24(i)   "v_mad_u32_u24 %" #i ", %" #i ", %16, %17\n"
#define I_XOR(i)       "v_xor_b32 %" #i ", %16, %" #i "\n"
#define I_LSHL_ADD(i)  "v_lshl_add_u32 %" #i ", %" #i ", 3, %16\n"
#define I_CVT_U(i)     "v_cvt_f32_u32 %" #i ", %" #i "\n"
#define I_READLANE(i)  "v_readlane_b32 s20, %" #i ", 3\n"
#define I_PK_FMA(i)    "v_pk_fma_f32 %" #i ", %" #i ", %16, %17\n"
#define I_PK_MUL(i)    "v_pk_mul_f32 %" #i ", %" #i ", %16\n"
#define I_PK_ADD(i)    "v_pk_add_f32 %" #i ", %" #i ", %16\n"
#define I_FMA64(i)     "v_fma_f64 %" #i ", %" #i ", %16, %17\n"
#define I_MAD64(i)     "v_mad_u64_u32 %" #i ", vcc, %16, %17, %" #i "\n"
#define I_CNDMASK_E64V(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %16, vcc\n"
#define I_CMP_CND_VCC(i) "v_cmp_lt_f32 vcc, %16, %" #i "\nv_cndmask_b32 %" #i ", %" #i ", %17, vcc\n"
#define I_CMP_CND_S(i)   "v_cmp_lt_f32 s[20:21], %16, %" #i "\nv_cndmask_b32 %" #i ", %" #i ", %17, s[20:21]\n"
#define I_AND(i)       "v_and_b32 %" #i ", %16, %" #i "\n"
#define I_OR(i)        "v_or_b32 %" #i ", %16, %" #i "\n"
#define I_LSHL(i)      "v_lshlrev_b32 %" #i ", 3, %" #i "\n"
#define I_ADD_U(i)     "v_add_u32 %" #i ", %16, %" #i "\n"
#define I_SUB_F(i)     "v_sub_f32 %" #i ", %16, %" #i "\n"
#define I_MAX3(i)      "v_max3_f32 %" #i ", %" #i ", %16, %17\n"
#define I_FMA_MOD(i)   "v_fma_f32 %" #i ", -%" #i ", |%16|, -%17\n"
#define I_BFE(i)       "v_bfe_u32 %" #i ", %" #i ", 3, 24\n"
#define I_EXP(i)       "v_exp_f32 %" #i ", %" #i "\n"
#define I_BFI(i)       "v_bfi_b32 %" #i ", %16, %" #i ", %17\n"
#define I_ADD3(i)      "v_add3_u32 %" #i ", %" #i ", %16, %17\n"
#define I_OR3(i)       "v_or3_b32 %" #i ", %" #i ", %16, %17\n"
#define I_XAD(i)       "v_xad_u32 %" #i ", %" #i ", %16, %17\n"
#define I_LSHR(i)      "v_lshrrev_b32 %" #i ", 3, %" #i "\n"
#define I_PERM(i)      "v_perm_b32 %" #i ", %" #i ", %16, %17\n"
#define I_ALIGNBIT(i)  "v_alignbit_b32 %" #i ", %" #i ", %16, 7\n"
#define I_MAXU(i)      "v_max_u32 %" #i ", %16, %" #i "\n"
#define I_SUBU(i)      "v_sub_u32 %" #i ", %16, %" #i "\n"
#define I_MED3F(i)     "v_med3_f32 %" #i ", %" #i ", %16, %17\n"
#define I_FLOOR(i)     "v_floor_f32 %" #i ", %" #i "\n"
#define I_FRACT(i)     "v_fract_f32 %" #i ", %" #i "\n"
#define I_CVT_UB(i)    "v_cvt_f32_ubyte1 %" #i ", %" #i "\n"
#define I_CVT_FU(i)    "v_cvt_u32_f32 %" #i ", %" #i "\n"
#define I_LDEXP(i)     "v_ldexp_f32 %" #i ", %" #i ", 3\n"
#define I_ADD_DPP(i)   "v_add_f32_dpp %" #i ", %" #i ", %16 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_MUL_CLAMP(i) "v_mul_f32_e64 %" #i ", %16, %" #i " clamp\n"
#define I_AND_SDWA(i)  "v_and_b32_sdwa %" #i ", %16, %" #i " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
// select patterns (per GROUP as written)
#define I_CND32_FMA(i)   "v_cndmask_b32 %" #i ", %" #i ", %16, vcc\nv_fma_f32 %" #i ", %" #i ", %16, %17\n"
#define I_CND32_2FMA(i)  "v_cndmask_b32 %" #i ", %" #i ", %16, vcc\nv_fma_f32 %" #i ", %" #i ", %16, %17\nv_fma_f32 %" #i ", %" #i ", %16, %17\n"
#define I_CMP_3CND32(i)  "v_cmp_lt_f32 vcc, %16, %" #i "\nv_cndmask_b32 %" #i ", %" #i ", %17, vcc\nv_cndmask_b32 %" #i ", %" #i ", %16, vcc\nv_cndmask_b32 %" #i ", %" #i ", %17, vcc\n"
#define I_CMP_3CND64(i)  "v_cmp_lt_f32 s[20:21], %16, %" #i "\nv_cndmask_b32 %" #i ", %" #i ", %17, s[20:21]\nv_cndmask_b32 %" #i ", %" #i ", %16, s[20:21]\nv_cndmask_b32 %" #i ", %" #i ", %17, s[20:21]\n"
#define I_CMP_3CND64V(i) "v_cmp_lt_f32 vcc, %16, %" #i "\nv_cndmask_b32_e64 %" #i ", %" #i ", %17, vcc\nv_cndmask_b32_e64 %" #i ", %" #i ", %16, vcc\nv_cndmask_b32_e64 %" #i ", %" #i ", %17, vcc\n"
#define I_FMA_WAITCNT(i) "v_fma_f32 %" #i ", %" #i ", %16, %17\ns_waitcnt lgkmcnt(0)\n"
#define I_FMA_NOP(i)     "v_fma_f32 %" #i ", %" #i ", %16, %17\ns_nop 0\n"
#define I_FMA_SMOV(i)    "v_fma_f32 %" #i ", %" #i ", %16, %17\ns_mov_b32 s20, s21\n"
#define I_FMA_SAVEEXEC(i) "v_fma_f32 %" #i ", %" #i ", %16, %17\ns_and_saveexec_b64 s[20:21], exec\n"
// mixes
#define I_FMA_SALU(i)  "v_fma_f32 %" #i ", %" #i ", %16, %17\ns_add_u32 s20, s20, 1\n"
#define I_FMA_2SALU(i) "v_fma_f32 %" #i ", %" #i ", %16, %17\ns_add_u32 s20, s20, 1\ns_and_b32 s21, s21, s20\n"
#define I_FMA_TRANS(i) "v_fma_f32 %" #i ", %" #i ", %16, %17\n"      /* + one v_rcp per 4, see kernel */
// dependent chain: every instruction reads the previous one's result
#define I_FMA_DEP(i)   "v_fma_f32 %0, %0, %16, %17\n"

enum Op { FMA_VVV, FMA_VSV, FMA_VCC, FMA_SELF, FMAC, MUL_S, MUL_V, ADD_V, MIN_V, MAX_S, MOV, CNDMASK, CNDMASK_S, AND_OR, MED3U, MINU,
          CMP_VCC, CMP_SGPR, RCP, RSQ, SQRT, MUL_LO, MUL_HI, MAD_U24, XOR, LSHL_ADD, CVT_U, READLANE, PK_FMA, PK_MUL, PK_ADD,
          FMA64, MAD64, FMA_SALU, FMA_2SALU, FMA_DEP, CNDMASK_VCCSET, CNDMASK_E64V, CMP_CND_VCC, CMP_CND_S, AND_, OR_, LSHL, ADD_U, SUB_F, MAX3, FMA_MOD, BFE, EXP, CND32_FMA, CND32_2FMA, CMP_3CND32, CMP_3CND64, CMP_3CND64V, FMA_WAITCNT, FMA_NOP, FMA_SMOV, FMA_SAVEEXEC,
          BFI, ADD3, OR3, XAD, LSHR, PERM, ALIGNBIT, MAXU, SUBU, MED3F, FLOOR, FRACT, CVT_UB, CVT_FU, LDEXP, ADD_DPP, MUL_CLAMP, AND_SDWA, N_OPS };

#define BODY32(M, extra_in...) asm volatile(R16(M) R16(M) : REGS16(a) : extra_in : "vcc", "scc", "s20", "s21", "s22", "s23")

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned long long *stamps, float *sink, float seed, int iters) {
    const float s = seed * 0.999f, t = seed * 1e-3f;
    unsigned long long t0 = 0, t1 = 0;
    float r = 0;
    if constexpr (OP == PK_FMA || OP == PK_MUL || OP == PK_ADD) {
        v2f a[16];
        for (int i = 0; i < 16; ++i) a[i] = v2f{seed + i + threadIdx.x * 1e-3f, seed * 0.5f + i};
        const v2f ps = {s, s}, pt = {t, t};
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
            if constexpr (OP == PK_FMA) BODY32(I_PK_FMA, "v"(ps), "v"(pt));
            if constexpr (OP == PK_MUL) BODY32(I_PK_MUL, "v"(ps), "v"(pt));
            if constexpr (OP == PK_ADD) BODY32(I_PK_ADD, "v"(pt), "v"(ps));
        }
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 16; ++i) r += a[i].x + a[i].y;
    } else if constexpr (OP == FMA64) {
        double a[16];
        for (int i = 0; i < 16; ++i) a[i] = seed + i + threadIdx.x * 1e-3;
        const double ds = s, dt = t;
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) BODY32(I_FMA64, "v"(ds), "v"(dt));
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 16; ++i) r += (float)a[i];
    } else if constexpr (OP == MAD64) {
        unsigned long long a[16];
        for (int i = 0; i < 16; ++i) a[i] = (unsigned long long)(seed * 977.0f) + i + threadIdx.x;
        const unsigned m0 = 0xD2511F53u + threadIdx.x, m1 = 0xCD9E8D57u ^ threadIdx.x;
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) BODY32(I_MAD64, "v"(m0), "v"(m1));
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 16; ++i) r += (float)a[i];
    } else if constexpr (OP == MUL_LO || OP == MUL_HI || OP == MAD_U24 || OP == XOR || OP == LSHL_ADD || OP == AND_OR || OP == MED3U ||
                         OP == MINU || OP == CVT_U || OP == AND_ || OP == OR_ || OP == LSHL || OP == ADD_U || OP == BFE || OP == BFI || OP == ADD3 ||
                         OP == OR3 || OP == XAD || OP == LSHR || OP == PERM || OP == ALIGNBIT || OP == MAXU || OP == SUBU || OP == CVT_UB || OP == AND_SDWA) {
        unsigned a[16];
        for (int i = 0; i < 16; ++i) a[i] = ((unsigned)(seed * 977.0f) + 31u * i + threadIdx.x) | 1u;
        const unsigned m0 = 0xD2511F53u + threadIdx.x, m1 = 0x00FFF0F0u ^ threadIdx.x;
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
            if constexpr (OP == MUL_LO) BODY32(I_MUL_LO, "v"(m0), "v"(m1));
            if constexpr (OP == MUL_HI) BODY32(I_MUL_HI, "v"(m0), "v"(m1));
            if constexpr (OP == MAD_U24) BODY32(I_MAD_U24, "v"(m0), "v"(m1));
            if constexpr (OP == XOR) BODY32(I_XOR, "v"(m0), "v"(m1));
            if constexpr (OP == LSHL_ADD) BODY32(I_LSHL_ADD, "v"(m0), "v"(m1));
            if constexpr (OP == AND_OR) BODY32(I_AND_OR, "v"(m0), "s"(__builtin_amdgcn_readfirstlane(m1)));
            if constexpr (OP == MED3U) BODY32(I_MED3U, "v"(m0), "v"(m1));
            if constexpr (OP == MINU) BODY32(I_MINU, "v"(m0), "v"(m1));
            if constexpr (OP == CVT_U) BODY32(I_CVT_U, "v"(m0), "v"(m1));
            if constexpr (OP == AND_) BODY32(I_AND, "v"(m0), "v"(m1));
            if constexpr (OP == OR_) BODY32(I_OR, "v"(m0), "v"(m1));
            if constexpr (OP == LSHL) BODY32(I_LSHL, "v"(m0), "v"(m1));
            if constexpr (OP == ADD_U) BODY32(I_ADD_U, "v"(m0), "v"(m1));
            if constexpr (OP == BFE) BODY32(I_BFE, "v"(m0), "v"(m1));
            if constexpr (OP == BFI) BODY32(I_BFI, "v"(m0), "v"(m1));
            if constexpr (OP == ADD3) BODY32(I_ADD3, "v"(m0), "v"(m1));
            if constexpr (OP == OR3) BODY32(I_OR3, "v"(m0), "v"(m1));
            if constexpr (OP == XAD) BODY32(I_XAD, "v"(m0), "v"(m1));
            if constexpr (OP == LSHR) BODY32(I_LSHR, "v"(m0), "v"(m1));
            if constexpr (OP == PERM) BODY32(I_PERM, "v"(m0), "v"(m1));
            if constexpr (OP == ALIGNBIT) BODY32(I_ALIGNBIT, "v"(m0), "v"(m1));
            if constexpr (OP == MAXU) BODY32(I_MAXU, "v"(m0), "v"(m1));
            if constexpr (OP == SUBU) BODY32(I_SUBU, "v"(m0), "v"(m1));
            if constexpr (OP == CVT_UB) BODY32(I_CVT_UB, "v"(m0), "v"(m1));
            if constexpr (OP == AND_SDWA) BODY32(I_AND_SDWA, "v"(m0), "v"(m1));
        }
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 16; ++i) r += (float)a[i];
    } else {
        float a[16];
        for (int i = 0; i < 16; ++i) a[i] = seed + i + threadIdx.x * 1e-3f;
        const float ss = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s)));
        const unsigned long long mask = 0x5555AAAA3333CCCCull;
        if constexpr (OP == CNDMASK_VCCSET || OP == CNDMASK_E64V) asm volatile("s_mov_b64 vcc, %0" :: "s"(mask) : "vcc");
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
            if constexpr (OP == CNDMASK_VCCSET) asm volatile(R16(I_CNDMASK) R16(I_CNDMASK) : REGS16(a) : "v"(s), "v"(t) : "scc", "s20", "s21");
            if constexpr (OP == CNDMASK_E64V) asm volatile(R16(I_CNDMASK_E64V) R16(I_CNDMASK_E64V) : REGS16(a) : "v"(s), "v"(t) : "scc", "s20", "s21");
            if constexpr (OP == CMP_CND_VCC) BODY32(I_CMP_CND_VCC, "v"(s), "v"(t));
            if constexpr (OP == CMP_CND_S) BODY32(I_CMP_CND_S, "v"(s), "v"(t));
            if constexpr (OP == CND32_FMA) BODY32(I_CND32_FMA, "v"(s), "v"(t));
            if constexpr (OP == CND32_2FMA) BODY32(I_CND32_2FMA, "v"(s), "v"(t));
            if constexpr (OP == CMP_3CND32) BODY32(I_CMP_3CND32, "v"(s), "v"(t));
            if constexpr (OP == CMP_3CND64) BODY32(I_CMP_3CND64, "v"(s), "v"(t));
            if constexpr (OP == CMP_3CND64V) BODY32(I_CMP_3CND64V, "v"(s), "v"(t));
            if constexpr (OP == FMA_WAITCNT) BODY32(I_FMA_WAITCNT, "v"(s), "v"(t));
            if constexpr (OP == FMA_NOP) BODY32(I_FMA_NOP, "v"(s), "v"(t));
            if constexpr (OP == FMA_SMOV) BODY32(I_FMA_SMOV, "v"(s), "v"(t));
            if constexpr (OP == FMA_SAVEEXEC) BODY32(I_FMA_SAVEEXEC, "v"(s), "v"(t));
            if constexpr (OP == SUB_F) BODY32(I_SUB_F, "v"(s), "v"(t));
            if constexpr (OP == MAX3) BODY32(I_MAX3, "v"(s), "v"(t));
            if constexpr (OP == FMA_MOD) BODY32(I_FMA_MOD, "v"(s), "v"(t));
            if constexpr (OP == EXP) BODY32(I_EXP, "v"(s), "v"(t));
            if constexpr (OP == FMA_VVV) BODY32(I_FMA_VVV, "v"(s), "v"(t));
            if constexpr (OP == FMA_VSV) BODY32(I_FMA_VSV, "s"(ss), "v"(t));
            if constexpr (OP == FMA_VCC) BODY32(I_FMA_VCC, "v"(s), "v"(t));
            if constexpr (OP == FMA_SELF) BODY32(I_FMA_SELF, "v"(s), "v"(t));
            if constexpr (OP == FMAC) BODY32(I_FMAC, "v"(s), "v"(t));
            if constexpr (OP == MUL_S) BODY32(I_MUL_S, "s"(ss), "v"(t));
            if constexpr (OP == MUL_V) BODY32(I_MUL_V, "v"(s), "v"(t));
            if constexpr (OP == ADD_V) BODY32(I_ADD_V, "v"(t), "v"(s));
            if constexpr (OP == MIN_V) BODY32(I_MIN_V, "v"(s), "v"(t));
            if constexpr (OP == MAX_S) BODY32(I_MAX_S, "s"(ss), "v"(t));
            if constexpr (OP == MOV) BODY32(I_MOV, "v"(s), "v"(t));
            if constexpr (OP == CNDMASK) BODY32(I_CNDMASK, "v"(s), "v"(t));
            if constexpr (OP == CNDMASK_S) BODY32(I_CNDMASK_S, "v"(s), "s"(mask));
            if constexpr (OP == CMP_VCC) BODY32(I_CMP_VCC, "v"(s), "v"(t));
            if constexpr (OP == CMP_SGPR) BODY32(I_CMP_SGPR, "v"(s), "v"(t));
            if constexpr (OP == RCP) BODY32(I_RCP, "v"(s), "v"(t));
            if constexpr (OP == RSQ) BODY32(I_RSQ, "v"(s), "v"(t));
            if constexpr (OP == SQRT) BODY32(I_SQRT, "v"(s), "v"(t));
            if constexpr (OP == READLANE) BODY32(I_READLANE, "v"(s), "v"(t));
            if constexpr (OP == FMA_SALU) BODY32(I_FMA_SALU, "v"(s), "v"(t));
            if constexpr (OP == FMA_2SALU) BODY32(I_FMA_2SALU, "v"(s), "v"(t));
            if constexpr (OP == FMA_DEP) BODY32(I_FMA_DEP, "v"(s), "v"(t));
            if constexpr (OP == MED3F) BODY32(I_MED3F, "v"(s), "v"(t));
            if constexpr (OP == FLOOR) BODY32(I_FLOOR, "v"(s), "v"(t));
            if constexpr (OP == FRACT) BODY32(I_FRACT, "v"(s), "v"(t));
            if constexpr (OP == CVT_FU) BODY32(I_CVT_FU, "v"(s), "v"(t));
            if constexpr (OP == LDEXP) BODY32(I_LDEXP, "v"(s), "v"(t));
            if constexpr (OP == ADD_DPP) BODY32(I_ADD_DPP, "v"(s), "v"(t));
            if constexpr (OP == MUL_CLAMP) BODY32(I_MUL_CLAMP, "v"(s), "v"(t));
        }
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 16; ++i) r += a[i];
    }
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) { stamps[2 * wave] = t0; stamps[2 * wave + 1] = t1; }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

struct Row { const char *name; double cyc[4]; double ginst_s; };

template <int OP>
Row run(const char *name, int n_cu, unsigned long long *d_stamps, float *d_sink) {
    Row row{name, {0, 0, 0, 0}, 0};
    const int iters = 2048;
    const int waves_per_simd[4] = {1, 2, 4, 8};
    std::vector<unsigned long long> h;
    for (int w = 0; w < 4; ++w) {
        const int blocks = n_cu * waves_per_simd[w];
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_stamps, d_sink, 1.0001f, 64);   // warm-up
        CHECK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_stamps, d_sink, 1.0001f, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const int n_waves = blocks * 4;
        h.resize(2 * (size_t)n_waves);
        CHECK(hipMemcpy(h.data(), d_stamps, h.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> dt(n_waves);
        for (int i = 0; i < n_waves; ++i) dt[i] = (double)(h[2 * i + 1] - h[2 * i]);
        std::nth_element(dt.begin(), dt.begin() + n_waves / 2, dt.end());
        const double per_wave = (double)iters * 32.0;
        row.cyc[w] = dt[n_waves / 2] / (per_wave * waves_per_simd[w]);
        if (w == 3) row.ginst_s = per_wave * n_waves / (ms * 1e-3) * 1e-9;
        CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    }
    printf("%-44s %7.2f %7.2f %7.2f %7.2f   %9.1f\n", name, row.cyc[0], row.cyc[1], row.cyc[2], row.cyc[3], row.ginst_s);
    fflush(stdout);
    return row;
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    printf("device %s  CUs %d  max clock %d kHz\n", prop.gcnArchName, n_cu, prop.clockRate);
    printf("shader cycles per wave64 instruction per SIMD (median wave; s_memtime), by waves per SIMD; last column: G wave-instructions/s, whole chip, 8 waves/SIMD\n");
    printf("%-44s %7s %7s %7s %7s   %9s\n", "instruction (32 independent copies per trip)", "1 w", "2 w", "4 w", "8 w", "Ginst/s");
    unsigned long long *d_stamps; float *d_sink;
    CHECK(hipMalloc(&d_stamps, sizeof(unsigned long long) * 2 * (size_t)n_cu * 8 * 4));
    CHECK(hipMalloc(&d_sink, sizeof(float) * (size_t)n_cu * 8 * 256));
    run<FMA_VVV>("v_fma_f32 v,v,v,v (2 shared sources)", n_cu, d_stamps, d_sink);
    run<FMA_VSV>("v_fma_f32 v,v,s,v", n_cu, d_stamps, d_sink);
    run<FMA_VCC>("v_fma_f32 v,v,0.5,1.0 (inline constants)", n_cu, d_stamps, d_sink);
    run<FMA_SELF>("v_fma_f32 v,v,v,v (one register)", n_cu, d_stamps, d_sink);
    run<FMAC>("v_fmac_f32 v,v,v (VOP2)", n_cu, d_stamps, d_sink);
    run<MUL_V>("v_mul_f32 v,v,v", n_cu, d_stamps, d_sink);
    run<MUL_S>("v_mul_f32 v,s,v", n_cu, d_stamps, d_sink);
    run<ADD_V>("v_add_f32 v,v,v", n_cu, d_stamps, d_sink);
    run<MIN_V>("v_min_f32 v,v,v", n_cu, d_stamps, d_sink);
    run<MAX_S>("v_max_f32 v,s,v", n_cu, d_stamps, d_sink);
    run<MOV>("v_mov_b32 v,v", n_cu, d_stamps, d_sink);
    run<CNDMASK>("v_cndmask_b32 v,v,v,vcc", n_cu, d_stamps, d_sink);
    run<CNDMASK_S>("v_cndmask_b32 v,v,v,s[2] (VOP3)", n_cu, d_stamps, d_sink);
    run<CMP_VCC>("v_cmp_lt_f32 vcc,v,v", n_cu, d_stamps, d_sink);
    run<CMP_SGPR>("v_cmp_lt_f32 s[2],v,v (VOP3)", n_cu, d_stamps, d_sink);
    run<AND_OR>("v_and_or_b32 v,v,v,s", n_cu, d_stamps, d_sink);
    run<MED3U>("v_med3_u32 v,v,v,v", n_cu, d_stamps, d_sink);
    run<MINU>("v_min_u32 v,v,v", n_cu, d_stamps, d_sink);
    run<XOR>("v_xor_b32 v,v,v", n_cu, d_stamps, d_sink);
    run<LSHL_ADD>("v_lshl_add_u32 v,v,3,v", n_cu, d_stamps, d_sink);
    run<CVT_U>("v_cvt_f32_u32 v,v", n_cu, d_stamps, d_sink);
    run<MUL_LO>("v_mul_lo_u32 v,v,v", n_cu, d_stamps, d_sink);
    run<MUL_HI>("v_mul_hi_u32 v,v,v", n_cu, d_stamps, d_sink);
    run<MAD_U24>("v_mad_u32_u24 v,v,v,v", n_cu, d_stamps, d_sink);
    run<MAD64>("v_mad_u64_u32 v[2],vcc,v,v,v[2]", n_cu, d_stamps, d_sink);
    run<RCP>("v_rcp_f32 v,v", n_cu, d_stamps, d_sink);
    run<RSQ>("v_rsq_f32 v,v", n_cu, d_stamps, d_sink);
    run<SQRT>("v_sqrt_f32 v,v", n_cu, d_stamps, d_sink);
    run<READLANE>("v_readlane_b32 s,v,3", n_cu, d_stamps, d_sink);
    run<PK_FMA>("v_pk_fma_f32 v[2],v[2],v[2],v[2]", n_cu, d_stamps, d_sink);
    run<PK_MUL>("v_pk_mul_f32 v[2],v[2],v[2]", n_cu, d_stamps, d_sink);
    run<PK_ADD>("v_pk_add_f32 v[2],v[2],v[2]", n_cu, d_stamps, d_sink);
    run<FMA64>("v_fma_f64 v[2],v[2],v[2],v[2]", n_cu, d_stamps, d_sink);
    run<CNDMASK_VCCSET>("v_cndmask_b32 v,v,v,vcc (vcc set before loop)", n_cu, d_stamps, d_sink);
    run<CNDMASK_E64V>("v_cndmask_b32_e64 v,v,v,vcc", n_cu, d_stamps, d_sink);
    run<CMP_CND_VCC>("v_cmp vcc + v_cndmask vcc (per PAIR)", n_cu, d_stamps, d_sink);
    run<CMP_CND_S>("v_cmp s[2] + v_cndmask s[2] (per PAIR)", n_cu, d_stamps, d_sink);
    run<AND_>("v_and_b32 v,v,v", n_cu, d_stamps, d_sink);
    run<OR_>("v_or_b32 v,v,v", n_cu, d_stamps, d_sink);
    run<LSHL>("v_lshlrev_b32 v,3,v", n_cu, d_stamps, d_sink);
    run<ADD_U>("v_add_u32 v,v,v", n_cu, d_stamps, d_sink);
    run<BFE>("v_bfe_u32 v,v,3,24", n_cu, d_stamps, d_sink);
    run<SUB_F>("v_sub_f32 v,v,v", n_cu, d_stamps, d_sink);
    run<MAX3>("v_max3_f32 v,v,v,v", n_cu, d_stamps, d_sink);
    run<FMA_MOD>("v_fma_f32 v,-v,|v|,-v (modifiers)", n_cu, d_stamps, d_sink);
    run<EXP>("v_exp_f32 v,v", n_cu, d_stamps, d_sink);
    run<CND32_FMA>("[v_cndmask e32 vcc, v_fma] (per GROUP of 2)", n_cu, d_stamps, d_sink);
    run<CND32_2FMA>("[v_cndmask e32 vcc, 2 v_fma] (per GROUP of 3)", n_cu, d_stamps, d_sink);
    run<CMP_3CND32>("[v_cmp vcc, 3 v_cndmask e32 vcc] (per GROUP of 4)", n_cu, d_stamps, d_sink);
    run<CMP_3CND64>("[v_cmp s[2], 3 v_cndmask s[2]] (per GROUP of 4)", n_cu, d_stamps, d_sink);
    run<CMP_3CND64V>("[v_cmp vcc, 3 v_cndmask_e64 vcc] (per GROUP of 4)", n_cu, d_stamps, d_sink);
    run<FMA_WAITCNT>("[v_fma, s_waitcnt lgkmcnt(0)] (per GROUP of 2)", n_cu, d_stamps, d_sink);
    run<FMA_NOP>("[v_fma, s_nop 0] (per GROUP of 2)", n_cu, d_stamps, d_sink);
    run<FMA_SMOV>("[v_fma, s_mov_b32] (per GROUP of 2)", n_cu, d_stamps, d_sink);
    run<FMA_SAVEEXEC>("[v_fma, s_and_saveexec_b64] (per GROUP of 2)", n_cu, d_stamps, d_sink);
    run<FMA_SALU>("v_fma_f32 + 1 s_add_u32 (per VALU op)", n_cu, d_stamps, d_sink);
    run<FMA_2SALU>("v_fma_f32 + 2 SALU (per VALU op)", n_cu, d_stamps, d_sink);
    run<FMA_DEP>("v_fma_f32 dependent chain (latency)", n_cu, d_stamps, d_sink);
    run<BFI>("v_bfi_b32 v,v,v,v", n_cu, d_stamps, d_sink);
    run<ADD3>("v_add3_u32 v,v,v,v", n_cu, d_stamps, d_sink);
    run<OR3>("v_or3_b32 v,v,v,v", n_cu, d_stamps, d_sink);
    run<XAD>("v_xad_u32 v,v,v,v", n_cu, d_stamps, d_sink);
    run<LSHR>("v_lshrrev_b32 v,3,v", n_cu, d_stamps, d_sink);
    run<PERM>("v_perm_b32 v,v,v,v", n_cu, d_stamps, d_sink);
    run<ALIGNBIT>("v_alignbit_b32 v,v,v,7", n_cu, d_stamps, d_sink);
    run<MAXU>("v_max_u32 v,v,v", n_cu, d_stamps, d_sink);
    run<SUBU>("v_sub_u32 v,v,v", n_cu, d_stamps, d_sink);
    run<MED3F>("v_med3_f32 v,v,v,v", n_cu, d_stamps, d_sink);
    run<FLOOR>("v_floor_f32 v,v", n_cu, d_stamps, d_sink);
    run<FRACT>("v_fract_f32 v,v", n_cu, d_stamps, d_sink);
    run<CVT_UB>("v_cvt_f32_ubyte1 v,v", n_cu, d_stamps, d_sink);
    run<CVT_FU>("v_cvt_u32_f32 v,v", n_cu, d_stamps, d_sink);
    run<LDEXP>("v_ldexp_f32 v,v,3", n_cu, d_stamps, d_sink);
    run<ADD_DPP>("v_add_f32_dpp v,v,v row_shr:1", n_cu, d_stamps, d_sink);
    run<MUL_CLAMP>("v_mul_f32_e64 v,v,v clamp", n_cu, d_stamps, d_sink);
    run<AND_SDWA>("v_and_b32_sdwa v,v,v src0_sel:BYTE_1", n_cu, d_stamps, d_sink);
    return 0;
}
