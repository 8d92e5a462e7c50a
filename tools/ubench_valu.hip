// ubench_valu.hip -- VALU issue rates on the GPU at hand, for k_blend's instruction-mix roofline.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 -o /tmp/ubench_valu tools/ubench_valu.hip && /tmp/ubench_valu
// Every kernel runs ITER x 32 independent instructions of one kind per wave, 8 waves per SIMD on every SIMD,
// and reports wave-instructions per microsecond per SIMD (= 1 / issue interval).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
#define ITER 4096
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define REP8(X) X X X X X X X X
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)

__global__ void __launch_bounds__(256) k_pk_fma(float* out, float s)
{
    v2f a0 = {s, s + 1}, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f, a4 = a0 + 4.0f, a5 = a0 + 5.0f, a6 = a0 + 6.0f, a7 = a0 + 7.0f;
    const v2f m = {1.0000001f, 0.9999999f}, c = {1e-9f, -1e-9f};
    for (int i = 0; i < ITER; ++i) {
#define X asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n" \
                       "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n" \
                       : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        X X X X
#undef X
    }
    v2f r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (r.x + r.y == 12345.678f) out[threadIdx.x] = r.x;
}

#define SCALAR_KERNEL(NAME, INSN)                                                                                         \
__global__ void __launch_bounds__(256) NAME(float* out, float s)                                                          \
{                                                                                                                         \
    float a0 = s, a1 = s + 1, a2 = s + 2, a3 = s + 3, a4 = s + 4, a5 = s + 5, a6 = s + 6, a7 = s + 7;                      \
    const float m = 1.0000001f, c = 1e-9f;                                                                                \
    for (int i = 0; i < ITER; ++i) {                                                                                      \
        REP4_(asm volatile(INSN(0) INSN(1) INSN(2) INSN(3) INSN(4) INSN(5) INSN(6) INSN(7)                                \
              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)        \
    }                                                                                                                     \
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                                                      \
    if (r == 12345.678f) out[threadIdx.x] = r;                                                                            \
}
#define REP4_(X) X X X X
#define I_FMA(k) "v_fma_f32 %" #k ", %" #k ", %8, %9\n"
#define I_MUL(k) "v_mul_f32 %" #k ", %" #k ", %8\n"
#define I_MAX(k) "v_max_f32 %" #k ", |%" #k "|, |%8|\n"
#define I_RNDNE(k) "v_rndne_f32 %" #k ", %" #k "\n"
#define I_CVT(k) "v_cvt_i32_f32 %" #k ", %" #k "\n"
#define I_LSHLADD(k) "v_lshl_add_u32 %" #k ", %" #k ", 1, %8\n"
#define I_MOV(k) "v_mov_b32 %" #k ", %8\n"
#define I_CMP_CND(k) "v_cmp_le_f32 vcc, %" #k ", %8\n v_cndmask_b32 %" #k ", %9, %" #k ", vcc\n"
#define I_CMP_S(k) "v_cmp_le_f32 s[20:21], %" #k ", %8\n"
#define I_MAX_E32(k) "v_max_f32 %" #k ", %" #k ", %8\n"
#define I_MIN_E32(k) "v_min_f32 %" #k ", %" #k ", %8\n"
#define I_SUB(k) "v_sub_f32 %" #k ", %8, %" #k "\n"
#define I_FMAC(k) "v_fmac_f32 %" #k ", %8, %9\n"
#define I_MED3(k) "v_med3_f32 %" #k ", %" #k ", %8, %9\n"
#define I_MAX3(k) "v_max3_f32 %" #k ", %" #k ", %8, %9\n"
#define I_EXP(k) "v_exp_f32 %" #k ", %" #k "\n"
#define I_LDEXP(k) "v_ldexp_f32 %" #k ", %" #k ", %8\n"
#define I_ADDU(k) "v_add_u32 %" #k ", %" #k ", %8\n"
#define I_LSHLREV(k) "v_lshlrev_b32 %" #k ", 1, %" #k "\n"
#define I_AND(k) "v_and_b32 %" #k ", %" #k ", %8\n"
#define I_CMP_VCC(k) "v_cmp_le_f32 vcc, %" #k ", %8\n"
#define I_CND(k) "v_cndmask_b32 %" #k ", %9, %" #k ", vcc\n"
#define I_CMPX(k) "v_cmp_class_f32 vcc, %" #k ", %8\n"
#define I_DPP(k) "v_mov_b32_dpp %" #k ", %" #k " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define I_BFI(k) "v_bfi_b32 %" #k ", %8, %" #k ", %9\n"
#define I_MULLEGACY(k) "v_mul_legacy_f32 %" #k ", %" #k ", %8\n"
SCALAR_KERNEL(k_fma, I_FMA)
SCALAR_KERNEL(k_mul, I_MUL)
SCALAR_KERNEL(k_max, I_MAX)
SCALAR_KERNEL(k_rndne, I_RNDNE)
SCALAR_KERNEL(k_cvt, I_CVT)
SCALAR_KERNEL(k_lshladd, I_LSHLADD)
SCALAR_KERNEL(k_mov, I_MOV)
SCALAR_KERNEL(k_cmp_cnd, I_CMP_CND)
SCALAR_KERNEL(k_max_e32, I_MAX_E32)
SCALAR_KERNEL(k_min_e32, I_MIN_E32)
SCALAR_KERNEL(k_sub, I_SUB)
SCALAR_KERNEL(k_fmac, I_FMAC)
SCALAR_KERNEL(k_med3, I_MED3)
SCALAR_KERNEL(k_max3, I_MAX3)
SCALAR_KERNEL(k_exp, I_EXP)
SCALAR_KERNEL(k_ldexp, I_LDEXP)
SCALAR_KERNEL(k_addu, I_ADDU)
SCALAR_KERNEL(k_lshlrev, I_LSHLREV)
SCALAR_KERNEL(k_and, I_AND)
SCALAR_KERNEL(k_cmp_vcc, I_CMP_VCC)
SCALAR_KERNEL(k_cnd, I_CND)
SCALAR_KERNEL(k_cmp_class, I_CMPX)
SCALAR_KERNEL(k_dpp, I_DPP)
SCALAR_KERNEL(k_bfi, I_BFI)
SCALAR_KERNEL(k_mul_legacy, I_MULLEGACY)

#define PK_KERNEL(NAME, INSN)                                                                                             \
__global__ void __launch_bounds__(256) NAME(float* out, float s)                                                          \
{                                                                                                                         \
    v2f a0 = {s, s + 1}, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f, a4 = a0 + 4.0f, a5 = a0 + 5.0f, a6 = a0 + 6.0f, a7 = a0 + 7.0f; \
    const v2f m = {1.0000001f, 0.9999999f}, c = {1e-9f, -1e-9f};                                                          \
    for (int i = 0; i < ITER; ++i) {                                                                                      \
        REP4_(asm volatile(INSN(0) INSN(1) INSN(2) INSN(3) INSN(4) INSN(5) INSN(6) INSN(7)                                \
              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)        \
    }                                                                                                                     \
    v2f r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                                                        \
    if (r.x + r.y == 12345.678f) out[threadIdx.x] = r.x;                                                                  \
}
#define I_PKMUL(k) "v_pk_mul_f32 %" #k ", %" #k ", %8\n"
#define I_PKADD(k) "v_pk_add_f32 %" #k ", %" #k ", %9\n"
#define I_PKFMA_BCAST(k) "v_pk_fma_f32 %" #k ", %" #k ", %8, %9 op_sel_hi:[1,0,1]\n"
#define I_PKMOV(k) "v_pk_mov_b32 %" #k ", %8, %9\n"
PK_KERNEL(k_pk_mul, I_PKMUL)
PK_KERNEL(k_pk_add, I_PKADD)
PK_KERNEL(k_pk_fma_bcast, I_PKFMA_BCAST)
PK_KERNEL(k_pk_mov, I_PKMOV)

__global__ void __launch_bounds__(256) k_cmp_sgpr(float* out, float s)
{
    float a0 = s, a1 = s + 1, a2 = s + 2, a3 = s + 3, a4 = s + 4, a5 = s + 5, a6 = s + 6, a7 = s + 7;
    const float m = 1.0000001f;
    for (int i = 0; i < ITER; ++i) {
        REP4_(asm volatile(I_CMP_S(0) I_CMP_S(1) I_CMP_S(2) I_CMP_S(3) I_CMP_S(4) I_CMP_S(5) I_CMP_S(6) I_CMP_S(7)
              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "s20", "s21");)
    }
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <typename K>
static int run(const char* name, K kernel, double insn_per_iter, int cus, float* d_out)
{
    const int blocks = cus * 8;   // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d_out, 1.0f);
    CHK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CHK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d_out, 1.0f);
        CHK(hipEventRecord(e1, 0));
        CHK(hipEventSynchronize(e1));
        float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double waves_per_simd = 8.0;
    const double insn_per_simd = waves_per_simd * ITER * insn_per_iter;
    const double rate = insn_per_simd / (best * 1e3);   // wave-instructions per microsecond per SIMD
    printf("%-28s %8.3f ms   %8.1f wave-instr/us/SIMD   (%.2f ns per wave-instruction)\n", name, best, rate, 1e3 / rate);
    return 0;
}

int main()
{
    hipDeviceProp_t p;
    CHK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("%s: %d CUs, clockRate %d kHz\n", p.name, cus, p.clockRate);
    float* d_out = nullptr;
    CHK(hipMalloc(reinterpret_cast<void**>(&d_out), 4096));
    if (run("v_pk_fma_f32", k_pk_fma, 32, cus, d_out)) return 1;
    if (run("v_fma_f32", k_fma, 32, cus, d_out)) return 1;
    if (run("v_mul_f32", k_mul, 32, cus, d_out)) return 1;
    if (run("v_max_f32 (abs mods)", k_max, 32, cus, d_out)) return 1;
    if (run("v_rndne_f32", k_rndne, 32, cus, d_out)) return 1;
    if (run("v_cvt_i32_f32", k_cvt, 32, cus, d_out)) return 1;
    if (run("v_lshl_add_u32", k_lshladd, 32, cus, d_out)) return 1;
    if (run("v_mov_b32", k_mov, 32, cus, d_out)) return 1;
    if (run("v_cmp_le_f32 -> SGPR pair", k_cmp_sgpr, 32, cus, d_out)) return 1;
    if (run("v_cmp + v_cndmask (pairs)", k_cmp_cnd, 64, cus, d_out)) return 1;
    if (run("v_max_f32 e32 (no mods)", k_max_e32, 32, cus, d_out)) return 1;
    if (run("v_min_f32 e32", k_min_e32, 32, cus, d_out)) return 1;
    if (run("v_sub_f32 e32", k_sub, 32, cus, d_out)) return 1;
    if (run("v_fmac_f32 e32", k_fmac, 32, cus, d_out)) return 1;
    if (run("v_med3_f32", k_med3, 32, cus, d_out)) return 1;
    if (run("v_max3_f32", k_max3, 32, cus, d_out)) return 1;
    if (run("v_exp_f32", k_exp, 32, cus, d_out)) return 1;
    if (run("v_ldexp_f32", k_ldexp, 32, cus, d_out)) return 1;
    if (run("v_add_u32 e32", k_addu, 32, cus, d_out)) return 1;
    if (run("v_lshlrev_b32 e32", k_lshlrev, 32, cus, d_out)) return 1;
    if (run("v_and_b32 e32", k_and, 32, cus, d_out)) return 1;
    if (run("v_cmp_le_f32 e32 -> vcc", k_cmp_vcc, 32, cus, d_out)) return 1;
    if (run("v_cndmask_b32 e32 (vcc)", k_cnd, 32, cus, d_out)) return 1;
    if (run("v_cmp_class_f32 -> vcc", k_cmp_class, 32, cus, d_out)) return 1;
    if (run("v_mov_b32 dpp quad_perm", k_dpp, 32, cus, d_out)) return 1;
    if (run("v_bfi_b32", k_bfi, 32, cus, d_out)) return 1;
    if (run("v_mul_legacy_f32", k_mul_legacy, 32, cus, d_out)) return 1;
    if (run("v_pk_mul_f32", k_pk_mul, 32, cus, d_out)) return 1;
    if (run("v_pk_add_f32", k_pk_add, 32, cus, d_out)) return 1;
    if (run("v_pk_fma_f32 op_sel bcast", k_pk_fma_bcast, 32, cus, d_out)) return 1;
    if (run("v_pk_mov_b32", k_pk_mov, 32, cus, d_out)) return 1;
    return 0;
}
