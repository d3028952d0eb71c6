// Issue rate of common vector instructions on gfx950, one op at a time: 16 independent registers per lane, 1 / 2 / 4 / 8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates.bin scripts/micro/valu_rates.hip && ./valu_rates.bin
// Prints cycles per wave-instruction per SIMD (at the clock measured with s_memtime-free arithmetic: assumes 2.4 GHz nominal; compare rows).
#include <hip/hip_runtime.h>
#include <cstdio>

#define OPS(X) \
    X(0, "v_fma_f32", "v_fma_f32 %0, %1, %2, %0") \
    X(1, "v_mul_f32", "v_mul_f32 %0, %1, %0") \
    X(2, "v_add_f32", "v_add_f32 %0, %1, %0") \
    X(3, "v_max_f32", "v_max_f32 %0, %1, %0") \
    X(4, "v_mov_b32", "v_mov_b32 %0, %1") \
    X(5, "v_add_u32", "v_add_u32 %0, %1, %0") \
    X(6, "v_and_b32", "v_and_b32 %0, %1, %0") \
    X(7, "v_exp_f32", "v_exp_f32 %0, %0") \
    X(8, "v_rcp_f32", "v_rcp_f32 %0, %0") \
    X(9, "v_cndmask_b32", "v_cndmask_b32 %0, %1, %0, vcc") \
    X(10, "v_cmp_lt_f32", "v_cmp_lt_f32 vcc, %1, %0") \
    X(11, "v_mul_lo_u32", "v_mul_lo_u32 %0, %1, %0") \
    X(12, "v_mad_u32_u24", "v_mad_u32_u24 %0, %1, %2, %0") \
    X(13, "v_lshlrev_b32", "v_lshlrev_b32 %0, 1, %0") \
    X(14, "v_add_f32 dpp", "v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf") \
    X(15, "v_cvt_f32_u32", "v_cvt_f32_u32 %0, %0") \
    X(16, "v_fmac_f32", "v_fmac_f32 %0, %1, %2") \
    X(17, "v_sub_f32", "v_sub_f32 %0, %1, %0") \
    X(18, "v_mul_f32 + v_fma_f32 mix", "v_mul_f32 %0, %1, %0\n v_fma_f32 %0, %1, %2, %0") \
    X(19, "v_fma_f32 sgpr operand", "v_fma_f32 %0, %1, s4, %0") \
    X(20, "v_bfe_u32", "v_bfe_u32 %0, %0, 1, 7") \
    X(21, "v_min_f32", "v_min_f32 %0, %1, %0") \
    X(22, "v_mbcnt_lo", "v_mbcnt_lo_u32_b32 %0, -1, %0") \
    X(23, "v_readfirstlane", "v_readfirstlane_b32 s6, %0") \
    X(24, "ds_bpermute_b32", "ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)") \
    X(25, "v_add_co_u32 64-bit pair", "v_add_co_u32 %0, vcc, %1, %0\n v_addc_co_u32 %0, vcc, 0, %0, vcc") \
    X(26, "v_cndmask_b32 e64 sgpr mask", "v_cndmask_b32_e64 %0, %1, %0, s[4:5]") \
    X(27, "v_cmp + v_cndmask pair", "v_cmp_lt_f32 vcc, %1, %0\n v_cndmask_b32 %0, %1, %0, vcc") \
    X(28, "v_cndmask_b32 0, v, vcc", "v_cndmask_b32 %0, 0, %0, vcc") \
    X(29, "v_pk_fma_f32", "v_pk_fma_f32 %3, %4, %4, %3") \
    X(30, "v_pk_mul_f32", "v_pk_mul_f32 %3, %4, %3") \
    X(31, "v_fma_f32 neg modifier", "v_fma_f32 %0, -%1, %2, %0") \
    X(32, "v_fma_f32 inline const", "v_fma_f32 %0, %1, 2.0, %0") \
    X(33, "v_fmaak_f32 literal", "v_fmaak_f32 %0, %1, %0, 0x3f7fbe77") \
    X(34, "v_mul_f32 sgpr operand", "v_mul_f32 %0, s4, %0") \
    X(35, "v_mul_f32 e64 |abs|", "v_mul_f32_e64 %0, |%1|, %0") \
    X(36, "v_med3_f32", "v_med3_f32 %0, %1, %2, %0") \
    X(37, "v_log_f32", "v_log_f32 %0, %0") \
    X(38, "v_sub_u32", "v_sub_u32 %0, %1, %0") \
    X(39, "v_or_b32", "v_or_b32 %0, %1, %0") \
    X(40, "v_xor_b32", "v_xor_b32 %0, %1, %0") \
    X(41, "v_add3_u32", "v_add3_u32 %0, %1, %2, %0") \
    X(42, "v_lshl_add_u32", "v_lshl_add_u32 %0, %1, 2, %0") \
    X(43, "v_mov_b32 dpp", "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf") \
    X(44, "v_mul_f32 then dependent v_fma", "v_mul_f32 %0, %1, %0\n v_fma_f32 %0, %0, %2, %0") \
    X(45, "v_cmp e64 to sgpr pair", "v_cmp_lt_f32_e64 s[4:5], %1, %0") \
    X(46, "v_max_f32 e64 clamp-style", "v_max_f32 %0, 0, %0") \
    X(47, "v_add_f32 sgpr operand", "v_add_f32 %0, s4, %0") \
    X(48, "v_exp_f32 (indep. of fma) mix", "v_exp_f32 %0, %0\n v_fma_f32 %0, %1, %2, %0\n v_fma_f32 %0, %1, %2, %0\n v_fma_f32 %0, %1, %2, %0") \
    X(49, "v_accvgpr_write", "v_accvgpr_write_b32 a0, %0")

template <int OP>
__global__ void __launch_bounds__(256) k(float *out, int iters, float w)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    float a[16];
    f2 b[16];
    for (int i = 0; i < 16; i++) { a[i] = (float)threadIdx.x * 0.001f + i; b[i] = f2{a[i], a[i] + 1.f}; }
    const f2 x2 = {1.0001f, 0.9999f};
    const float x = 1.0001f + threadIdx.x * 1e-9f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int i = 0; i < 16; i++) {
#define X(ID, NAME, ASM) if (OP == ID) asm volatile(ASM : "+v"(a[i]) : "v"(x), "v"(w), "v"(b[i]), "v"(x2) : "vcc", "s4", "s5", "s6", "a0");
                OPS(X)
#undef X
            }
    }
    float s = 0.f;
    for (int i = 0; i < 16; i++) s += a[i] + b[i].x + b[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP> void run(float *out, const char *name, int per)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 1000;
    printf("%-28s", name);
    for (int blocks : {256, 512, 1024, 2048}) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            (void)hipEventRecord(e0);
            k<OP><<<blocks, 256>>>(out, iters, 0.999f);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        const double instr_per_simd = (double)blocks / 256 * iters * 64 * per;      // wave-instructions one SIMD executes
        printf("  %5.2f", best * 1e-3 * 2.4e9 / instr_per_simd);
    }
    printf("\n");
}

int main()
{
    float *out;
    (void)hipMalloc(&out, 4096 * 256 * 4);
    printf("cycles (at 2.4 GHz) per wave-instruction per SIMD     1     2     4     8  waves per SIMD\n");
#define X(ID, NAME, ASM) run<ID>(out, NAME, (ID == 48) ? 4 : (ID == 18 || ID == 25 || ID == 27 || ID == 44) ? 2 : 1);
    OPS(X)
#undef X
    return 0;
}
