// What the backward blend's 64-lane affine scan costs per call and SIMD (the asm block of rast_render_bwd.hip::affine_scan_with_products),
// against single DPP flavours, at 1 / 2 / 4 / 8 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -o scan_cost.bin scripts/micro/scan_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ void __launch_bounds__(256) k(float *out, int iters, float w)
{
    float m = 0.999f + threadIdx.x * 1e-7f, b = 0.001f * threadIdx.x, dx = 0.3f, dy = 0.7f, A = 1.1f, B = 0.2f, C = 0.9f;
    float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f, o4 = 0.f, o5 = 0.f;
    for (int it = 0; it < iters; it++) {
        if (OP == 0) {
            asm volatile(
                "v_mul_f32 %2, %8, %8\n\t" "v_mul_f32 %3, %8, %9\n\t" "v_rcp_f32 %7, %1\n\t"
                "v_fmac_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t" "v_mul_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                "v_mul_f32 %4, %9, %9\n\t"
                "v_fmac_f32_dpp %0, %0, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t" "v_mul_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                "v_mul_f32 %5, %10, %8\n\t"
                "v_fmac_f32_dpp %0, %0, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t" "v_mul_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                "v_mul_f32 %6, %12, %9\n\t"
                "v_fmac_f32_dpp %0, %0, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t" "v_mul_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f32 %5, %11, %9\n\t"
                "v_fmac_f32_dpp %0, %0, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" "v_mul_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                "v_fmac_f32 %6, %11, %8\n\t"
                "v_fmac_f32_dpp %0, %0, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t" "v_mul_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf"
                : "+v"(b), "+v"(m), "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3), "=&v"(o4), "=&v"(o5) : "v"(dx), "v"(dy), "v"(A), "v"(B), "v"(C));
        } else if (OP == 1) {           // 20 independent fast-class instructions (the same count)
#pragma unroll
            for (int r = 0; r < 20; r++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r & 1 ? b : m) : "v"(dx), "v"(dy));
        } else if (OP == 2) {           // 12 DPP instructions, row_shr:1 only, two independent chains
#pragma unroll
            for (int r = 0; r < 6; r++) {
                asm volatile("v_fmac_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(b) : "v"(dx));
                asm volatile("v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(m));
            }
        } else if (OP == 3) {           // 12 DPP instructions, row_bcast:15 / 31 flavours
#pragma unroll
            for (int r = 0; r < 3; r++) {
                asm volatile("v_fmac_f32_dpp %0, %0, %1 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(b) : "v"(dx));
                asm volatile("v_mul_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(m));
                asm volatile("v_fmac_f32_dpp %0, %0, %1 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(b) : "v"(dx));
                asm volatile("v_mul_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(m));
            }
        } else if (OP == 4) {           // the scan's 12 DPP steps alone, dependent as in the scan (b reads m of the step before)
            asm volatile(
                "v_fmac_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t" "v_mul_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t" "s_nop 0\n\t"
                "v_fmac_f32_dpp %0, %0, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t" "v_mul_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t" "s_nop 0\n\t"
                "v_fmac_f32_dpp %0, %0, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t" "v_mul_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t" "s_nop 0\n\t"
                "v_fmac_f32_dpp %0, %0, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t" "v_mul_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t" "s_nop 0\n\t"
                "v_fmac_f32_dpp %0, %0, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" "v_mul_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" "s_nop 0\n\t"
                "v_fmac_f32_dpp %0, %0, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t" "v_mul_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t" "s_nop 1"
                : "+v"(b), "+v"(m));
        }
        m = m * 0.5f + 0.4995f;       // keep the values in range (2 more fast instructions per trip)
    }
    out[blockIdx.x * 256 + threadIdx.x] = m + b + o0 + o1 + o2 + o3 + o4 + o5;
}

template <int OP> void run(float *out, const char *name)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    printf("%-58s", name);
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
        printf("  %6.1f", best * 1e-3 * 2.4e9 / ((double)blocks / 256 * iters));
    }
    printf("\n");
}

int main()
{
    float *out;
    (void)hipMalloc(&out, 4096 * 256 * 4);
    printf("cycles (at 2.4 GHz) per trip and SIMD (+ 2 fast instructions per trip)            1       2       4       8  waves per SIMD\n");
    run<0>(out, "the scan with its six products and the reciprocal (21 instr.)");
    run<1>(out, "20 v_fma_f32");
    run<2>(out, "12 DPP (row_shr:1), two chains");
    run<3>(out, "12 DPP (row_bcast:15 / 31), two chains");
    run<4>(out, "the scan's 12 DPP steps alone (+ 6 s_nop)");
    return 0;
}
