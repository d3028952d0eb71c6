// Micro-benchmark: how many instructions per cycle does one gfx950 SIMD issue when its wavefronts mix VALU and SALU work?
// Each wavefront runs ITER trips of an unrolled block of NV independent-ish v_fma_f32 and NS s_add_u32; the grid keeps
// `waves_per_simd` wavefronts on every SIMD (1024 SIMDs).  Prints cycles per trip per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int NV, int NS>
__global__ void __launch_bounds__(64) mix_kernel(float *out, int iters)
{
    float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f;
    unsigned s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < NV / 4; k++) {
            asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        }
#pragma unroll
        for (int k = 0; k < NS / 4; k++) {
            asm volatile("s_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 1\n\ts_add_u32 %2, %2, 1\n\ts_add_u32 %3, %3, 1"
                         : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        }
    }
    if (a0 + a1 + a2 + a3 == 12345.f || s0 + s1 + s2 + s3 == 77u) out[0] = a0;
}
// interleaved: one SALU after every VALU
template <int N>
__global__ void __launch_bounds__(64) interleaved_kernel(float *out, int iters)
{
    float a0 = threadIdx.x, a1 = 1.f;
    unsigned s0 = blockIdx.x, s1 = 1;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < N / 2; k++) {
            asm volatile("v_fma_f32 %0, %0, %0, %0\n\ts_add_u32 %2, %2, 1\n\tv_fma_f32 %1, %1, %1, %1\n\ts_add_u32 %3, %3, 1"
                         : "+v"(a0), "+v"(a1), "+s"(s0), "+s"(s1) : : "scc");
        }
    }
    if (a0 + a1 == 12345.f || s0 + s1 == 77u) out[0] = a0;
}

template <typename K>
static double time_kernel(K kernel, int grid, int iters, float *out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(64), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(64), 0, 0, out, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3;
}

int main()
{
    float *out;
    hipMalloc(&out, 64);
    int clk_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    const int iters = 2000;
    printf("clock %d MHz; per trip: 128 instructions per wavefront\n", clk_khz / 1000);
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int grid = 1024 * wps;
        const double v = time_kernel(mix_kernel<128, 0>, grid, iters, out);
        const double s = time_kernel(mix_kernel<0, 128>, grid, iters, out);
        const double m = time_kernel(mix_kernel<64, 64>, grid, iters, out);
        const double il = time_kernel(interleaved_kernel<128>, grid, iters, out);
        const double cyc = clk_khz * 1e-3;      // cycles per us
        printf("waves/SIMD %d: cycles per trip per SIMD -- 128 VALU: %.0f, 128 SALU: %.0f, 64 VALU then 64 SALU: %.0f, 64+64 interleaved: %.0f\n", wps,
               v * cyc / iters, s * cyc / iters, m * cyc / iters, il * cyc / iters);
    }
    return 0;
}
