// Issue rate of v_fma_f32 against v_pk_fma_f32 on gfx950: N independent accumulator chains per lane, W waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pk_rate scripts/micro/pk_rate.hip && /tmp/pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int PK>
__global__ void __launch_bounds__(256) k(float *out, int iters, float w)
{
    f2 a[8];
    for (int i = 0; i < 8; i++) a[i] = f2{(float)threadIdx.x + i, 1.f + i};
    const f2 x = {1.0001f, 0.9999f};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (PK) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(f2{w, w}));
                else {
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(x.x), "v"(w));
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].y) : "v"(x.y), "v"(w));
                }
            }
    }
    float s = 0.f;
    for (int i = 0; i < 8; i++) s += a[i].x + a[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    float *out;
    hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int blocks : {256, 512, 1024, 2048}) {
        for (int pk = 0; pk < 2; pk++) {
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (pk) k<1><<<blocks, 256>>>(out, iters, 0.999f); else k<0><<<blocks, 256>>>(out, iters, 0.999f);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                const double fma = (double)blocks * 256 * iters * 64 * 2;     // scalar FMAs
                if (rep) printf("blocks %4d (%d waves/SIMD)  %-12s  %.3f ms  %.1f TFLOP/s\n", blocks, blocks / 256, pk ? "v_pk_fma_f32" : "v_fma_f32 x2", ms, 2 * fma / ms * 1e-9);
            }
        }
    }
    return 0;
}
