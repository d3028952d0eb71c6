// Micro-benchmark: what does a launch cost before / beside the work of its wavefronts?  An (almost) empty kernel with the resource
// shape of the blend kernels (256 threads, ~20 KB of LDS per workgroup) for several grid sizes, back to back on one stream; then the
// same with every workgroup writing N bytes (the write-back at the end of a kernel).
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void __launch_bounds__(256) empty_kernel(float *out, int never)
{
    __shared__ float lds[5000];
    if (never) { lds[threadIdx.x] = out[threadIdx.x]; __syncthreads(); out[blockIdx.x] = lds[(threadIdx.x * 7) % 5000]; }
}
__global__ void __launch_bounds__(256) write_kernel(float4 *out, size_t n4)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) out[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

int main()
{
    float *buf;
    const size_t bytes = 512ull << 20;
    hipMalloc(&buf, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 200;
    for (int grid : {1, 256, 1024, 4096, 8192, 32768, 131072}) {
        hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(256), 0, 0, buf, 0);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int r = 0; r < reps; r++) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(256), 0, 0, buf, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        printf("empty kernel, %6d workgroups of 256 threads + 20 KB LDS: %.2f us per launch back to back\n", grid, ms * 1e3 / reps);
    }
    for (size_t mb : {1, 8, 32, 128, 512}) {
        const size_t n4 = (mb << 20) / 16;
        hipLaunchKernelGGL(write_kernel, dim3(2048), dim3(256), 0, 0, (float4 *)buf, n4);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int r = 0; r < 50; r++) hipLaunchKernelGGL(write_kernel, dim3(2048), dim3(256), 0, 0, (float4 *)buf, n4);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        printf("kernel that writes %4zu MB: %.1f us per launch back to back = %.0f GB/s\n", mb, ms * 1e3 / 50, mb / 1024.0 / (ms * 1e-3 / 50));
    }
    return 0;
}
