// row_handover.hip -- prices the two ways a backward blend can hand its per-(block, entry) sums over to the per-Gaussian stage
// (VERDICT r5 item 2), at the shape of BASELINE config C3: 4 frames x 100 000 Gaussians, 825 000 live (block, entry) rows of 64 bytes
// per frame (3.3 M per 4-frame launch: TCC_EA0_ATOMIC of round 5), rows per Gaussian skewed like tile counts (log-normal).
//   A  float atomics into acc[frame][P][16], four WHOLE rows per wave-instruction (what rast_render_bwd.hip does);
//   B  plain stores of the same rows into private slots (a wavefront takes 64 consecutive slots per batch), then
//   C  a per-Gaussian sum in a FIXED order through an inverted index (CSR: offsets[P + 1], row ids), float64 in registers,
//      16 lanes per Gaussian (one per component: every row is read as one 64-byte segment);
//   D  the index itself: count (integer atomics on P counters), exclusive scan, scatter of the row ids.
// Nothing here is product code; build: hipcc --offload-arch=gfx950 -O3 row_handover.hip -o row_handover.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int F = 4, P = 100000, ROWS = 825000;

// lane = (one of four rows of the instruction, component); 16 instructions per batch of 64 rows
__global__ void __launch_bounds__(64) atomics_kernel(const uint32_t *gid, const float *val, float *acc, int rows_per_frame)
{
    const int lane = threadIdx.x, r4 = lane >> 4, q = lane & 15;
    const int frame = blockIdx.y;
    const size_t batch = (size_t)blockIdx.x * 64;
    const uint32_t *g = gid + (size_t)frame * rows_per_frame;
    float *a = acc + (size_t)frame * P * 16;
    const float v = val[lane];
    for (int k = 0; k < 16; k++) {
        const size_t row = batch + 4 * k + r4;
        if (row < (size_t)rows_per_frame && q < 13) atomicAdd(a + (size_t)g[row] * 16 + q, v);
    }
}
__global__ void __launch_bounds__(64) stores_kernel(const uint32_t *gid, const float *val, float *rows, int rows_per_frame)
{
    const int lane = threadIdx.x, r4 = lane >> 4, q = lane & 15;
    const int frame = blockIdx.y;
    const size_t batch = (size_t)blockIdx.x * 64;
    float *dst = rows + (size_t)frame * rows_per_frame * 16;
    const float v = val[lane];
    for (int k = 0; k < 16; k++) {
        const size_t row = batch + 4 * k + r4;
        if (row < (size_t)rows_per_frame) dst[row * 16 + q] = v;
    }
}
// 16 lanes per Gaussian, rows of one Gaussian in index order
__global__ void __launch_bounds__(256) gather_kernel(const uint32_t *offsets, const uint32_t *ids, const float *rows, float *out, int rows_per_frame)
{
    const int frame = blockIdx.y;
    const int g = blockIdx.x * 16 + (threadIdx.x >> 4), q = threadIdx.x & 15;
    if (g >= P) return;
    const uint32_t *off = offsets + (size_t)frame * (P + 1);
    const uint32_t *id = ids + (size_t)frame * rows_per_frame;
    const float *src = rows + (size_t)frame * rows_per_frame * 16;
    double s = 0.0;
    for (uint32_t k = off[g]; k < off[g + 1]; k++) s += (double)src[(size_t)id[k] * 16 + q];
    out[((size_t)frame * P + g) * 16 + q] = (float)s;
}
__global__ void count_kernel(const uint32_t *gid, uint32_t *count, int rows_per_frame)
{
    const int frame = blockIdx.y;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < (size_t)rows_per_frame) atomicAdd(count + (size_t)frame * (P + 1) + gid[(size_t)frame * rows_per_frame + i], 1u);
}
// one workgroup per frame: exclusive scan of P counters (1024 threads, sequential chunks)
__global__ void __launch_bounds__(1024) scan_kernel(uint32_t *count)
{
    __shared__ uint32_t part[1024];
    uint32_t *c = count + (size_t)blockIdx.x * (P + 1);
    const int per = (P + 1023) / 1024, t = threadIdx.x, lo = t * per, hi = min(P, lo + per);
    uint32_t s = 0;
    for (int i = lo; i < hi; i++) s += c[i];
    part[t] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) { uint32_t v = t >= d ? part[t - d] : 0u; __syncthreads(); part[t] += v; __syncthreads(); }
    uint32_t run = t ? part[t - 1] : 0u;
    for (int i = lo; i < hi; i++) { const uint32_t v = c[i]; c[i] = run; run += v; }
    if (t == 1023) c[P] = run;
}
__global__ void scatter_kernel(const uint32_t *gid, uint32_t *cursor, uint32_t *ids, int rows_per_frame)
{
    const int frame = blockIdx.y;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < (size_t)rows_per_frame) {
        const uint32_t at = atomicAdd(cursor + (size_t)frame * (P + 1) + gid[(size_t)frame * rows_per_frame + i], 1u);
        ids[(size_t)frame * rows_per_frame + at] = (uint32_t)i;
    }
}

int main()
{
    std::mt19937 rng(1);
    // rows per Gaussian ~ log-normal (a few wide splats own thousands of rows), drawn as a weighted choice of the Gaussian per row;
    // rows that follow each other in a batch belong to neighbours in the list = Gaussians drawn independently
    std::vector<double> w(P);
    std::lognormal_distribution<double> ln(0.0, 1.2);
    for (auto &x : w) x = ln(rng);
    std::discrete_distribution<uint32_t> pick(w.begin(), w.end());
    std::vector<uint32_t> gid((size_t)F * ROWS);
    for (auto &g : gid) g = pick(rng);
    std::vector<uint32_t> cnt(P, 0);
    for (int i = 0; i < ROWS; i++) cnt[gid[i]]++;
    std::vector<uint32_t> sorted(cnt);
    std::sort(sorted.begin(), sorted.end());
    printf("rows per Gaussian (frame 0): median %u, p99 %u, max %u, mean %.2f\n", sorted[P / 2], sorted[P * 99 / 100], sorted[P - 1], (double)ROWS / P);

    uint32_t *d_gid, *d_count, *d_cursor, *d_ids;
    float *d_val, *d_acc, *d_rows, *d_out;
    CK(hipMalloc(&d_gid, gid.size() * 4));
    CK(hipMemcpy(d_gid, gid.data(), gid.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_val, 64 * 4));
    CK(hipMemset(d_val, 0, 64 * 4));
    CK(hipMalloc(&d_acc, (size_t)F * P * 16 * 4));
    CK(hipMalloc(&d_rows, (size_t)F * ROWS * 16 * 4));
    CK(hipMalloc(&d_out, (size_t)F * P * 16 * 4));
    CK(hipMalloc(&d_count, (size_t)F * (P + 1) * 4));
    CK(hipMalloc(&d_cursor, (size_t)F * (P + 1) * 4));
    CK(hipMalloc(&d_ids, (size_t)F * ROWS * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const dim3 gb((ROWS + 63) / 64, F), gr((ROWS + 255) / 256, F), gg((P + 15) / 16, F);
    auto time = [&](const char *name, auto fn, double bytes) {
        for (int i = 0; i < 3; i++) fn();
        (void)hipEventRecord(e0);
        const int reps = 20;
        for (int i = 0; i < reps; i++) fn();
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double us = 1e3 * ms / reps;
        printf("%-46s %8.1f us per 4-frame launch   %6.2f TB/s of %0.0f MB\n", name, us, bytes / us / 1e6, bytes / 1e6);
        return us;
    };
    const double row_bytes = (double)F * ROWS * 64;
    time("A  atomics, 4 whole rows per instruction", [&] { hipLaunchKernelGGL(atomics_kernel, gb, dim3(64), 0, 0, d_gid, d_val, d_acc, ROWS); }, row_bytes);
    time("B  plain stores into private slots", [&] { hipLaunchKernelGGL(stores_kernel, gb, dim3(64), 0, 0, d_gid, d_val, d_rows, ROWS); }, row_bytes);
    auto index = [&] {
        (void)hipMemsetAsync(d_count, 0, (size_t)F * (P + 1) * 4, 0);
        hipLaunchKernelGGL(count_kernel, gr, dim3(256), 0, 0, d_gid, d_count, ROWS);
        hipLaunchKernelGGL(scan_kernel, dim3(F), dim3(1024), 0, 0, d_count);
        (void)hipMemcpyAsync(d_cursor, d_count, (size_t)F * (P + 1) * 4, hipMemcpyDeviceToDevice, 0);
        hipLaunchKernelGGL(scatter_kernel, gr, dim3(256), 0, 0, d_gid, d_cursor, d_ids, ROWS);
    };
    time("D  inverted index: count + scan + scatter", index, (double)F * ROWS * 12);
    index();
    time("C  fixed-order gather-sum, float64 registers", [&] { hipLaunchKernelGGL(gather_kernel, gg, dim3(256), 0, 0, d_count, d_ids, d_rows, d_out, ROWS); }, row_bytes);
    CK(hipDeviceSynchronize());
    return 0;
}
