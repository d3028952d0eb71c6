// optim.hip -- the optimizer step of the per-frame training path: Adam on the per-Gaussian leaves, one launch for all of them.
//
// The reference ends every training step with torch.optim.Adam(eps=1e-15).step() over the parameter groups of the Gaussian model
// (TS/geometry/surfel_base.py:596-681 training_setup: one group per leaf with its own learning rate;
// TS/system/gaussian_surfel_mvdream.py:471-472 optimizer.step()).  Between two steps the positions move by ~lr: the KNN blend
// weights of the next step are recomputed from the new positions (lbs_knn.hip).  torch's optimizer costs one multi-tensor
// launch chain and ~100 us of host time per step; here the update of all leaves is one kernel over a table of rows
// {parameter, gradient, first / second moment, count, learning rate}, with the step counter and its bias corrections kept on
// the device so that the launch can sit in a captured graph.
//
// Arithmetic = torch.optim.Adam (no weight decay, no amsgrad, maximize = False):
//   m = m + (g - m) (1 - beta1);  v = beta2 v + (1 - beta2) g g;
//   p = p - (lr / (1 - beta1^t)) m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
#include "soar_common.h"

#include <cmath>
#include <cstdint>

namespace soar {

namespace {

constexpr int ADAM_MAX_ROWS = 8;
struct AdamTable {
    SoarAdamRow row[ADAM_MAX_ROWS];
    int64_t first_block[ADAM_MAX_ROWS + 1];      // blocks of 1024 elements, row after row
    int n;
};
struct AdamState {           // 16 bytes of device memory owned by the caller
    int32_t step;
    float bias_correction1, bias_correction2_sqrt;
    int32_t pad;
};

__global__ void adam_tick_kernel(AdamState *st, double beta1, double beta2)
{
    const int t = st->step + 1;
    st->step = t;
    st->bias_correction1 = (float)(1.0 - pow(beta1, (double)t));
    st->bias_correction2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)t));
}

// one_minus_b1 / one_minus_b2: 1 - beta worked out in double by the host and rounded once, as torch does with its Python floats
// (1.f - 0.9f is 0.100000024, float(1 - 0.9) is 0.1: the lerp weight would differ in its last bits)
__global__ void __launch_bounds__(256) adam_update_kernel(AdamTable tab, const AdamState *__restrict__ st_dev, AdamState st_host, float beta2,
                                                          float one_minus_b1, float one_minus_b2, float eps)
{
    const AdamState *st = st_dev ? st_dev : &st_host;      // the step's bias corrections: from the device counter, or worked out by the host
    // rows of a step that was never started (advance = 0 on a fresh, zeroed state): 1 - beta1^0 = 0 would make the step size infinite
    if (st->step <= 0) return;
    int r = 0;
    while (r + 1 < tab.n && (int64_t)blockIdx.x >= tab.first_block[r + 1]) r++;
    const SoarAdamRow row = tab.row[r];
    const int64_t i0 = ((int64_t)blockIdx.x - tab.first_block[r]) * 1024 + threadIdx.x * 4;
    const float step_size = row.lr / st->bias_correction1, bc2s = st->bias_correction2_sqrt;
    auto update = [&](float g, float &p, float &m, float &v) {
#pragma clang fp contract(off)
        m = m + (g - m) * one_minus_b1;
        v = beta2 * v + one_minus_b2 * (g * g);        // (torch squares first: a gradient beyond 1.8e19 makes v infinite and the value stops moving -- kept)
        const float denom = sqrtf(v) / bc2s + eps;
        p = p - step_size * (m / denom);
    };
    // four consecutive elements per thread: one 16-byte load / store per array where the row allows it (every leaf of a model whose
    // size is a multiple of four; the slices of the flat gradient buffer start on 16-byte boundaries then)
    const bool vec = i0 + 3 < row.count && (((uintptr_t)row.param | (uintptr_t)row.grad | (uintptr_t)row.exp_avg | (uintptr_t)row.exp_avg_sq) & 15u) == 0u;
    if (vec) {
        const float4 g = *reinterpret_cast<const float4 *>(row.grad + i0);
        float4 p = *reinterpret_cast<const float4 *>(row.param + i0), m = *reinterpret_cast<const float4 *>(row.exp_avg + i0),
               v = *reinterpret_cast<const float4 *>(row.exp_avg_sq + i0);
        update(g.x, p.x, m.x, v.x); update(g.y, p.y, m.y, v.y); update(g.z, p.z, m.z, v.z); update(g.w, p.w, m.w, v.w);
        *reinterpret_cast<float4 *>(row.param + i0) = p;
        *reinterpret_cast<float4 *>(row.exp_avg + i0) = m;
        *reinterpret_cast<float4 *>(row.exp_avg_sq + i0) = v;
        return;
    }
    for (int k = 0; k < 4; k++) {
        const int64_t i = i0 + k;
        if (i >= row.count) break;
        float p = row.param[i], m = row.exp_avg[i], v = row.exp_avg_sq[i];
        update(row.grad[i], p, m, v);
        row.param[i] = p;
        row.exp_avg[i] = m;
        row.exp_avg_sq[i] = v;
    }
}

}  // namespace

}  // namespace soar

using namespace soar;

extern "C" int soar_adam_step(int32_t n_rows, const SoarAdamRow *rows_host, double beta1, double beta2, double eps, void *state_dev,
                              void *stream_)
{
    return soar_adam_step_rows(n_rows, rows_host, beta1, beta2, eps, state_dev, 1, stream_);
}

// The same update with the step number kept by the caller (as torch.optim.Adam does: its bias corrections are Python floats): no
// device counter, no launch to advance it.  Not for a captured graph -- a replay would repeat the same step number.
extern "C" int soar_adam_step_at(int32_t n_rows, const SoarAdamRow *rows_host, double beta1, double beta2, double eps, int64_t step,
                                 void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n_rows < 0 || n_rows > ADAM_MAX_ROWS || (n_rows && !rows_host) || step < 1) {
        set_error("soar_adam_step_at: 0 <= n_rows <= %d, rows must be given, step >= 1", ADAM_MAX_ROWS);
        return 1;
    }
    AdamTable tab;
    tab.n = n_rows;
    int64_t blocks = 0;
    for (int r = 0; r < n_rows; r++) {
        const SoarAdamRow &w = rows_host[r];
        if (w.count < 0 || (w.count && (!w.param || !w.grad || !w.exp_avg || !w.exp_avg_sq))) {
            set_error("soar_adam_step_at: row %d has a NULL pointer or a negative count", r);
            return 1;
        }
        tab.row[r] = w;
        tab.first_block[r] = blocks;
        blocks += (w.count + 1023) / 1024;
    }
    for (int r = n_rows; r <= ADAM_MAX_ROWS; r++) tab.first_block[r] = blocks;
    AdamState st;
    st.step = (int32_t)step;
    st.bias_correction1 = (float)(1.0 - pow(beta1, (double)step));
    st.bias_correction2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)step));
    st.pad = 0;
    StageTimer timer(ST_OPTIMIZER, stream);
    if (blocks > 0)
        hipLaunchKernelGGL(adam_update_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, tab, (const AdamState *)nullptr, st, (float)beta2,
                           (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps);
    SOAR_LAUNCH_OK("adam_step_at", stream, 0);
    return 0;
}

extern "C" int soar_adam_step_rows(int32_t n_rows, const SoarAdamRow *rows_host, double beta1, double beta2, double eps, void *state_dev,
                                   int32_t advance, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n_rows < 0 || n_rows > ADAM_MAX_ROWS || (n_rows && !rows_host) || !state_dev) {
        set_error("soar_adam_step: 0 <= n_rows <= %d, rows and the 16-byte device state must be given", ADAM_MAX_ROWS);
        return 1;
    }
    AdamTable tab;
    tab.n = n_rows;
    int64_t blocks = 0;
    for (int r = 0; r < n_rows; r++) {
        const SoarAdamRow &w = rows_host[r];
        if (w.count < 0 || (w.count && (!w.param || !w.grad || !w.exp_avg || !w.exp_avg_sq))) {
            set_error("soar_adam_step: row %d has a NULL pointer or a negative count", r);
            return 1;
        }
        tab.row[r] = w;
        tab.first_block[r] = blocks;
        blocks += (w.count + 1023) / 1024;
    }
    for (int r = n_rows; r <= ADAM_MAX_ROWS; r++) tab.first_block[r] = blocks;
    AdamState *st = static_cast<AdamState *>(state_dev);
    StageTimer timer(ST_OPTIMIZER, stream);
    if (advance) hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, stream, st, beta1, beta2);
    if (blocks > 0)
        hipLaunchKernelGGL(adam_update_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, tab, st, AdamState{}, (float)beta2, (float)(1.0 - beta1),
                           (float)(1.0 - beta2), (float)eps);
    SOAR_LAUNCH_OK("adam_step", stream, 0);
    return 0;
}
