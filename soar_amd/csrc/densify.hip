// densify.hip -- the densification / pruning state machine of the surfel model.
//
// Restates TS/geometry/surfel_base.py:850-1136,1198-1230 as three kernels over the original point index:
//   densify_stats_kernel : update_states' per-view body + add_densification_stats (:1102-1128, 1208-1216), one pass
//   densify_plan_kernel  : adaptive_prune (:1067-1087) + adaptive_densify / densify_and_clone / densify_and_split masks
//                          (:982-1000, 1032-1046, 1089-1100) -> a flag byte per point and three packed counters,
//                          followed by ONE 64-bit exclusive scan (rocPRIM) that yields every destination row
//   densify_apply_kernel : all parameter tensors, their Adam moments, in one launch: kept rows moved, clones appended,
//                          split children sampled (:1001-1014) -- the layout the reference reaches through
//                          prune_points -> cat (clone) -> cat (split) -> prune_points:
//                              [kept & not split | clones | split children rep 0 | split children rep 1 ...]
// The reference does this with ~150 boolean-index / cat / repeat launches and one optimizer-state rebuild per tensor.
// The state machine is deterministic given the accumulators and the normal samples, so ranks of a frame-data-parallel job
// that all-reduce the accumulators and share the noise generator stay identical (soar_amd/densify.py).
// Built with -ffp-contract=off: thresholds decide integer layout.
#include "soar_common.h"

#include <rocprim/device/device_scan.hpp>

namespace soar {

namespace {

constexpr uint8_t F_PRUNE = 1, F_CLONE = 2, F_SPLIT = 4;
constexpr int CNT_BITS = 21;                         // three 21-bit counters in one u64: P < 2^21 points
constexpr uint64_t CNT_MASK = (1ull << CNT_BITS) - 1;
constexpr int MAX_ROWS = 24;

struct PlanBuf {
    uint8_t *flags;        // [P]
    uint64_t *packed;      // [P]  kept | clone << 21 | split << 42 (then exclusive-scanned in place into offs)
    uint64_t *offs;        // [P]
    uint64_t *totals;      // [1]  packed totals
    void *scan_temp;
    size_t scan_bytes;
};

inline size_t plan_scan_bytes(int P)
{
    size_t bytes = 0;
    (void)rocprim::exclusive_scan((void *)nullptr, bytes, (uint64_t *)nullptr, (uint64_t *)nullptr, (uint64_t)0,
                                  (size_t)(P > 0 ? P : 1), rocprim::plus<uint64_t>(), (hipStream_t)0);
    return bytes;
}

inline size_t carve_plan(PlanBuf &b, void *base, int P)
{
    char *p = static_cast<char *>(base);
    auto take = [&](size_t n) { char *q = p; p += (n + 255) & ~(size_t)255; return q; };
    b.flags = reinterpret_cast<uint8_t *>(take((size_t)P));
    b.packed = reinterpret_cast<uint64_t *>(take((size_t)P * 8));
    b.offs = reinterpret_cast<uint64_t *>(take((size_t)P * 8));
    b.totals = reinterpret_cast<uint64_t *>(take(8));
    b.scan_bytes = plan_scan_bytes(P);
    b.scan_temp = take(b.scan_bytes);
    return (size_t)(p - static_cast<char *>(base));
}

__global__ void __launch_bounds__(256) densify_stats_kernel(int P, const int *__restrict__ radii, const float *__restrict__ grad2d,
                                                            int grad_stride, const float *__restrict__ scaling_grad,
                                                            const float *__restrict__ rotation, const float *__restrict__ opacity,
                                                            float *__restrict__ accum, float *__restrict__ max_radii)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const int r = radii[i];
    max_radii[i] = fmaxf(max_radii[i], (float)r);
    if (r <= 0) return;                                   // update_filter = radii > 0
    const float gx = grad2d[(size_t)i * grad_stride], gy = grad2d[(size_t)i * grad_stride + 1];
    accum[i] += sqrtf(gx * gx + gy * gy);
    accum[(size_t)P + i] += scaling_grad[3 * i] + scaling_grad[3 * i + 1];
    const float4 q = reinterpret_cast<const float4 *>(rotation)[i];
    accum[2 * (size_t)P + i] += sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);   // sic: |rotation|, not its gradient
    accum[3 * (size_t)P + i] += opacity[i];                                                // sic: the raw opacity
    accum[4 * (size_t)P + i] += 1.f;
}

struct PlanArgs {
    int P, do_prune, do_densify;
    const float *accum, *scaling, *opacity;
    float min_opacity, prune_scale_max, prune_area_min, max_grad, dense_scale;
    uint8_t *flags;
    uint64_t *packed;
};

__global__ void __launch_bounds__(256) densify_plan_kernel(PlanArgs a)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.P) return;
    const size_t P = a.P;
    const float sx = expf(a.scaling[3 * i]), sy = expf(a.scaling[3 * i + 1]), sz = expf(a.scaling[3 * i + 2]);
    const float denom = a.accum[4 * P + i];
    uint8_t f = 0;
    if (a.do_prune) {
        const float op = 1.f / (1.f + expf(-a.opacity[i]));
        const float smin = fminf(sx, sy), smax = fmaxf(sx, sy);
        if (op < a.min_opacity || denom == 0.f || smax > a.prune_scale_max || smin * smax < a.prune_area_min) f = F_PRUNE;
    }
    if (a.do_densify && !f) {
        // grad = accum / denom with NaN -> 0 (0/0); x/0 = inf stays, as in the reference
        float gp = a.accum[i] / denom, gs = a.accum[P + i] / denom, go = a.accum[3 * P + i] / denom;
        if (gp != gp) gp = 0.f;
        if (gs != gs) gs = 0.f;
        if (go != go) go = 0.f;
        const bool big = fmaxf(fmaxf(sx, sy), sz) > a.dense_scale;
        const bool hot = fabsf(gp) >= a.max_grad;              // torch.norm over the single column
        if (hot && !big && go <= 2.f && gs <= 1e-7f) f |= F_CLONE;
        if (gp >= a.max_grad && big) f |= F_SPLIT;             // the split test uses the signed padded gradient
    }
    a.flags[i] = f;
    const uint64_t kept = !(f & (F_PRUNE | F_SPLIT)), cl = (f & F_CLONE) ? 1 : 0, sp = (f & F_SPLIT) ? 1 : 0;
    a.packed[i] = kept | (cl << CNT_BITS) | (sp << (2 * CNT_BITS));
}

__global__ void densify_totals_kernel(int P, const uint64_t *__restrict__ packed, const uint64_t *__restrict__ offs,
                                      uint64_t *__restrict__ totals)
{
    totals[0] = P > 0 ? offs[P - 1] + packed[P - 1] : 0;
}

struct Row {
    const float *src;
    float *dst;
    int width, mode;            // 0 copy, 1 moments (new rows zero), 2 xyz (children sampled), 3 scaling (children shrunk)
};

struct ApplyArgs {
    int P, N, n_rows, surface;
    const uint8_t *flags;
    const uint64_t *offs, *totals;
    const float *scaling, *rotation, *xyz, *noise;
    Row rows[MAX_ROWS];
};

__global__ void __launch_bounds__(256) densify_apply_kernel(ApplyArgs a)
{
    const Row row = a.rows[blockIdx.y];
    const int W = row.width;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)a.P * W) return;
    const int i = (int)(e / W), c = (int)(e - (long long)i * W);
    const uint8_t f = a.flags[i];
    const uint64_t off = a.offs[i], tot = a.totals[0];
    const size_t n_keep = tot & CNT_MASK, n_clone = (tot >> CNT_BITS) & CNT_MASK, n_split = (tot >> (2 * CNT_BITS)) & CNT_MASK;
    const float v = row.src[e];
    if (!(f & (F_PRUNE | F_SPLIT))) row.dst[(off & CNT_MASK) * W + c] = v;
    if (f & F_CLONE) row.dst[(n_keep + ((off >> CNT_BITS) & CNT_MASK)) * W + c] = row.mode == 1 ? 0.f : v;
    if (f & F_SPLIT) {
        const size_t so = (off >> (2 * CNT_BITS)) & CNT_MASK;
        for (int r = 0; r < a.N; r++) {
            const size_t child = (size_t)r * n_split + so;
            float out = v;
            if (row.mode == 1) out = 0.f;
            else if (row.mode == 3) {
                // new_scaling = log(exp(s) / (0.8 N)); surface: last column -1e10   (:1005-1009)
                out = (a.surface && c == W - 1) ? -1e10f : logf(expf(v) / (0.8f * (float)a.N));
            } else if (row.mode == 2) {
                // new_xyz = R(q / |q|) (noise * exp(scaling)) + xyz   (:1001-1004, build_rotation general_utils.py:100-123)
                const float4 q0 = reinterpret_cast<const float4 *>(a.rotation)[i];
                const float nrm = sqrtf(q0.x * q0.x + q0.y * q0.y + q0.z * q0.z + q0.w * q0.w);
                const float w = q0.x / nrm, x = q0.y / nrm, y = q0.z / nrm, z = q0.w / nrm;
                float R0, R1, R2;
                if (c == 0) { R0 = 1.f - 2.f * (y * y + z * z); R1 = 2.f * (x * y - w * z); R2 = 2.f * (x * z + w * y); }
                else if (c == 1) { R0 = 2.f * (x * y + w * z); R1 = 1.f - 2.f * (x * x + z * z); R2 = 2.f * (y * z - w * x); }
                else { R0 = 2.f * (x * z - w * y); R1 = 2.f * (y * z + w * x); R2 = 1.f - 2.f * (x * x + y * y); }
                const float n0 = a.noise ? a.noise[child * 3] : 0.f, n1 = a.noise ? a.noise[child * 3 + 1] : 0.f,
                            n2 = a.noise ? a.noise[child * 3 + 2] : 0.f;
                const float s0 = n0 * expf(a.scaling[3 * i]), s1 = n1 * expf(a.scaling[3 * i + 1]), s2 = n2 * expf(a.scaling[3 * i + 2]);
                out = ((R0 * s0 + R1 * s1) + R2 * s2) + v;
            }
            row.dst[(n_keep + n_clone + child) * W + c] = out;
        }
    }
}

}  // namespace

}  // namespace soar

using namespace soar;

extern "C" int soar_densify_stats(int32_t P, const int32_t *radii, const float *grad2d, int32_t grad_stride,
                                  const float *scaling_grad, const float *rotation, const float *opacity, float *accum,
                                  float *max_radii2D, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (P < 0 || grad_stride < 2) { set_error("soar_densify_stats: bad P / grad_stride"); return 1; }
    if (P == 0) return 0;
    if (!radii || !grad2d || !scaling_grad || !rotation || !opacity || !accum || !max_radii2D) {
        set_error("soar_densify_stats: NULL argument");
        return 1;
    }
    hipLaunchKernelGGL(densify_stats_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, radii, grad2d, grad_stride,
                       scaling_grad, rotation, opacity, accum, max_radii2D);
    SOAR_LAUNCH_OK("densify_stats", stream, 0);
    return 0;
}

extern "C" int soar_densify_plan_bytes(int32_t P, size_t *bytes)
{
    if (P < 0 || !bytes) { set_error("soar_densify_plan_bytes: bad arguments"); return 1; }
    PlanBuf b;
    *bytes = carve_plan(b, nullptr, P > 0 ? P : 1) + 256;
    return 0;
}

extern "C" int soar_densify_plan(int32_t P, const float *accum, const float *scaling, const float *opacity, int32_t do_prune,
                                 int32_t do_densify, float min_opacity, float prune_scale_max, float prune_area_min,
                                 float max_grad, float dense_scale, void *plan, int64_t *counts_host, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (P <= 0 || P >= (1 << CNT_BITS)) { set_error("soar_densify_plan: need 0 < P < 2^%d (P=%d)", CNT_BITS, P); return 1; }
    if (!accum || !scaling || !opacity || !plan || ((uintptr_t)plan & 255)) {
        set_error("soar_densify_plan: NULL argument or plan buffer not 256-byte aligned");
        return 1;
    }
    PlanBuf b;
    carve_plan(b, plan, P);
    PlanArgs a = {P, do_prune, do_densify, accum, scaling, opacity, min_opacity, prune_scale_max, prune_area_min, max_grad,
                  dense_scale, b.flags, b.packed};
    hipLaunchKernelGGL(densify_plan_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, a);
    size_t bytes = b.scan_bytes;
    SOAR_HIP_OK(rocprim::exclusive_scan(b.scan_temp, bytes, b.packed, b.offs, (uint64_t)0, (size_t)P, rocprim::plus<uint64_t>(),
                                        stream));
    hipLaunchKernelGGL(densify_totals_kernel, dim3(1), dim3(1), 0, stream, P, b.packed, b.offs, b.totals);
    SOAR_LAUNCH_OK("densify_plan", stream, 0);
    if (counts_host) {
        uint64_t tot = 0;
        SOAR_HIP_OK(hipMemcpyAsync(&tot, b.totals, 8, hipMemcpyDeviceToHost, stream));
        SOAR_HIP_OK(hipStreamSynchronize(stream));
        counts_host[0] = (int64_t)(tot & CNT_MASK);
        counts_host[1] = (int64_t)((tot >> CNT_BITS) & CNT_MASK);
        counts_host[2] = (int64_t)((tot >> (2 * CNT_BITS)) & CNT_MASK);
    }
    return 0;
}

extern "C" int soar_densify_flags(int32_t P, const void *plan, uint8_t *flags_out, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (P <= 0 || !plan || !flags_out) { set_error("soar_densify_flags: bad arguments"); return 1; }
    PlanBuf b;
    carve_plan(b, const_cast<void *>(plan), P);
    SOAR_HIP_OK(hipMemcpyAsync(flags_out, b.flags, (size_t)P, hipMemcpyDeviceToDevice, stream));
    return 0;
}

extern "C" int soar_densify_apply(int32_t P, int32_t N, const void *plan, int32_t n_rows, const SoarDensifyRow *rows,
                                  const float *scaling, const float *rotation, const float *noise, int32_t surface,
                                  void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (P <= 0 || N < 1 || n_rows < 1 || n_rows > MAX_ROWS || !plan || !rows) {
        set_error("soar_densify_apply: need P > 0, N >= 1, 1 <= n_rows <= %d", MAX_ROWS);
        return 1;
    }
    PlanBuf b;
    carve_plan(b, const_cast<void *>(plan), P);
    ApplyArgs a = {};
    a.P = P; a.N = N; a.n_rows = n_rows; a.surface = surface;
    a.flags = b.flags; a.offs = b.offs; a.totals = b.totals;
    a.scaling = scaling; a.rotation = rotation; a.noise = noise;
    int wmax = 1;
    for (int k = 0; k < n_rows; k++) {
        if (!rows[k].src || !rows[k].dst || rows[k].width < 1 || rows[k].mode < 0 || rows[k].mode > 3) {
            set_error("soar_densify_apply: row %d is malformed", k);
            return 1;
        }
        if ((rows[k].mode == 2 && (rows[k].width != 3 || !scaling || !rotation)) || (rows[k].mode == 3 && !scaling)) {
            set_error("soar_densify_apply: row %d (mode %d) needs scaling / rotation and width 3 for xyz", k, rows[k].mode);
            return 1;
        }
        a.rows[k] = {rows[k].src, rows[k].dst, rows[k].width, rows[k].mode};
        wmax = rows[k].width > wmax ? rows[k].width : wmax;
    }
    const long long blocks = ((long long)P * wmax + 255) / 256;
    hipLaunchKernelGGL(densify_apply_kernel, dim3((unsigned)blocks, n_rows), dim3(256), 0, stream, a);
    SOAR_LAUNCH_OK("densify_apply", stream, 0);
    return 0;
}
