// ref_shim.hip -- TEST INFRASTRUCTURE ONLY.  A C-ABI caller of the reference's own C++ entry points
// (CudaRasterizer::Rasterizer::forward / backward / markVisible, DGR/cuda_rasterizer/rasterizer.h:24-106), linked against the
// reference's rasterizer sources built by oracle/ref_build/build_ref.sh into oracle/_ref/libref_rasterizer.so.
// It plays the role of DGR/rasterize_points.cu (the torch binding) without torch: buffers come from hipMalloc.
// Nothing under soar_amd/ may load this library; tests use it to pin oracle/rasterizer_oracle.c and the HIP path.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <functional>

#include "rasterizer.h"
#include "rasterizer_impl.h"

namespace {

struct Chunk {
    char *ptr = nullptr;
    size_t size = 0;
    char *resize(size_t n)
    {
        if (n > size) {
            if (ptr) (void)hipFree(ptr);
            ptr = nullptr;
            if (hipMalloc(&ptr, n) != hipSuccess) { size = 0; return nullptr; }
            size = n;
        }
        return ptr;
    }
    ~Chunk() { if (ptr) (void)hipFree(ptr); }
};

struct RefRast {
    Chunk geom, binning, img;
    int P = 0, R = 0, W = 0, H = 0;
};

}  // namespace

extern "C" void *ref_rast_create() { return new RefRast(); }
extern "C" void ref_rast_destroy(void *h) { delete static_cast<RefRast *>(h); }

extern "C" int ref_rast_forward(void *h_, int P, int D, int M, const float *background, int W, int H, const float *means3D,
                                const float *shs, const float *colors_precomp, const float *opacities, const float *scales,
                                float scale_modifier, const float *rotations, const float *cov3D_precomp,
                                const float *viewmatrix, const float *projmatrix, const float *prcppoint,
                                const float *patchbbox, const float *campos, float tan_fovx, float tan_fovy, int prefiltered,
                                int render_front, int sort_descending, float *config, float *out_color, float *out_normal,
                                float *out_depth, float *out_opac, int *radii, int debug)
{
    RefRast *h = static_cast<RefRast *>(h_);
    h->P = P; h->W = W; h->H = H;
    h->R = CudaRasterizer::Rasterizer::forward(
        [h](size_t n) { return h->geom.resize(n); }, [h](size_t n) { return h->binning.resize(n); },
        [h](size_t n) { return h->img.resize(n); }, P, D, M, background, W, H, means3D, shs, colors_precomp, opacities, scales,
        scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, prcppoint, patchbbox, campos, tan_fovx, tan_fovy,
        prefiltered != 0, render_front != 0, sort_descending != 0, config, out_color, out_normal, out_depth, out_opac, radii,
        debug != 0);
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return h->R;
}

extern "C" int ref_rast_backward(void *h_, int D, int M, const float *background, const float *means3D, const float *shs,
                                 const float *colors_precomp, const float *scales, float scale_modifier,
                                 const float *rotations, const float *cov3D_precomp, const float *viewmatrix,
                                 const float *projmatrix, const float *campos, const float *prcppoint, const float *patchbbox,
                                 float tan_fovx, float tan_fovy, const int *radii, const float *dL_dpixcolor,
                                 const float *dL_dpixnormal, const float *dL_dpixdepth, const float *dL_dpixopac,
                                 float *dL_dmean2D, float *dL_dconic, float *dL_dopacity, float *dL_dcolor, float *dL_dnormal,
                                 float *dL_ddepth, float *dL_dmean3D, float *dL_dcov3D, float *dL_dsh, float *dL_dscale,
                                 float *dL_drot, float *dL_dviewmat, float *dL_dprojmat, float *dL_dcampos, int debug,
                                 float *config)
{
    RefRast *h = static_cast<RefRast *>(h_);
    CudaRasterizer::Rasterizer::backward(h->P, D, M, h->R, background, h->W, h->H, means3D, shs, colors_precomp, scales,
                                         scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, campos, prcppoint,
                                         patchbbox, tan_fovx, tan_fovy, radii, h->geom.ptr, h->binning.ptr, h->img.ptr,
                                         dL_dpixcolor, dL_dpixnormal, dL_dpixdepth, dL_dpixopac, dL_dmean2D, dL_dconic,
                                         dL_dopacity, dL_dcolor, dL_dnormal, dL_ddepth, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale,
                                         dL_drot, dL_dviewmat, dL_dprojmat, dL_dcampos, debug != 0, config);
    return hipDeviceSynchronize() == hipSuccess ? 0 : -1;
}

extern "C" int ref_rast_mark_visible(int P, float *means3D, float *viewmatrix, float *projmatrix, bool *present)
{
    CudaRasterizer::Rasterizer::markVisible(P, means3D, viewmatrix, projmatrix, present);
    return hipDeviceSynchronize() == hipSuccess ? 0 : -1;
}

// Copies the reference's internal state (located with its own fromChunk carving) into caller device buffers; any may be NULL.
extern "C" int ref_rast_state(void *h_, float *means2D, float *depths, float *conic_opacity, float *normal, float *rgb,
                              float *cov3D, uint32_t *tiles_touched, uint32_t *point_offsets, uint64_t *keys_unsorted,
                              uint32_t *vals_unsorted, uint64_t *keys_sorted, uint32_t *point_list, uint32_t *ranges,
                              float *final_T, uint32_t *n_contrib)
{
    RefRast *h = static_cast<RefRast *>(h_);
    const size_t P = h->P, R = h->R, N = (size_t)h->W * h->H;
    const size_t T = (size_t)((h->W + 15) / 16) * ((h->H + 15) / 16);
    char *c = h->geom.ptr;
    CudaRasterizer::GeometryState g = CudaRasterizer::GeometryState::fromChunk(c, P);
    c = h->img.ptr;
    CudaRasterizer::ImageState im = CudaRasterizer::ImageState::fromChunk(c, N);
    auto cp = [](void *dst, const void *src, size_t n) { return !dst || n == 0 || hipMemcpy(dst, src, n, hipMemcpyDeviceToDevice) == hipSuccess; };
    bool ok = cp(means2D, g.means2D, P * 8) && cp(depths, g.depths, P * 4) && cp(conic_opacity, g.conic_opacity, P * 16) &&
              cp(normal, g.normal, P * 12) && cp(rgb, g.rgb, P * 12) && cp(cov3D, g.cov3D, P * 24) &&
              cp(tiles_touched, g.tiles_touched, P * 4) && cp(point_offsets, g.point_offsets, P * 4) &&
              cp(ranges, im.ranges, T * 8) && cp(final_T, im.accum_alpha, N * 4) && cp(n_contrib, im.n_contrib, N * 4);
    if (R > 0 && h->binning.ptr) {
        c = h->binning.ptr;
        CudaRasterizer::BinningState b = CudaRasterizer::BinningState::fromChunk(c, R);
        ok = ok && cp(keys_unsorted, b.point_list_keys_unsorted, R * 8) && cp(vals_unsorted, b.point_list_unsorted, R * 4) &&
             cp(keys_sorted, b.point_list_keys, R * 8) && cp(point_list, b.point_list, R * 4);
    }
    return ok ? 0 : -1;
}

// ================================================================================================================================
// The third corner of the accuracy triangle (round 6): the formulas of the reference's backward blend (DGR/cuda_rasterizer/backward.cu:529-858,
// restated line by line -- one thread per pixel, no staging: the block-cooperative fetch of :629-647 only affects scheduling) evaluated over
// the reference's OWN forward state (its GeometryState / BinningState / ImageState, not a copy), with the per-Gaussian sums in float64 and
// the per-pixel arithmetic either in the reference's float (mode 1: "order-free" -- what the reference would give if its float atomics had no
// order) or in float64 as well (mode 2: "f64" -- the value the formulas define).  The skip decisions (:653-680) are the FLOAT ones in both
// modes, with the device expf the reference's own kernels call: they must be the decisions the forward made when it counted n_contrib.
// The 13 sums, rounded once to float, then go through the reference's own BACKWARD::preprocess (float, as the reference runs it).
// TEST INFRASTRUCTURE: scripts/r6_c5_triangle.py, tests/test_reference_build_gpu.py.
// ================================================================================================================================
#include "backward.h"

namespace {

__device__ __forceinline__ double wave_sum(double v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

template <typename R>
__global__ void __launch_bounds__(256) wide_backward_kernel(
    const uint2 *ranges, const uint32_t *point_list, int W, int H, const float *bg_color, const float2 *points_xy_image,
    const float4 *conic_opacity, const float *colors, const float *normal, const float *depth, const float *Jinv,
    const float *final_Ts, const float *final_D, const uint32_t *n_contrib, const float *dL_dpixcolor, const float *dL_dpixnormal,
    const float *dL_dpixdepth, const float *dL_dpixopac, double *acc /* [P][13] */, const float *config)
{
    constexpr int C = 3;
    const uint32_t horizontal_blocks = (W + 15) / 16;
    const uint32_t px = blockIdx.x * 16 + threadIdx.x, py = blockIdx.y * 16 + threadIdx.y;
    const bool inside = px < (uint32_t)W && py < (uint32_t)H;
    const uint32_t pix_id = inside ? W * py + px : 0;
    const float pixf_x = (float)px, pixf_y = (float)py;
    const uint2 range = ranges[blockIdx.y * horizontal_blocks + blockIdx.x];
    const int toDo = range.y - range.x;
    const bool surface = config[0] > 0, per_pixel_depth = config[2] > 0, normalize_depth = config[1] > 0;

    const R T_final = inside ? final_Ts[pix_id] : 0;                                  // :595
    const R D_final = inside && normalize_depth ? final_D[pix_id] : 0;                // :597
    R T = T_final;
    uint32_t contributor = toDo;                                                      // :603
    const int last_contributor = inside ? n_contrib[pix_id] : 0;                      // :604
    R accum_rec[C] = {0}, accum_rec_n[3] = {0}, accum_rec_d = 0;
    R dL_dpixC[C] = {0}, dL_dpixN[3] = {0}, dL_dpixD = 0, dL_dpixO = 0;
    if (inside) {
        for (int i = 0; i < C; i++) dL_dpixC[i] = dL_dpixcolor[i * H * W + pix_id];
        for (int i = 0; i < 3; i++) dL_dpixN[i] = dL_dpixnormal[i * H * W + pix_id];
        dL_dpixD = dL_dpixdepth[pix_id];
        dL_dpixO = dL_dpixopac[pix_id];
    }
    R last_alpha = 0, last_color[C] = {0}, last_normal[3] = {0}, last_depth = 0;
    const R ddelx_dx = (R)0.5 * W, ddely_dy = (R)0.5 * H;                             // :622-623

    for (int k = 0; k < toDo; k++) {                                                  // every lane walks the whole list: the sums are taken
        contributor--;                                                                // over the wavefront before they leave
        const uint32_t global_id = point_list[range.y - 1 - k];                       // :634
        double v[13];
        for (int q = 0; q < 13; q++) v[q] = 0.0;
        bool live = inside && (int64_t)contributor < (int64_t)last_contributor;       // :653
        float2 xy = points_xy_image[global_id];
        const float4 con_o = conic_opacity[global_id];
        const float dxf = xy.x - pixf_x, dyf = xy.y - pixf_y;
        // the decisions, in the reference's float arithmetic (:660-680)
        const float dist_f = (con_o.x * dxf * dxf + con_o.z * dyf * dyf) + 2 * con_o.y * dxf * dyf;
        const float power_f = -0.5f * dist_f;
        if (power_f > 0.0f) live = false;
        const float G_f = exp(power_f);
        const float alpha_f = min(0.99f, con_o.w * G_f);
        if (alpha_f < 1.0f / 255.0f) live = false;
        if (live) {
            R dx, dy, G, alpha;
            if constexpr (sizeof(R) == 4) { dx = dxf; dy = dyf; G = G_f; alpha = alpha_f; }
            else {
                dx = (double)xy.x - (double)pixf_x; dy = (double)xy.y - (double)pixf_y;
                const double dist = ((double)con_o.x * dx * dx + (double)con_o.z * dy * dy) + 2 * (double)con_o.y * dx * dy;
                G = exp(-0.5 * dist);
                alpha = fmin(0.99, (double)con_o.w * G);
            }
            T = T / ((R)1 - alpha);                                                   // :683
            const R dchannel_dcolor = alpha * T;
            R Jv[10] = {0};
            if (surface && per_pixel_depth) for (int ch = 0; ch < 10; ch++) Jv[ch] = Jinv[global_id * 10 + ch];
            R dL_dalpha = 0;
            for (int ch = 0; ch < C; ch++) {                                          // :698-713
                const R c_cur = colors[global_id * C + ch];
                accum_rec[ch] = last_alpha * last_color[ch] + ((R)1 - last_alpha) * accum_rec[ch];
                last_color[ch] = c_cur;
                const R dL_dchannel = dL_dpixC[ch];
                dL_dalpha += (c_cur - accum_rec[ch]) * dL_dchannel;
                v[6 + ch] = dchannel_dcolor * dL_dchannel;
            }
            if (surface) {                                                            // :715-731
                for (int ch = 0; ch < 3; ch++) {
                    const R n_cur = normal[global_id * 3 + ch];
                    accum_rec_n[ch] = last_alpha * last_normal[ch] + ((R)1 - last_alpha) * accum_rec_n[ch];
                    last_normal[ch] = n_cur;
                    const R dL_dchannel = dL_dpixN[ch];
                    dL_dalpha += (n_cur - accum_rec_n[ch]) * dL_dchannel;
                    v[9 + ch] = dchannel_dcolor * dL_dchannel * 10;
                }
            }
            {                                                                         // :758-784
                R d_cur = depth[global_id];
                if (surface && per_pixel_depth) {                                     // auxiliary.h:390-397 (.z)
                    const R dif_u0 = dx * Jv[0] + dy * Jv[1], dif_u1 = dx * Jv[2] + dy * Jv[3];
                    d_cur -= dif_u0 * Jv[6] + dif_u1 * Jv[9];
                }
                accum_rec_d = last_alpha * last_depth + ((R)1 - last_alpha) * accum_rec_d;
                last_depth = d_cur;
                R dL_dchannel = dL_dpixD, dL_dalpha_depth = 0;
                if (normalize_depth) {
                    dL_dchannel /= ((R)1 - T_final);
                    dL_dalpha_depth += dL_dpixD * D_final / ((R)1 - T_final) / ((R)1 - T_final) * -T_final / ((R)1 - alpha) / T;
                }
                dL_dalpha_depth += (d_cur - accum_rec_d) * dL_dchannel;
                v[12] = dchannel_dcolor * dL_dchannel * 1;
                dL_dalpha += dL_dalpha_depth;
            }
            dL_dalpha *= T;                                                           // :788
            dL_dalpha += dL_dpixO * T_final / ((R)1 - alpha);                         // :791
            last_alpha = alpha;
            R bg_dot_dpixel = 0;
            for (int i = 0; i < C; i++) bg_dot_dpixel += (R)bg_color[i] * dL_dpixC[i];
            dL_dalpha += (-T_final / ((R)1 - alpha)) * bg_dot_dpixel;                 // :801
            if (!normalize_depth) dL_dalpha += (-T_final / ((R)1 - alpha)) * (10 * dL_dpixD);
            R dL_ddist = 0;
            dL_ddist += dL_dalpha * (R)con_o.w * (R)-0.5f * G;                        // :823
            R ndc_x = dL_ddist * 2 * ((R)con_o.x * dx + (R)con_o.y * dy) * ddelx_dx;
            R ndc_y = dL_ddist * 2 * ((R)con_o.z * dy + (R)con_o.y * dx) * ddely_dy;
            if (surface && per_pixel_depth) {                                         // :838-841
                ndc_x += 1 * -dL_dpixD * (Jv[6] * Jv[0] + Jv[9] * Jv[2]);
                ndc_y += 1 * -dL_dpixD * (Jv[6] * Jv[1] + Jv[9] * Jv[3]);
            }
            v[0] = ndc_x; v[1] = ndc_y;
            v[2] = dL_ddist * (dx * dx); v[3] = dL_ddist * (1 * dx * dy); v[4] = dL_ddist * (dy * dy);
            v[5] = G * dL_dalpha;                                                     // :854
        }
        if (__ballot(live) == 0ull) continue;
        for (int q = 0; q < 13; q++) {
            const double s = wave_sum(v[q]);
            if ((threadIdx.x + 16 * threadIdx.y) % 64 == 0 && s != 0.0) atomicAdd(acc + (size_t)global_id * 13 + q, s);
        }
    }
}

__global__ void wide_narrow_kernel(int P, const double *acc, float *dL_dmean2D, float *dL_dconic, float *dL_dopacity, float *dL_dcolor,
                                   float *dL_dnormal, float *dL_ddepth)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const double *a = acc + (size_t)i * 13;
    dL_dmean2D[3 * i + 0] = (float)a[0]; dL_dmean2D[3 * i + 1] = (float)a[1]; dL_dmean2D[3 * i + 2] = 0.f;
    dL_dconic[4 * i + 0] = (float)a[2]; dL_dconic[4 * i + 1] = (float)a[3]; dL_dconic[4 * i + 2] = 0.f; dL_dconic[4 * i + 3] = (float)a[4];
    dL_dopacity[i] = (float)a[5];
    for (int ch = 0; ch < 3; ch++) { dL_dcolor[3 * i + ch] = (float)a[6 + ch]; dL_dnormal[3 * i + ch] = (float)a[9 + ch]; }
    dL_ddepth[i] = (float)a[12];
}

}  // namespace

// mode 1: the reference's float arithmetic per pixel, float64 sums; mode 2: float64 per pixel as well.  Same arguments as ref_rast_backward.
extern "C" int ref_rast_backward_wide(void *h_, int mode, int D, int M, const float *background, const float *means3D, const float *shs,
                                      const float *colors_precomp, const float *scales, float scale_modifier,
                                      const float *rotations, const float *cov3D_precomp, const float *viewmatrix,
                                      const float *projmatrix, const float *campos, const float *prcppoint, const float *patchbbox,
                                      float tan_fovx, float tan_fovy, const int *radii, const float *dL_dpixcolor,
                                      const float *dL_dpixnormal, const float *dL_dpixdepth, const float *dL_dpixopac,
                                      float *dL_dmean2D, float *dL_dconic, float *dL_dopacity, float *dL_dcolor, float *dL_dnormal,
                                      float *dL_ddepth, float *dL_dmean3D, float *dL_dcov3D, float *dL_dsh, float *dL_dscale,
                                      float *dL_drot, float *dL_dviewmat, float *dL_dprojmat, float *dL_dcampos, int debug,
                                      float *config)
{
    RefRast *h = static_cast<RefRast *>(h_);
    const int P = h->P, W = h->W, H = h->H;
    char *c = h->geom.ptr;
    CudaRasterizer::GeometryState g = CudaRasterizer::GeometryState::fromChunk(c, P);
    c = h->img.ptr;
    CudaRasterizer::ImageState im = CudaRasterizer::ImageState::fromChunk(c, (size_t)W * H);
    c = h->binning.ptr;
    CudaRasterizer::BinningState b = CudaRasterizer::BinningState::fromChunk(c, h->R);
    if (radii == nullptr) radii = g.internal_radii;
    double *acc = nullptr;
    if (hipMalloc(&acc, (size_t)P * 13 * sizeof(double)) != hipSuccess) return -1;
    (void)hipMemset(acc, 0, (size_t)P * 13 * sizeof(double));
    const dim3 grid((W + 15) / 16, (H + 15) / 16), block(16, 16);
    const float *color_ptr = colors_precomp ? colors_precomp : g.rgb;
    if (h->R > 0) {
        if (mode == 1)
            hipLaunchKernelGGL(wide_backward_kernel<float>, grid, block, 0, 0, im.ranges, b.point_list, W, H, background, g.means2D,
                               g.conic_opacity, color_ptr, g.normal, g.depths, g.Jinv, im.accum_alpha, im.accum_depth, im.n_contrib,
                               dL_dpixcolor, dL_dpixnormal, dL_dpixdepth, dL_dpixopac, acc, config);
        else
            hipLaunchKernelGGL(wide_backward_kernel<double>, grid, block, 0, 0, im.ranges, b.point_list, W, H, background, g.means2D,
                               g.conic_opacity, color_ptr, g.normal, g.depths, g.Jinv, im.accum_alpha, im.accum_depth, im.n_contrib,
                               dL_dpixcolor, dL_dpixnormal, dL_dpixdepth, dL_dpixopac, acc, config);
    }
    hipLaunchKernelGGL(wide_narrow_kernel, dim3((P + 255) / 256), dim3(256), 0, 0, P, acc, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor,
                       dL_dnormal, dL_ddepth);
    const float focal_y = H / (2.0f * tan_fovy), focal_x = W / (2.0f * tan_fovx);
    const float *cov3D_ptr = cov3D_precomp ? cov3D_precomp : g.cov3D;
    BACKWARD::preprocess(P, D, M, (float3 *)means3D, radii, shs, g.clamped, (glm::vec3 *)scales, (glm::vec4 *)rotations, scale_modifier,
                         cov3D_ptr, viewmatrix, projmatrix, focal_x, focal_y, tan_fovx, tan_fovy, (glm::vec3 *)campos,
                         (float3 *)dL_dmean2D, dL_dconic, (glm::vec3 *)dL_dmean3D, dL_dcolor, dL_dnormal, dL_ddepth, dL_dcov3D, dL_dsh,
                         (glm::vec3 *)dL_dscale, (glm::vec4 *)dL_drot, dL_dviewmat, dL_dprojmat, dL_dcampos, config);
    const bool ok = hipDeviceSynchronize() == hipSuccess;
    (void)hipFree(acc);
    return ok ? 0 : -1;
}
