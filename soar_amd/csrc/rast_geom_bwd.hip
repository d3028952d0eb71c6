// rast_geom_bwd.hip -- per-Gaussian backward stage for gfx950: one fused kernel that turns the 64-byte
// accumulation rows of the backward blend into the gradients of the rasterizer inputs.
//
// Replaces computeCov2DCUDA (DGR/cuda_rasterizer/backward.cu:163-322), preprocessCUDA backward (:437-526),
// computeCov3D backward (:326-432) and the SH backward (:20-158).  The reference runs two kernels that
// communicate through dL_dmeans / dL_dcov3D in HBM; here one thread keeps everything in registers and every
// output element is written exactly once (also zeros for culled Gaussians, so the caller need not memset).
#include "geom_bwd_point.h"

namespace soar {

namespace {

__global__ void __launch_bounds__(256) geometry_backward_kernel(Batch<GeomBwdArgs> batch)
{
    const GeomBwdArgs &a = batch.v[blockIdx.y];
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = idx < a.P;
    const bool active = in_range && a.radii[idx] > 0;    // backward.cu:178, :465

    GeomBwdPoint o;
    float cam_view[16], cam_proj[16], cam_pos[3];
    if (a.lrn_cam) {
#pragma unroll
        for (int k = 0; k < 16; k++) { cam_view[k] = 0.f; cam_proj[k] = 0.f; }
        cam_pos[0] = cam_pos[1] = cam_pos[2] = 0.f;
    }
    geometry_backward_point<true>(a, idx, active, o, cam_view, cam_proj, cam_pos);
    const float (&acc)[ACC_STRIDE] = o.acc;
    const float (&g_mean)[3] = o.g_mean;
    const float (&g_cov)[6] = o.g_cov;
    const float (&g_scale)[3] = o.g_scale;
    const float (&g_rot)[4] = o.g_rot;
    if (a.dL_docc && in_range) a.dL_docc[idx] = acc[13];      // (every Gaussian: nothing else writes this output)

    if (in_range) {
        a.dL_dmeans2D[3 * idx + 0] = acc[0];
        a.dL_dmeans2D[3 * idx + 1] = acc[1];
        a.dL_dmeans2D[3 * idx + 2] = 0.f;                    // z is never written by the reference (Appendix A #26)
        a.dL_dcolors[3 * idx + 0] = acc[6];
        a.dL_dcolors[3 * idx + 1] = acc[7];
        a.dL_dcolors[3 * idx + 2] = acc[8];
        a.dL_dopacity[idx] = acc[5];
#pragma unroll
        for (int k = 0; k < 3; k++) a.dL_dmeans3D[3 * idx + k] = g_mean[k];
#pragma unroll
        for (int k = 0; k < 6; k++) a.dL_dcov3D[6 * idx + k] = g_cov[k];
#pragma unroll
        for (int k = 0; k < 3; k++) a.dL_dscales[3 * idx + k] = g_scale[k];
        reinterpret_cast<float4 *>(a.dL_drots)[idx] = make_float4(g_rot[0], g_rot[1], g_rot[2], g_rot[3]);
        if (a.dL_dsh && !(active && a.shs)) {
            float *dsh = a.dL_dsh + (size_t)idx * a.M * 3;
            for (int k = 0; k < a.M * 3; k++) dsh[k] = 0.f;
        } else if (a.dL_dsh && active && a.shs) {
            // coefficients above the active degree are not touched by the SH backward: zero them
            const int used = (a.D + 1) * (a.D + 1);
            float *dsh = a.dL_dsh + (size_t)idx * a.M * 3;
            for (int k = used * 3; k < a.M * 3; k++) dsh[k] = 0.f;
        }
    }

    if (a.lrn_cam) {   // wave-level sums, one atomic per wave and element (the reference: one per Gaussian)
#pragma unroll
        for (int k = 0; k < 16; k++) {
            wave_atomic_add(a.dL_dviewmat + k, cam_view[k]);
            wave_atomic_add(a.dL_dprojmat + k, cam_proj[k]);
        }
#pragma unroll
        for (int k = 0; k < 3; k++) wave_atomic_add(a.dL_dcampos + k, cam_pos[k]);
    }
}

}  // namespace

int launch_geometry_backward(const SoarRastParams &prm, const float *means3D, const int32_t *radii, const float *shs,
                             const float *scales, const float *rotations, const float *cov3D_precomp, const GeomBuf &g,
                             const float *acc, float *dL_dmeans2D, float *dL_dcolors, float *dL_dopacity,
                             float *dL_dmeans3D, float *dL_dcov3D, float *dL_dsh, float *dL_dscales, float *dL_drotations,
                             float *dL_dviewmat, float *dL_dprojmat, float *dL_dcampos, bool zero_camera_grads, hipStream_t stream,
                             float *dL_docc)
{
    GeomBwdArgs a;
    fill_geom_bwd_args(a, prm, means3D, radii, shs, scales, rotations, cov3D_precomp, g, acc);
    a.dL_dmeans2D = dL_dmeans2D; a.dL_dcolors = dL_dcolors; a.dL_dopacity = dL_dopacity; a.dL_dmeans3D = dL_dmeans3D;
    a.dL_dcov3D = dL_dcov3D; a.dL_dsh = dL_dsh; a.dL_dscales = dL_dscales; a.dL_drots = dL_drotations;
    a.dL_dviewmat = dL_dviewmat; a.dL_dprojmat = dL_dprojmat; a.dL_dcampos = dL_dcampos;
    a.dL_docc = dL_docc;
    if (zero_camera_grads) {
        const ZeroRange zr[3] = {{dL_dviewmat, 16 * sizeof(float)}, {dL_dprojmat, 16 * sizeof(float)}, {dL_dcampos, 3 * sizeof(float)}};
        if (launch_zero_ranges(zr, 3, stream)) return 1;
    }
    StageTimer timer(ST_GEOM_BWD, stream);
    SOAR_LAUNCH_BATCHED(geometry_backward_kernel, dim3((prm.P + 255) / 256), dim3(256), 0, stream, a);
    SOAR_LAUNCH_OK("geometry_backward", stream, prm.debug);
    return 0;
}

}  // namespace soar
