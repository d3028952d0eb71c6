// rast_preprocess.hip -- per-Gaussian forward stage of the Gaussian-surfel rasterizer for gfx950.
//
// Replaces preprocessCUDA (DGR/cuda_rasterizer/forward.cu:205-385) together with its helpers
// (forward.cu:20-202, auxiliary.h:42-388).  One thread per Gaussian; the output is one 64-byte
// GaussRec per Gaussian (the gather granule of the blend kernels) plus cov3D / radii / tile counts.
//
// This translation unit is compiled with -ffp-contract=off: every value that feeds an integer
// decision (culls, radius, tile rectangle, depth key) is evaluated in exactly the operation order of
// the reference source, so radii / tiles_touched / sort keys are bit-identical to an IEEE evaluation
// of the reference (SURVEY.md section 7 "Bit-exact indexing").
#include "preprocess_point.h"

namespace soar {

namespace {

__global__ void __launch_bounds__(256) preprocess_kernel(Batch<PreArgs> batch)
{
    const PreArgs &a = batch.v[blockIdx.y];
    // lanes past the end redo the last Gaussian (identical stores) so that whole wavefronts reach the reduction of the statistics
    const int idx_raw = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = idx_raw < a.P;
    const int idx = in_range ? idx_raw : a.P - 1;
    const float px3 = a.means3D[3 * idx], py3 = a.means3D[3 * idx + 1], pz3 = a.means3D[3 * idx + 2];
    float4 q = make_float4(1.f, 0.f, 0.f, 0.f);
    if (a.rotations) q = reinterpret_cast<const float4 *>(a.rotations)[idx];
    PrePoint o;
    preprocess_point(a, idx, in_range, px3, py3, pz3, a.rotations != nullptr, q, o);
    // (one statistics row per wavefront = per 64 consecutive Gaussians)
    const int first = idx_raw - (int)(threadIdx.x & 63);
    preprocess_store(a, idx, in_range, first < a.P ? first / WAVE : -1, o);
    if (!a.prefiltered && blockIdx.x == 0 && threadIdx.x == 0) a.header[H_PREFILTER_VIOLATIONS] = 0u;    // (nobody zeroed the header)
}

}  // namespace

int launch_preprocess(const SoarRastParams &prm, const float *means3D, const float *shs, const float *colors_precomp,
                      const float *opacities, const float *scales, const float *rotations, const float *cov3D_precomp,
                      GeomBuf &g, int32_t *radii, hipStream_t stream)
{
    PreArgs a;
    fill_pre_args(a, prm, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, g, radii);
    const int threads = 256;
    const int blocks = (prm.P + threads - 1) / threads;
    StageTimer timer(ST_PREPROCESS, stream);
    SOAR_LAUNCH_BATCHED(preprocess_kernel, dim3(blocks), dim3(threads), 0, stream, a);
    SOAR_LAUNCH_OK("preprocess", stream, prm.debug);
    return 0;
}

}  // namespace soar
