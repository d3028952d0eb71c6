// geom_bwd_point.h -- the per-Gaussian backward of the rasterizer as a device function: what geometry_backward_kernel
// (rast_geom_bwd.hip) runs per thread, shared with the fused kernel of lbs.hip that carries the result straight on through the warp's
// backward (geom_warp_backward_frames_kernel, round 6) without a round trip through memory.
//
// Replaces computeCov2DCUDA (DGR/cuda_rasterizer/backward.cu:163-322), preprocessCUDA backward (:437-526), computeCov3D backward
// (:326-432) and the SH backward (:20-158).
//
// Floating point: every function here evaluates its expressions as written -- no FMA contraction (`#pragma clang fp contract(off)`
// at the top of each body; rast_geom_bwd.hip is compiled with -ffp-contract=off as a whole, lbs.hip is not): the quaternion gradient
// has cancellation-prone expressions whose rounding must follow an IEEE evaluation of the reference source, in whichever kernel the
// function is inlined.
#pragma once
#include "soar_common.h"
#include "soar_m3.h"

namespace soar {

namespace {

__constant__ float bSH_C0 = 0.28209479177387814f;
__constant__ float bSH_C1 = 0.4886025119029199f;
__constant__ float bSH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                                0.5462742152960396f};
__constant__ float bSH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                                -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

struct GeomBwdArgs {
    int P, D, M, W, H;
    int surface, lrn_cam;
    float tanfovx, tanfovy, h_x, h_y, scale_modifier;
    const float *means3D, *shs, *scales, *rotations, *cov3D;   // cov3D: precomputed input or the forward's
    const int32_t *radii;
    const uint8_t *clamped;
    const float *view, *proj, *campos;
    const float *acc;
    float *dL_dmeans2D, *dL_dcolors, *dL_dopacity, *dL_dmeans3D, *dL_dcov3D, *dL_dsh, *dL_dscales, *dL_drots;
    float *dL_dviewmat, *dL_dprojmat, *dL_dcampos;
    float *dL_docc;          // != NULL: slot 13 of the rows is the gradient of the Gaussian's occlusion value (soar_rast_backward_occ)
};

// sum a per-thread value over the wave and let one lane issue the global atomic (camera gradients only)
__device__ __forceinline__ void wave_atomic_add(float *dst, float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0 && v != 0.f) atomicAdd(dst, v);
}

// what one Gaussian's backward leaves: the accumulation row as read, and the gradients of its (posed) inputs
struct GeomBwdPoint {
    float acc[ACC_STRIDE];
    float g_mean[3], g_cov[6], g_scale[3], g_rot[4];
};

// CAM_SH = false: explicit colours, no camera gradients (the per-frame training path) -- the SH and camera branches are not compiled in
template <bool CAM_SH>
__device__ __forceinline__ void geometry_backward_point(const GeomBwdArgs &a, const int idx, const bool active, GeomBwdPoint &o,
                                                        float *cam_view, float *cam_proj, float *cam_pos)
{
#pragma clang fp contract(off)
    float (&acc)[ACC_STRIDE] = o.acc;
    float (&g_mean)[3] = o.g_mean;
    float (&g_cov)[6] = o.g_cov;
    float (&g_scale)[3] = o.g_scale;
    float (&g_rot)[4] = o.g_rot;
#pragma unroll
    for (int k = 0; k < ACC_STRIDE; k++) acc[k] = 0.f;
    if (active) {
        const float4 *row = reinterpret_cast<const float4 *>(a.acc + (size_t)idx * ACC_STRIDE);
        float4 r0 = row[0], r1 = row[1], r2 = row[2], r3 = row[3];
        acc[0] = r0.x; acc[1] = r0.y; acc[2] = r0.z; acc[3] = r0.w;
        acc[4] = r1.x; acc[5] = r1.y; acc[6] = r1.z; acc[7] = r1.w;
        acc[8] = r2.x; acc[9] = r2.y; acc[10] = r2.z; acc[11] = r2.w;
        acc[12] = r3.x; acc[13] = r3.y;
    }
#pragma unroll
    for (int k = 0; k < 3; k++) { g_mean[k] = 0.f; g_scale[k] = 0.f; }
#pragma unroll
    for (int k = 0; k < 6; k++) g_cov[k] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; k++) g_rot[k] = 0.f;
    if (active) {
        const float *V = a.view, *PM = a.proj;
        const float mx = a.means3D[3 * idx], my = a.means3D[3 * idx + 1], mz = a.means3D[3 * idx + 2];

        // ---------------- conic -> cov2D -> cov3D / mean (backward.cu:163-322) ----------------
        const float *cov3D = a.cov3D + 6 * idx;
        const float dcon0 = acc[2], dcon1 = acc[3], dcon2 = acc[4];          // float4 slots x, y, w (:187)
        float t0 = V[0] * mx + V[4] * my + V[8] * mz + V[12];
        float t1 = V[1] * mx + V[5] * my + V[9] * mz + V[13];
        float t2 = V[2] * mx + V[6] * my + V[10] * mz + V[14];
        const float limx = 1.3f * a.tanfovx, limy = 1.3f * a.tanfovy;
        const float txtz = t0 / t2, tytz = t1 / t2;
        t0 = fminf(limx, fmaxf(-limx, txtz)) * t2;
        t1 = fminf(limy, fmaxf(-limy, tytz)) * t2;
        const float x_grad_mul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
        const float y_grad_mul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
        const float J0 = a.h_x / t2, J1 = -(a.h_x * t0) / (t2 * t2), J2 = a.h_y / t2, J3 = -(a.h_y * t1) / (t2 * t2);
        M3 Jm, Wm, Vrk;
        Jm.e[0][0] = J0; Jm.e[0][1] = 0.f; Jm.e[0][2] = J1;
        Jm.e[1][0] = 0.f; Jm.e[1][1] = J2; Jm.e[1][2] = J3;
        Jm.e[2][0] = 0.f; Jm.e[2][1] = 0.f; Jm.e[2][2] = 0.f;
        Wm.e[0][0] = V[0]; Wm.e[0][1] = V[4]; Wm.e[0][2] = V[8];
        Wm.e[1][0] = V[1]; Wm.e[1][1] = V[5]; Wm.e[1][2] = V[9];
        Wm.e[2][0] = V[2]; Wm.e[2][1] = V[6]; Wm.e[2][2] = V[10];
        Vrk.e[0][0] = cov3D[0]; Vrk.e[0][1] = cov3D[1]; Vrk.e[0][2] = cov3D[2];
        Vrk.e[1][0] = cov3D[1]; Vrk.e[1][1] = cov3D[3]; Vrk.e[1][2] = cov3D[4];
        Vrk.e[2][0] = cov3D[2]; Vrk.e[2][1] = cov3D[4]; Vrk.e[2][2] = cov3D[5];
        const M3 T = m3mul(Wm, Jm);
        const M3 c2 = m3mul(m3mul(m3t(T), m3t(Vrk)), T);
        const float ca = c2.e[0][0] + 0.3f, cb = c2.e[0][1], cc = c2.e[1][1] + 0.3f;
        const float denom = ca * cc - cb * cb;
        float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
        const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        if (denom2inv != 0.f) {
            dL_da = denom2inv * (-cc * cc * dcon0 + 2 * cb * cc * dcon1 + (denom - ca * cc) * dcon2);
            dL_dc = denom2inv * (-ca * ca * dcon2 + 2 * ca * cb * dcon1 + (denom - ca * cc) * dcon0);
            dL_db = denom2inv * 2 * (cb * cc * dcon0 - (denom + 2 * cb * cb) * dcon1 + ca * cb * dcon2);
            g_cov[0] = (T.e[0][0] * T.e[0][0] * dL_da + T.e[0][0] * T.e[1][0] * dL_db + T.e[1][0] * T.e[1][0] * dL_dc);
            g_cov[3] = (T.e[0][1] * T.e[0][1] * dL_da + T.e[0][1] * T.e[1][1] * dL_db + T.e[1][1] * T.e[1][1] * dL_dc);
            g_cov[5] = (T.e[0][2] * T.e[0][2] * dL_da + T.e[0][2] * T.e[1][2] * dL_db + T.e[1][2] * T.e[1][2] * dL_dc);
            g_cov[1] = 2 * T.e[0][0] * T.e[0][1] * dL_da + (T.e[0][0] * T.e[1][1] + T.e[0][1] * T.e[1][0]) * dL_db +
                       2 * T.e[1][0] * T.e[1][1] * dL_dc;
            g_cov[2] = 2 * T.e[0][0] * T.e[0][2] * dL_da + (T.e[0][0] * T.e[1][2] + T.e[0][2] * T.e[1][0]) * dL_db +
                       2 * T.e[1][0] * T.e[1][2] * dL_dc;
            g_cov[4] = 2 * T.e[0][2] * T.e[0][1] * dL_da + (T.e[0][1] * T.e[1][2] + T.e[0][2] * T.e[1][1]) * dL_db +
                       2 * T.e[1][1] * T.e[1][2] * dL_dc;
        }
        // dL/dT (upper 2x3), :260-271
        float dT0[3], dT1[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float r0 = T.e[0][0] * Vrk.e[k][0] + T.e[0][1] * Vrk.e[k][1] + T.e[0][2] * Vrk.e[k][2];
            const float r1 = T.e[1][0] * Vrk.e[k][0] + T.e[1][1] * Vrk.e[k][1] + T.e[1][2] * Vrk.e[k][2];
            dT0[k] = 2 * r0 * dL_da + r1 * dL_db;
            dT1[k] = 2 * r1 * dL_dc + r0 * dL_db;
        }
        const float dL_dJ00 = Wm.e[0][0] * dT0[0] + Wm.e[0][1] * dT0[1] + Wm.e[0][2] * dT0[2];
        const float dL_dJ02 = Wm.e[2][0] * dT0[0] + Wm.e[2][1] * dT0[1] + Wm.e[2][2] * dT0[2];
        const float dL_dJ11 = Wm.e[1][0] * dT1[0] + Wm.e[1][1] * dT1[1] + Wm.e[1][2] * dT1[2];
        const float dL_dJ12 = Wm.e[2][0] * dT1[0] + Wm.e[2][1] * dT1[1] + Wm.e[2][2] * dT1[2];
        const float tz = 1.f / t2, tz2 = tz * tz, tz3 = tz2 * tz;
        if ((CAM_SH && a.lrn_cam)) {   // :286-302
            cam_view[0] += dT0[0] * J0; cam_view[1] += dT1[0] * J2; cam_view[2] += dT0[0] * J1 + dT1[0] * J3;
            cam_view[4] += dT0[1] * J0; cam_view[5] += dT1[1] * J2; cam_view[6] += dT0[1] * J1 + dT1[1] * J3;
            cam_view[8] += dT0[2] * J0; cam_view[9] += dT1[2] * J2; cam_view[10] += dT0[2] * J1 + dT1[2] * J3;
        }
        const float dL_dtx = x_grad_mul * -a.h_x * tz2 * dL_dJ02;
        const float dL_dty = y_grad_mul * -a.h_y * tz2 * dL_dJ12;
        const float dL_dtz = -a.h_x * tz2 * dL_dJ00 - a.h_y * tz2 * dL_dJ11 + (2 * a.h_x * t0) * tz3 * dL_dJ02 +
                             (2 * a.h_y * t1) * tz3 * dL_dJ12;
        g_mean[0] = V[0] * dL_dtx + V[1] * dL_dty + V[2] * dL_dtz;          // transformVec4x3Transpose, :316
        g_mean[1] = V[4] * dL_dtx + V[5] * dL_dty + V[6] * dL_dtz;
        g_mean[2] = V[8] * dL_dtx + V[9] * dL_dty + V[10] * dL_dtz;

        // ---------------- mean2D / depth -> mean3D (backward.cu:468-496) ----------------
        const float m_hw = PM[3] * mx + PM[7] * my + PM[11] * mz + PM[15];
        const float m_w = 1.0f / (m_hw + 0.0000001f);
        const float mul1 = (PM[0] * mx + PM[4] * my + PM[8] * mz + PM[12]) * m_w * m_w;
        const float mul2 = (PM[1] * mx + PM[5] * my + PM[9] * mz + PM[13]) * m_w * m_w;
        const float d2x = acc[0], d2y = acc[1];
        const float dL_dd = acc[12];
        g_mean[0] += ((PM[0] * m_w - PM[3] * mul1) * d2x + (PM[1] * m_w - PM[3] * mul2) * d2y) + dL_dd * V[2];
        g_mean[1] += ((PM[4] * m_w - PM[7] * mul1) * d2x + (PM[5] * m_w - PM[7] * mul2) * d2y) + dL_dd * V[6];
        g_mean[2] += ((PM[8] * m_w - PM[11] * mul1) * d2x + (PM[9] * m_w - PM[11] * mul2) * d2y) + dL_dd * V[10];
        if ((CAM_SH && a.lrn_cam)) {   // :499-517
            const float mm[3] = {mx, my, mz};
#pragma unroll
            for (int k = 0; k < 3; k++) {
                cam_proj[4 * k + 0] += d2x * mm[k] * m_w;
                cam_proj[4 * k + 1] += d2y * mm[k] * m_w;
                cam_proj[4 * k + 3] += d2x * -mul1 * mm[k] + d2y * -mul2 * mm[k];
                cam_view[4 * k + 2] += dL_dd * mm[k];
            }
            cam_proj[12] += d2x * m_w; cam_proj[13] += d2y * m_w; cam_proj[15] += d2x * -mul1 + d2y * -mul2;
            cam_view[14] += dL_dd;
        }

        // ---------------- colour -> SH (backward.cu:20-158) ----------------
        if (CAM_SH && a.shs) {
            const float *campos = a.campos;
            const float ox = mx - campos[0], oy = my - campos[1], oz = mz - campos[2];
            const float len = sqrtf(ox * ox + oy * oy + oz * oz);
            const float x = ox / len, y = oy / len, z = oz / len;
            const float *sh = a.shs + (size_t)idx * a.M * 3;
            float *dsh = a.dL_dsh + (size_t)idx * a.M * 3;
            float dRGB[3];
#pragma unroll
            for (int c = 0; c < 3; c++) dRGB[c] = a.clamped[3 * idx + c] ? 0.f : acc[6 + c];
            float ddir[3] = {0.f, 0.f, 0.f};
            const int deg = a.D;
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float g = dRGB[c];
                float ddx = 0.f, ddy = 0.f, ddz = 0.f;
                dsh[c] = bSH_C0 * g;
                if (deg > 0) {
                    dsh[3 + c] = -bSH_C1 * y * g; dsh[6 + c] = bSH_C1 * z * g; dsh[9 + c] = -bSH_C1 * x * g;
                    ddx = -bSH_C1 * sh[9 + c]; ddy = -bSH_C1 * sh[3 + c]; ddz = bSH_C1 * sh[6 + c];
                    if (deg > 1) {
                        dsh[12 + c] = bSH_C2[0] * xy * g; dsh[15 + c] = bSH_C2[1] * yz * g;
                        dsh[18 + c] = bSH_C2[2] * (2.f * zz - xx - yy) * g; dsh[21 + c] = bSH_C2[3] * xz * g;
                        dsh[24 + c] = bSH_C2[4] * (xx - yy) * g;
                        ddx += bSH_C2[0] * y * sh[12 + c] + bSH_C2[2] * 2.f * -x * sh[18 + c] + bSH_C2[3] * z * sh[21 + c] +
                               bSH_C2[4] * 2.f * x * sh[24 + c];
                        ddy += bSH_C2[0] * x * sh[12 + c] + bSH_C2[1] * z * sh[15 + c] + bSH_C2[2] * 2.f * -y * sh[18 + c] +
                               bSH_C2[4] * 2.f * -y * sh[24 + c];
                        ddz += bSH_C2[1] * y * sh[15 + c] + bSH_C2[2] * 2.f * 2.f * z * sh[18 + c] + bSH_C2[3] * x * sh[21 + c];
                        if (deg > 2) {
                            dsh[27 + c] = bSH_C3[0] * y * (3.f * xx - yy) * g; dsh[30 + c] = bSH_C3[1] * xy * z * g;
                            dsh[33 + c] = bSH_C3[2] * y * (4.f * zz - xx - yy) * g;
                            dsh[36 + c] = bSH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * g;
                            dsh[39 + c] = bSH_C3[4] * x * (4.f * zz - xx - yy) * g; dsh[42 + c] = bSH_C3[5] * z * (xx - yy) * g;
                            dsh[45 + c] = bSH_C3[6] * x * (xx - 3.f * yy) * g;
                            ddx += (bSH_C3[0] * sh[27 + c] * 3.f * 2.f * xy + bSH_C3[1] * sh[30 + c] * yz +
                                    bSH_C3[2] * sh[33 + c] * -2.f * xy + bSH_C3[3] * sh[36 + c] * -3.f * 2.f * xz +
                                    bSH_C3[4] * sh[39 + c] * (-3.f * xx + 4.f * zz - yy) + bSH_C3[5] * sh[42 + c] * 2.f * xz +
                                    bSH_C3[6] * sh[45 + c] * 3.f * (xx - yy));
                            ddy += (bSH_C3[0] * sh[27 + c] * 3.f * (xx - yy) + bSH_C3[1] * sh[30 + c] * xz +
                                    bSH_C3[2] * sh[33 + c] * (-3.f * yy + 4.f * zz - xx) + bSH_C3[3] * sh[36 + c] * -3.f * 2.f * yz +
                                    bSH_C3[4] * sh[39 + c] * -2.f * xy + bSH_C3[5] * sh[42 + c] * -2.f * yz +
                                    bSH_C3[6] * sh[45 + c] * -3.f * 2.f * xy);
                            ddz += (bSH_C3[1] * sh[30 + c] * xy + bSH_C3[2] * sh[33 + c] * 4.f * 2.f * yz +
                                    bSH_C3[3] * sh[36 + c] * 3.f * (2.f * zz - xx - yy) + bSH_C3[4] * sh[39 + c] * 4.f * 2.f * xz +
                                    bSH_C3[5] * sh[42 + c] * (xx - yy));
                        }
                    }
                }
                ddir[0] += ddx * g; ddir[1] += ddy * g; ddir[2] += ddz * g;
            }
            // through the direction normalisation (dnormvdv, auxiliary.h:114-124)
            const float sum2 = ox * ox + oy * oy + oz * oz;
            const float inv32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
            const float gx_ = ((sum2 - ox * ox) * ddir[0] - oy * ox * ddir[1] - oz * ox * ddir[2]) * inv32;
            const float gy_ = (-ox * oy * ddir[0] + (sum2 - oy * oy) * ddir[1] - oz * oy * ddir[2]) * inv32;
            const float gz_ = (-ox * oz * ddir[0] - oy * oz * ddir[1] + (sum2 - oz * oz) * ddir[2]) * inv32;
            g_mean[0] += gx_; g_mean[1] += gy_; g_mean[2] += gz_;
            if ((CAM_SH && a.lrn_cam)) { cam_pos[0] -= gx_; cam_pos[1] -= gy_; cam_pos[2] -= gz_; }
        }

        // ---------------- cov3D (+ normal) -> scale, quaternion (backward.cu:326-432) ----------------
        if (a.scales) {
            const float4 q = reinterpret_cast<const float4 *>(a.rotations)[idx];
            const float r = q.x, x = q.y, y = q.z, z = q.w;
            M3 R, S;
            R.e[0][0] = 1.f - 2.f * (y * y + z * z); R.e[0][1] = 2.f * (x * y - r * z); R.e[0][2] = 2.f * (x * z + r * y);
            R.e[1][0] = 2.f * (x * y + r * z); R.e[1][1] = 1.f - 2.f * (x * x + z * z); R.e[1][2] = 2.f * (y * z - r * x);
            R.e[2][0] = 2.f * (x * z - r * y); R.e[2][1] = 2.f * (y * z + r * x); R.e[2][2] = 1.f - 2.f * (x * x + y * y);
            const float s0 = a.scale_modifier * a.scales[3 * idx], s1 = a.scale_modifier * a.scales[3 * idx + 1],
                        s2 = a.scale_modifier * a.scales[3 * idx + 2];   // all three axes, also in surface mode (:354)
#pragma unroll
            for (int c = 0; c < 3; c++)
#pragma unroll
                for (int k = 0; k < 3; k++) S.e[c][k] = 0.f;
            S.e[0][0] = s0; S.e[1][1] = s1; S.e[2][2] = s2;
            M3 Mm = m3mul(S, R);
            M3 dSig;
            dSig.e[0][0] = g_cov[0]; dSig.e[0][1] = 0.5f * g_cov[1]; dSig.e[0][2] = 0.5f * g_cov[2];
            dSig.e[1][0] = 0.5f * g_cov[1]; dSig.e[1][1] = g_cov[3]; dSig.e[1][2] = 0.5f * g_cov[4];
            dSig.e[2][0] = 0.5f * g_cov[2]; dSig.e[2][1] = 0.5f * g_cov[4]; dSig.e[2][2] = g_cov[5];
#pragma unroll
            for (int c = 0; c < 3; c++)
#pragma unroll
                for (int k = 0; k < 3; k++) Mm.e[c][k] = 2.0f * Mm.e[c][k];
            const M3 dM = m3mul(Mm, dSig);
            const M3 Rt = m3t(R);
            M3 dRt = m3t(dM);                                   // dL_dMt
            g_scale[0] = Rt.e[0][0] * dRt.e[0][0] + Rt.e[0][1] * dRt.e[0][1] + Rt.e[0][2] * dRt.e[0][2];
            g_scale[1] = Rt.e[1][0] * dRt.e[1][0] + Rt.e[1][1] * dRt.e[1][1] + Rt.e[1][2] * dRt.e[1][2];
            g_scale[2] = a.surface ? 0.f : Rt.e[2][0] * dRt.e[2][0] + Rt.e[2][1] * dRt.e[2][1] + Rt.e[2][2] * dRt.e[2][2];
#pragma unroll
            for (int k = 0; k < 3; k++) { dRt.e[0][k] *= s0; dRt.e[1][k] *= s1; dRt.e[2][k] *= s2; }
            // view-space normal gradient enters the third rotation column (:394-402); zero when not in surface mode
            const float cn0 = a.surface ? acc[9] : 0.f, cn1 = a.surface ? acc[10] : 0.f, cn2 = a.surface ? acc[11] : 0.f;
            dRt.e[2][0] += cn0 * V[0] + cn1 * V[1] + cn2 * V[2];
            dRt.e[2][1] += cn0 * V[4] + cn1 * V[5] + cn2 * V[6];
            dRt.e[2][2] += cn0 * V[8] + cn1 * V[9] + cn2 * V[10];
            if ((CAM_SH && a.lrn_cam)) {   // :404-414
                const float wn[3] = {R.e[0][2], R.e[1][2], R.e[2][2]};
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    cam_view[4 * i + 0] += cn0 * wn[i]; cam_view[4 * i + 1] += cn1 * wn[i]; cam_view[4 * i + 2] += cn2 * wn[i];
                }
            }
#define DR(i, j) dRt.e[i][j]
            g_rot[0] = 2 * z * (DR(0, 1) - DR(1, 0)) + 2 * y * (DR(2, 0) - DR(0, 2)) + 2 * x * (DR(1, 2) - DR(2, 1));
            g_rot[1] = 2 * y * (DR(1, 0) + DR(0, 1)) + 2 * z * (DR(2, 0) + DR(0, 2)) + 2 * r * (DR(1, 2) - DR(2, 1)) -
                       4 * x * (DR(2, 2) + DR(1, 1));
            g_rot[2] = 2 * x * (DR(1, 0) + DR(0, 1)) + 2 * r * (DR(2, 0) - DR(0, 2)) + 2 * z * (DR(1, 2) + DR(2, 1)) -
                       4 * y * (DR(2, 2) + DR(0, 0));
            g_rot[3] = 2 * r * (DR(0, 1) - DR(1, 0)) + 2 * x * (DR(2, 0) + DR(0, 2)) + 2 * y * (DR(1, 2) + DR(2, 1)) -
                       4 * z * (DR(1, 1) + DR(0, 0));
#undef DR
        }
    }

}

// the argument block of one frame's per-Gaussian backward (rasterizer_impl.cu:339-340 for the focal terms)
inline void fill_geom_bwd_args(GeomBwdArgs &a, const SoarRastParams &prm, const float *means3D, const int32_t *radii, const float *shs,
                               const float *scales, const float *rotations, const float *cov3D_precomp, const GeomBuf &g, const float *acc)
{
    a.P = prm.P; a.D = prm.sh_degree; a.M = prm.M; a.W = prm.W; a.H = prm.H;
    a.surface = prm.cfg_surface; a.lrn_cam = prm.cfg_lrn_cam;
    a.tanfovx = prm.tanfovx; a.tanfovy = prm.tanfovy;
    a.h_y = prm.H / (2.0f * prm.tanfovy);
    a.h_x = prm.W / (2.0f * prm.tanfovx);
    a.scale_modifier = prm.scale_modifier;
    a.means3D = means3D; a.shs = shs; a.scales = scales; a.rotations = rotations;
    a.cov3D = cov3D_precomp ? cov3D_precomp : g.cov3D;
    a.radii = radii; a.clamped = g.clamped;
    a.view = prm.viewmatrix_dev; a.proj = prm.projmatrix_dev; a.campos = prm.campos_dev;
    a.acc = acc;
}

}  // namespace

}  // namespace soar
