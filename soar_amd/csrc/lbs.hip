// lbs.hip -- SMPL-X linear-blend skinning of canonical Gaussian surfels and nearest-neighbour helpers, gfx950.
//
//  (the K-NN blend-weight kernel lives in lbs_knn.hip)
//  * warp_forward_kernel  : blend (TS/utils/smpl.py:613) fused with the apply step of DiffGaussian.forward
//                           (TS/renderer/diff_gaussian_rasterizer.py:103-114, :138-149): one kernel instead of an
//                           einsum + ~8 small torch kernels; weight rows are staged through LDS with coalesced loads,
//                           joint matrices are wave-uniform (scalar loads).
//  * warp_backward_kernel : analytic gradient w.r.t. canonical xyz and quaternion.
//  * dist2_knn3_kernel    : simple-knn distCUDA2 semantics (mean of the 3 smallest squared distances, self excluded).
#include "soar_common.h"
#include "geom_bwd_point.h"
#include "preprocess_point.h"

// Products and sums contract to FMAs within one expression only, as written: the same point has to come out bit-identical
// from the single-frame kernels and from the all-frames ones, whatever each kernel's surroundings let the backend fuse.
#pragma clang fp contract(on)

#ifndef SOAR_LBS_SKIP_UNUSED_JOINTS
#define SOAR_LBS_SKIP_UNUSED_JOINTS 1
#endif

namespace soar {

namespace {

constexpr int KNN_TILE = 1024;

// ------------------------------------------------------------------------------------------------
// quaternion <-> matrix helpers (row-major 3x3, m[r*3+c]); public pytorch3d conventions, real part first
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void quat_to_mat(const float q[4], float m[9])
{
    const float r = q[0], i = q[1], j = q[2], k = q[3];
    const float two_s = 2.0f / (r * r + i * i + j * j + k * k);
    m[0] = 1 - two_s * (j * j + k * k); m[1] = two_s * (i * j - k * r); m[2] = two_s * (i * k + j * r);
    m[3] = two_s * (i * j + k * r); m[4] = 1 - two_s * (i * i + k * k); m[5] = two_s * (j * k - i * r);
    m[6] = two_s * (i * k - j * r); m[7] = two_s * (j * k + i * r); m[8] = 1 - two_s * (i * i + j * j);
}

// candidate table of matrix_to_quaternion: returns best index, fills cand[4] and a = q_abs[best]
__device__ __forceinline__ int mat_to_quat_candidates(const float m[9], float cand[4], float &a_best, float &x_best)
{
    const float m00 = m[0], m01 = m[1], m02 = m[2], m10 = m[3], m11 = m[4], m12 = m[5], m20 = m[6], m21 = m[7], m22 = m[8];
    const float xs[4] = {1.0f + m00 + m11 + m22, 1.0f + m00 - m11 - m22, 1.0f - m00 + m11 - m22, 1.0f - m00 - m11 + m22};
    float qa[4];
    int best = 0;
#pragma unroll
    for (int t = 0; t < 4; t++) qa[t] = xs[t] > 0.f ? sqrtf(xs[t]) : 0.f;
#pragma unroll
    for (int t = 1; t < 4; t++)
        if (qa[t] > qa[best]) best = t;
    a_best = qa[best];
    x_best = xs[best];
    const float sq = a_best * a_best;
    if (best == 0) { cand[0] = sq; cand[1] = m21 - m12; cand[2] = m02 - m20; cand[3] = m10 - m01; }
    else if (best == 1) { cand[0] = m21 - m12; cand[1] = sq; cand[2] = m10 + m01; cand[3] = m02 + m20; }
    else if (best == 2) { cand[0] = m02 - m20; cand[1] = m10 + m01; cand[2] = sq; cand[3] = m12 + m21; }
    else { cand[0] = m10 - m01; cand[1] = m20 + m02; cand[2] = m21 + m12; cand[3] = sq; }
    return best;
}

struct WarpArgs {
    int P, J;
    const float *xyz, *rot, *weights, *joint_mats, *offsets, *axis_perm;
    float *xyz_out, *rot_out, *pt_mats_out;
    const float *g_xyz_out, *g_rot_out;
    float *g_xyz, *g_rot;
    // batch of frames (blockIdx.y): the canonical model and the blend weights are shared, joint transforms and outputs per frame
    size_t mats_stride, xyz_stride, rot_stride;      // floats between two frames (0: one frame)
};

__device__ __forceinline__ void select_frame(WarpArgs &a)
{
    const size_t f = blockIdx.y;
    if (f == 0) return;
    a.joint_mats += f * a.mats_stride;
    if (a.xyz_out) a.xyz_out += f * a.xyz_stride;
    if (a.rot_out) a.rot_out += f * a.rot_stride;
    if (a.g_xyz_out) a.g_xyz_out += f * a.xyz_stride;
    if (a.g_rot_out) a.g_rot_out += f * a.rot_stride;
    if (a.g_xyz) a.g_xyz += f * a.xyz_stride;
    if (a.g_rot) a.g_rot += f * a.rot_stride;
}

constexpr int WARP_THREADS = 256;
constexpr int WARP_MAXJ = 64;

// blended 3x4 transform of one Gaussian: M[r*4+c] = sum_j w_j * A_j[r][c]; weights via LDS (coalesced tile load).
// With a.weights == nullptr, a.joint_mats holds one ready-made 4x4 per Gaussian (the pt_mats tensor that
// SMPL_Guidance.__call__ returns) and is simply loaded.
__device__ __forceinline__ void blend_frame_matrix(const float *wrow, const float *mats, int J, float M[12]);
__device__ __forceinline__ void blend_matrix(const WarpArgs &a, float *wtile, int p0, int tid, float M[12])
{
    const int J = a.J;
    const int nrows = min(WARP_THREADS, a.P - p0);
#pragma unroll
    for (int c = 0; c < 12; c++) M[c] = 0.f;
    if (a.weights == nullptr) {
        if (tid < nrows) {
            const float4 *m = reinterpret_cast<const float4 *>(a.joint_mats + (size_t)(p0 + tid) * 16);
            const float4 r0 = m[0], r1 = m[1], r2 = m[2];
            M[0] = r0.x; M[1] = r0.y; M[2] = r0.z; M[3] = r0.w;
            M[4] = r1.x; M[5] = r1.y; M[6] = r1.z; M[7] = r1.w;
            M[8] = r2.x; M[9] = r2.y; M[10] = r2.z; M[11] = r2.w;
        }
        return;
    }
    __syncthreads();
    for (int t = tid; t < nrows * J; t += WARP_THREADS) wtile[t] = a.weights[(size_t)p0 * J + t];
    __syncthreads();
    if (tid < nrows) blend_frame_matrix(wtile + tid * J, a.joint_mats, J, M);      // row stride J = 55 floats: odd => conflict-free
}

// the warp of one Gaussian given its blended 3x4 transform M
__device__ __forceinline__ void forward_point_vals(const WarpArgs &a, int p, const float M[12], float pos[3], float4 &quat)
{
    // position: p' = M3 p + t (+ offsets), then optional axis permutation p' <- p' T
    const float x = a.xyz[3 * p], y = a.xyz[3 * p + 1], z = a.xyz[3 * p + 2];
    float px = M[0] * x + M[1] * y + M[2] * z + M[3];
    float py = M[4] * x + M[5] * y + M[6] * z + M[7];
    float pz = M[8] * x + M[9] * y + M[10] * z + M[11];
    if (a.offsets) { px += a.offsets[3 * p]; py += a.offsets[3 * p + 1]; pz += a.offsets[3 * p + 2]; }

    // rotation: R' = M3 R(q)
    const float4 qv = reinterpret_cast<const float4 *>(a.rot)[p];
    const float q[4] = {qv.x, qv.y, qv.z, qv.w};
    float R[9], Rp[9];
    quat_to_mat(q, R);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) Rp[r * 3 + c] = M[r * 4 + 0] * R[c] + M[r * 4 + 1] * R[3 + c] + M[r * 4 + 2] * R[6 + c];

    if (a.axis_perm) {
        const float *T = a.axis_perm;                      // row-major 3x3
        const float tx = px * T[0] + py * T[3] + pz * T[6];
        const float ty = px * T[1] + py * T[4] + pz * T[7];
        const float tz = px * T[2] + py * T[5] + pz * T[8];
        px = tx; py = ty; pz = tz;
        float Rt[9];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) Rt[r * 3 + c] = T[0 * 3 + r] * Rp[c] + T[1 * 3 + r] * Rp[3 + c] + T[2 * 3 + r] * Rp[6 + c];
#pragma unroll
        for (int k = 0; k < 9; k++) Rp[k] = Rt[k];
    }
    pos[0] = px; pos[1] = py; pos[2] = pz;

    // q' = normalize(standardize(matrix_to_quaternion(R')))
    float cand[4], a_best, x_best;
    mat_to_quat_candidates(Rp, cand, a_best, x_best);
    const float inv = 1.0f / (2.0f * fmaxf(a_best, 0.1f));
    float o[4] = {cand[0] * inv, cand[1] * inv, cand[2] * inv, cand[3] * inv};
    if (o[0] < 0.f) { o[0] = -o[0]; o[1] = -o[1]; o[2] = -o[2]; o[3] = -o[3]; }
    const float nrm = fmaxf(sqrtf(o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3]), 1e-12f);
    quat = make_float4(o[0] / nrm, o[1] / nrm, o[2] / nrm, o[3] / nrm);
}
__device__ __forceinline__ void forward_point(const WarpArgs &a, int p, const float M[12], float *xyz_out, float *rot_out)
{
    float pos[3];
    float4 quat;
    forward_point_vals(a, p, M, pos, quat);
    xyz_out[3 * p] = pos[0]; xyz_out[3 * p + 1] = pos[1]; xyz_out[3 * p + 2] = pos[2];
    reinterpret_cast<float4 *>(rot_out)[p] = quat;
}

__global__ void __launch_bounds__(WARP_THREADS) warp_forward_kernel(WarpArgs a)
{
    extern __shared__ float wtile[];
    select_frame(a);
    const int tid = threadIdx.x;
    const int p0 = blockIdx.x * WARP_THREADS;
    const int p = p0 + tid;
    float M[12];
    blend_matrix(a, wtile, p0, tid, M);
    if (p >= a.P) return;

    if (a.pt_mats_out && a.weights) {
        float4 *o = reinterpret_cast<float4 *>(a.pt_mats_out + (size_t)p * 16);
        // bottom row is the blend of the joints' [0,0,0,1] rows = sum of weights
        float wsum = 0.f;
        const float *wrow = wtile + tid * a.J;
        float b0 = 0.f, b1 = 0.f, b2 = 0.f;
        for (int j = 0; j < a.J; j++) {
            const float *A = a.joint_mats + 16 * j + 12;
            b0 += wrow[j] * A[0]; b1 += wrow[j] * A[1]; b2 += wrow[j] * A[2]; wsum += wrow[j] * A[3];
        }
        o[0] = make_float4(M[0], M[1], M[2], M[3]);
        o[1] = make_float4(M[4], M[5], M[6], M[7]);
        o[2] = make_float4(M[8], M[9], M[10], M[11]);
        o[3] = make_float4(b0, b1, b2, wsum);
    }

    forward_point(a, p, M, a.xyz_out, a.rot_out);
}

// gradient of one Gaussian's canonical position / quaternion given its blended 3x4 transform M and the upstream gradients
__device__ __forceinline__ void backward_point_vals(const WarpArgs &a, int p, const float M[12], float gx, float gy, float gz, const float4 gq,
                                                   float dxyz[3], float4 &drot)
{
    const float *T = a.axis_perm;
    // ---- position: dL/dp = M3^T (T g)
    if (T) {
        const float tx = T[0] * gx + T[1] * gy + T[2] * gz;
        const float ty = T[3] * gx + T[4] * gy + T[5] * gz;
        const float tz = T[6] * gx + T[7] * gy + T[8] * gz;
        gx = tx; gy = ty; gz = tz;
    }
    dxyz[0] = M[0] * gx + M[4] * gy + M[8] * gz;
    dxyz[1] = M[1] * gx + M[5] * gy + M[9] * gz;
    dxyz[2] = M[2] * gx + M[6] * gy + M[10] * gz;

    // ---- rotation: recompute the forward
    const float4 qv = reinterpret_cast<const float4 *>(a.rot)[p];
    const float q[4] = {qv.x, qv.y, qv.z, qv.w};
    float R[9], B[9], Rp[9];
    quat_to_mat(q, R);
    // B = T^T M3 (or M3): R' = B R
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++)
            B[r * 3 + c] = T ? (T[0 * 3 + r] * M[0 * 4 + c] + T[1 * 3 + r] * M[1 * 4 + c] + T[2 * 3 + r] * M[2 * 4 + c]) : M[r * 4 + c];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) Rp[r * 3 + c] = B[r * 3 + 0] * R[c] + B[r * 3 + 1] * R[3 + c] + B[r * 3 + 2] * R[6 + c];

    float cand[4], a_best, x_best;
    const int best = mat_to_quat_candidates(Rp, cand, a_best, x_best);
    const float Dn = 2.0f * fmaxf(a_best, 0.1f);
    float o[4] = {cand[0] / Dn, cand[1] / Dn, cand[2] / Dn, cand[3] / Dn};
    const float sgn = (o[0] < 0.f) ? -1.f : 1.f;
    float u[4] = {sgn * o[0], sgn * o[1], sgn * o[2], sgn * o[3]};
    const float un = sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2] + u[3] * u[3]);
    float g[4] = {gq.x, gq.y, gq.z, gq.w};
    // through F.normalize: g_u = (g - (g.n) n) / |u|   (eps branch: plain scale)
    float gu[4];
    if (un > 1e-12f) {
        const float n[4] = {u[0] / un, u[1] / un, u[2] / un, u[3] / un};
        const float gn = g[0] * n[0] + g[1] * n[1] + g[2] * n[2] + g[3] * n[3];
#pragma unroll
        for (int k = 0; k < 4; k++) gu[k] = (g[k] - gn * n[k]) / un;
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) gu[k] = g[k] / 1e-12f;
    }
    // through standardize and the division by D = 2 max(a, 0.1)
    float gc[4];
    float gD = 0.f;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float go = sgn * gu[k];
        gc[k] = go / Dn;
        gD -= go * cand[k] / (Dn * Dn);
    }
    // cand[best] = a^2 = x (x > 0), D = 2a when a > 0.1:  dL/dx = gc[best] + gD * 2 * (1 / (2a))
    float gxb = (x_best > 0.f) ? gc[best] : 0.f;
    if (a_best > 0.1f && x_best > 0.f) gxb += gD / a_best;
    // scatter to dL/dR' (row-major); x_best = 1 +- m00 +- m11 +- m22
    float gR[9];
#pragma unroll
    for (int k = 0; k < 9; k++) gR[k] = 0.f;
    const float s00 = (best == 0 || best == 1) ? 1.f : -1.f;
    const float s11 = (best == 0 || best == 2) ? 1.f : -1.f;
    const float s22 = (best == 0 || best == 3) ? 1.f : -1.f;
    gR[0] = s00 * gxb; gR[4] = s11 * gxb; gR[8] = s22 * gxb;
    // off-diagonal linear terms (m[r*3+c]): m01=1 m02=2 m10=3 m12=5 m20=6 m21=7
    if (best == 0) {
        gR[7] += gc[1]; gR[5] -= gc[1]; gR[2] += gc[2]; gR[6] -= gc[2]; gR[3] += gc[3]; gR[1] -= gc[3];
    } else if (best == 1) {
        gR[7] += gc[0]; gR[5] -= gc[0]; gR[3] += gc[2]; gR[1] += gc[2]; gR[2] += gc[3]; gR[6] += gc[3];
    } else if (best == 2) {
        gR[2] += gc[0]; gR[6] -= gc[0]; gR[3] += gc[1]; gR[1] += gc[1]; gR[5] += gc[3]; gR[7] += gc[3];
    } else {
        gR[3] += gc[0]; gR[1] -= gc[0]; gR[6] += gc[1]; gR[2] += gc[1]; gR[7] += gc[2]; gR[5] += gc[2];
    }
    // dL/dR(q) = B^T dL/dR'
    float G[9];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) G[r * 3 + c] = B[0 * 3 + r] * gR[c] + B[1 * 3 + r] * gR[3 + c] + B[2 * 3 + r] * gR[6 + c];
    // R = I + s Q(q), s = 2/|q|^2:  dL/dq_m = s <G, dQ/dq_m> - s^2 q_m <G, Q>
    const float r = q[0], i = q[1], j = q[2], k = q[3];
    const float nn = r * r + i * i + j * j + k * k;
    const float s = 2.0f / nn;
    const float Q[9] = {-(j * j + k * k), i * j - k * r, i * k + j * r, i * j + k * r, -(i * i + k * k), j * k - i * r,
                        i * k - j * r, j * k + i * r, -(i * i + j * j)};
    float GQ = 0.f;
#pragma unroll
    for (int t = 0; t < 9; t++) GQ += G[t] * Q[t];
    const float dr = -k * G[1] + j * G[2] + k * G[3] - i * G[5] - j * G[6] + i * G[7];
    const float di = j * G[1] + k * G[2] + j * G[3] - 2 * i * G[4] - r * G[5] + k * G[6] + r * G[7] - 2 * i * G[8];
    const float dj = -2 * j * G[0] + i * G[1] + r * G[2] + i * G[3] + k * G[5] - r * G[6] + k * G[7] - 2 * j * G[8];
    const float dk = -2 * k * G[0] - r * G[1] + i * G[2] + r * G[3] - 2 * k * G[4] + j * G[5] + i * G[6] + j * G[7];
    drot =
        make_float4(s * dr - s * s * r * GQ, s * di - s * s * i * GQ, s * dj - s * s * j * GQ, s * dk - s * s * k * GQ);
}

__device__ __forceinline__ void backward_point(const WarpArgs &a, int p, const float M[12], const float *g_xyz_out, const float *g_rot_out,
                                              float dxyz[3], float4 &drot)
{
    backward_point_vals(a, p, M, g_xyz_out[3 * p], g_xyz_out[3 * p + 1], g_xyz_out[3 * p + 2], reinterpret_cast<const float4 *>(g_rot_out)[p],
                        dxyz, drot);
}

__global__ void __launch_bounds__(WARP_THREADS) warp_backward_kernel(WarpArgs a)
{
    extern __shared__ float wtile[];
    select_frame(a);
    const int tid = threadIdx.x;
    const int p0 = blockIdx.x * WARP_THREADS;
    const int p = p0 + tid;
    float M[12];
    blend_matrix(a, wtile, p0, tid, M);
    if (p >= a.P) return;

    float dxyz[3];
    float4 drot;
    backward_point(a, p, M, a.g_xyz_out, a.g_rot_out, dxyz, drot);
    a.g_xyz[3 * p] = dxyz[0]; a.g_xyz[3 * p + 1] = dxyz[1]; a.g_xyz[3 * p + 2] = dxyz[2];
    reinterpret_cast<float4 *>(a.g_rot)[p] = drot;
}

// ---- all frames of a step in one launch: a workgroup takes 64 Gaussians, its wavefront k the frames k, k + WARP_NF, ...  The
//      weights rows of the 64 Gaussians are staged once and serve every frame; the joint transforms are wavefront-uniform
//      (scalar loads), exactly as in the single-frame kernels -- per wavefront this IS the single-frame kernel's work, with
//      WARP_NF times the wavefronts in flight to hide its latencies behind.  The backward form adds the frames' gradients in
//      frame order (through LDS) and can sum other per-frame gradient blocks of the same points on the way.
constexpr int WARP_NF = WARP_THREADS / WAVE;                 // frames in flight per workgroup
struct FrameSums {
    int n_extra;
    const float *src[2];         // [n][P][width]
    float *dst[2];               // [P][width]
    int width[2];
};
__device__ __forceinline__ void stage_weight_rows(const WarpArgs &a, float *wtile, int tid, int p0)
{
    // p0 * J * 4 bytes is a multiple of 16 (p0 is one of 64): float4 copies, the odd tail of the last workgroup one by one
    const int count = min(WAVE, a.P - p0) * a.J;
    const float *src = a.weights + (size_t)p0 * a.J;
    for (int t = tid; t < count / 4; t += WARP_THREADS) reinterpret_cast<float4 *>(wtile)[t] = reinterpret_cast<const float4 *>(src)[t];
    if (tid < count % 4) wtile[count - 1 - tid] = src[count - 1 - tid];
    __syncthreads();
}
#ifndef SOAR_LBS_WCHUNK
#define SOAR_LBS_WCHUNK 8
#endif
__device__ __forceinline__ void blend_frame_matrix(const float *wrow, const float *mats, int J, float M[12])
{
#pragma unroll
    for (int c = 0; c < 12; c++) M[c] = 0.f;
    // Skinning weights are sparse -- a vertex follows at most a handful of the 55 joints, a Gaussian the union of its 30 nearest
    // vertices' joints, and the 64 Gaussians of a wavefront (neighbours in space when the model is in spatial order) share most of
    // them: a joint NO lane of the wavefront follows is left out (round 6).  Exact: its terms are w A = +-0 added to sums that
    // started at +0 -- M does not change by a bit, whatever the lanes left out of the test hold.
    // (the weights of SOAR_LBS_WCHUNK joints are read together: one LDS round trip per chunk instead of one per joint in front of
    // every test; the joints are still taken in ascending order)
    constexpr int WC = SOAR_LBS_WCHUNK;
    for (int j0 = 0; j0 < J; j0 += WC) {
        float w[WC];
#pragma unroll
        for (int u = 0; u < WC; u++) w[u] = j0 + u < J ? wrow[j0 + u] : 0.f;
#pragma unroll
        for (int u = 0; u < WC; u++) {
#if SOAR_LBS_SKIP_UNUSED_JOINTS
            if (__ballot(w[u] != 0.f) == 0ull) continue;
#else
            if (j0 + u >= J) continue;
#endif
            const float *A = mats + 16 * (j0 + u);     // wavefront-uniform address -> scalar loads
#pragma unroll
            for (int c = 0; c < 12; c++) M[c] += w[u] * A[c];
        }
    }
}
__global__ void __launch_bounds__(WARP_THREADS) warp_forward_frames_kernel(WarpArgs a, int n)
{
    extern __shared__ float wtile[];
    const int tid = threadIdx.x, k = __builtin_amdgcn_readfirstlane(tid / WAVE), lane = tid % WAVE;
    const int p0 = blockIdx.x * WAVE, p = p0 + lane;
    stage_weight_rows(a, wtile, tid, p0);
    const int f = blockIdx.y * WARP_NF + k;               // no loop over frame groups: a store before the matrix loads of a
    if (p >= a.P || f >= n) return;                       // later group would turn them into vector loads
    float M[12];
    blend_frame_matrix(wtile + lane * a.J, a.joint_mats + (size_t)f * a.mats_stride, a.J, M);
    forward_point(a, p, M, a.xyz_out + (size_t)f * a.xyz_stride, a.rot_out + (size_t)f * a.rot_stride);
}
__global__ void __launch_bounds__(WARP_THREADS) warp_backward_frames_kernel(WarpArgs a, int n, FrameSums fs)
{
    extern __shared__ float wtile[];
    float *red = wtile + WAVE * a.J;                             // [WARP_NF][7][WAVE]
    const int tid = threadIdx.x, k = __builtin_amdgcn_readfirstlane(tid / WAVE), lane = tid % WAVE;
    const int p0 = blockIdx.x * WAVE, p = p0 + lane;
    stage_weight_rows(a, wtile, tid, p0);
    float acc[2] = {0.f, 0.f};                                   // thread tid owns outputs tid and tid + WARP_THREADS of the 7 x 64
    for (int f0 = 0; f0 < n; f0 += WARP_NF) {
        const int f = f0 + k;
        float d[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (p < a.P && f < n) {
            float M[12], dx[3];
            float4 dq;
            blend_frame_matrix(wtile + lane * a.J, a.joint_mats + (size_t)f * a.mats_stride, a.J, M);
            backward_point(a, p, M, a.g_xyz_out + (size_t)f * a.xyz_stride, a.g_rot_out + (size_t)f * a.rot_stride, dx, dq);
            d[0] = dx[0]; d[1] = dx[1]; d[2] = dx[2]; d[3] = dq.x; d[4] = dq.y; d[5] = dq.z; d[6] = dq.w;
        }
        if (f0) __syncthreads();                                 // the previous group's sums have been read
#pragma unroll
        for (int c = 0; c < 7; c++) red[(k * 7 + c) * WAVE + lane] = d[c];
        __syncthreads();
        // frame order: (((acc + g_f0) + g_f0+1) + g_f0+2) + g_f0+3 (frames past n contribute exact zeros)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int o = tid + h * WARP_THREADS;
            if (o < 7 * WAVE)
#pragma unroll
                for (int kk = 0; kk < WARP_NF; kk++) acc[h] += red[kk * 7 * WAVE + o];
        }
    }
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int o = tid + h * WARP_THREADS, c = o / WAVE, q = p0 + o % WAVE;
        if (o < 7 * WAVE && q < a.P) {
            if (c < 3) a.g_xyz[3 * q + c] = acc[h];
            else a.g_rot[4 * q + c - 3] = acc[h];
        }
    }
    if (p < a.P && k >= 1 && k - 1 < fs.n_extra) {                // wavefronts 1, 2: the extra blocks of these points
        const int e = k - 1, w = fs.width[e];
        const size_t count = (size_t)a.P * w;
        const float *src = fs.src[e] + (size_t)p * w;
        float *dst = fs.dst[e] + (size_t)p * w;
        // every load of the (up to) 4 x 4 values in flight before the first add; the sums still run in frame order
        for (int c0 = 0; c0 < w; c0 += 4) {
            float v[4][WARP_NF];
#pragma unroll
            for (int c = 0; c < 4; c++)
#pragma unroll
                for (int f = 0; f < WARP_NF; f++) v[c][f] = (c0 + c < w && f < n) ? src[(size_t)f * count + c0 + c] : 0.f;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                if (c0 + c >= w) break;
                float s = v[c][0];
#pragma unroll
                for (int f = 1; f < WARP_NF; f++) s = f < n ? s + v[c][f] : s;
                for (int f = WARP_NF; f < n; f++) s += src[(size_t)f * count + c0 + c];
                dst[c0 + c] = s;
            }
        }
    }
}

// ---- the warp and the per-Gaussian forward stage of the rasterizer in ONE kernel (round 6) ------------------------------------------
// warp_forward_frames_kernel wrote the posed position and quaternion of every (frame, Gaussian) and preprocess_kernel read them
// back one launch later.  Here wavefront k of a workgroup takes frame k of 64 Gaussians: blended joint transform -> forward_point ->
// preprocess_point (preprocess_point.h: the same function, the same un-contracted arithmetic -- radii, tile rectangles and depth
// keys bit for bit those of the two kernels) in registers.  The posed values are still written (the backward's tail reads them).
__global__ void __launch_bounds__(WARP_THREADS) warp_preprocess_frames_kernel(WarpArgs a, int n, Batch<PreArgs> fr)
{
    extern __shared__ float wtile[];
    const int tid = threadIdx.x, k = __builtin_amdgcn_readfirstlane(tid / WAVE), lane = tid % WAVE;
    const int p0 = blockIdx.x * WAVE, p_raw = p0 + lane;
    stage_weight_rows(a, wtile, tid, p0);
    const int f = blockIdx.y * WARP_NF + k;
    if (f >= n) return;
    // lanes past the end redo the last Gaussian (identical stores) so that the whole wavefront reaches the statistics' reduction
    const bool in_range = p_raw < a.P;
    const int p = in_range ? p_raw : a.P - 1;
    float M[12], pos[3];
    float4 quat;
    blend_frame_matrix(wtile + (p - p0) * a.J, a.joint_mats + (size_t)f * a.mats_stride, a.J, M);
    forward_point_vals(a, p, M, pos, quat);
    float *xyz_out = a.xyz_out + (size_t)f * a.xyz_stride, *rot_out = a.rot_out + (size_t)f * a.rot_stride;
    xyz_out[3 * p] = pos[0]; xyz_out[3 * p + 1] = pos[1]; xyz_out[3 * p + 2] = pos[2];
    reinterpret_cast<float4 *>(rot_out)[p] = quat;
    const PreArgs &g = fr.v[f];
    PrePoint o;
    preprocess_point(g, p, in_range, pos[0], pos[1], pos[2], true, quat, o);
    preprocess_store(g, p, in_range, blockIdx.x, o);
    if (!g.prefiltered && blockIdx.x == 0 && lane == 0) g.header[H_PREFILTER_VIOLATIONS] = 0u;
}

// ---- the per-Gaussian backward of the rasterizer and the warp's backward in ONE kernel (round 6) ----------------------------------
// geometry_backward_kernel wrote, per frame and Gaussian, the gradients of the posed position / quaternion / scales / colours (52
// bytes, plus 28 nobody reads) and warp_backward_frames_kernel read them back one launch later: 99 us of four launches per 4-frame
// step at C3 for 53 us worth of bytes.  Here wavefront k of a workgroup takes frame k of 64 Gaussians: the accumulation row of the
// backward blend -> geometry_backward_point (geom_bwd_point.h: the same function, the same un-contracted arithmetic) -> the blended
// joint transform -> backward_point, all in registers; the frames' results are added in frame order through LDS exactly like the
// unfused kernel's, the scale / colour / occlusion gradients on the way.  Explicit colours, scales + quaternions, no camera gradients
// (what the per-frame training path uses; the general entry points keep the two kernels).
constexpr int TAIL_C = 14;        // xyz 3 | quaternion 4 | scales 3 | colours 3 | occlusion value 1
struct TailOut {
    float *dL_dmeans2D[MAX_BATCH];   // per frame [P,3]
    float *dL_dscales, *dL_dcolors, *dL_docc;      // sums over the frames: [P,3], [P,3], [P] or NULL
};
#ifndef SOAR_TAIL_WPE
#define SOAR_TAIL_WPE 4
#endif
__global__ void __launch_bounds__(WARP_THREADS) __attribute__((amdgpu_waves_per_eu(SOAR_TAIL_WPE, 8))) geom_warp_backward_frames_kernel(WarpArgs a, int n, Batch<GeomBwdArgs> fr, TailOut out)
{
    extern __shared__ float wtile[];
    float *red = wtile + WAVE * a.J;                             // [WARP_NF][TAIL_C][WAVE]
    const int tid = threadIdx.x, k = __builtin_amdgcn_readfirstlane(tid / WAVE), lane = tid % WAVE;
    const int p0 = blockIdx.x * WAVE, p = p0 + lane;
    stage_weight_rows(a, wtile, tid, p0);
    constexpr int PER = (TAIL_C * WAVE + WARP_THREADS - 1) / WARP_THREADS;       // outputs per thread: o = tid + h * WARP_THREADS
    float acc[PER];
#pragma unroll
    for (int h = 0; h < PER; h++) acc[h] = 0.f;
    for (int f0 = 0; f0 < n; f0 += WARP_NF) {
        const int f = f0 + k;
        float d[TAIL_C];
#pragma unroll
        for (int c = 0; c < TAIL_C; c++) d[c] = 0.f;
        if (p < a.P && f < n) {
            const GeomBwdArgs &g = fr.v[f];
            GeomBwdPoint o;
            geometry_backward_point<false>(g, p, g.radii[p] > 0, o, nullptr, nullptr, nullptr);
            float *m2 = out.dL_dmeans2D[f];
            m2[3 * p + 0] = o.acc[0]; m2[3 * p + 1] = o.acc[1]; m2[3 * p + 2] = 0.f;        // (z is never written by the reference)
            float M[12], dx[3];
            float4 dq;
            blend_frame_matrix(wtile + lane * a.J, a.joint_mats + (size_t)f * a.mats_stride, a.J, M);
            backward_point_vals(a, p, M, o.g_mean[0], o.g_mean[1], o.g_mean[2], make_float4(o.g_rot[0], o.g_rot[1], o.g_rot[2], o.g_rot[3]), dx, dq);
            d[0] = dx[0]; d[1] = dx[1]; d[2] = dx[2]; d[3] = dq.x; d[4] = dq.y; d[5] = dq.z; d[6] = dq.w;
            d[7] = o.g_scale[0]; d[8] = o.g_scale[1]; d[9] = o.g_scale[2];
            d[10] = o.acc[6]; d[11] = o.acc[7]; d[12] = o.acc[8];
            d[13] = o.acc[13];
        }
        if (f0) __syncthreads();                                 // the previous group's sums have been read
#pragma unroll
        for (int c = 0; c < TAIL_C; c++) red[(k * TAIL_C + c) * WAVE + lane] = d[c];
        __syncthreads();
        // frame order: (((acc + g_f0) + g_f0+1) + g_f0+2) + g_f0+3 (frames past n contribute exact zeros)
#pragma unroll
        for (int h = 0; h < PER; h++) {
            const int o = tid + h * WARP_THREADS;
            if (o < TAIL_C * WAVE)
#pragma unroll
                for (int kk = 0; kk < WARP_NF; kk++) acc[h] += red[kk * TAIL_C * WAVE + o];
        }
    }
#pragma unroll
    for (int h = 0; h < PER; h++) {
        const int o = tid + h * WARP_THREADS, c = o / WAVE, q = p0 + o % WAVE;
        if (o < TAIL_C * WAVE && q < a.P) {
            if (c < 3) a.g_xyz[3 * q + c] = acc[h];
            else if (c < 7) a.g_rot[4 * q + c - 3] = acc[h];
            else if (c < 10) out.dL_dscales[3 * q + c - 7] = acc[h];
            else if (c < 13) out.dL_dcolors[3 * q + c - 10] = acc[h];
            else if (out.dL_docc) out.dL_docc[q] = acc[h];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// mean squared distance to the 3 nearest other points
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) dist2_knn3_kernel(const float *__restrict__ pts, int N, float *__restrict__ out)
{
    __shared__ float tile[KNN_TILE * 3];
    const int tid = threadIdx.x;
    const int p = blockIdx.x * 256 + tid;
    const bool valid = p < N;
    float x = 0.f, y = 0.f, z = 0.f;
    if (valid) { x = pts[3 * p]; y = pts[3 * p + 1]; z = pts[3 * p + 2]; }
    float b0 = 3.402823466e+38f, b1 = b0, b2 = b0;            // FLT_MAX, ascending
    for (int base = 0; base < N; base += KNN_TILE) {
        const int n = min(KNN_TILE, N - base);
        __syncthreads();
        for (int t = tid; t < n * 3; t += 256) tile[t] = pts[(size_t)base * 3 + t];
        __syncthreads();
        if (!valid) continue;
        for (int v = 0; v < n; v++) {
            const float ddx = x - tile[3 * v], ddy = y - tile[3 * v + 1], ddz = z - tile[3 * v + 2];
            float d = ddx * ddx + ddy * ddy + ddz * ddz;
            if (base + v == p) continue;
            if (d < b2) {
                b2 = d;
                if (b2 < b1) { const float t = b1; b1 = b2; b2 = t; }
                if (b1 < b0) { const float t = b0; b0 = b1; b1 = t; }
            }
        }
    }
    if (valid) out[p] = (b0 + b1 + b2) / 3.0f;
}

}  // namespace

}  // namespace soar

using namespace soar;

extern "C" {

static int warp_check(const float *xyz, const float *rot, const float *weights, const float *joint_mats, int32_t P, int32_t J)
{
    if (P < 0 || (weights && (J <= 0 || J > WARP_MAXJ))) { set_error("soar_lbs_warp: bad sizes P=%d J=%d (max J %d)", P, J, WARP_MAXJ); return 1; }
    if (P > 0 && (!xyz || !rot || !joint_mats)) { set_error("soar_lbs_warp: NULL pointer"); return 1; }
    return 0;
}

int soar_lbs_warp_forward(const float *xyz, const float *rot, const float *weights, const float *joint_mats,
                          const float *offsets, const float *axis_perm, int32_t P, int32_t J, float *xyz_out,
                          float *rot_out, float *pt_mats_out, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (warp_check(xyz, rot, weights, joint_mats, P, J)) return 1;
    if (P == 0) return 0;
    if (!xyz_out || !rot_out) { set_error("soar_lbs_warp_forward: NULL output"); return 1; }
    WarpArgs a{};
    a.P = P; a.J = J; a.xyz = xyz; a.rot = rot; a.weights = weights; a.joint_mats = joint_mats; a.offsets = offsets;
    a.axis_perm = axis_perm; a.xyz_out = xyz_out; a.rot_out = rot_out; a.pt_mats_out = pt_mats_out;
    const size_t lds = weights ? sizeof(float) * WARP_THREADS * (size_t)J : 0;
    StageTimer timer(ST_LBS_WARP_FWD, stream);
    hipLaunchKernelGGL(warp_forward_kernel, dim3((P + WARP_THREADS - 1) / WARP_THREADS), dim3(WARP_THREADS), lds, stream, a);
    SOAR_LAUNCH_OK("lbs_warp_forward", stream, 0);
    return 0;
}

int soar_lbs_warp_backward(const float *xyz, const float *rot, const float *weights, const float *joint_mats,
                           const float *axis_perm, int32_t P, int32_t J, const float *dL_dxyz_out,
                           const float *dL_drot_out, float *dL_dxyz, float *dL_drot, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (warp_check(xyz, rot, weights, joint_mats, P, J)) return 1;
    if (P == 0) return 0;
    if (!dL_dxyz_out || !dL_drot_out || !dL_dxyz || !dL_drot) { set_error("soar_lbs_warp_backward: NULL pointer"); return 1; }
    WarpArgs a{};
    a.P = P; a.J = J; a.xyz = xyz; a.rot = rot; a.weights = weights; a.joint_mats = joint_mats; a.axis_perm = axis_perm;
    a.g_xyz_out = dL_dxyz_out; a.g_rot_out = dL_drot_out; a.g_xyz = dL_dxyz; a.g_rot = dL_drot;
    const size_t lds = weights ? sizeof(float) * WARP_THREADS * (size_t)J : 0;
    StageTimer timer(ST_LBS_WARP_BWD, stream);
    hipLaunchKernelGGL(warp_backward_kernel, dim3((P + WARP_THREADS - 1) / WARP_THREADS), dim3(WARP_THREADS), lds, stream, a);
    SOAR_LAUNCH_OK("lbs_warp_backward", stream, 0);
    return 0;
}

// The warps of the n frames of a step in one launch each way (soar_amd/step_plan.py): the canonical model and the blend weights are
// shared (the weights tile is staged once per workgroup), joint_mats [n][J][16], xyz_out / dL_dxyz_out [n][P][3],
// rot_out / dL_drot_out [n][P][4].  The backward form returns the SUM over the frames (frame order), dL_dxyz [P][3] and
// dL_drot [P][4], and adds up to two more per-frame gradient blocks of the same points on the way (extra_src[e] = [n][P][width],
// extra_dst[e] = [P][width]).
int soar_lbs_warp_forward_batch(const float *xyz, const float *rot, const float *weights, const float *joint_mats, int32_t n,
                                int32_t P, int32_t J, float *xyz_out, float *rot_out, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n <= 0 || !weights) { set_error("soar_lbs_warp_forward_batch: need n > 0 and blend weights"); return 1; }
    if (warp_check(xyz, rot, weights, joint_mats, P, J)) return 1;
    if (P == 0) return 0;
    if (!xyz_out || !rot_out) { set_error("soar_lbs_warp_forward_batch: NULL output"); return 1; }
    WarpArgs a{};
    a.P = P; a.J = J; a.xyz = xyz; a.rot = rot; a.weights = weights; a.joint_mats = joint_mats;
    a.xyz_out = xyz_out; a.rot_out = rot_out;
    a.mats_stride = (size_t)J * 16; a.xyz_stride = (size_t)P * 3; a.rot_stride = (size_t)P * 4;
    const size_t lds = sizeof(float) * WAVE * (size_t)J;      // the weights rows of 64 Gaussians
    StageTimer timer(ST_LBS_WARP_FWD, stream);
    hipLaunchKernelGGL(warp_forward_frames_kernel, dim3((P + WAVE - 1) / WAVE, (n + WARP_NF - 1) / WARP_NF), dim3(WARP_THREADS), lds, stream, a, (int)n);
    SOAR_LAUNCH_OK("lbs_warp_forward_batch", stream, 0);
    return 0;
}

static int warp_backward_frames(const float *xyz, const float *rot, const float *weights, const float *joint_mats, const float *axis_perm,
                                size_t mats_stride, int32_t n, int32_t P, int32_t J, const float *dL_dxyz_out, const float *dL_drot_out,
                                float *dL_dxyz, float *dL_drot, int32_t n_extra, const float *const *extra_src, float *const *extra_dst,
                                const int32_t *extra_width, void *stream_);

int soar_lbs_warp_backward_sum(const float *xyz, const float *rot, const float *weights, const float *joint_mats, int32_t n,
                               int32_t P, int32_t J, const float *dL_dxyz_out, const float *dL_drot_out, float *dL_dxyz,
                               float *dL_drot, int32_t n_extra, const float *const *extra_src, float *const *extra_dst,
                               const int32_t *extra_width, void *stream_)
{
    return warp_backward_frames(xyz, rot, weights, joint_mats, nullptr, (size_t)J * 16, n, P, J, dL_dxyz_out, dL_drot_out, dL_dxyz, dL_drot,
                                n_extra, extra_src, extra_dst, extra_width, stream_);
}

// n views of ONE pose (soar_views_backward): the frames share the joint transforms; the plugin's SDS views carry an axis permutation
int soar_lbs_warp_backward_views(const float *xyz, const float *rot, const float *weights, const float *joint_mats, const float *axis_perm,
                                 int32_t n, int32_t P, int32_t J, const float *dL_dxyz_out, const float *dL_drot_out, float *dL_dxyz,
                                 float *dL_drot, int32_t n_extra, const float *const *extra_src, float *const *extra_dst,
                                 const int32_t *extra_width, void *stream_)
{
    return warp_backward_frames(xyz, rot, weights, joint_mats, axis_perm, 0, n, P, J, dL_dxyz_out, dL_drot_out, dL_dxyz, dL_drot, n_extra,
                                extra_src, extra_dst, extra_width, stream_);
}

static int warp_backward_frames(const float *xyz, const float *rot, const float *weights, const float *joint_mats, const float *axis_perm,
                                size_t mats_stride, int32_t n, int32_t P, int32_t J, const float *dL_dxyz_out, const float *dL_drot_out,
                                float *dL_dxyz, float *dL_drot, int32_t n_extra, const float *const *extra_src, float *const *extra_dst,
                                const int32_t *extra_width, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n <= 0 || !weights) { set_error("soar_lbs_warp_backward_sum: need n > 0 and blend weights"); return 1; }
    if (n_extra < 0 || n_extra > 2 || (n_extra > 0 && (!extra_src || !extra_dst || !extra_width))) {
        set_error("soar_lbs_warp_backward_sum: at most two extra blocks");
        return 1;
    }
    if (warp_check(xyz, rot, weights, joint_mats, P, J)) return 1;
    if (P == 0) return 0;
    if (!dL_dxyz_out || !dL_drot_out || !dL_dxyz || !dL_drot) { set_error("soar_lbs_warp_backward_sum: NULL pointer"); return 1; }
    WarpArgs a{};
    a.P = P; a.J = J; a.xyz = xyz; a.rot = rot; a.weights = weights; a.joint_mats = joint_mats;
    a.g_xyz_out = dL_dxyz_out; a.g_rot_out = dL_drot_out; a.g_xyz = dL_dxyz; a.g_rot = dL_drot;
    a.axis_perm = axis_perm;
    a.mats_stride = mats_stride; a.xyz_stride = (size_t)P * 3; a.rot_stride = (size_t)P * 4;
    FrameSums fs{};
    fs.n_extra = n_extra;
    for (int e = 0; e < n_extra; e++) {
        if (!extra_src[e] || !extra_dst[e] || extra_width[e] <= 0) { set_error("soar_lbs_warp_backward_sum: bad extra block %d", e); return 1; }
        fs.src[e] = extra_src[e]; fs.dst[e] = extra_dst[e]; fs.width[e] = extra_width[e];
    }
    const size_t lds = sizeof(float) * (WAVE * (size_t)J + WARP_NF * 7 * WAVE);
    StageTimer timer(ST_LBS_WARP_BWD, stream);
    hipLaunchKernelGGL(warp_backward_frames_kernel, dim3((P + WAVE - 1) / WAVE), dim3(WARP_THREADS), lds, stream, a, (int)n, fs);
    SOAR_LAUNCH_OK("lbs_warp_backward_sum", stream, 0);
    return 0;
}

int soar_frames_warp_preprocess(int32_t n, const SoarFrameHead *frames, const float *xyz, const float *rot, const float *weights,
                                const float *joint_mats, int32_t P, int32_t J, const float *colors, const float *opacities, const float *scales,
                                float *xyz_out, float *rot_out, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const char *who = "soar_frames_warp_preprocess";
    if (n < 1 || n > MAX_BATCH || !frames) { set_error("%s: 1 <= n <= %d frames", who, MAX_BATCH); return 1; }
    if (!weights) { set_error("%s: needs the blend weights", who); return 1; }
    if (warp_check(xyz, rot, weights, joint_mats, P, J)) return 1;
    if (P == 0) return 0;
    if (!colors || !opacities || !scales || !xyz_out || !rot_out) { set_error("%s: a required pointer is NULL", who); return 1; }
    WarpArgs a{};
    a.P = P; a.J = J; a.xyz = xyz; a.rot = rot; a.weights = weights; a.joint_mats = joint_mats;
    a.xyz_out = xyz_out; a.rot_out = rot_out;
    a.mats_stride = (size_t)J * 16; a.xyz_stride = (size_t)P * 3; a.rot_stride = (size_t)P * 4;
    Batch<PreArgs> fr{};
    for (int f = 0; f < n; f++) {
        const SoarFrameHead &t = frames[f];
        if (!t.prm || !t.geom_buffer || !t.radii) { set_error("%s: frame %d: a pointer is NULL", who, f); return 1; }
        if (t.prm->P != P || t.prm->M != 0 || t.prm->prefiltered != 0) {
            set_error("%s: frame %d: needs P = %d Gaussians with explicit colours (M = 0), not prefiltered", who, f, P);
            return 1;
        }
        GeomBuf g;
        carve_geom(t.geom_buffer, P, 0, &g);
        fill_pre_args(fr.v[f], *t.prm, xyz_out + (size_t)f * a.xyz_stride, nullptr, colors, opacities, scales, rot_out + (size_t)f * a.rot_stride,
                      nullptr, g, t.radii);
    }
    const size_t lds = sizeof(float) * WAVE * (size_t)J;
    StageTimer timer(ST_LBS_WARP_FWD, stream);
    hipLaunchKernelGGL(warp_preprocess_frames_kernel, dim3((P + WAVE - 1) / WAVE, (n + WARP_NF - 1) / WARP_NF), dim3(WARP_THREADS), lds, stream,
                       a, (int)n, fr);
    SOAR_LAUNCH_OK("frames_warp_preprocess", stream, 0);
    return 0;
}

int soar_frames_geometry_warp_backward(int32_t n, const SoarFrameTail *frames, const float *xyz, const float *rot, const float *weights,
                                       const float *joint_mats, int32_t P, int32_t J, const float *scales, float *dL_dxyz, float *dL_drot,
                                       float *dL_dscales, float *dL_dcolors, float *dL_docc, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const char *who = "soar_frames_geometry_warp_backward";
    if (n < 1 || n > MAX_BATCH || !frames) { set_error("%s: 1 <= n <= %d frames", who, MAX_BATCH); return 1; }
    if (!weights) { set_error("%s: needs the blend weights", who); return 1; }
    if (warp_check(xyz, rot, weights, joint_mats, P, J)) return 1;
    if (P == 0) return 0;
    if (!scales || !dL_dxyz || !dL_drot || !dL_dscales || !dL_dcolors) { set_error("%s: a required pointer is NULL", who); return 1; }
    WarpArgs a{};
    a.P = P; a.J = J; a.xyz = xyz; a.rot = rot; a.weights = weights; a.joint_mats = joint_mats;
    a.g_xyz = dL_dxyz; a.g_rot = dL_drot;
    a.mats_stride = (size_t)J * 16;
    Batch<GeomBwdArgs> fr{};
    TailOut out{};
    out.dL_dscales = dL_dscales; out.dL_dcolors = dL_dcolors; out.dL_docc = dL_docc;
    for (int f = 0; f < n; f++) {
        const SoarFrameTail &t = frames[f];
        if (!t.prm || !t.means3D || !t.rotations || !t.radii || !t.geom_buffer || !t.workspace || !t.dL_dmeans2D) {
            set_error("%s: frame %d: a pointer is NULL", who, f);
            return 1;
        }
        if (t.prm->P != P || t.prm->M != 0 || t.prm->cfg_lrn_cam != 0) {
            set_error("%s: frame %d: needs P = %d Gaussians with explicit colours (M = 0) and cfg_lrn_cam = 0", who, f, P);
            return 1;
        }
        GeomBuf g;
        carve_geom(const_cast<void *>(t.geom_buffer), P, 0, &g);
        fill_geom_bwd_args(fr.v[f], *t.prm, t.means3D, t.radii, nullptr, scales, t.rotations, nullptr, g, static_cast<const float *>(t.workspace));
        out.dL_dmeans2D[f] = t.dL_dmeans2D;
    }
    const size_t lds = sizeof(float) * (WAVE * (size_t)J + WARP_NF * TAIL_C * WAVE);
    StageTimer timer(ST_LBS_WARP_BWD, stream);
    hipLaunchKernelGGL(geom_warp_backward_frames_kernel, dim3((P + WAVE - 1) / WAVE), dim3(WARP_THREADS), lds, stream, a, (int)n, fr, out);
    SOAR_LAUNCH_OK("frames_geometry_warp_backward", stream, 0);
    return 0;
}

int soar_dist2_knn3(const float *points, int32_t N, float *out, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (N < 0) { set_error("soar_dist2_knn3: N < 0"); return 1; }
    if (N == 0) return 0;
    if (!points || !out) { set_error("soar_dist2_knn3: NULL pointer"); return 1; }
    StageTimer timer(ST_DIST2, stream);
    hipLaunchKernelGGL(dist2_knn3_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, points, N, out);
    SOAR_LAUNCH_OK("dist2_knn3", stream, 0);
    return 0;
}

}  // extern "C"
