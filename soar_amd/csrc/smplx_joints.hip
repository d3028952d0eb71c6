// smplx_joints.hip -- the SMPL-X joint transforms of B frames in one launch.
//
// Restates, per frame, what the per-frame path consumes of the vendored body model:
//   rest joints   J = J_regressor (v_template + shapedirs betas) = J_template + J_dirs betas   (TS/utils/smplx/lbs.py:201-206)
//   batch_rodrigues                                                                              (lbs.py:293-327)
//   batch_rigid_transform: local = [R | J - J_parent], world_j = world_parent(j) local_j,
//                          A_j = world_j - [0 | world_j J_j]                                     (lbs.py:343-396)
//   + transl on the translation column                                                           (body_models.py:1383)
//   cano2live_j = A_j inv(A_cano)_j                                                              (TS/utils/smpl.py:601-609)
// One 64-lane workgroup per frame, lane = joint (J <= 64): the reference's 55-step Python loop of 4x4 matmuls (~300 tiny
// launches per frame with the Rodrigues / regressor ops around it) becomes one launch for all the frames of a step.
// The chain product is associated exactly like the reference's loop, ((L_root L_a) L_b) ... L_j: every lane multiplies down
// its own root path from the locals in LDS, so no level-by-level barriers are needed.
#include "soar_common.h"

namespace soar {

namespace {

constexpr int MAX_J = 64;
constexpr int MAX_DEPTH = 32;

struct JointArgs {
    int B, J, NB, betas_batch;
    const float *betas;       // [betas_batch, NB]
    const float *J_template;  // [J,3]
    const float *J_dirs;      // [J,3,NB]
    const int *parents;       // [J], parents[0] < 0
    const float *full_pose;   // [B, J*3] axis-angle
    const float *transl;      // [B,3] or nullptr
    const float *right;       // [J,4,4] or nullptr: out_j = A_j right_j
    float *out;               // [B,J,4,4]
};

__global__ void __launch_bounds__(64) smplx_joint_mats_kernel(JointArgs a)
{
    __shared__ float loc[MAX_J][12];     // local transforms, rows 0..2 of [R | t]
    __shared__ float jrest[MAX_J][3];
    const int b = blockIdx.x, j = threadIdx.x;
    const bool live = j < a.J;
    float Jr[3] = {0.f, 0.f, 0.f};
    int par = -1;
    if (live) {
        par = a.parents[j];
        const float *be = a.betas + (size_t)(a.betas_batch > 1 ? b : 0) * a.NB;
        for (int k = 0; k < 3; k++) {
            float s = 0.f;
            for (int l = 0; l < a.NB; l++) s += be[l] * a.J_dirs[((size_t)j * 3 + k) * a.NB + l];
            Jr[k] = a.J_template[j * 3 + k] + s;
            jrest[j][k] = Jr[k];
        }
    }
    __syncthreads();
    if (live) {
        // batch_rodrigues: angle = |v + 1e-8|, K = skew(v / angle), R = I + sin K + (1 - cos) K K
        const float *v = a.full_pose + ((size_t)b * a.J + j) * 3;
        const float vx = v[0], vy = v[1], vz = v[2];
        const float ex = vx + 1e-8f, ey = vy + 1e-8f, ez = vz + 1e-8f;
        const float angle = sqrtf(ex * ex + ey * ey + ez * ez);
        const float rx = vx / angle, ry = vy / angle, rz = vz / angle;
        const float sn = sinf(angle), cs = 1.f - cosf(angle);
        const float K[9] = {0.f, -rz, ry, rz, 0.f, -rx, -ry, rx, 0.f};
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) {
                float kk = 0.f;
                for (int m = 0; m < 3; m++) kk += K[r * 3 + m] * K[m * 3 + c];
                loc[j][r * 4 + c] = (r == c ? 1.f : 0.f) + sn * K[r * 3 + c] + cs * kk;
            }
        for (int k = 0; k < 3; k++) loc[j][k * 4 + 3] = par >= 0 ? Jr[k] - jrest[par][k] : Jr[k];
    }
    __syncthreads();
    if (!live) return;

    // root path of this joint, then the product from the root down (the reference's association order)
    int path[MAX_DEPTH];
    int depth = 0;
    for (int p = j; p >= 0 && depth < MAX_DEPTH; p = a.parents[p]) path[depth++] = p;
    float W[12];
    for (int k = 0; k < 12; k++) W[k] = loc[path[depth - 1]][k];
    for (int d = depth - 2; d >= 0; d--) {
        const float *L = loc[path[d]];
        float N[12];
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 3; c++) N[r * 4 + c] = W[r * 4] * L[c] + W[r * 4 + 1] * L[4 + c] + W[r * 4 + 2] * L[8 + c];
            N[r * 4 + 3] = W[r * 4] * L[3] + W[r * 4 + 1] * L[7] + W[r * 4 + 2] * L[11] + W[r * 4 + 3];
        }
        for (int k = 0; k < 12; k++) W[k] = N[k];
    }
    // A = world - [0 | world (J, 0)]  (+ transl)
    for (int r = 0; r < 3; r++) {
        W[r * 4 + 3] -= W[r * 4] * Jr[0] + W[r * 4 + 1] * Jr[1] + W[r * 4 + 2] * Jr[2];
        if (a.transl) W[r * 4 + 3] += a.transl[b * 3 + r];
    }
    float *o = a.out + ((size_t)b * a.J + j) * 16;
    if (a.right) {
        const float *Rm = a.right + (size_t)j * 16;
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 4; c++)
                o[r * 4 + c] = W[r * 4] * Rm[c] + W[r * 4 + 1] * Rm[4 + c] + W[r * 4 + 2] * Rm[8 + c] + W[r * 4 + 3] * Rm[12 + c];
        for (int c = 0; c < 4; c++) o[12 + c] = Rm[12 + c];          // bottom row of A is (0,0,0,1)
    } else {
        for (int k = 0; k < 12; k++) o[k] = W[k];
        o[12] = 0.f; o[13] = 0.f; o[14] = 0.f; o[15] = 1.f;
    }
}

}  // namespace

}  // namespace soar

using namespace soar;

extern "C" int soar_smplx_joint_mats(int32_t B, int32_t J, int32_t NB, const float *betas, int32_t betas_batch,
                                     const float *J_template, const float *J_dirs, const int32_t *parents,
                                     const float *full_pose, const float *transl, const float *right_mats, float *out,
                                     void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (B < 0 || J <= 0 || J > MAX_J || NB < 0 || (betas_batch != 1 && betas_batch != B)) {
        set_error("soar_smplx_joint_mats: need 0 < J <= %d, betas_batch in {1, B} (B=%d J=%d NB=%d betas_batch=%d)", MAX_J, B, J, NB,
                  betas_batch);
        return 1;
    }
    if (B == 0) return 0;
    if ((NB > 0 && (!betas || !J_dirs)) || !J_template || !parents || !full_pose || !out) {
        set_error("soar_smplx_joint_mats: NULL argument");
        return 1;
    }
    JointArgs a = {B, J, NB, betas_batch, betas, J_template, J_dirs, parents, full_pose, transl, right_mats, out};
    hipLaunchKernelGGL(smplx_joint_mats_kernel, dim3(B), dim3(64), 0, stream, a);
    SOAR_LAUNCH_OK("smplx_joint_mats", stream, 0);
    return 0;
}
