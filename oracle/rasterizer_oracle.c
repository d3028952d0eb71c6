/*
 * oracle/rasterizer_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See rasterizer_oracle.h.
 *
 * PARITY STATUS: **parity unpinned** by the rules of this build.  The reference holds no golden vectors, known-answer tests or
 * fixtures for the rasterizer, and its CUDA kernels cannot be compiled here with their own toolchain.  What this restatement IS
 * checked against (tests/test_reference_build_gpu.py) is the reference's own kernel sources translated by hipify-perl and built
 * for gfx950 (oracle/_ref, recipe oracle/ref_build/build_ref.sh) -- the strongest evidence available in this image, disclosed in
 * DESIGN.md section 3, but a translated build is not the reference compiled with its own toolchain.
 *
 * Scalar CPU restatement of hangg7/soar's Gaussian-surfel rasterizer.  Citations are
 * file:line relative to /root/reference/submodules/diff-gaussian-rasterization/ ("DGR/").
 * Build: gcc -O2 -ffp-contract=off (no -ffast-math, no -march=native): every fp32
 * expression is evaluated in the order the reference source writes it.
 */
#include "rasterizer_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define BLOCK_X 16            /* DGR/cuda_rasterizer/config.h:14-16 */
#define BLOCK_Y 16
#define BLOCK_SIZE (BLOCK_X * BLOCK_Y)
#define NUM_CHANNELS 3

/* ------------------------------------------------------------------------------------------
 * small helpers
 * ---------------------------------------------------------------------------------------- */

/* float -> int conversion with the GPU's semantics (saturating, NaN -> 0); on the value ranges the
 * rasterizer produces this equals the plain C cast. */
static inline int f2i(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int)v;
}

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* glm::mat3 is column-major: m[c][r].  Multiplication order of operations follows
 * DGR/third_party/glm/glm/detail/type_mat3x3.inl:486-519. */
typedef struct { float m[3][3]; } mat3;

static inline mat3 mat3_cols(float a0, float a1, float a2, float b0, float b1, float b2,
                             float c0, float c1, float c2)
{
    mat3 r;
    r.m[0][0] = a0; r.m[0][1] = a1; r.m[0][2] = a2;
    r.m[1][0] = b0; r.m[1][1] = b1; r.m[1][2] = b2;
    r.m[2][0] = c0; r.m[2][1] = c1; r.m[2][2] = c2;
    return r;
}

static inline mat3 mat3_mul(mat3 a, mat3 b)
{
    mat3 r;
    for (int c = 0; c < 3; c++)
        for (int row = 0; row < 3; row++)
            r.m[c][row] = a.m[0][row] * b.m[c][0] + a.m[1][row] * b.m[c][1] + a.m[2][row] * b.m[c][2];
    return r;
}

static inline mat3 mat3_transpose(mat3 a)
{
    mat3 r;
    for (int c = 0; c < 3; c++)
        for (int row = 0; row < 3; row++)
            r.m[c][row] = a.m[row][c];
    return r;
}

/* DGR/cuda_rasterizer/auxiliary.h:42-46 -- evaluated in double, rounded to float on return. */
static inline float ndc2Pix(float v, int S, float prcp)
{
    return (float)(((v + 1.0) * S - 1.0) * 0.5 + S * (prcp - 0.5));
}

/* DGR/cuda_rasterizer/auxiliary.h:53-63 */
static inline void getRect(float px, float py, int max_radius, uint32_t *rect_min, uint32_t *rect_max,
                           uint32_t grid_x, uint32_t grid_y)
{
    rect_min[0] = (uint32_t)imin((int)grid_x, imax(0, f2i((px - max_radius) / BLOCK_X)));
    rect_min[1] = (uint32_t)imin((int)grid_y, imax(0, f2i((py - max_radius) / BLOCK_Y)));
    rect_max[0] = (uint32_t)imin((int)grid_x, imax(0, f2i((px + max_radius + BLOCK_X - 1) / BLOCK_X)));
    rect_max[1] = (uint32_t)imin((int)grid_y, imax(0, f2i((py + max_radius + BLOCK_Y - 1) / BLOCK_Y)));
}

/* DGR/cuda_rasterizer/auxiliary.h:65-104 */
static inline void transformPoint4x3(const float *p, const float *m, float *o)
{
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
static inline void transformPoint4x4(const float *p, const float *m, float *o)
{
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}
static inline void transformVec4x3(const float *p, const float *m, float *o)
{
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2];
}
static inline void transformVec4x3Transpose(const float *p, const float *m, float *o)
{
    o[0] = m[0] * p[0] + m[1] * p[1] + m[2] * p[2];
    o[1] = m[4] * p[0] + m[5] * p[1] + m[6] * p[2];
    o[2] = m[8] * p[0] + m[9] * p[1] + m[10] * p[2];
}

/* DGR/cuda_rasterizer/auxiliary.h:114-124 */
static inline void dnormvdv3(const float *v, const float *dv, float *o)
{
    float sum2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    o[0] = ((+sum2 - v[0] * v[0]) * dv[0] - v[1] * v[0] * dv[1] - v[2] * v[0] * dv[2]) * invsum32;
    o[1] = (-v[0] * v[1] * dv[0] + (sum2 - v[1] * v[1]) * dv[1] - v[2] * v[1] * dv[2]) * invsum32;
    o[2] = (-v[0] * v[2] * dv[0] - v[1] * v[2] * dv[1] + (sum2 - v[2] * v[2]) * dv[2]) * invsum32;
}

/* DGR/cuda_rasterizer/auxiliary.h:236-242 */
static inline float normalize3(float *v)
{
    float mod = fmaxf(sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), (float)0.00000001);
    v[0] /= mod; v[1] /= mod; v[2] /= mod;
    return mod;
}

/* DGR/cuda_rasterizer/auxiliary.h:291-388.  Returns 1 when the surfel is seen at a grazing angle. */
static int local_homo(const float *p_view, const float *n_view, float fx, float fy,
                      const float *ax0, const float *ax1, float *res)
{
    float p_prj[2] = { p_view[0] / p_view[2], p_view[1] / p_view[2] };
    float S_fix = 1000, Svp = (fx + fy) / 2;
    float dir_x0[3], dir_x1[3];
    dir_x0[0] = p_prj[0] + 1 / S_fix; dir_x0[1] = p_prj[1]; dir_x0[2] = 1;
    float dir_x0_mod = normalize3(dir_x0);
    dir_x1[0] = p_prj[0]; dir_x1[1] = p_prj[1] + 1 / S_fix; dir_x1[2] = 1;
    float dir_x1_mod = normalize3(dir_x1);

    float prj_x0, prj_x1, thrsh_prj = (float)0.01;
    prj_x0 = dir_x0[0] * n_view[0] + dir_x0[1] * n_view[1] + dir_x0[2] * n_view[2];
    prj_x1 = dir_x1[0] * n_view[0] + dir_x1[1] * n_view[1] + dir_x1[2] * n_view[2];
    int cond_prj = (fabsf(prj_x0 / dir_x0_mod) < thrsh_prj) || (fabsf(prj_x1 / dir_x1_mod) < thrsh_prj);
    if (cond_prj) return 1;

    float t_temp, t_x0, t_x1, xu0[3], xu1[3], u0[3], u1[3];
    t_temp = p_view[0] * n_view[0] + p_view[1] * n_view[1] + p_view[2] * n_view[2];
    t_x0 = t_temp / prj_x0;
    t_x1 = t_temp / prj_x1;
    for (int i = 0; i < 3; i++) {
        xu0[i] = dir_x0[i] * t_x0 - p_view[i];
        xu1[i] = dir_x1[i] * t_x1 - p_view[i];
    }
    /* auxiliary.h:349-363: the Surface-Splatting basis is computed and then overwritten by the
     * view-space rotation axes, so only the latter survives. */
    for (int i = 0; i < 3; i++) { u0[i] = ax0[i]; u1[i] = ax1[i]; }

    float J_inv[4];
    J_inv[0] = xu0[0] * u0[0] + xu0[1] * u0[1] + xu0[2] * u0[2];
    J_inv[1] = xu1[0] * u0[0] + xu1[1] * u0[1] + xu1[2] * u0[2];
    J_inv[2] = xu0[0] * u1[0] + xu0[1] * u1[1] + xu0[2] * u1[2];
    J_inv[3] = xu1[0] * u1[0] + xu1[1] * u1[1] + xu1[2] * u1[2];
    for (int i = 0; i < 4; i++) J_inv[i] /= (Svp / S_fix);
    for (int i = 0; i < 4; i++) res[i] = J_inv[i];
    for (int i = 0; i < 3; i++) { res[4 + i] = u0[i]; res[7 + i] = u1[i]; }
    return 0;
}

/* DGR/cuda_rasterizer/auxiliary.h:390-397 (only .z is consumed by the renderers). */
static inline float depth_differencing_z(float dx, float dy, const float *J)
{
    float dif_u0 = dx * J[0] + dy * J[1];
    float dif_u1 = dx * J[2] + dy * J[3];
    return dif_u0 * J[6] + dif_u1 * J[9];
}

/* DGR/cuda_rasterizer/rasterizer_impl.cu:35-48 */
uint32_t oracle_get_higher_msb(uint32_t n)
{
    uint32_t msb = sizeof(n) * 4;
    uint32_t step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step; else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

/* SH constants, DGR/cuda_rasterizer/auxiliary.h:23-40 */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = { 1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                -1.0925484305920792f, 0.5462742152960396f };
static const float SH_C3[7] = { -0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                -0.5900435899266435f };

/* DGR/cuda_rasterizer/forward.cu:20-71 */
static void computeColorFromSH_fwd(int idx, int deg, int max_coeffs, const float *means, const float *campos,
                                   const float *shs, uint8_t *clamped, float *out)
{
    float dir[3] = { means[3 * idx] - campos[0], means[3 * idx + 1] - campos[1], means[3 * idx + 2] - campos[2] };
    float len = sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
    dir[0] = dir[0] / len; dir[1] = dir[1] / len; dir[2] = dir[2] / len;
    const float *sh = shs + (size_t)idx * max_coeffs * 3;
    float x = dir[0], y = dir[1], z = dir[2];
    for (int c = 0; c < 3; c++) {
#define SH(k) sh[(k) * 3 + c]
        float result = SH_C0 * SH(0);
        if (deg > 0) {
            result = result - SH_C1 * y * SH(1) + SH_C1 * z * SH(2) - SH_C1 * x * SH(3);
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z;
                float xy = x * y, yz = y * z, xz = x * z;
                result = result + SH_C2[0] * xy * SH(4) + SH_C2[1] * yz * SH(5) +
                         SH_C2[2] * (2.0f * zz - xx - yy) * SH(6) + SH_C2[3] * xz * SH(7) +
                         SH_C2[4] * (xx - yy) * SH(8);
                if (deg > 2) {
                    result = result + SH_C3[0] * y * (3.0f * xx - yy) * SH(9) + SH_C3[1] * xy * z * SH(10) +
                             SH_C3[2] * y * (4.0f * zz - xx - yy) * SH(11) +
                             SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SH(12) +
                             SH_C3[4] * x * (4.0f * zz - xx - yy) * SH(13) + SH_C3[5] * z * (xx - yy) * SH(14) +
                             SH_C3[6] * x * (xx - 3.0f * yy) * SH(15);
                }
            }
        }
#undef SH
        result += 0.5f;
        clamped[3 * idx + c] = (result < 0);
        out[c] = fmaxf(result, 0.0f);
    }
}

/* ------------------------------------------------------------------------------------------
 * forward: preprocess (DGR/cuda_rasterizer/forward.cu:205-385) + inclusive scan
 * (DGR/cuda_rasterizer/rasterizer_impl.cu:242-252)
 * ---------------------------------------------------------------------------------------- */
int64_t oracle_preprocess(const OracleParams *prm,
                          const float *means3D, const float *scales, const float *rotations,
                          const float *opacities, const float *shs, const float *cov3D_precomp,
                          const float *colors_precomp, OracleGeom *g)
{
    const int P = prm->P, W = prm->W, H = prm->H;
    /* rasterizer_impl.cu:201-202 */
    const float focal_y = H / (2.0f * prm->tanfovy);
    const float focal_x = W / (2.0f * prm->tanfovx);
    const uint32_t grid_x = (W + BLOCK_X - 1) / BLOCK_X, grid_y = (H + BLOCK_Y - 1) / BLOCK_Y;
    const float *viewmatrix = prm->viewmatrix, *projmatrix = prm->projmatrix;
    const int surface = prm->config[0] > 0, pix_depth = prm->config[2] > 0;   /* forward.cu:275 */

    memset(g->radii, 0, sizeof(int32_t) * P);
    memset(g->means2D, 0, sizeof(float) * 2 * P);
    memset(g->depths, 0, sizeof(float) * P);
    memset(g->cov3D, 0, sizeof(float) * 6 * P);
    memset(g->conic_opacity, 0, sizeof(float) * 4 * P);
    if (g->rgb) memset(g->rgb, 0, sizeof(float) * 3 * P);
    if (g->clamped) memset(g->clamped, 0, 3 * (size_t)P);
    memset(g->normal, 0, sizeof(float) * 3 * P);
    memset(g->Jinv, 0, sizeof(float) * 10 * P);
    memset(g->viewCos, 0, sizeof(float) * P);
    memset(g->tiles_touched, 0, sizeof(uint32_t) * P);

    for (int idx = 0; idx < P; idx++) {
        /* forward.cu:249-250: radii / tiles_touched start at 0 (done by memset above) */
        const float p_orig[3] = { means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2] };
        float p_hom[4];
        transformPoint4x4(p_orig, projmatrix, p_hom);
        float p_w = 1.0f / (p_hom[3] + 0.0000001f);
        float p_proj[3] = { p_hom[0] * p_w, p_hom[1] * p_w, p_hom[2] * p_w };
        float p_view[3];
        transformPoint4x3(p_orig, viewmatrix, p_view);

        float point_image[2] = { ndc2Pix(p_proj[0], W, prm->prcppoint[0]), ndc2Pix(p_proj[1], H, prm->prcppoint[1]) };
        {   /* in_frustum, auxiliary.h:146-171 (prefiltered => device trap in the reference; treated as cull) */
            float x0 = prm->patchbbox[1], y0 = prm->patchbbox[0], x1 = prm->patchbbox[3], y1 = prm->patchbbox[2];
            float w = x1 - x0, h = y1 - y0;
            float expand = (float)0.2;
            if (p_view[2] < 0 || point_image[0] < x0 - w * expand || point_image[0] >= x1 + w * expand ||
                point_image[1] < y0 - h * expand || point_image[1] >= y1 + h * expand)
                continue;
        }

        /* quaternion2rotmat, forward.cu:141-156 (no normalisation) */
        float r = rotations ? rotations[4 * idx] : 1.f, x = rotations ? rotations[4 * idx + 1] : 0.f,
              y = rotations ? rotations[4 * idx + 2] : 0.f, z = rotations ? rotations[4 * idx + 3] : 0.f;
        mat3 R = mat3_cols(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                           2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                           2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));

        if (surface) {   /* forward.cu:278-309 */
            float nw[3] = { R.m[0][2], R.m[1][2], R.m[2][2] };
            float a0[3] = { R.m[0][0], R.m[1][0], R.m[2][0] };
            float a1[3] = { R.m[0][1], R.m[1][1], R.m[2][1] };
            float n_view[3], ax0_view[3], ax1_view[3];
            transformVec4x3(nw, viewmatrix, n_view);
            transformVec4x3(a0, viewmatrix, ax0_view);
            transformVec4x3(a1, viewmatrix, ax1_view);
            /* front_facing, auxiliary.h:173-208 */
            float dot = p_view[0] * n_view[0] + p_view[1] * n_view[1] + p_view[2] * n_view[2];
            int front = !((double)dot > -0.01);   /* auxiliary.h:181: double literal => double compare */
            if (front) g->viewCos[idx] = dot;
            if (prm->render_front && !front) continue;
            g->normal[idx * 3 + 0] = n_view[0];
            g->normal[idx * 3 + 1] = n_view[1];
            g->normal[idx * 3 + 2] = n_view[2];
            if (pix_depth) {
                float Jinv_u0_u1[10];
                int grazing = local_homo(p_view, n_view, focal_x, focal_y, ax0_view, ax1_view, Jinv_u0_u1);
                if (grazing) continue;
                for (int i = 0; i < 10; i++) g->Jinv[idx * 10 + i] = Jinv_u0_u1[i];
            }
        }

        /* computeCov3D, forward.cu:162-202 (note the precedence quirk at :168) */
        const float *cov3D;
        if (cov3D_precomp != NULL) {
            cov3D = cov3D_precomp + idx * 6;
        } else {
            const float mod = prm->scale_modifier;
            mat3 S = mat3_cols(1, 0, 0, 0, 1, 0, 0, 0, 1);
            S.m[0][0] = mod * scales[3 * idx + 0];
            S.m[1][1] = mod * scales[3 * idx + 1];
            S.m[2][2] = ((mod * (surface ? 1.0f : 0.0f)) != 0.0f) ? 0 : scales[3 * idx + 2];
            mat3 Mm = mat3_mul(S, R);
            mat3 Sigma = mat3_mul(mat3_transpose(Mm), Mm);
            float *c = g->cov3D + idx * 6;
            c[0] = Sigma.m[0][0]; c[1] = Sigma.m[0][1]; c[2] = Sigma.m[0][2];
            c[3] = Sigma.m[1][1]; c[4] = Sigma.m[1][2]; c[5] = Sigma.m[2][2];
            cov3D = c;
        }

        /* computeCov2D on the VIEW-space point, forward.cu:74-139,329 */
        float cov[3];
        {
            float t[3] = { p_view[0], p_view[1], p_view[2] };
            const float limx = 1.3f * prm->tanfovx, limy = 1.3f * prm->tanfovy;
            const float txtz = t[0] / t[2], tytz = t[1] / t[2];
            t[0] = fminf(limx, fmaxf(-limx, txtz)) * t[2];
            t[1] = fminf(limy, fmaxf(-limy, tytz)) * t[2];
            mat3 J = mat3_cols(focal_x / t[2], 0.0f, -(focal_x * t[0]) / (t[2] * t[2]),
                               0.0f, focal_y / t[2], -(focal_y * t[1]) / (t[2] * t[2]),
                               0, 0, 0);
            mat3 Wm = mat3_cols(viewmatrix[0], viewmatrix[4], viewmatrix[8],
                                viewmatrix[1], viewmatrix[5], viewmatrix[9],
                                viewmatrix[2], viewmatrix[6], viewmatrix[10]);
            mat3 T = mat3_mul(Wm, J);
            mat3 Vrk = mat3_cols(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4],
                                 cov3D[2], cov3D[4], cov3D[5]);
            mat3 c2 = mat3_mul(mat3_mul(mat3_transpose(T), mat3_transpose(Vrk)), T);
            c2.m[0][0] += 0.3f;
            c2.m[1][1] += 0.3f;
            cov[0] = c2.m[0][0]; cov[1] = c2.m[0][1]; cov[2] = c2.m[1][1];
        }

        /* forward.cu:337-355 */
        float det = (cov[0] * cov[2] - cov[1] * cov[1]);
        if (det == 0.0f) continue;
        float det_inv = 1.f / det;
        float conic[3] = { cov[2] * det_inv, -cov[1] * det_inv, cov[0] * det_inv };
        float mid = 0.5f * (cov[0] + cov[2]);
        float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
        float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
        float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
        uint32_t rect_min[2], rect_max[2];
        getRect(point_image[0], point_image[1], f2i(my_radius), rect_min, rect_max, grid_x, grid_y);
        if ((rect_max[0] - rect_min[0]) * (rect_max[1] - rect_min[1]) == 0) continue;

        /* forward.cu:359-365 */
        if (colors_precomp == NULL) {
            float res[3];
            computeColorFromSH_fwd(idx, prm->sh_degree, prm->M, means3D, prm->campos, shs, g->clamped, res);
            g->rgb[idx * 3 + 0] = res[0]; g->rgb[idx * 3 + 1] = res[1]; g->rgb[idx * 3 + 2] = res[2];
        }

        /* forward.cu:377-383 */
        g->depths[idx] = p_view[2];
        g->radii[idx] = f2i(my_radius);
        g->means2D[2 * idx] = point_image[0];
        g->means2D[2 * idx + 1] = point_image[1];
        g->conic_opacity[4 * idx + 0] = conic[0];
        g->conic_opacity[4 * idx + 1] = conic[1];
        g->conic_opacity[4 * idx + 2] = conic[2];
        g->conic_opacity[4 * idx + 3] = opacities[idx];
        g->tiles_touched[idx] = (rect_max[1] - rect_min[1]) * (rect_max[0] - rect_min[0]);
    }

    /* InclusiveSum, rasterizer_impl.cu:242-245; num_rendered = point_offsets[P-1] (:250) */
    uint32_t acc = 0;
    for (int i = 0; i < P; i++) { acc += g->tiles_touched[i]; g->point_offsets[i] = acc; }
    return P > 0 ? (int64_t)g->point_offsets[P - 1] : 0;
}

/* ------------------------------------------------------------------------------------------
 * forward: duplicateWithKeys (rasterizer_impl.cu:66-99), stable radix sort on bits [0, 32+bit)
 * (rasterizer_impl.cu:266-285; cub::DeviceRadixSort is stable, also when descending),
 * identifyTileRanges (rasterizer_impl.cu:104-124, 287-295)
 * ---------------------------------------------------------------------------------------- */
void oracle_bin(const OracleParams *prm, const OracleGeom *g, int64_t R,
                uint64_t *keys_unsorted, uint32_t *vals_unsorted,
                uint64_t *keys_sorted, uint32_t *vals_sorted, uint32_t *ranges)
{
    const int P = prm->P;
    const uint32_t grid_x = (prm->W + BLOCK_X - 1) / BLOCK_X, grid_y = (prm->H + BLOCK_Y - 1) / BLOCK_Y;

    for (int idx = 0; idx < P; idx++) {
        if (g->radii[idx] > 0) {
            uint32_t off = (idx == 0) ? 0 : g->point_offsets[idx - 1];
            uint32_t rect_min[2], rect_max[2];
            getRect(g->means2D[2 * idx], g->means2D[2 * idx + 1], g->radii[idx], rect_min, rect_max, grid_x, grid_y);
            uint32_t depth_bits;
            memcpy(&depth_bits, &g->depths[idx], 4);
            for (int y = (int)rect_min[1]; y < (int)rect_max[1]; y++) {
                for (int x = (int)rect_min[0]; x < (int)rect_max[0]; x++) {
                    uint64_t key = (uint64_t)(y * grid_x + x);
                    key <<= 32;
                    key |= depth_bits;
                    keys_unsorted[off] = key;
                    vals_unsorted[off] = (uint32_t)idx;
                    off++;
                }
            }
        }
    }

    /* stable LSD radix sort, 8-bit digits, restricted to bits [0, 32+bit) */
    const int end_bit = 32 + (int)oracle_get_higher_msb(grid_x * grid_y);
    uint64_t *ka = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(R > 0 ? R : 1));
    uint32_t *va = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(R > 0 ? R : 1));
    uint64_t *kb = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(R > 0 ? R : 1));
    uint32_t *vb = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(R > 0 ? R : 1));
    memcpy(ka, keys_unsorted, sizeof(uint64_t) * (size_t)R);
    memcpy(va, vals_unsorted, sizeof(uint32_t) * (size_t)R);
    for (int shift = 0; shift < end_bit; shift += 8) {
        int nbits = end_bit - shift < 8 ? end_bit - shift : 8;
        uint32_t mask = (1u << nbits) - 1u;
        int64_t count[257];
        memset(count, 0, sizeof(count));
        for (int64_t i = 0; i < R; i++) {
            uint32_t d = (uint32_t)(ka[i] >> shift) & mask;
            if (prm->sort_descending) d = mask - d;
            count[d + 1]++;
        }
        for (int d = 0; d < 256; d++) count[d + 1] += count[d];
        for (int64_t i = 0; i < R; i++) {
            uint32_t d = (uint32_t)(ka[i] >> shift) & mask;
            if (prm->sort_descending) d = mask - d;
            int64_t dst = count[d]++;
            kb[dst] = ka[i]; vb[dst] = va[i];
        }
        uint64_t *tk = ka; ka = kb; kb = tk;
        uint32_t *tv = va; va = vb; vb = tv;
    }
    memcpy(keys_sorted, ka, sizeof(uint64_t) * (size_t)R);
    memcpy(vals_sorted, va, sizeof(uint32_t) * (size_t)R);
    free(ka); free(va); free(kb); free(vb);

    memset(ranges, 0, sizeof(uint32_t) * 2 * (size_t)grid_x * grid_y);
    for (int64_t idx = 0; idx < R; idx++) {
        uint32_t currtile = (uint32_t)(keys_sorted[idx] >> 32);
        if (idx == 0) ranges[2 * currtile] = 0;
        else {
            uint32_t prevtile = (uint32_t)(keys_sorted[idx - 1] >> 32);
            if (currtile != prevtile) {
                ranges[2 * prevtile + 1] = (uint32_t)idx;
                ranges[2 * currtile] = (uint32_t)idx;
            }
        }
        if (idx == R - 1) ranges[2 * currtile + 1] = (uint32_t)R;
    }
}

/* ------------------------------------------------------------------------------------------
 * forward: renderCUDA (DGR/cuda_rasterizer/forward.cu:390-692).  One pixel at a time: the
 * block-cooperative fetch (:479-494) and the block-wide vote (:475) only affect scheduling.
 * ---------------------------------------------------------------------------------------- */
void oracle_render_forward(const OracleParams *prm, const OracleGeom *g, const float *features,
                           const uint32_t *point_list, const uint32_t *ranges,
                           float *final_T, float *final_D, uint32_t *n_contrib,
                           float *out_color, float *out_normal, float *out_depth, float *out_opac,
                           int n_threads)
{
    const int W = prm->W, H = prm->H;
    const int grid_x = (W + BLOCK_X - 1) / BLOCK_X, grid_y = (H + BLOCK_Y - 1) / BLOCK_Y;
    const int surface = prm->config[0] > 0, per_pixel_depth = prm->config[2] > 0, normalize_depth = prm->config[1] > 0;
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(n_threads > 1 ? n_threads : 1)
#endif
    for (int tile = 0; tile < grid_x * grid_y; tile++) {
        const int tx = tile % grid_x, ty = tile / grid_x;
        const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        for (int ly = 0; ly < BLOCK_Y; ly++)
            for (int lx = 0; lx < BLOCK_X; lx++) {
                const int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
                if (!(px < W && py < H)) continue;                   /* :435-438 */
                const uint32_t pix_id = (uint32_t)W * py + px;
                const float pixf[2] = { (float)px, (float)py };
                float T = 1.0f, test_T;
                uint32_t contributor = 0, last_contributor = 0;
                float C[NUM_CHANNELS] = { 0 }, N[3] = { 0 }, D = 0;
                for (uint32_t i = r0; i < r1; i++) {
                    contributor++;                                     /* :500 */
                    const uint32_t id = point_list[i];
                    const float xy[2] = { g->means2D[2 * id], g->means2D[2 * id + 1] };
                    const float d[2] = { xy[0] - pixf[0], xy[1] - pixf[1] };
                    const float *con_o = g->conic_opacity + 4 * id;
                    float dist = (con_o[0] * d[0] * d[0] + con_o[2] * d[1] * d[1]) + 2 * con_o[1] * d[0] * d[1];
                    float power = -0.5f * dist;
                    if (power > 0.0f) continue;                        /* :512 */
                    float alpha = fminf(0.99f, con_o[3] * expf(power)); /* :519 */
                    if (alpha < 1.0f / 255.0f) continue;               /* :545 */
                    test_T = T * (1 - alpha);
                    if (test_T < 0.0001f) break;                       /* :549-552 (done) */
                    float w = alpha * T;
                    float depth_temp = g->depths[id];
                    if (surface && per_pixel_depth)
                        depth_temp -= depth_differencing_z(d[0], d[1], g->Jinv + 10 * id);   /* :556-577 */
                    D += depth_temp * w;
                    for (int ch = 0; ch < NUM_CHANNELS; ch++) C[ch] += features[id * NUM_CHANNELS + ch] * w;
                    if (surface) for (int ch = 0; ch < 3; ch++) N[ch] += g->normal[id * 3 + ch] * w;
                    T = test_T;
                    last_contributor = contributor;
                }
                /* epilogue :618-633 */
                T = fminf((float)(1 - 0.000001), T);
                final_T[pix_id] = T;
                n_contrib[pix_id] = last_contributor;
                for (int ch = 0; ch < NUM_CHANNELS; ch++)
                    out_color[(size_t)ch * H * W + pix_id] = C[ch] + T * prm->bg[ch];
                out_normal[(size_t)0 * H * W + pix_id] = surface ? N[0] : 0;
                out_normal[(size_t)1 * H * W + pix_id] = surface ? N[1] : 0;
                out_normal[(size_t)2 * H * W + pix_id] = surface ? N[2] : 0;
                out_depth[pix_id] = normalize_depth ? D / (1 - T) : D + T * 10;
                out_opac[pix_id] = 1 - T;
                if (normalize_depth) final_D[pix_id] = D;
            }
    }
}

/* ------------------------------------------------------------------------------------------
 * backward: renderCUDA (DGR/cuda_rasterizer/backward.cu:529-858)
 * ---------------------------------------------------------------------------------------- */
typedef struct { double v[13]; } PairAcc;   /* mean2D.xy, conic.x,y,w, opacity, color rgb, normal xyz, depth */

void oracle_render_backward(const OracleParams *prm, const OracleGeom *g, const float *features,
                            const uint32_t *point_list, const uint32_t *ranges,
                            const float *final_Ts, const float *final_Ds, const uint32_t *n_contrib,
                            const float *dL_dpixcolor, const float *dL_dpixnormal,
                            const float *dL_dpixdepth, const float *dL_dpixopac,
                            float *dL_dmean2D, float *dL_dconic, float *dL_dopacity,
                            float *dL_dcolors, float *dL_dnormal, float *dL_ddepth,
                            int n_threads)
{
    const int W = prm->W, H = prm->H, P = prm->P;
    const int grid_x = (W + BLOCK_X - 1) / BLOCK_X, grid_y = (H + BLOCK_Y - 1) / BLOCK_Y;
    const int surface = prm->config[0] > 0, per_pixel_depth = prm->config[2] > 0, normalize_depth = prm->config[1] > 0;
    const int C = NUM_CHANNELS;
    PairAcc *acc = (PairAcc *)calloc((size_t)(P > 0 ? P : 1), sizeof(PairAcc));
    const float ddelx_dx = (float)(0.5 * W), ddely_dy = (float)(0.5 * H);     /* :622-623 */
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(n_threads > 1 ? n_threads : 1)
#endif
    for (int tile = 0; tile < grid_x * grid_y; tile++) {
        const int tx = tile % grid_x, ty = tile / grid_x;
        const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        const int toDo = (int)(r1 - r0);
        for (int ly = 0; ly < BLOCK_Y; ly++)
            for (int lx = 0; lx < BLOCK_X; lx++) {
                const int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
                if (!(px < W && py < H)) continue;
                const uint32_t pix_id = (uint32_t)W * py + px;
                const float pixf[2] = { (float)px, (float)py };
                const float T_final = final_Ts[pix_id];                        /* :595 */
                const float D_final = normalize_depth ? final_Ds[pix_id] : 0;  /* :597 */
                float T = T_final;
                uint32_t contributor = (uint32_t)toDo;                         /* :603 */
                const int last_contributor = (int)n_contrib[pix_id];
                float accum_rec[3] = { 0 }, accum_rec_n[3] = { 0 }, accum_rec_d = 0;
                float dL_dpixC[3], dL_dpixN[3], dL_dpixD, dL_dpixO;
                for (int i = 0; i < C; i++) dL_dpixC[i] = dL_dpixcolor[(size_t)i * H * W + pix_id];
                for (int i = 0; i < 3; i++) dL_dpixN[i] = dL_dpixnormal[(size_t)i * H * W + pix_id];
                dL_dpixD = dL_dpixdepth[pix_id] * 1;
                dL_dpixO = dL_dpixopac[pix_id];
                float last_alpha = 0, last_color[3] = { 0 }, last_normal[3] = { 0 }, last_depth = 0;

                for (int k = 0; k < toDo; k++) {
                    contributor--;                                             /* :652 */
                    if ((int64_t)contributor >= (int64_t)last_contributor) continue;
                    const uint32_t global_id = point_list[r1 - 1 - (uint32_t)k];  /* :634 */
                    const float xy[2] = { g->means2D[2 * global_id], g->means2D[2 * global_id + 1] };
                    const float d[2] = { xy[0] - pixf[0], xy[1] - pixf[1] };
                    const float *con_o = g->conic_opacity + 4 * global_id;
                    const float dist = (con_o[0] * d[0] * d[0] + con_o[2] * d[1] * d[1]) + 2 * con_o[1] * d[0] * d[1];
                    const float power = -0.5f * dist;
                    if (power > 0.0f) continue;
                    const float G = expf(power);
                    float alpha = fminf(0.99f, con_o[3] * G);
                    if (alpha < 1.0f / 255.0f) continue;

                    T = T / (1.f - alpha);                                      /* :683 */
                    const float dchannel_dcolor = alpha * T;
                    float Jv[10] = { 0 };
                    if (surface && per_pixel_depth) for (int ch = 0; ch < 10; ch++) Jv[ch] = g->Jinv[10 * global_id + ch];

                    float dL_dalpha = 0.0f;
                    PairAcc *a = &acc[global_id];
                    for (int ch = 0; ch < C; ch++) {                            /* :698-713 */
                        const float c_cur = features[global_id * C + ch];
                        accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                        last_color[ch] = c_cur;
                        float dL_dchannel = dL_dpixC[ch], dL_dalpha_color = 0;
                        dL_dalpha_color += (c_cur - accum_rec[ch]) * dL_dchannel;
                        float contrib = dchannel_dcolor * dL_dchannel;
#ifdef _OPENMP
#pragma omp atomic
#endif
                        a->v[6 + ch] += contrib;
                        dL_dalpha += dL_dalpha_color;
                    }
                    if (surface) {                                              /* :715-731 (x10 gain at :727) */
                        for (int ch = 0; ch < 3; ch++) {
                            const float n_cur = g->normal[global_id * 3 + ch];
                            accum_rec_n[ch] = last_alpha * last_normal[ch] + (1.f - last_alpha) * accum_rec_n[ch];
                            last_normal[ch] = n_cur;
                            float dL_dchannel = dL_dpixN[ch], dL_dalpha_normal = 0;
                            dL_dalpha_normal += (n_cur - accum_rec_n[ch]) * dL_dchannel;
                            float contrib = dchannel_dcolor * dL_dchannel * 10;
#ifdef _OPENMP
#pragma omp atomic
#endif
                            a->v[9 + ch] += contrib;
                            dL_dalpha += dL_dalpha_normal;
                        }
                    }
                    {                                                           /* :758-784 */
                        float d_cur = g->depths[global_id];
                        if (surface && per_pixel_depth) d_cur -= depth_differencing_z(d[0], d[1], Jv);
                        accum_rec_d = last_alpha * last_depth + (1.f - last_alpha) * accum_rec_d;
                        last_depth = d_cur;
                        float dL_dchannel = dL_dpixD, dL_dalpha_depth = 0;
                        if (normalize_depth) {
                            dL_dchannel /= (1.f - T_final);
                            dL_dalpha_depth += dL_dpixD * D_final / (1.f - T_final) / (1.f - T_final) * -T_final / (1 - alpha) / T;
                        }
                        dL_dalpha_depth += (d_cur - accum_rec_d) * dL_dchannel;
                        float contrib = dchannel_dcolor * dL_dchannel * 1;
#ifdef _OPENMP
#pragma omp atomic
#endif
                        a->v[12] += contrib;
                        dL_dalpha += dL_dalpha_depth;
                    }

                    dL_dalpha *= T;                                             /* :788 */
                    dL_dalpha += dL_dpixO * T_final / (1 - alpha);              /* :791 */
                    last_alpha = alpha;
                    float bg_dot_dpixel = 0;
                    for (int i = 0; i < C; i++) bg_dot_dpixel += prm->bg[i] * dL_dpixC[i];
                    dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;    /* :801 */
                    if (!normalize_depth) dL_dalpha += (-T_final / (1.f - alpha)) * (10 * dL_dpixD);   /* :802 */

                    float dL_ddist = 0;
                    dL_ddist += dL_dalpha * con_o[3] * -0.5f * G;               /* :823 */
                    float dL_dNDC[2] = {
                        dL_ddist * 2 * (con_o[0] * d[0] + con_o[1] * d[1]) * ddelx_dx,
                        dL_ddist * 2 * (con_o[2] * d[1] + con_o[1] * d[0]) * ddely_dy };
                    float dL_dcon[3] = { dL_ddist * (d[0] * d[0]), dL_ddist * (1 * d[0] * d[1]), dL_ddist * (d[1] * d[1]) };
                    if (surface && per_pixel_depth) {                           /* :838-841 */
                        dL_dNDC[0] += 1 * -dL_dpixD * (Jv[6] * Jv[0] + Jv[9] * Jv[2]);
                        dL_dNDC[1] += 1 * -dL_dpixD * (Jv[6] * Jv[1] + Jv[9] * Jv[3]);
                    }
                    float dL_dopac = G * dL_dalpha;                             /* :854 */
                    const float vals[6] = { dL_dNDC[0], dL_dNDC[1], dL_dcon[0], dL_dcon[1], dL_dcon[2], dL_dopac };
                    for (int q = 0; q < 6; q++) {
#ifdef _OPENMP
#pragma omp atomic
#endif
                        a->v[q] += vals[q];
                    }
                }
            }
    }
    for (int i = 0; i < P; i++) {
        const PairAcc *a = &acc[i];
        dL_dmean2D[3 * i + 0] = (float)a->v[0];
        dL_dmean2D[3 * i + 1] = (float)a->v[1];
        dL_dmean2D[3 * i + 2] = 0.f;                                            /* z never written */
        dL_dconic[4 * i + 0] = (float)a->v[2];
        dL_dconic[4 * i + 1] = (float)a->v[3];
        dL_dconic[4 * i + 2] = 0.f;
        dL_dconic[4 * i + 3] = (float)a->v[4];                                  /* :849-851: slots x, y, w */
        dL_dopacity[i] = (float)a->v[5];
        for (int ch = 0; ch < 3; ch++) dL_dcolors[3 * i + ch] = (float)a->v[6 + ch];
        for (int ch = 0; ch < 3; ch++) dL_dnormal[3 * i + ch] = (float)a->v[9 + ch];
        dL_ddepth[i] = (float)a->v[12];
    }
    free(acc);
}

/* ------------------------------------------------------------------------------------------
 * backward: per-Gaussian kernels
 * ---------------------------------------------------------------------------------------- */

/* SH backward, DGR/cuda_rasterizer/backward.cu:20-158 */
static void computeColorFromSH_bwd(int idx, int deg, int max_coeffs, const float *means, const float *campos,
                                   const float *shs, const uint8_t *clamped, const float *dL_dcolor,
                                   float *dL_dmeans, float *dL_dshs, int lrn_cam, float *dL_dcampos)
{
    float dir_orig[3] = { means[3 * idx] - campos[0], means[3 * idx + 1] - campos[1], means[3 * idx + 2] - campos[2] };
    float len = sqrtf(dir_orig[0] * dir_orig[0] + dir_orig[1] * dir_orig[1] + dir_orig[2] * dir_orig[2]);
    float dir[3] = { dir_orig[0] / len, dir_orig[1] / len, dir_orig[2] / len };
    const float *sh = shs + (size_t)idx * max_coeffs * 3;
    float dL_dRGB[3] = { dL_dcolor[3 * idx], dL_dcolor[3 * idx + 1], dL_dcolor[3 * idx + 2] };
    for (int c = 0; c < 3; c++) dL_dRGB[c] *= clamped[3 * idx + c] ? 0 : 1;
    float x = dir[0], y = dir[1], z = dir[2];
    float *dL_dsh = dL_dshs + (size_t)idx * max_coeffs * 3;
    float dRGBdx[3] = { 0, 0, 0 }, dRGBdy[3] = { 0, 0, 0 }, dRGBdz[3] = { 0, 0, 0 };
#define SH(k, c) sh[(k) * 3 + (c)]
#define DSH(k, coef) do { float cf = (coef); for (int c = 0; c < 3; c++) dL_dsh[(k) * 3 + c] = cf * dL_dRGB[c]; } while (0)
    DSH(0, SH_C0);
    if (deg > 0) {
        DSH(1, -SH_C1 * y); DSH(2, SH_C1 * z); DSH(3, -SH_C1 * x);
        for (int c = 0; c < 3; c++) {
            dRGBdx[c] = -SH_C1 * SH(3, c);
            dRGBdy[c] = -SH_C1 * SH(1, c);
            dRGBdz[c] = SH_C1 * SH(2, c);
        }
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z;
            float xy = x * y, yz = y * z, xz = x * z;
            DSH(4, SH_C2[0] * xy); DSH(5, SH_C2[1] * yz); DSH(6, SH_C2[2] * (2.f * zz - xx - yy));
            DSH(7, SH_C2[3] * xz); DSH(8, SH_C2[4] * (xx - yy));
            for (int c = 0; c < 3; c++) {
                dRGBdx[c] += SH_C2[0] * y * SH(4, c) + SH_C2[2] * 2.f * -x * SH(6, c) + SH_C2[3] * z * SH(7, c) + SH_C2[4] * 2.f * x * SH(8, c);
                dRGBdy[c] += SH_C2[0] * x * SH(4, c) + SH_C2[1] * z * SH(5, c) + SH_C2[2] * 2.f * -y * SH(6, c) + SH_C2[4] * 2.f * -y * SH(8, c);
                dRGBdz[c] += SH_C2[1] * y * SH(5, c) + SH_C2[2] * 2.f * 2.f * z * SH(6, c) + SH_C2[3] * x * SH(7, c);
            }
            if (deg > 2) {
                DSH(9, SH_C3[0] * y * (3.f * xx - yy)); DSH(10, SH_C3[1] * xy * z);
                DSH(11, SH_C3[2] * y * (4.f * zz - xx - yy)); DSH(12, SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy));
                DSH(13, SH_C3[4] * x * (4.f * zz - xx - yy)); DSH(14, SH_C3[5] * z * (xx - yy));
                DSH(15, SH_C3[6] * x * (xx - 3.f * yy));
                for (int c = 0; c < 3; c++) {
                    dRGBdx[c] += (SH_C3[0] * SH(9, c) * 3.f * 2.f * xy + SH_C3[1] * SH(10, c) * yz +
                                  SH_C3[2] * SH(11, c) * -2.f * xy + SH_C3[3] * SH(12, c) * -3.f * 2.f * xz +
                                  SH_C3[4] * SH(13, c) * (-3.f * xx + 4.f * zz - yy) + SH_C3[5] * SH(14, c) * 2.f * xz +
                                  SH_C3[6] * SH(15, c) * 3.f * (xx - yy));
                    dRGBdy[c] += (SH_C3[0] * SH(9, c) * 3.f * (xx - yy) + SH_C3[1] * SH(10, c) * xz +
                                  SH_C3[2] * SH(11, c) * (-3.f * yy + 4.f * zz - xx) + SH_C3[3] * SH(12, c) * -3.f * 2.f * yz +
                                  SH_C3[4] * SH(13, c) * -2.f * xy + SH_C3[5] * SH(14, c) * -2.f * yz +
                                  SH_C3[6] * SH(15, c) * -3.f * 2.f * xy);
                    dRGBdz[c] += (SH_C3[1] * SH(10, c) * xy + SH_C3[2] * SH(11, c) * 4.f * 2.f * yz +
                                  SH_C3[3] * SH(12, c) * 3.f * (2.f * zz - xx - yy) + SH_C3[4] * SH(13, c) * 4.f * 2.f * xz +
                                  SH_C3[5] * SH(14, c) * (xx - yy));
                }
            }
        }
    }
#undef SH
#undef DSH
    float dL_ddir[3] = {
        dRGBdx[0] * dL_dRGB[0] + dRGBdx[1] * dL_dRGB[1] + dRGBdx[2] * dL_dRGB[2],
        dRGBdy[0] * dL_dRGB[0] + dRGBdy[1] * dL_dRGB[1] + dRGBdy[2] * dL_dRGB[2],
        dRGBdz[0] * dL_dRGB[0] + dRGBdz[1] * dL_dRGB[1] + dRGBdz[2] * dL_dRGB[2] };
    float dL_dmean[3];
    dnormvdv3(dir_orig, dL_ddir, dL_dmean);
    if (lrn_cam) {
        dL_dcampos[0] += -dL_dmean[0]; dL_dcampos[1] += -dL_dmean[1]; dL_dcampos[2] += -dL_dmean[2];
    }
    dL_dmeans[3 * idx + 0] += dL_dmean[0];
    dL_dmeans[3 * idx + 1] += dL_dmean[1];
    dL_dmeans[3 * idx + 2] += dL_dmean[2];
}

/* computeCov3D backward, DGR/cuda_rasterizer/backward.cu:326-432 */
static void computeCov3D_bwd(int idx, const float *scale, float mod, const float *rot, const float *dL_dcov3Ds,
                             float *dL_dscales, float *dL_drots, const float *dL_dnormal, const float *view,
                             int surface, int lrn_cam, float *dL_dviewmat)
{
    float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    mat3 R = mat3_cols(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                       2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                       2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
    mat3 S = mat3_cols(1, 0, 0, 0, 1, 0, 0, 0, 1);
    float s[3] = { mod * scale[0], mod * scale[1], mod * scale[2] };
    S.m[0][0] = s[0]; S.m[1][1] = s[1]; S.m[2][2] = s[2];
    mat3 M = mat3_mul(S, R);
    const float *d = dL_dcov3Ds + 6 * idx;
    mat3 dL_dSigma = mat3_cols(d[0], 0.5f * d[1], 0.5f * d[2], 0.5f * d[1], d[3], 0.5f * d[4], 0.5f * d[2], 0.5f * d[4], d[5]);
    mat3 M2 = M;
    for (int c = 0; c < 3; c++) for (int rr = 0; rr < 3; rr++) M2.m[c][rr] = 2.0f * M.m[c][rr];
    mat3 dL_dM = mat3_mul(M2, dL_dSigma);
    mat3 Rt = mat3_transpose(R);
    mat3 dL_dMt = mat3_transpose(dL_dM);
    float *dL_dscale = dL_dscales + 3 * idx;
#define DOT3(a, b) ((a)[0] * (b)[0] + (a)[1] * (b)[1] + (a)[2] * (b)[2])
    dL_dscale[0] = DOT3(Rt.m[0], dL_dMt.m[0]);
    dL_dscale[1] = DOT3(Rt.m[1], dL_dMt.m[1]);
    dL_dscale[2] = surface ? 0 : DOT3(Rt.m[2], dL_dMt.m[2]);
#undef DOT3
    mat3 dL_dRt = dL_dMt;
    for (int k = 0; k < 3; k++) { dL_dRt.m[0][k] *= s[0]; dL_dRt.m[1][k] *= s[1]; dL_dRt.m[2][k] *= s[2]; }
    const float *cN = dL_dnormal + 3 * idx;
    float wN[3] = { cN[0] * view[0] + cN[1] * view[1] + cN[2] * view[2],
                    cN[0] * view[4] + cN[1] * view[5] + cN[2] * view[6],
                    cN[0] * view[8] + cN[1] * view[9] + cN[2] * view[10] };
    dL_dRt.m[2][0] += wN[0]; dL_dRt.m[2][1] += wN[1]; dL_dRt.m[2][2] += wN[2];
    if (lrn_cam) {   /* :404-414 */
        float wrdN[3] = { R.m[0][2], R.m[1][2], R.m[2][2] };
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
                dL_dviewmat[4 * i + j] += cN[j] * wrdN[i];
    }
    float *q = dL_drots + 4 * idx;
#define D(a, b) dL_dRt.m[a][b]
    q[0] = 2 * z * (D(0, 1) - D(1, 0)) + 2 * y * (D(2, 0) - D(0, 2)) + 2 * x * (D(1, 2) - D(2, 1));
    q[1] = 2 * y * (D(1, 0) + D(0, 1)) + 2 * z * (D(2, 0) + D(0, 2)) + 2 * r * (D(1, 2) - D(2, 1)) - 4 * x * (D(2, 2) + D(1, 1));
    q[2] = 2 * x * (D(1, 0) + D(0, 1)) + 2 * r * (D(2, 0) - D(0, 2)) + 2 * z * (D(1, 2) + D(2, 1)) - 4 * y * (D(2, 2) + D(0, 0));
    q[3] = 2 * r * (D(0, 1) - D(1, 0)) + 2 * x * (D(2, 0) + D(0, 2)) + 2 * y * (D(1, 2) + D(2, 1)) - 4 * z * (D(1, 1) + D(0, 0));
#undef D
}

void oracle_preprocess_backward(const OracleParams *prm, const int32_t *radii,
                                const float *means3D, const float *scales, const float *rotations,
                                const float *shs, const uint8_t *clamped, const float *cov3Ds,
                                const float *dL_dmean2D, const float *dL_dconics,
                                float *dL_dcolor, const float *dL_dnormal, const float *dL_ddepth,
                                float *dL_dmeans3D, float *dL_dcov, float *dL_dsh,
                                float *dL_dscales, float *dL_drots,
                                float *dL_dviewmat, float *dL_dprojmat, float *dL_dcampos)
{
    const int P = prm->P;
    const float h_y = prm->H / (2.0f * prm->tanfovy);   /* rasterizer_impl.cu:339-340 */
    const float h_x = prm->W / (2.0f * prm->tanfovx);
    const float *view_matrix = prm->viewmatrix, *proj = prm->projmatrix;
    const int surface = prm->config[0] > 0, lrn_cam = prm->config[3] > 0;

    /* computeCov2DCUDA, backward.cu:163-322 */
    for (int idx = 0; idx < P; idx++) {
        if (!(radii[idx] > 0)) continue;
        const float *cov3D = cov3Ds + 6 * idx;
        const float mean[3] = { means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2] };
        const float dL_dconic[3] = { dL_dconics[4 * idx], dL_dconics[4 * idx + 1], dL_dconics[4 * idx + 3] };
        float t[3];
        transformPoint4x3(mean, view_matrix, t);
        const float limx = 1.3f * prm->tanfovx, limy = 1.3f * prm->tanfovy;
        const float txtz = t[0] / t[2], tytz = t[1] / t[2];
        t[0] = fminf(limx, fmaxf(-limx, txtz)) * t[2];
        t[1] = fminf(limy, fmaxf(-limy, tytz)) * t[2];
        const float x_grad_mul = txtz < -limx || txtz > limx ? 0 : 1;
        const float y_grad_mul = tytz < -limy || tytz > limy ? 0 : 1;
        float J0 = h_x / t[2], J1 = -(h_x * t[0]) / (t[2] * t[2]), J2 = h_y / t[2], J3 = -(h_y * t[1]) / (t[2] * t[2]);
        mat3 J = mat3_cols(J0, 0.0f, J1, 0.0f, J2, J3, 0, 0, 0);
        mat3 Wm = mat3_cols(view_matrix[0], view_matrix[4], view_matrix[8],
                            view_matrix[1], view_matrix[5], view_matrix[9],
                            view_matrix[2], view_matrix[6], view_matrix[10]);
        mat3 Vrk = mat3_cols(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
        mat3 T = mat3_mul(Wm, J);
        mat3 cov2D = mat3_mul(mat3_mul(mat3_transpose(T), mat3_transpose(Vrk)), T);
        float a = cov2D.m[0][0] += 0.3f;
        float b = cov2D.m[0][1];
        float c = cov2D.m[1][1] += 0.3f;
        float denom = a * c - b * b;
        float dL_da = 0, dL_db = 0, dL_dc = 0;
        float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        if (denom2inv != 0) {
            dL_da = denom2inv * (-c * c * dL_dconic[0] + 2 * b * c * dL_dconic[1] + (denom - a * c) * dL_dconic[2]);
            dL_dc = denom2inv * (-a * a * dL_dconic[2] + 2 * a * b * dL_dconic[1] + (denom - a * c) * dL_dconic[0]);
            dL_db = denom2inv * 2 * (b * c * dL_dconic[0] - (denom + 2 * b * b) * dL_dconic[1] + a * b * dL_dconic[2]);
#define T_(i, j) T.m[i][j]
            dL_dcov[6 * idx + 0] = (T_(0, 0) * T_(0, 0) * dL_da + T_(0, 0) * T_(1, 0) * dL_db + T_(1, 0) * T_(1, 0) * dL_dc);
            dL_dcov[6 * idx + 3] = (T_(0, 1) * T_(0, 1) * dL_da + T_(0, 1) * T_(1, 1) * dL_db + T_(1, 1) * T_(1, 1) * dL_dc);
            dL_dcov[6 * idx + 5] = (T_(0, 2) * T_(0, 2) * dL_da + T_(0, 2) * T_(1, 2) * dL_db + T_(1, 2) * T_(1, 2) * dL_dc);
            dL_dcov[6 * idx + 1] = 2 * T_(0, 0) * T_(0, 1) * dL_da + (T_(0, 0) * T_(1, 1) + T_(0, 1) * T_(1, 0)) * dL_db + 2 * T_(1, 0) * T_(1, 1) * dL_dc;
            dL_dcov[6 * idx + 2] = 2 * T_(0, 0) * T_(0, 2) * dL_da + (T_(0, 0) * T_(1, 2) + T_(0, 2) * T_(1, 0)) * dL_db + 2 * T_(1, 0) * T_(1, 2) * dL_dc;
            dL_dcov[6 * idx + 4] = 2 * T_(0, 2) * T_(0, 1) * dL_da + (T_(0, 1) * T_(1, 2) + T_(0, 2) * T_(1, 1)) * dL_db + 2 * T_(1, 1) * T_(1, 2) * dL_dc;
        } else {
            for (int i = 0; i < 6; i++) dL_dcov[6 * idx + i] = 0;
        }
#define V_(i, j) Vrk.m[i][j]
        float dL_dT00 = 2 * (T_(0, 0) * V_(0, 0) + T_(0, 1) * V_(0, 1) + T_(0, 2) * V_(0, 2)) * dL_da +
                        (T_(1, 0) * V_(0, 0) + T_(1, 1) * V_(0, 1) + T_(1, 2) * V_(0, 2)) * dL_db;
        float dL_dT01 = 2 * (T_(0, 0) * V_(1, 0) + T_(0, 1) * V_(1, 1) + T_(0, 2) * V_(1, 2)) * dL_da +
                        (T_(1, 0) * V_(1, 0) + T_(1, 1) * V_(1, 1) + T_(1, 2) * V_(1, 2)) * dL_db;
        float dL_dT02 = 2 * (T_(0, 0) * V_(2, 0) + T_(0, 1) * V_(2, 1) + T_(0, 2) * V_(2, 2)) * dL_da +
                        (T_(1, 0) * V_(2, 0) + T_(1, 1) * V_(2, 1) + T_(1, 2) * V_(2, 2)) * dL_db;
        float dL_dT10 = 2 * (T_(1, 0) * V_(0, 0) + T_(1, 1) * V_(0, 1) + T_(1, 2) * V_(0, 2)) * dL_dc +
                        (T_(0, 0) * V_(0, 0) + T_(0, 1) * V_(0, 1) + T_(0, 2) * V_(0, 2)) * dL_db;
        float dL_dT11 = 2 * (T_(1, 0) * V_(1, 0) + T_(1, 1) * V_(1, 1) + T_(1, 2) * V_(1, 2)) * dL_dc +
                        (T_(0, 0) * V_(1, 0) + T_(0, 1) * V_(1, 1) + T_(0, 2) * V_(1, 2)) * dL_db;
        float dL_dT12 = 2 * (T_(1, 0) * V_(2, 0) + T_(1, 1) * V_(2, 1) + T_(1, 2) * V_(2, 2)) * dL_dc +
                        (T_(0, 0) * V_(2, 0) + T_(0, 1) * V_(2, 1) + T_(0, 2) * V_(2, 2)) * dL_db;
#undef V_
#undef T_
        float dL_dJ00 = Wm.m[0][0] * dL_dT00 + Wm.m[0][1] * dL_dT01 + Wm.m[0][2] * dL_dT02;
        float dL_dJ02 = Wm.m[2][0] * dL_dT00 + Wm.m[2][1] * dL_dT01 + Wm.m[2][2] * dL_dT02;
        float dL_dJ11 = Wm.m[1][0] * dL_dT10 + Wm.m[1][1] * dL_dT11 + Wm.m[1][2] * dL_dT12;
        float dL_dJ12 = Wm.m[2][0] * dL_dT10 + Wm.m[2][1] * dL_dT11 + Wm.m[2][2] * dL_dT12;
        float tz = 1.f / t[2];
        float tz2 = tz * tz;
        float tz3 = tz2 * tz;
        if (lrn_cam) {   /* :286-305 */
            float dL_dW[16] = {
                dL_dT00 * J0, dL_dT10 * J2, dL_dT00 * J1 + dL_dT10 * J3, 0,
                dL_dT01 * J0, dL_dT11 * J2, dL_dT01 * J1 + dL_dT11 * J3, 0,
                dL_dT02 * J0, dL_dT12 * J2, dL_dT02 * J1 + dL_dT12 * J3, 0,
                0, 0, 0, 0 };
            for (int i = 0; i < 16; i++) dL_dviewmat[i] += dL_dW[i];
        }
        float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
        float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
        float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t[0]) * tz3 * dL_dJ02 + (2 * h_y * t[1]) * tz3 * dL_dJ12;
        float dL_dt[3] = { dL_dtx, dL_dty, dL_dtz }, dL_dmean[3];
        transformVec4x3Transpose(dL_dt, view_matrix, dL_dmean);
        dL_dmeans3D[3 * idx + 0] = dL_dmean[0];   /* assignment, :321 */
        dL_dmeans3D[3 * idx + 1] = dL_dmean[1];
        dL_dmeans3D[3 * idx + 2] = dL_dmean[2];
    }

    /* preprocessCUDA (backward), backward.cu:437-526 */
    for (int idx = 0; idx < P; idx++) {
        if (!(radii[idx] > 0)) continue;
        const float m[3] = { means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2] };
        float m_hom[4];
        transformPoint4x4(m, proj, m_hom);
        float m_w = 1.0f / (m_hom[3] + 0.0000001f);
        float mul1 = (proj[0] * m[0] + proj[4] * m[1] + proj[8] * m[2] + proj[12]) * m_w * m_w;
        float mul2 = (proj[1] * m[0] + proj[5] * m[1] + proj[9] * m[2] + proj[13]) * m_w * m_w;
        float d2[2] = { dL_dmean2D[3 * idx], dL_dmean2D[3 * idx + 1] };
        float dL_dmean[3];
        dL_dmean[0] = (proj[0] * m_w - proj[3] * mul1) * d2[0] + (proj[1] * m_w - proj[3] * mul2) * d2[1];
        dL_dmean[1] = (proj[4] * m_w - proj[7] * mul1) * d2[0] + (proj[5] * m_w - proj[7] * mul2) * d2[1];
        dL_dmean[2] = (proj[8] * m_w - proj[11] * mul1) * d2[0] + (proj[9] * m_w - proj[11] * mul2) * d2[1];
        float dL_dd = dL_ddepth[idx];
        float fromD[3] = { dL_dd * view_matrix[2], dL_dd * view_matrix[6], dL_dd * view_matrix[10] };
        for (int k = 0; k < 3; k++) dL_dmeans3D[3 * idx + k] += dL_dmean[k] + fromD[k];
        if (lrn_cam) {   /* :499-517 */
            float pp[16] = {
                d2[0] * m[0] * m_w, d2[1] * m[0] * m_w, 0, d2[0] * -mul1 * m[0] + d2[1] * -mul2 * m[0],
                d2[0] * m[1] * m_w, d2[1] * m[1] * m_w, 0, d2[0] * -mul1 * m[1] + d2[1] * -mul2 * m[1],
                d2[0] * m[2] * m_w, d2[1] * m[2] * m_w, 0, d2[0] * -mul1 * m[2] + d2[1] * -mul2 * m[2],
                d2[0] * m_w, d2[1] * m_w, 0, d2[0] * -mul1 + d2[1] * -mul2 };
            for (int i = 0; i < 16; i++) dL_dprojmat[i] += pp[i];
            float vd[16] = { 0, 0, dL_dd * m[0], 0, 0, 0, dL_dd * m[1], 0, 0, 0, dL_dd * m[2], 0, 0, 0, dL_dd, 0 };
            for (int i = 0; i < 16; i++) dL_dviewmat[i] += vd[i];
        }
        if (shs)
            computeColorFromSH_bwd(idx, prm->sh_degree, prm->M, means3D, prm->campos, shs, clamped, dL_dcolor,
                                   dL_dmeans3D, dL_dsh, lrn_cam, dL_dcampos);
        if (scales)
            computeCov3D_bwd(idx, scales + 3 * idx, prm->scale_modifier, rotations + 4 * idx, dL_dcov, dL_dscales,
                             dL_drots, dL_dnormal, view_matrix, surface, lrn_cam, dL_dviewmat);
    }
}
