// preprocess_point.h -- the per-Gaussian forward stage of the rasterizer as a device function: what preprocess_kernel
// (rast_preprocess.hip) runs per thread, shared with the fused kernel of lbs.hip that feeds it the posed position and quaternion
// straight from the warp's registers (warp_preprocess_frames_kernel, round 6).
//
// Replaces preprocessCUDA (DGR/cuda_rasterizer/forward.cu:205-385) together with its helpers (forward.cu:20-202,
// auxiliary.h:42-388).
//
// Floating point: every function here evaluates its expressions as written -- no FMA contraction (`#pragma clang fp contract(off)` at
// the top of each body; rast_preprocess.hip is compiled with -ffp-contract=off as a whole, lbs.hip is not): every value that feeds an
// integer decision (culls, radius, tile rectangle, depth key) follows an IEEE evaluation of the reference source, so radii /
// tiles_touched / sort keys are bit-identical to it in whichever kernel the function is inlined.
#pragma once
#include "soar_common.h"
#include "soar_m3.h"

namespace soar {

namespace {

__device__ __forceinline__ float pix_from_ndc(float v, int S, float prcp)
{
#pragma clang fp contract(off)
    // double arithmetic on purpose (auxiliary.h:42-46)
    return (float)(((v + 1.0) * S - 1.0) * 0.5 + S * (prcp - 0.5));
}

__device__ __forceinline__ float unit3(float *v)
{
#pragma clang fp contract(off)
    float mod = fmaxf(sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), (float)0.00000001);
    v[0] /= mod;
    v[1] /= mod;
    v[2] /= mod;
    return mod;
}

__constant__ float kSH_C0 = 0.28209479177387814f;
__constant__ float kSH_C1 = 0.4886025119029199f;
__constant__ float kSH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                                0.5462742152960396f};
__constant__ float kSH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                                -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

// SH -> RGB (forward.cu:20-71); returns clamp flags in bits 0..2
__device__ inline unsigned sh_to_rgb(int idx, int deg, int M, const float *means, const float *campos, const float *shs,
                                     float *rgb)
{
#pragma clang fp contract(off)
    float dx = means[3 * idx] - campos[0], dy = means[3 * idx + 1] - campos[1], dz = means[3 * idx + 2] - campos[2];
    float len = sqrtf(dx * dx + dy * dy + dz * dz);
    float x = dx / len, y = dy / len, z = dz / len;
    const float *sh = shs + (size_t)idx * M * 3;
    unsigned flags = 0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float res = kSH_C0 * sh[c];
        if (deg > 0) {
            res = res - kSH_C1 * y * sh[3 + c] + kSH_C1 * z * sh[6 + c] - kSH_C1 * x * sh[9 + c];
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                res = res + kSH_C2[0] * xy * sh[12 + c] + kSH_C2[1] * yz * sh[15 + c] +
                      kSH_C2[2] * (2.0f * zz - xx - yy) * sh[18 + c] + kSH_C2[3] * xz * sh[21 + c] +
                      kSH_C2[4] * (xx - yy) * sh[24 + c];
                if (deg > 2) {
                    res = res + kSH_C3[0] * y * (3.0f * xx - yy) * sh[27 + c] + kSH_C3[1] * xy * z * sh[30 + c] +
                          kSH_C3[2] * y * (4.0f * zz - xx - yy) * sh[33 + c] +
                          kSH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[36 + c] +
                          kSH_C3[4] * x * (4.0f * zz - xx - yy) * sh[39 + c] + kSH_C3[5] * z * (xx - yy) * sh[42 + c] +
                          kSH_C3[6] * x * (xx - 3.0f * yy) * sh[45 + c];
                }
            }
        }
        res += 0.5f;
        if (res < 0) flags |= 1u << c;
        rgb[c] = fmaxf(res, 0.0f);
    }
    return flags;
}

// tile rectangle of a splat (auxiliary.h:53-63); float->int conversions saturate on the GPU
__device__ __forceinline__ void tile_rect(float px, float py, int rad, int gx, int gy, int &x0, int &y0, int &x1, int &y1)
{
#pragma clang fp contract(off)
    x0 = min(gx, max(0, (int)((px - rad) / TILE)));
    y0 = min(gy, max(0, (int)((py - rad) / TILE)));
    x1 = min(gx, max(0, (int)((px + rad + TILE - 1) / TILE)));
    y1 = min(gy, max(0, (int)((py + rad + TILE - 1) / TILE)));
}

struct PreArgs {
    int P, D, M, W, H, gx, gy;
    int prefiltered, render_front, surface, pix_depth;
    float tanfovx, tanfovy, focal_x, focal_y, scale_modifier;
    const float *means3D, *shs, *colors, *opacities, *scales, *rotations, *cov3D_precomp;
    const float *view, *proj, *prcp, *bbox, *campos;
    GaussRec *rec;
    float *cov3D;
    uint32_t *tiles_touched;
    uint8_t *clamped;
    float *front_out;
    uint2 *rect_out;
    uint32_t *depth_key_out;
    uint32_t *blk_stats;
    int32_t *radii;
    uint32_t *header;        // GeomBuf::header (H_PREFILTER_VIOLATIONS)
};

// what one Gaussian's forward stage leaves (the caller stores it)
struct PrePoint {
    GaussRec rec;
    uint2 rect;
    uint32_t depth_key, tiles, key_bits;
    int radius;
    bool alive, faces_camera, prefilter_violation;
};

// (px3, py3, pz3), q: the position and -- has_rot -- the quaternion (r, x, y, z) of Gaussian idx; everything else is read through `a`
__device__ __forceinline__ void preprocess_point(const PreArgs &a, const int idx, const bool in_range, const float px3, const float py3,
                                                 const float pz3, const bool has_rot, const float4 q, PrePoint &o)
{
#pragma clang fp contract(off)
    // culled unless proven otherwise (forward.cu:249-250)
    int out_radius = 0;
    uint32_t out_tiles = 0;
    uint2 out_rect = make_uint2(0u, 0u);        // empty rectangle: not visible
    GaussRec rec;
    rec.q0 = make_float4(0.f, 0.f, 0.f, 0.f);
    rec.q1 = rec.q0;
    rec.q2 = rec.q0;
    rec.q3 = rec.q0;

    const float *V = a.view, *PM = a.proj;

    // clip-space / view-space position (auxiliary.h:65-84, forward.cu:254-264)
    float hx = PM[0] * px3 + PM[4] * py3 + PM[8] * pz3 + PM[12];
    float hy = PM[1] * px3 + PM[5] * py3 + PM[9] * pz3 + PM[13];
    float hw = PM[3] * px3 + PM[7] * py3 + PM[11] * pz3 + PM[15];
    float p_w = 1.0f / (hw + 0.0000001f);
    float ndc_x = hx * p_w, ndc_y = hy * p_w;
    float vx = V[0] * px3 + V[4] * py3 + V[8] * pz3 + V[12];
    float vy = V[1] * px3 + V[5] * py3 + V[9] * pz3 + V[13];
    float vz = V[2] * px3 + V[6] * py3 + V[10] * pz3 + V[14];

    const float pix_x = pix_from_ndc(ndc_x, a.W, a.prcp[0]);
    const float pix_y = pix_from_ndc(ndc_y, a.H, a.prcp[1]);

    bool alive = true;
    {   // patch-bbox frustum test, 20 % margin, view-space z (auxiliary.h:146-171)
        float x0 = a.bbox[1], y0 = a.bbox[0], x1 = a.bbox[3], y1 = a.bbox[2];
        float w = x1 - x0, h = y1 - y0;
        float expand = (float)0.2;
        if (vz < 0 || pix_x < x0 - w * expand || pix_x >= x1 + w * expand || pix_y < y0 - h * expand ||
            pix_y >= y1 + h * expand)
            alive = false;
    }
    // The reference prints and traps when a point is culled although the caller promised a prefiltered set (auxiliary.h:163-167,
    // 195-199).  A trap takes the whole context down; here the violations are counted (one atomic per wavefront that has any) and
    // reported as an error by the host: soar_rast_forward_geometry in debug mode, soar_rast_prefilter_violations on request
    bool prefilter_violation = a.prefiltered && in_range && !alive;

    M3 R;
    if (alive) {
        // quaternion (r,x,y,z) used as given, no normalisation (forward.cu:141-156)
        float r = 1.f, x = 0.f, y = 0.f, z = 0.f;
        if (has_rot) { r = q.x; x = q.y; y = q.z; z = q.w; }
        R.e[0][0] = 1.f - 2.f * (y * y + z * z); R.e[0][1] = 2.f * (x * y - r * z); R.e[0][2] = 2.f * (x * z + r * y);
        R.e[1][0] = 2.f * (x * y + r * z); R.e[1][1] = 1.f - 2.f * (x * x + z * z); R.e[1][2] = 2.f * (y * z - r * x);
        R.e[2][0] = 2.f * (x * z - r * y); R.e[2][1] = 2.f * (y * z + r * x); R.e[2][2] = 1.f - 2.f * (x * x + y * y);
    }

    float nview[3] = {0.f, 0.f, 0.f};
    float plane_a = 0.f, plane_b = 0.f;
    bool faces_camera = true;          // what a render_front pass keeps (all splats when not in surface mode)
    if (alive && a.surface) {
        // surfel normal and tangent axes in view space (forward.cu:283-285)
        float ax0[3], ax1[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            nview[k] = V[k] * R.e[0][2] + V[4 + k] * R.e[1][2] + V[8 + k] * R.e[2][2];
            ax0[k] = V[k] * R.e[0][0] + V[4 + k] * R.e[1][0] + V[8 + k] * R.e[2][0];
            ax1[k] = V[k] * R.e[0][1] + V[4 + k] * R.e[1][1] + V[8 + k] * R.e[2][1];
        }
        // back-face test (auxiliary.h:173-208): the literal -0.01 is a double
        float dot = vx * nview[0] + vy * nview[1] + vz * nview[2];
        bool front = !((double)dot > -0.01);
        faces_camera = front;
        if (a.render_front && !front) {
            alive = false;
            prefilter_violation = prefilter_violation || (a.prefiltered && in_range);
        }

        if (alive && a.pix_depth) {
            // local homography between the image plane and the surfel plane (auxiliary.h:291-388)
            float prj_x = vx / vz, prj_y = vy / vz;
            float S_fix = 1000, Svp = (a.focal_x + a.focal_y) / 2;
            float d0[3] = {prj_x + 1 / S_fix, prj_y, 1.f};
            float d0_mod = unit3(d0);
            float d1[3] = {prj_x, prj_y + 1 / S_fix, 1.f};
            float d1_mod = unit3(d1);
            float thr = (float)0.01;
            float c0 = d0[0] * nview[0] + d0[1] * nview[1] + d0[2] * nview[2];
            float c1 = d1[0] * nview[0] + d1[1] * nview[1] + d1[2] * nview[2];
            if ((fabsf(c0 / d0_mod) < thr) || (fabsf(c1 / d1_mod) < thr)) {
                alive = false;   // grazing view of the surfel
            } else {
                float tt = vx * nview[0] + vy * nview[1] + vz * nview[2];
                float t0 = tt / c0, t1 = tt / c1;
                float pv[3] = {vx, vy, vz};
                float xu0[3], xu1[3];
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    xu0[k] = d0[k] * t0 - pv[k];
                    xu1[k] = d1[k] * t1 - pv[k];
                }
                float J[4];
                J[0] = xu0[0] * ax0[0] + xu0[1] * ax0[1] + xu0[2] * ax0[2];
                J[1] = xu1[0] * ax0[0] + xu1[1] * ax0[1] + xu1[2] * ax0[2];
                J[2] = xu0[0] * ax1[0] + xu0[1] * ax1[1] + xu0[2] * ax1[2];
                J[3] = xu1[0] * ax1[0] + xu1[1] * ax1[1] + xu1[2] * ax1[2];
                float sc = Svp / S_fix;
                J[0] /= sc; J[1] /= sc; J[2] /= sc; J[3] /= sc;
                // only the z row of the tangent basis is ever consumed: fold Jinv[10] into two numbers
                plane_a = ax0[2] * J[0] + ax1[2] * J[2];   // J6*J0 + J9*J2
                plane_b = ax0[2] * J[1] + ax1[2] * J[3];   // J6*J1 + J9*J3
            }
        }
    }

    float cv[6] = {0, 0, 0, 0, 0, 0};
    float conic[3] = {0, 0, 0};
    if (alive) {
        // 3D covariance (forward.cu:162-202); scale.z is replaced by 0 in surface mode (precedence quirk at :168)
        if (a.cov3D_precomp) {
#pragma unroll
            for (int k = 0; k < 6; k++) cv[k] = a.cov3D_precomp[6 * idx + k];
        } else {
            const float mod = a.scale_modifier;
            M3 S;
#pragma unroll
            for (int c = 0; c < 3; c++)
#pragma unroll
                for (int k = 0; k < 3; k++) S.e[c][k] = (c == k) ? 1.f : 0.f;
            S.e[0][0] = mod * a.scales[3 * idx + 0];
            S.e[1][1] = mod * a.scales[3 * idx + 1];
            S.e[2][2] = ((mod * (a.surface ? 1.0f : 0.0f)) != 0.0f) ? 0.f : a.scales[3 * idx + 2];
            M3 Mm = m3mul(S, R);
            M3 Sg = m3mul(m3t(Mm), Mm);
            cv[0] = Sg.e[0][0]; cv[1] = Sg.e[0][1]; cv[2] = Sg.e[0][2];
            cv[3] = Sg.e[1][1]; cv[4] = Sg.e[1][2]; cv[5] = Sg.e[2][2];
#pragma unroll
            for (int k = 0; k < 6; k++) a.cov3D[6 * idx + k] = cv[k];
        }

        // EWA projection of the covariance at the VIEW-space point (forward.cu:74-139, :329)
        float t0 = vx, t1 = vy, t2 = vz;
        const float limx = 1.3f * a.tanfovx, limy = 1.3f * a.tanfovy;
        const float txtz = t0 / t2, tytz = t1 / t2;
        t0 = fminf(limx, fmaxf(-limx, txtz)) * t2;
        t1 = fminf(limy, fmaxf(-limy, tytz)) * t2;
        M3 Jm, Wm, Vrk;
        Jm.e[0][0] = a.focal_x / t2; Jm.e[0][1] = 0.0f; Jm.e[0][2] = -(a.focal_x * t0) / (t2 * t2);
        Jm.e[1][0] = 0.0f; Jm.e[1][1] = a.focal_y / t2; Jm.e[1][2] = -(a.focal_y * t1) / (t2 * t2);
        Jm.e[2][0] = 0.f; Jm.e[2][1] = 0.f; Jm.e[2][2] = 0.f;
        Wm.e[0][0] = V[0]; Wm.e[0][1] = V[4]; Wm.e[0][2] = V[8];
        Wm.e[1][0] = V[1]; Wm.e[1][1] = V[5]; Wm.e[1][2] = V[9];
        Wm.e[2][0] = V[2]; Wm.e[2][1] = V[6]; Wm.e[2][2] = V[10];
        Vrk.e[0][0] = cv[0]; Vrk.e[0][1] = cv[1]; Vrk.e[0][2] = cv[2];
        Vrk.e[1][0] = cv[1]; Vrk.e[1][1] = cv[3]; Vrk.e[1][2] = cv[4];
        Vrk.e[2][0] = cv[2]; Vrk.e[2][1] = cv[4]; Vrk.e[2][2] = cv[5];
        M3 T = m3mul(Wm, Jm);
        M3 c2 = m3mul(m3mul(m3t(T), m3t(Vrk)), T);
        float cxx = c2.e[0][0] + 0.3f, cxy = c2.e[0][1], cyy = c2.e[1][1] + 0.3f;   // low-pass (:119-120)

        float det = (cxx * cyy - cxy * cxy);
        if (det == 0.0f) {
            alive = false;
        } else {
            float det_inv = 1.f / det;
            conic[0] = cyy * det_inv;
            conic[1] = -cxy * det_inv;
            conic[2] = cxx * det_inv;
            float mid = 0.5f * (cxx + cyy);
            float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
            float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
            float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
            int rad = (int)my_radius;
            int x0, y0, x1, y1;
            tile_rect(pix_x, pix_y, rad, a.gx, a.gy, x0, y0, x1, y1);
            uint32_t ntiles = (uint32_t)(y1 - y0) * (uint32_t)(x1 - x0);
            if (ntiles == 0) {
                alive = false;
            } else {
                out_radius = rad;
                out_tiles = ntiles;
                out_rect = make_uint2((uint32_t)x0 | ((uint32_t)x1 << 16), (uint32_t)y0 | ((uint32_t)y1 << 16));
            }
        }
    }

    if (alive) {
        float rgb[3];
        if (a.colors) {
            rgb[0] = a.colors[3 * idx]; rgb[1] = a.colors[3 * idx + 1]; rgb[2] = a.colors[3 * idx + 2];
        } else {
            unsigned f = sh_to_rgb(idx, a.D, a.M, a.means3D, a.campos, a.shs, rgb);
            a.clamped[3 * idx + 0] = f & 1u;
            a.clamped[3 * idx + 1] = (f >> 1) & 1u;
            a.clamped[3 * idx + 2] = (f >> 2) & 1u;
        }
        rec.q0 = make_float4(pix_x, pix_y, conic[0], conic[1]);
        rec.q1 = make_float4(conic[2], a.opacities[idx], vz, plane_a);
        rec.q2 = make_float4(plane_b, rgb[0], rgb[1], rgb[2]);
        rec.q3 = make_float4(nview[0], nview[1], nview[2], splat_cull_threshold(a.opacities[idx]));
    }
    o.rec = rec; o.rect = out_rect; o.depth_key = alive ? __float_as_uint(vz) : 0xFFFFFFFFu; o.key_bits = __float_as_uint(vz);
    o.tiles = out_tiles; o.radius = out_radius; o.alive = alive; o.faces_camera = faces_camera; o.prefilter_violation = prefilter_violation;
}

// stores of one Gaussian's results and the statistics row of its wavefront (64 consecutive Gaussians: row idx / 64 of
// GeomBuf::blk_stats -- the depth sort and the tile binning fold the rows; no atomics)
// (stats_row < 0: a wavefront wholly past the end of the model -- its lanes repeat the last Gaussian's stores -- leaves no row)
__device__ __forceinline__ void preprocess_store(const PreArgs &a, const int idx, const bool in_range, const int stats_row, const PrePoint &o)
{
    a.rec[idx] = o.rec;
    a.rect_out[idx] = o.rect;
    a.depth_key_out[idx] = o.depth_key;
    {
        const bool vis = o.alive && in_range;
        const uint32_t key = o.key_bits;
        uint32_t v[BLK_STATS] = {vis ? key : 0u, vis ? ~key : 0u, vis ? (o.rect.x >> 16) : 0u, vis ? (o.rect.y >> 16) : 0u,
                                 vis ? ~(o.rect.x & 0xFFFFu) : 0u, vis ? ~(o.rect.y & 0xFFFFu) : 0u};
#pragma unroll
        for (int k = 0; k < BLK_STATS; k++) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v[k] = max(v[k], (uint32_t)__shfl_xor((int)v[k], off));
        }
        if (stats_row >= 0 && (threadIdx.x & 63) < BLK_STATS) {
            const int k = threadIdx.x & 63;
            a.blk_stats[stats_row * BLK_STATS + k] = k == 0 ? v[0] : k == 1 ? v[1] : k == 2 ? v[2] : k == 3 ? v[3] : k == 4 ? v[4] : v[5];
        }
    }
    a.front_out[idx] = o.faces_camera ? 1.f : 0.f;
    a.radii[idx] = o.radius;
    a.tiles_touched[idx] = o.tiles;
    if (a.prefiltered) {                                         // (uniform: SOAR never sets it)
        const unsigned long long bad = __ballot(o.prefilter_violation);
        if (bad != 0ull && (threadIdx.x & 63) == 0) atomicAdd(a.header + H_PREFILTER_VIOLATIONS, (uint32_t)__builtin_popcountll(bad));
    }
}

inline void fill_pre_args(PreArgs &a, const SoarRastParams &prm, const float *means3D, const float *shs, const float *colors_precomp,
                          const float *opacities, const float *scales, const float *rotations, const float *cov3D_precomp, GeomBuf &g,
                          int32_t *radii)
{
    a.P = prm.P; a.D = prm.sh_degree; a.M = prm.M; a.W = prm.W; a.H = prm.H;
    a.gx = (prm.W + TILE - 1) / TILE; a.gy = (prm.H + TILE - 1) / TILE;
    a.prefiltered = prm.prefiltered; a.render_front = prm.render_front;
    a.surface = prm.cfg_surface; a.pix_depth = prm.cfg_perpix_depth;
    a.tanfovx = prm.tanfovx; a.tanfovy = prm.tanfovy;
    a.focal_y = prm.H / (2.0f * prm.tanfovy);   // rasterizer_impl.cu:201-202
    a.focal_x = prm.W / (2.0f * prm.tanfovx);
    a.scale_modifier = prm.scale_modifier;
    a.means3D = means3D; a.shs = shs; a.colors = colors_precomp; a.opacities = opacities;
    a.scales = scales; a.rotations = rotations; a.cov3D_precomp = cov3D_precomp;
    a.view = prm.viewmatrix_dev; a.proj = prm.projmatrix_dev; a.prcp = prm.prcppoint_dev;
    a.bbox = prm.patchbbox_dev; a.campos = prm.campos_dev;
    a.rec = g.rec; a.cov3D = g.cov3D; a.tiles_touched = g.tiles_touched; a.clamped = g.clamped; a.front_out = g.front; a.rect_out = g.rect;
    a.depth_key_out = g.depth_key; a.blk_stats = g.blk_stats; a.radii = radii; a.header = g.header;
}

}  // namespace

}  // namespace soar
