// rast_render_bwd.hip -- back-to-front gradient blend for gfx950.
//
// Replaces renderCUDA<3> backward (DGR/cuda_rasterizer/backward.cu:529-858).
//
// The reference issues 13 global atomicAdd per contributing (pixel, Gaussian) pair, all 256 pixels of a
// tile hitting the same addresses.  Here:
//  * each wavefront owns an 8x8 pixel quad (same mapping as the forward kernel) and walks the tile list
//    backwards starting at the deepest contributor of ITS 64 pixels (wave max of n_contrib), not at the end
//    of the tile list;
//  * the 13 per-pair terms are summed across the 64 lanes in registers with DPP row shifts / row broadcasts
//    (6 v_add_f32_dpp per term), the totals are moved to 13 different lanes and leave as ONE
//    global_atomic_add_f32 wave-instruction into a 64-byte accumulation row acc[gaussian][16]
//    (one memory-side atomic request per (quad, Gaussian) instead of 13 * 64);
//  * records are staged exactly as in the forward kernel (64-entry chunks, wave-private LDS slab, no barrier).
// The per-Gaussian rows are consumed by geometry_backward_kernel (rast_geom_bwd.hip).
#include "soar_common.h"

namespace soar {

namespace {

struct BwdArgs {
    int W, H, gx, gy, ntiles;
    int normalize_depth;
    const uint2 *ranges;
    const uint32_t *point_list;
    const GaussRec *rec;
    const float *bg;
    const float *final_T;
    const float *final_D;
    const uint32_t *n_contrib;
    const float *dL_dcolor, *dL_dnormal, *dL_ddepth, *dL_dopac;
    float *acc;
};

__device__ __forceinline__ int xcd_tile(int bid, int n)
{
    const int q = n >> 3, r = n & 7;
    const int xcd = bid & 7, within = bid >> 3;
    return xcd * q + min(xcd, r) + within;
}

// DPP controls (GFX9 encoding)
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118;
constexpr int DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v)
{
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true);
    return v + __int_as_float(moved);
}

// sum over the 64 lanes; the total is valid in lane 63
__device__ __forceinline__ float wave_sum_to_lane63(float v)
{
    v = dpp_add<DPP_ROW_SHR1, 0xf>(v);
    v = dpp_add<DPP_ROW_SHR2, 0xf>(v);
    v = dpp_add<DPP_ROW_SHR4, 0xf>(v);
    v = dpp_add<DPP_ROW_SHR8, 0xf>(v);
    v = dpp_add<DPP_ROW_BCAST15, 0xa>(v);
    v = dpp_add<DPP_ROW_BCAST31, 0xc>(v);
    return v;
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, off));
    return v;
}

__global__ void __launch_bounds__(256) render_backward_kernel(BwdArgs a)
{
    __shared__ GaussRec slab[4][WAVE];
    __shared__ uint32_t slab_id[4][WAVE];

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tile = xcd_tile(blockIdx.x, a.ntiles);
    const int tx = tile % a.gx, ty = tile / a.gx;
    const int px = tx * TILE + (wave & 1) * 8 + (lane & 7);
    const int py = ty * TILE + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < a.W && py < a.H;
    const float fx = (float)px, fy = (float)py;
    const size_t pix = (size_t)a.W * py + px;
    const size_t hw = (size_t)a.H * a.W;

    const uint2 range = a.ranges[tile];
    const uint32_t last = inside ? a.n_contrib[pix] : 0u;           // backward.cu:604
    const uint32_t deepest = wave_max_u32(last);                     // wave-uniform
    if (deepest == 0u) return;

    // per-pixel constants (backward.cu:595-623)
    const float T_final = inside ? a.final_T[pix] : 0.f;
    const float D_final = (inside && a.normalize_depth) ? a.final_D[pix] : 0.f;
    float dC0 = 0.f, dC1 = 0.f, dC2 = 0.f, dN0 = 0.f, dN1 = 0.f, dN2 = 0.f, dD = 0.f, dO = 0.f;
    if (inside) {
        dC0 = a.dL_dcolor[pix]; dC1 = a.dL_dcolor[hw + pix]; dC2 = a.dL_dcolor[2 * hw + pix];
        dN0 = a.dL_dnormal[pix]; dN1 = a.dL_dnormal[hw + pix]; dN2 = a.dL_dnormal[2 * hw + pix];
        dD = a.dL_ddepth[pix];
        dO = a.dL_dopac[pix];
    }
    const float bg_dot = a.bg[0] * dC0 + a.bg[1] * dC1 + a.bg[2] * dC2;      // :798-800
    const float ddelx_dx = 0.5f * a.W, ddely_dy = 0.5f * a.H;                 // :622-623
    const float inv_1mTf = 1.f / (1.f - T_final);
    const float dD_ch = a.normalize_depth ? dD * inv_1mTf : dD;               // :772
    // dL_dalpha terms that only depend on the pixel and on 1/(1-alpha):  (:791, :801, :802)
    const float tail = dO * T_final - T_final * bg_dot - (a.normalize_depth ? 0.f : T_final * (10.f * dD));
    const float norm_depth_k = a.normalize_depth ? dD * D_final * inv_1mTf * inv_1mTf * -T_final : 0.f;   // :773

    float T = T_final;
    float last_alpha = 0.f;
    float lc0 = 0.f, lc1 = 0.f, lc2 = 0.f, ln0 = 0.f, ln1 = 0.f, ln2 = 0.f, ld = 0.f;     // last_* (:617-618)
    float ac0 = 0.f, ac1 = 0.f, ac2 = 0.f, an0 = 0.f, an1 = 0.f, an2 = 0.f, ad = 0.f;     // accum_rec* (:607)

    const float quad_x0 = (float)(tx * TILE + (wave & 1) * 8), quad_y0 = (float)(ty * TILE + (wave >> 1) * 8);
    GaussRec *my = slab[wave];
    uint32_t *my_id = slab_id[wave];
    const float4 *myq = reinterpret_cast<const float4 *>(my);

    // positions [0, deepest) of the tile list, walked from the back in 64-entry chunks
    for (int cbase = (int)((deepest - 1u) & ~63u); cbase >= 0; cbase -= WAVE) {
        const int n = min(WAVE, (int)deepest - cbase);
        bool relevant = false;
        if (lane < n) {
            const uint32_t id = a.point_list[range.x + cbase + lane];
            const float4 *src = reinterpret_cast<const float4 *>(a.rec + id);
            float4 r0 = src[0], r1 = src[1], r2 = src[2], r3 = src[3];
            float4 *dst = reinterpret_cast<float4 *>(my + lane);
            dst[0] = r0; dst[1] = r1; dst[2] = r2; dst[3] = r3;
            my_id[lane] = id;
            relevant = splat_may_touch_quad(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, quad_x0, quad_y0);
        }
        unsigned long long todo = __ballot(relevant);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        while (todo != 0ull) {
            const int j = 63 - __builtin_clzll(todo);       // back to front
            todo &= ~(1ull << j);
            const uint32_t pos = (uint32_t)(cbase + j);
            const float4 q0 = myq[4 * j + 0];
            const float4 q1 = myq[4 * j + 1];
            const float dx = q0.x - fx, dy = q0.y - fy;
            const float power = falloff_power(q0.z, q0.w, q1.x, dx, dy);
            const float G = exp_nonpositive(power);
            const float alpha = fminf(0.99f, q1.y * G);
            const bool live = (pos < last) && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);   // :653-680
            if (__ballot(live) == 0ull) continue;
            const float4 q2 = myq[4 * j + 2];
            const float4 q3 = myq[4 * j + 3];

            float v[13];
#pragma unroll
            for (int k = 0; k < 13; k++) v[k] = 0.f;
            if (live) {
                const float one_m_alpha = 1.f - alpha;
                T = T / one_m_alpha;                                            // :683
                const float wgt = alpha * T;                                    // dchannel_dcolor
                const float keep = 1.f - last_alpha;
                // colour (:698-713)
                ac0 = last_alpha * lc0 + keep * ac0; lc0 = q2.y;
                ac1 = last_alpha * lc1 + keep * ac1; lc1 = q2.z;
                ac2 = last_alpha * lc2 + keep * ac2; lc2 = q2.w;
                float dL_dalpha = (q2.y - ac0) * dC0 + (q2.z - ac1) * dC1 + (q2.w - ac2) * dC2;
                v[6] = wgt * dC0; v[7] = wgt * dC1; v[8] = wgt * dC2;
                // normal, gain 10 on the per-Gaussian gradient (:715-731)
                an0 = last_alpha * ln0 + keep * an0; ln0 = q3.x;
                an1 = last_alpha * ln1 + keep * an1; ln1 = q3.y;
                an2 = last_alpha * ln2 + keep * an2; ln2 = q3.z;
                dL_dalpha += (q3.x - an0) * dN0 + (q3.y - an1) * dN1 + (q3.z - an2) * dN2;
                v[9] = wgt * dN0 * 10.f; v[10] = wgt * dN1 * 10.f; v[11] = wgt * dN2 * 10.f;
                // depth (:758-784)
                const float d_cur = q1.z - (dx * q1.w + dy * q2.x);
                ad = last_alpha * ld + keep * ad; ld = d_cur;
                dL_dalpha += norm_depth_k / one_m_alpha / T + (d_cur - ad) * dD_ch;
                v[12] = wgt * dD_ch;

                dL_dalpha *= T;                                                 // :788
                dL_dalpha += tail / one_m_alpha;                                // :791-802
                last_alpha = alpha;

                const float dL_ddist = dL_dalpha * q1.y * -0.5f * G;            // :823
                v[0] = dL_ddist * 2.f * (q0.z * dx + q0.w * dy) * ddelx_dx - dD * q1.w;   // :828, :839
                v[1] = dL_ddist * 2.f * (q1.x * dy + q0.w * dx) * ddely_dy - dD * q2.x;   // :829, :840
                v[2] = dL_ddist * (dx * dx);                                    // :831-835
                v[3] = dL_ddist * (dx * dy);
                v[4] = dL_ddist * (dy * dy);
                v[5] = G * dL_dalpha;                                           // :854
            }

            // 64 -> 1 in registers, then one 13-lane atomic into the Gaussian's accumulation row
            float mine = 0.f;
#pragma unroll
            for (int k = 0; k < 13; k++) {
                const float total = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_sum_to_lane63(v[k])), 63));
                mine = (lane == k) ? total : mine;
            }
            const uint32_t gid = my_id[j];
            if (lane < 13) atomicAdd(a.acc + (size_t)gid * ACC_STRIDE + lane, mine);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

int launch_render_backward(const SoarRastParams &prm, const GeomBuf &g, const BinBuf &b, const ImageBuf &img,
                           const float *dL_dcolor, const float *dL_dnormal, const float *dL_ddepth, const float *dL_dopac,
                           float *acc, hipStream_t stream)
{
    BwdArgs a;
    a.W = prm.W; a.H = prm.H;
    a.gx = (prm.W + TILE - 1) / TILE; a.gy = (prm.H + TILE - 1) / TILE;
    a.ntiles = a.gx * a.gy;
    a.normalize_depth = prm.cfg_normalize_depth;
    a.ranges = img.ranges; a.point_list = b.vals_sorted; a.rec = g.rec; a.bg = prm.bg_dev;
    a.final_T = img.final_T; a.final_D = img.final_D; a.n_contrib = img.n_contrib;
    a.dL_dcolor = dL_dcolor; a.dL_dnormal = dL_dnormal; a.dL_ddepth = dL_ddepth; a.dL_dopac = dL_dopac;
    a.acc = acc;
    StageTimer timer(ST_RENDER_BWD, stream);
    hipLaunchKernelGGL(render_backward_kernel, dim3(a.ntiles), dim3(256), 0, stream, a);
    SOAR_LAUNCH_OK("render_backward", stream, prm.debug);
    return 0;
}

}  // namespace soar
