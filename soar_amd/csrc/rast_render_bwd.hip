// rast_render_bwd.hip -- back-to-front gradient blend for gfx950.
//
// Replaces renderCUDA<3> backward (DGR/cuda_rasterizer/backward.cu:529-858).
//
// The reference issues 13 global atomicAdd per contributing (pixel, Gaussian) pair, all 256 pixels of a tile hitting the
// same addresses, and every thread walks the whole tile list from its end.  Here the forward kernel's decomposition is
// reused (workgroup = 8x8 pixel quad of a tile, wavefront = 4x4 pixel block, lane = (pixel, slot)): the list is walked back
// to front from the deepest contributor of the wavefront's pixels in staged chunks, four surviving entries per step; the 13
// gradient terms of a step are summed over the 16 pixels with a transpose-reduce (v_permlane32/16_swap + DPP) and leave as
// ONE global_atomic_add_f32 wave-instruction into the 64-byte accumulation rows acc[gaussian][16], which
// geometry_backward_kernel (rast_geom_bwd.hip) consumes.  Same small fixed grid with a rank-stride walk of the tile order and the
// same LDS ring of survivors as the forward kernel (rast_render_fwd.hip).  Details at the kernel below.
//
// Two earlier layouts were measured and retired (profiles/README.md, negative results): per-lane entry pointers with
// ds_add_f32 accumulation rows (2190 us: neighbouring pixels pop the same entry in the same step and the LDS atomics
// serialise) and one wavefront per 8x8 quad walking the entries uniformly (1045 us).
#include "soar_common.h"

#include <cstdio>
#include <cstdlib>

#ifndef SOAR_BWD_DIV
#define SOAR_BWD_DIV 0   // 0: v_rcp_f32 x multiply; 1: + one residual step; 2: IEEE division (development A/B, profiles/README.md)
#endif


namespace soar {

namespace {

struct BwdArgs {
    int W, H, gx, gy, ntiles;
    int normalize_depth;
    const uint2 *ranges;
    const uint32_t *tile_order;
    const uint32_t *point_list;
    const GaussRec *rec;
    const float *bg;
    const float *final_T;
    const float *final_D;
    const uint32_t *n_contrib;
    const float *dL_dcolor, *dL_dnormal, *dL_ddepth, *dL_dopac;
    const float *grad_scale;         // optional device scalar the four image gradients are multiplied by
    float *acc;
    double *acc64;                   // order-insensitive mode: float64 accumulation rows (same layout)
    unsigned long long *stats;       // diagnostic build only (-DSOAR_BWD_STATS): cycle / work counters summed over the wavefronts
    const uint64_t *masks;           // BinBuf::block_masks (rast_blockmask.hip)
    size_t mask_plane;
};

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, off));
    return v;
}

// per-pixel constants (backward.cu:595-623) and the state of the back-to-front recurrences (:607-618)
struct PixelConsts {
    float fx, fy;
    float dC0, dC1, dC2, dN0, dN1, dN2, dD, dD_ch;
    float tail, norm_depth_k, ddelx_dx, ddely_dy;
    uint32_t last;
};
struct PixelState {
    float T, last_alpha;
    float lc0, lc1, lc2, ln0, ln1, ln2, ld;
    float ac0, ac1, ac2, an0, an1, an2, ad;
};
// one Gaussian as the blend sees it
struct Splat {
    float x, y, A, B, C, opacity, depth, plane_a, plane_b, r, g, b, nx, ny, nz;
};

__device__ __forceinline__ void load_pixel(const BwdArgs &a, int px, int py, bool inside, PixelConsts &c, PixelState &s)
{
    const size_t pix = (size_t)a.W * py + px;
    const size_t hw = (size_t)a.H * a.W;
    c.fx = (float)px; c.fy = (float)py;
    c.last = inside ? a.n_contrib[pix] : 0u;                                  // :604
    const float T_final = inside ? a.final_T[pix] : 0.f;
    const float D_final = (inside && a.normalize_depth) ? a.final_D[pix] : 0.f;
    float dO = 0.f;
    c.dC0 = c.dC1 = c.dC2 = c.dN0 = c.dN1 = c.dN2 = c.dD = 0.f;
    // a pixel nothing was blended into takes no part in the walk (it starts at n_contrib): its upstream gradients are not even
    // read -- producers may leave them unwritten (soar_frame_loss with an image buffer does)
    if (inside && c.last != 0u) {
        const float gs = a.grad_scale ? *a.grad_scale : 1.f;
        c.dC0 = gs * a.dL_dcolor[pix]; c.dC1 = gs * a.dL_dcolor[hw + pix]; c.dC2 = gs * a.dL_dcolor[2 * hw + pix];
        c.dN0 = gs * a.dL_dnormal[pix]; c.dN1 = gs * a.dL_dnormal[hw + pix]; c.dN2 = gs * a.dL_dnormal[2 * hw + pix];
        c.dD = gs * a.dL_ddepth[pix];
        dO = gs * a.dL_dopac[pix];
    }
    const float bg_dot = a.bg[0] * c.dC0 + a.bg[1] * c.dC1 + a.bg[2] * c.dC2;     // :798-800
    c.ddelx_dx = 0.5f * a.W; c.ddely_dy = 0.5f * a.H;                             // :622-623
    const float inv_1mTf = 1.f / (1.f - T_final);
    c.dD_ch = a.normalize_depth ? c.dD * inv_1mTf : c.dD;                         // :772
    // dL_dalpha terms that only depend on the pixel and on 1/(1-alpha)  (:791, :801, :802)
    c.tail = dO * T_final - T_final * bg_dot - (a.normalize_depth ? 0.f : T_final * (10.f * c.dD));
    c.norm_depth_k = a.normalize_depth ? c.dD * D_final * inv_1mTf * inv_1mTf * -T_final : 0.f;   // :773
    s.T = T_final;
    s.last_alpha = 0.f;
    s.lc0 = s.lc1 = s.lc2 = s.ln0 = s.ln1 = s.ln2 = s.ld = 0.f;
    s.ac0 = s.ac1 = s.ac2 = s.an0 = s.an1 = s.an2 = s.ad = 0.f;
}

// The same with every load issued at once: the one above first waits for n_contrib and only then asks for the eight gradient
// planes -- two round trips to memory in a row at the start of every wavefront.  What a pixel without contributors holds in the
// gradient planes may be anything (producers may leave it unwritten): selected away, never multiplied.
__device__ __forceinline__ void load_pixel_at_once(const BwdArgs &a, int px, int py, bool inside, PixelConsts &c, PixelState &s)
{
    const size_t hw = (size_t)a.H * a.W;
    const size_t pix = inside ? (size_t)a.W * py + px : 0;
    const uint32_t last = a.n_contrib[pix];
    const float Tf = a.final_T[pix];
    const float Df = a.normalize_depth ? a.final_D[pix] : 0.f;
    const float g0 = a.dL_dcolor[pix], g1 = a.dL_dcolor[hw + pix], g2 = a.dL_dcolor[2 * hw + pix];
    const float n0 = a.dL_dnormal[pix], n1 = a.dL_dnormal[hw + pix], n2 = a.dL_dnormal[2 * hw + pix];
    const float gd = a.dL_ddepth[pix], go = a.dL_dopac[pix];
    const float gs = a.grad_scale ? *a.grad_scale : 1.f;
    c.fx = (float)px; c.fy = (float)py;
    c.last = inside ? last : 0u;                                              // :604
    const bool on = c.last != 0u;
    const float T_final = inside ? Tf : 0.f;
    const float D_final = inside ? Df : 0.f;
    c.dC0 = on ? gs * g0 : 0.f; c.dC1 = on ? gs * g1 : 0.f; c.dC2 = on ? gs * g2 : 0.f;
    c.dN0 = on ? gs * n0 : 0.f; c.dN1 = on ? gs * n1 : 0.f; c.dN2 = on ? gs * n2 : 0.f;
    c.dD = on ? gs * gd : 0.f;
    const float dO = on ? gs * go : 0.f;
    const float bg_dot = a.bg[0] * c.dC0 + a.bg[1] * c.dC1 + a.bg[2] * c.dC2;     // :798-800
    c.ddelx_dx = 0.5f * a.W; c.ddely_dy = 0.5f * a.H;                             // :622-623
    const float inv_1mTf = 1.f / (1.f - T_final);
    c.dD_ch = a.normalize_depth ? c.dD * inv_1mTf : c.dD;                         // :772
    c.tail = dO * T_final - T_final * bg_dot - (a.normalize_depth ? 0.f : T_final * (10.f * c.dD));
    c.norm_depth_k = a.normalize_depth ? c.dD * D_final * inv_1mTf * inv_1mTf * -T_final : 0.f;   // :773
    s.T = T_final;
    s.last_alpha = 0.f;
    s.lc0 = s.lc1 = s.lc2 = s.ln0 = s.ln1 = s.ln2 = s.ld = 0.f;
    s.ac0 = s.ac1 = s.ac2 = s.an0 = s.an1 = s.an2 = s.ad = 0.f;
}

// ---- 16 values x 64 lanes -> 16 totals in one pass ("transpose-reduce") --------------------------------------------
// Summing 13 per-lane terms over the wave one after the other costs 13 x 6 dependent DPP adds.  Instead every
// halving step also halves the number of live registers: v_permlane32_swap / v_permlane16_swap exchange register
// halves between the lane halves (rows) so that ONE add reduces two values across the 32- (16-) lane boundary;
// inside the 16-lane rows the same idea uses DPP row_ror:8 and quad_perm.  36 VALU ops instead of 78 (+26 to gather
// the totals), and the totals end up in 16 different lane groups, ready for a single atomic wave-instruction:
// lane l holds the total of value q(l) = 8*bit5(l) + 4*bit4(l) + 2*bit3(l) + bit0(l).
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
constexpr int DPP_QUAD_XOR3 = 0x1B;
constexpr int DPP_ROW_ROR8 = 0x128, DPP_ROW_HALF_MIRROR = 0x141;

template <int CTRL>
__device__ __forceinline__ float dpp_move(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float swap32_add(float a, float b)
{   // lanes 0-31: a summed over both halves; lanes 32-63: b summed over both halves
    const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ float swap16_add(float a, float b)
{   // even rows: a summed over the row pair; odd rows: b summed over the row pair
    const u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
// ================================================================================================
// the kernel: lane = (pixel, slot) -- four list entries per pixel and step
// ================================================================================================
// Same decomposition as the forward kernel: workgroup = 8x8 quad of a tile, wavefront = 4x4 pixel block, the four
// lanes of a pixel take the four deepest remaining surviving entries.  Per step:
//   * every lane evaluates alpha / liveness of ITS entry for ITS pixel (backward.cu:653-680);
//   * the transmittance in front of each entry follows by dividing back to front through the four slots (quad
//     broadcasts, reference order T = T / (1 - alpha), :683);
//   * the "colour behind" recurrences (:701, :719, :766) are linear, and the gradient only needs their dot product
//     with the pixel's upstream gradient: ONE scalar recurrence P' = alpha (c.d) + (1 - alpha) P replaces seven;
//   * each lane forms the 13 gradient terms of its (pixel, entry) pair; the 16 pixels of the wavefront are summed
//     with the transpose-reduce (v_permlane32/16_swap + DPP) which leaves, in the 16 pixel-lanes of slot s, the 13 totals
//     of entry s: the whole step leaves as ONE global_atomic_add_f32 wave-instruction = four 52-byte row segments.
constexpr int BCHUNK = 256;
constexpr int DPP_Q_BCAST0 = 0x00, DPP_Q_BCAST1 = 0x55, DPP_Q_BCAST2 = 0xAA, DPP_Q_BCAST3 = 0xFF;

// 16 values x 16 pixel-lanes (lane bits 2..5) -> lane (q, slot) holds the total of value q = 8*b5 + 4*b4 + 2*b3 + b2
__device__ __forceinline__ float pixel_reduce16(float v[16], int lane)
{
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = swap32_add(v[k], v[k + 8]);
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = swap16_add(v[k], v[k + 4]);
    const bool b3 = (lane & 8) != 0;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const float keep = b3 ? v[k + 2] : v[k];
        const float send = b3 ? v[k] : v[k + 2];
        v[k] = keep + dpp_move<DPP_ROW_ROR8>(send);
    }
    const bool b2 = (lane & 4) != 0;
    const float keep = b2 ? v[1] : v[0];
    const float send = b2 ? v[0] : v[1];
    return keep + dpp_move<DPP_QUAD_XOR3>(dpp_move<DPP_ROW_HALF_MIRROR>(send));     // partner lane ^ 4
}
__device__ __forceinline__ int pixel_reduce16_slot(int lane)
{
    return ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
}

// device self-test of pixel_reduce16 (soar_selftest_wave_reduce): every lane contributes 16 known values
__global__ void selftest_wave_reduce_kernel(float *out)
{
    const int lane = threadIdx.x & 63;
    float v[16];
#pragma unroll
    for (int q = 0; q < 16; q++) v[q] = (float)((lane + 1) * (q + 1)) + 0.25f * (float)((lane * 7 + q * 3) % 5);
    out[lane] = pixel_reduce16(v, lane);
    out[64 + lane] = (float)pixel_reduce16_slot(lane);
}


template <bool WIDE>
__device__ __forceinline__ void backward_quad(const BwdArgs &a, const int rank, const int quad)
{
    __shared__ float4 sq0[BCHUNK + 1], sq1[BCHUNK + 1], sq2[BCHUNK + 1], sq3[BCHUNK + 1];   // +1: all-zero record
    __shared__ uint32_t sid[BCHUNK + 1];
    __shared__ uint32_t wave_deep[4];
    __shared__ unsigned short todo_ring[4][WAVE + 4];       // per wavefront: [0..3] the zero record, then the LDS slots of a sub-chunk's relevant entries

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t tile_u = a.tile_order[rank];
    if (tile_u == 0xFFFFFFFFu) return;
    const int tile = (int)tile_u;
    const int tx = tile % a.gx, ty = tile / a.gx;
    const int bx0 = tx * TILE + (quad & 1) * 8 + (wave & 1) * 4, by0 = ty * TILE + (quad >> 1) * 8 + (wave >> 1) * 4;
    const int pxl = lane >> 2, slot = lane & 3;
    const int px = bx0 + (pxl & 3), py = by0 + (pxl >> 2);
    const bool inside = px < a.W && py < a.H;

    const uint2 range = a.ranges[tile];
    if (range.x == range.y) return;              // nothing was blended in this tile (most of the image)
    set_wave_priority_by_length(range.y - range.x);
    PixelConsts c;
    PixelState s;
    load_pixel(a, px, py, inside, c, s);
    float T = s.T;                               // replicated in the four lanes of the pixel
    float P = 0.f;                               // (blend of everything behind) . (upstream gradient), replicated
    const uint32_t deepest_wave = wave_max_u32(c.last);
    if (lane == 0) wave_deep[wave] = deepest_wave;
    if (lane < 4) todo_ring[wave][lane] = (unsigned short)BCHUNK;
    if (tid == 0) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        sq0[BCHUNK] = z; sq1[BCHUNK] = z; sq2[BCHUNK] = z; sq3[BCHUNK] = z;
        sid[BCHUNK] = 0u;
    }
    lds_barrier();
    const uint32_t deepest = max(max(wave_deep[0], wave_deep[1]), max(wave_deep[2], wave_deep[3]));   // block-uniform
    if (deepest == 0u) return;

    const int qslot = pixel_reduce16_slot(lane);
    const float dN0x10 = c.dN0 * 10.f, dN1x10 = c.dN1 * 10.f, dN2x10 = c.dN2 * 10.f;          // per-pixel constants of the pair terms
    const float two_ddelx = 2.f * c.ddelx_dx, two_ddely = 2.f * c.ddely_dy;

    // software pipeline over chunks (back to front): records of the chunk in front requested while this one is processed, the
    // list ids one chunk further ahead still (the record gather never waits for its own id load)
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0, r3 = r0;
    uint32_t rid = 0, rid_next = 0;
    const int cfirst = (int)((deepest - 1u) / BCHUNK) * BCHUNK;
    if (cfirst + tid < (int)deepest) {
        rid = a.point_list[range.x + cfirst + tid];
        const float4 *src = reinterpret_cast<const float4 *>(a.rec + rid);
        r0 = src[0]; r1 = src[1]; r2 = src[2]; r3 = src[3];
    }
    if (cfirst >= BCHUNK) rid_next = a.point_list[range.x + cfirst - BCHUNK + tid];
    for (int cbase = cfirst; cbase >= 0; cbase -= BCHUNK) {
        const int n = min(BCHUNK, (int)deepest - cbase);
        if (tid < n) { sq0[tid] = r0; sq1[tid] = r1; sq2[tid] = r2; sq3[tid] = r3; sid[tid] = rid; }
        if (cbase >= BCHUNK) {
            rid = rid_next;
            const float4 *src = reinterpret_cast<const float4 *>(a.rec + rid);
            r0 = src[0]; r1 = src[1]; r2 = src[2]; r3 = src[3];
        }
        if (cbase >= 2 * BCHUNK) rid_next = a.point_list[range.x + cbase - 2 * BCHUNK + tid];
        lds_barrier();          // LDS only: next chunk's gathers and this chunk's atomics stay in flight

        if (cbase < (int)deepest_wave) {
            for (int sub = ((n - 1) / WAVE) * WAVE; sub >= 0; sub -= WAVE) {
                // phase A -- lanes = entries: conservative test against this wave's 4x4 block
                bool relevant = false;
                const int e = sub + lane;
                if (e < n && (uint32_t)(cbase + e) < deepest_wave) {
                    const float4 e0 = sq0[e], e1 = sq1[e];
                    relevant = splat_may_touch_rect(e0.x, e0.y, e0.z, e0.w, e1.x, sq3[e].w, (float)bx0, (float)by0, 3.f);
                }
                // the relevant entries' LDS slots, compacted in list order behind the four pad entries of the wavefront's ring
                // (ballot-prefix ranks): a scalar walk of the ballot's bits costs ~30 scalar instructions per step, and one SIMD
                // issues a scalar instruction only every ~4 cycles
                const unsigned long long todo = __ballot(relevant);
                const int n_todo = (int)__builtin_popcountll(todo);
                if (relevant) todo_ring[wave][4 + __builtin_amdgcn_mbcnt_hi((uint32_t)(todo >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)todo, 0u))] = (unsigned short)e;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

                // phase B -- lanes = (pixel, slot): the four deepest remaining entries per step, slot 3 = deepest; the last step
                // of a sub-chunk reaches into the pads (the zero record)
                for (int top = n_todo; top > 0; top -= 4) {
                    const int j = todo_ring[wave][top + slot];
                    const float4 q0 = sq0[j], q1 = sq1[j], q2 = sq2[j], q3 = sq3[j];
                    const uint32_t gid = sid[j];
                    Splat g;
                    g.x = q0.x; g.y = q0.y; g.A = q0.z; g.B = q0.w; g.C = q1.x; g.opacity = q1.y; g.depth = q1.z;
                    g.plane_a = q1.w; g.plane_b = q2.x; g.r = q2.y; g.g = q2.z; g.b = q2.w; g.nx = q3.x; g.ny = q3.y; g.nz = q3.z;
                    const float dx = g.x - c.fx, dy = g.y - c.fy;
                    const float power = falloff_power(g.A, g.B, g.C, dx, dy);
                    const float G = exp_nonpositive(power);
                    const float alpha = fminf(0.99f, g.opacity * G);
                    const bool live = (j < BCHUNK) && ((uint32_t)(cbase + j) < c.last) && !(power > 0.0f) &&
                                      !(alpha < 1.0f / 255.0f);                    // :653-680
                    const unsigned long long live_mask = __ballot(live);
                    if (live_mask == 0ull) continue;
                    const float a_eff = live ? alpha : 0.f;
                    const float om = 1.f - a_eff;
                    const float om0 = dpp_move<DPP_Q_BCAST0>(om), om1 = dpp_move<DPP_Q_BCAST1>(om),
                                om2 = dpp_move<DPP_Q_BCAST2>(om), om3 = dpp_move<DPP_Q_BCAST3>(om);
                    // transmittance in front of each slot's entry, back to front (:683, T = T / (1 - alpha)); om == 1 for
                    // dead entries.  One IEEE division per lane (r = 1 / om) shared through the quad instead of one per
                    // slot: T * r differs from T / om by one rounding, far inside the 1e-4 gradient tolerance.
                    // r = 1 / om: hardware reciprocal + one Newton step (< 1 ulp; om is in [0.01, 1])
                    float r_om = __builtin_amdgcn_rcpf(om);
                    r_om = __builtin_fmaf(__builtin_fmaf(-om, r_om, 1.0f), r_om, r_om);
                    const float r0 = dpp_move<DPP_Q_BCAST0>(r_om), r1 = dpp_move<DPP_Q_BCAST1>(r_om),
                                r2 = dpp_move<DPP_Q_BCAST2>(r_om), r3 = dpp_move<DPP_Q_BCAST3>(r_om);
                    const float T3 = T * r3, T2 = T3 * r2, T1 = T2 * r1, T0 = T1 * r0;
                    const float T_mine = slot == 0 ? T0 : slot == 1 ? T1 : slot == 2 ? T2 : T3;
                    T = T0;
                    // u = (this entry's colour / normal / depth) . (upstream gradient of the pixel)
                    const float d_cur = g.depth - (dx * g.plane_a + dy * g.plane_b);
                    const float u = g.r * c.dC0 + g.g * c.dC1 + g.b * c.dC2 + g.nx * c.dN0 + g.ny * c.dN1 + g.nz * c.dN2 +
                                    d_cur * c.dD_ch;
                    // P before each slot: P3 = P, P2 = om3 P3 + a3 u3, ...   (:701, :719, :766 folded into one scalar)
                    const float t = a_eff * u;
                    const float t0 = dpp_move<DPP_Q_BCAST0>(t), t1 = dpp_move<DPP_Q_BCAST1>(t),
                                t2 = dpp_move<DPP_Q_BCAST2>(t), t3 = dpp_move<DPP_Q_BCAST3>(t);
                    const float P3 = P, P2 = __builtin_fmaf(om3, P3, t3), P1 = __builtin_fmaf(om2, P2, t2),
                                P0 = __builtin_fmaf(om1, P1, t1);
                    const float P_mine = slot == 0 ? P0 : slot == 1 ? P1 : slot == 2 ? P2 : P3;
                    P = __builtin_fmaf(om0, P0, t0);

                    // per-pair gradient terms; a dead pair (live == false) contributes exact zeros through wgt = 0 and
                    // dL_dalpha = 0 (no branch, no zero-initialised array)
                    float v[16];
                    const float wgt = live ? alpha * T_mine : 0.f;                              // dchannel_dcolor
                    v[6] = wgt * c.dC0; v[7] = wgt * c.dC1; v[8] = wgt * c.dC2;                 // :711
                    v[9] = wgt * dN0x10; v[10] = wgt * dN1x10; v[11] = wgt * dN2x10;          // :727 (x10 normal gain)
                    v[12] = wgt * c.dD_ch;                                                      // :782
                    // ((u - P) + k / om / T) * T + tail / om  (:706,:723,:773,:776,:788,:791-802) with the divisions folded:
                    // k / om / T * T = k / om, and both 1 / om terms share r_om
                    const float dL_dalpha = live ? __builtin_fmaf(u - P_mine, T_mine, (c.norm_depth_k + c.tail) * r_om) : 0.f;
                    const float dL_ddist = dL_dalpha * g.opacity * -0.5f * G;                   // :823
                    const float dD_live = live ? c.dD : 0.f;
                    v[0] = dL_ddist * (g.A * dx + g.B * dy) * two_ddelx - dD_live * g.plane_a;     // :828, :839
                    v[1] = dL_ddist * (g.C * dy + g.B * dx) * two_ddely - dD_live * g.plane_b;     // :829, :840
                    v[2] = dL_ddist * (dx * dx);                                                // :831-835
                    v[3] = dL_ddist * (dx * dy);
                    v[4] = dL_ddist * (dy * dy);
                    v[5] = G * dL_dalpha;                                                       // :854
                    v[13] = 0.f; v[14] = 0.f; v[15] = 0.f;
                    const float total = pixel_reduce16(v, lane);
                    // entries of this step with at least one live pixel: bits slot, slot+4, ... of the ballot
                    const bool entry_live = ((live_mask >> slot) & 0x1111111111111111ull) != 0ull;
                    if (entry_live && qslot < 13) {
                        if (WIDE) atomicAdd(a.acc64 + (size_t)gid * ACC_STRIDE + qslot, (double)total);
                        else atomicAdd(a.acc + (size_t)gid * ACC_STRIDE + qslot, total);
                    }
                }
            }
        }
        lds_barrier();                                // LDS arrays are overwritten by the next chunk
    }
}


// ================================================================================================
// the kernel, second form (round 3, default): lane = list entry, loop over the pixels of the wavefront's block
// ================================================================================================
// The (pixel, slot) form above pays, per step of 4 entries x 16 pixels, for things that are not arithmetic of the gradient: five
// LDS reads of the entries' records per lane, sixteen quad broadcasts for the two recurrences, and the 36-instruction
// transpose-reduce that turns 64 per-pair terms into per-entry sums (~160 vector + ~28 scalar instructions per 64 pairs).
// Turned around -- lane = entry, the wavefront walks the (up to 16) pixels of its 4x4 block one after the other -- none of that
// is needed:
//   * a lane keeps its entry's record in registers for the whole batch of 64 surviving entries (read from LDS once);
//   * the pixel's constants are wave-uniform (broadcast LDS reads), its state (T, P) lives in lane `pixel` of two registers;
//   * the two back-to-front recurrences of a pixel (T = T / (1 - alpha), backward.cu:683; "colour behind me", :701-:766, folded
//     into the scalar P' = alpha u + (1 - alpha) P as above) become ONE inclusive scan over the lanes of the affine maps
//     P -> (1 - alpha_i) P + alpha_i u_i (12 DPP-fused instructions): its multiplier IS the product of the (1 - alpha) behind
//     and at the entry, i.e. T_in / T in front of the entry;
//   * the 13 sums of an entry over the pixels accumulate in the lane's own registers: no cross-lane reduction at all;
//   * pixels that have nothing in the batch (their deepest contributor lies in front of it, or they are outside the image) are
//     skipped by a scalar loop over the set bits of a ballot -- in the other form their lanes idle.
// Survivors of the conservative block test are collected across chunk boundaries (a carried, partly filled batch keeps its
// records in registers) so that batches are full except the last one of a block.  A batch's sums leave through a transposition
// in LDS as 13 atomic wave-instructions over consecutive floats of the 52-byte row segments (rows of 64 different Gaussians
// straight from the lanes would be 64 separate 4-byte requests per instruction).
// About 100 vector instructions per (pixel, 64 entries) against 160 + 28; T in front of an entry is T_in / (product) with one
// reciprocal instead of a chain of divisions -- inside the gradient tolerance like the shared reciprocal of the other form
// (the forward's T, n_contrib and final_T are not touched by any of this).
constexpr int DPP_WAVE_SHR1 = 0x138;

// v of the lane the DPP control names; `otherwise` where that lane does not exist or the row is masked out
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_or(float otherwise, float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(otherwise), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
// Inclusive scan, lane 0 first, of the affine maps P -> m P + b: afterwards lane i holds the composition of the maps of the lanes
// 0 .. i with lane 0's applied first.  (mine o theirs)(P) = m (m' P + b') + b.  Six steps (1, 2, 4, 8 lanes inside the rows of
// 16, then the last lane of the row / of the half in front), each ONE v_fmac_f32_dpp and ONE v_mul_f32_dpp: a lane whose source
// lane does not exist keeps its value (DPP without bound_ctrl disables the write), which is the identity the scan needs.  Written
// in assembly because the compiler does not fold a float identity into the DPP operand (it emits v_mov + v_mov_dpp + op, 36
// instructions); the s_nop cover the two wait states between a vector write and a DPP read of the same register, which the
// compiler's hazard pass cannot see inside an asm block.
__device__ __forceinline__ void affine_scan(float &m, float &b)
{
    asm volatile(
        "s_nop 1\n\t"
        "v_fmac_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_fmac_f32_dpp %0, %0, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_fmac_f32_dpp %0, %0, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_fmac_f32_dpp %0, %0, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_fmac_f32_dpp %0, %0, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_fmac_f32_dpp %0, %0, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(b), "+v"(m));
}
// The same scan with the wait states between its dependent DPP steps filled by six products the pair terms need anyway (and the
// reciprocal of 1 - alpha) instead of s_nop: a scalar instruction between vector ones costs a wavefront far more than its slot
// (tests/tools/issue_model: ~10 cycles per alternation, whatever the number of resident wavefronts).
struct PairProducts { float dxdx, dxdy, dydy, gA, gC, r_om; };
__device__ __forceinline__ PairProducts affine_scan_with_products(float &m, float &b, float dx, float dy, float A, float B, float C)
{
    PairProducts o;
    asm volatile(
        "v_mul_f32 %2, %8, %8\n\t"                                                    // dx dx
        "v_mul_f32 %3, %8, %9\n\t"                                                    // dx dy
        "v_rcp_f32 %7, %1\n\t"                                                        // 1 / (1 - alpha): m still is this lane's factor
        "v_fmac_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32 %4, %9, %9\n\t"                                                    // dy dy
        "v_fmac_f32_dpp %0, %0, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32 %5, %10, %8\n\t"                                                   // A dx
        "v_fmac_f32_dpp %0, %0, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32 %6, %12, %9\n\t"                                                   // C dy
        "v_fmac_f32_dpp %0, %0, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32 %5, %11, %9\n\t"                                                  // + B dy
        "v_fmac_f32_dpp %0, %0, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_fmac_f32 %6, %11, %8\n\t"                                                  // + B dx
        "v_fmac_f32_dpp %0, %0, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf"
        : "+v"(b), "+v"(m), "=&v"(o.dxdx), "=&v"(o.dxdy), "=&v"(o.dydy), "=&v"(o.gA), "=&v"(o.gC), "=&v"(o.r_om)
        : "v"(dx), "v"(dy), "v"(A), "v"(B), "v"(C));
    return o;
}
// device self-test of affine_scan (soar_selftest_affine_scan): out[lane] = {m, b} of the scan of known maps
__global__ void selftest_affine_scan_kernel(const float *m_in, const float *b_in, float *out)
{
    const int lane = threadIdx.x & 63;
    float m = m_in[lane], b = b_in[lane];
    affine_scan(m, b);
    out[lane] = m;
    out[64 + lane] = b;
    out[128 + lane] = dpp_or<DPP_WAVE_SHR1, 0xf>(-7.f, b);
}

__device__ __forceinline__ float lane_value(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }

#ifdef SOAR_BWD_STATS
#define STAT_T(var) const long long var = clock64()
#define STAT_ADD(k, v) st[k] += (unsigned long long)(v)
#else
#define STAT_T(var)
#define STAT_ADD(k, v)
#endif
template <bool WIDE>
__device__ __forceinline__ void backward_quad_entries(const BwdArgs &a, const int rank, const int quad)
{
#ifdef SOAR_BWD_STATS
    unsigned long long st[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const long long t_begin = clock64();
#endif
#ifdef SOAR_BWD_TIMELINE
    unsigned long long tl[8] = {(unsigned long long)wall_clock64(), 0, 0, 0, 0, 0, 0, 0};
#define TL_SET(k, v) tl[k] = (unsigned long long)(v)
#define TL_ADD(k, v) tl[k] += (unsigned long long)(v)
#else
#define TL_SET(k, v)
#define TL_ADD(k, v)
#endif
    __shared__ float4 sq0[BCHUNK], sq1[BCHUNK], sq2[BCHUNK], sq3[BCHUNK];
    __shared__ uint32_t sid[BCHUNK];
    __shared__ uint32_t wave_deep[4];
    __shared__ unsigned char ring[4][BCHUNK];            // per wavefront: LDS slots of the chunk's relevant entries, deepest first
    __shared__ float4 pixc[4][16][3];                    // per wavefront and pixel: {fx, fy, dC0, dC1 | dC2, dN0, dN1, dN2 | dD, dD_ch, tail terms, -}
    __shared__ float xpose[4][WAVE * 13];                // per wavefront: a batch's sums, [entry][13]
    __shared__ uint32_t xgid[4][WAVE];
#ifdef SOAR_EXP_LDS_PAD
    __shared__ uint32_t lds_pad[SOAR_EXP_LDS_PAD / 4];
    if (a.W < 0) lds_pad[threadIdx.x] = 1u;
    if (a.H < 0) xgid[0][0] = lds_pad[threadIdx.x + 1];
#endif

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t tile_u = a.tile_order[rank];
    if (tile_u == 0xFFFFFFFFu) return;
    const int tile = (int)tile_u;
    const int tx = tile % a.gx, ty = tile / a.gx;
    const int bx0 = tx * TILE + (quad & 1) * 8 + (wave & 1) * 4, by0 = ty * TILE + (quad >> 1) * 8 + (wave >> 1) * 4;
    const int p_own = lane & 15;                         // lanes 0..15 own the block's pixels (the others hold copies)
    const int px = bx0 + (p_own & 3), py = by0 + (p_own >> 2);
    const bool inside = px < a.W && py < a.H;

    const uint2 range = a.ranges[tile];
    if (range.x == range.y) return;              // nothing was blended in this tile (most of the image)
    set_wave_priority_by_length(range.y - range.x);
    PixelConsts c;
    PixelState s;
    load_pixel(a, px, py, inside, c, s);
    float vT = s.T, vP = 0.f;                    // lane p: transmittance behind / blend of everything behind . upstream gradient, pixel p
    const uint32_t vLast = c.last;
    if (lane < 16) {
        pixc[wave][lane][0] = make_float4(c.fx, c.fy, c.dC0, c.dC1);
        pixc[wave][lane][1] = make_float4(c.dC2, c.dN0, c.dN1, c.dN2);
        pixc[wave][lane][2] = make_float4(c.dD, c.dD_ch, c.norm_depth_k + c.tail, 0.f);
    }
    const uint32_t deepest_wave = wave_max_u32(vLast);
    if (lane == 0) wave_deep[wave] = deepest_wave;
    lds_barrier();
    const uint32_t deepest = max(max(wave_deep[0], wave_deep[1]), max(wave_deep[2], wave_deep[3]));   // block-uniform
    if (deepest == 0u) return;
    TL_SET(1, wall_clock64());
    TL_SET(4, range.y - range.x);
    TL_SET(5, deepest | ((unsigned long long)deepest_wave << 32));
    const float two_ddelx = 2.f * c.ddelx_dx, two_ddely = 2.f * c.ddely_dy;

    // this lane's entry of the batch being collected (lane 0 = deepest)
    float ex = 0.f, ey = 0.f, eA = 0.f, eB = 0.f, eC = 0.f, eop = 0.f, edepth = 0.f, epa = 0.f, epb = 0.f;
    float er = 0.f, eg = 0.f, eb = 0.f, enx = 0.f, eny = 0.f, enz = 0.f;
    uint32_t egid = 0u, epos = 0xFFFFFFFFu;      // list position; 0xFFFFFFFF: no entry in this lane
    int cnt = 0;                                 // wave-uniform: lanes [0, cnt) are filled

    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0, r3 = r0;
    uint32_t rid = 0, rid_next = 0;
    const int cfirst = (int)((deepest - 1u) / BCHUNK) * BCHUNK;
    if (cfirst + tid < (int)deepest) {
        rid = a.point_list[range.x + cfirst + tid];
        const float4 *src = reinterpret_cast<const float4 *>(a.rec + rid);
        r0 = src[0]; r1 = src[1]; r2 = src[2]; r3 = src[3];
    }
    if (cfirst >= BCHUNK) rid_next = a.point_list[range.x + cfirst - BCHUNK + tid];
    for (int cbase = cfirst; cbase >= 0; cbase -= BCHUNK) {
        const int n = min(BCHUNK, (int)deepest - cbase);
        if (tid < n) { sq0[tid] = r0; sq1[tid] = r1; sq2[tid] = r2; sq3[tid] = r3; sid[tid] = rid; }
        if (cbase >= BCHUNK) {
            rid = rid_next;
            const float4 *src = reinterpret_cast<const float4 *>(a.rec + rid);
            r0 = src[0]; r1 = src[1]; r2 = src[2]; r3 = src[3];
        }
        if (cbase >= 2 * BCHUNK) rid_next = a.point_list[range.x + cbase - 2 * BCHUNK + tid];
        STAT_T(t_b0);
        lds_barrier();          // LDS only: next chunk's gathers and this chunk's atomics stay in flight
        STAT_T(t_b1);
        STAT_ADD(1, t_b1 - t_b0);
        STAT_ADD(10, 1);
#ifdef SOAR_BWD_TIMELINE
        if (cbase == cfirst) tl[2] = wall_clock64();
#endif

        if ((uint32_t)cbase < deepest_wave) {
            // phase A -- lanes = entries: conservative test against the bounding box of the block's pixels that reach into this
            // chunk (deepest contributor behind its first entry)
            float rx0, ry0, rex, rey;
            {
                const uint32_t am = (uint32_t)__ballot(vLast > (uint32_t)cbase) & 0xFFFFu;      // bit p = pixel p = (row p >> 2, column p & 3)
                const uint32_t rows = ((am & 0xFu) ? 1u : 0u) | ((am & 0xF0u) ? 2u : 0u) | ((am & 0xF00u) ? 4u : 0u) | ((am & 0xF000u) ? 8u : 0u);
                const uint32_t cols = (am | (am >> 4) | (am >> 8) | (am >> 12)) & 0xFu;
                const int c0 = __builtin_ctz(cols | 16u), c1 = 31 - __builtin_clz(cols | 1u), q0 = __builtin_ctz(rows | 16u), q1 = 31 - __builtin_clz(rows | 1u);
                rx0 = (float)(bx0 + c0); ry0 = (float)(by0 + q0); rex = (float)max(c1 - c0, 0); rey = (float)max(q1 - q0, 0);
            }
            int n_ring = 0;
            for (int sub = ((n - 1) / WAVE) * WAVE; sub >= 0; sub -= WAVE) {
                bool relevant = false;
                const int e = sub + lane;
                if (e < n && (uint32_t)(cbase + e) < deepest_wave) {
                    const float4 e0 = sq0[e], e1 = sq1[e];
                    relevant = splat_may_touch_rect(e0.x, e0.y, e0.z, e0.w, e1.x, sq3[e].w, rx0, ry0, rex, rey);
                }
                const unsigned long long todo = __ballot(relevant);
                const int k = (int)__builtin_popcountll(todo);
                const int below = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(todo >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)todo, 0u));
                if (relevant) ring[wave][n_ring + (k - 1 - below)] = (unsigned char)e;
                n_ring += k;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            STAT_T(t_a);
            STAT_ADD(2, t_a - t_b1);
            STAT_ADD(11, n_ring);
            // phase B -- lanes = entries of a batch, deepest in lane 0
            int consumed = 0;
            for (;;) {
                STAT_T(t_f0);
                const int take = min(WAVE - cnt, n_ring - consumed);
                {
                    const int k = lane - cnt;
                    if (k >= 0 && k < take) {
                        const int j = ring[wave][consumed + k];
                        const float4 q0 = sq0[j], q1 = sq1[j], q2 = sq2[j], q3 = sq3[j];
                        ex = q0.x; ey = q0.y; eA = q0.z; eB = q0.w; eC = q1.x; eop = q1.y; edepth = q1.z; epa = q1.w;
                        epb = q2.x; er = q2.y; eg = q2.z; eb = q2.w; enx = q3.x; eny = q3.y; enz = q3.z;
                        egid = sid[j]; epos = (uint32_t)(cbase + j);
                    }
                }
                consumed += take; cnt += take;
                STAT_T(t_f1);
                STAT_ADD(3, t_f1 - t_f0);
                if (cnt == 0 || (cnt < WAVE && cbase != 0)) break;         // a partly filled batch waits for the next chunk's survivors
                STAT_ADD(6, 1);
                STAT_ADD(8, cnt);
                TL_ADD(6, 1);

                // ---- one batch: cnt entries x the pixels that reach it
                const uint32_t nearest = (uint32_t)__builtin_amdgcn_readlane((int)epos, cnt - 1);
                uint32_t act = (uint32_t)__ballot(vLast > nearest) & 0xFFFFu;
                float acc[13];
#pragma unroll
                for (int q = 0; q < 13; q++) acc[q] = 0.f;
                float sdD = 0.f;
                bool any_live = false;
                const float oh = -0.5f * eop;
#ifdef SOAR_EXP_NO_PHASEB
                act = 0;
#endif
                STAT_ADD(7, __builtin_popcount(act));
                TL_ADD(7, __builtin_popcount(act));
                while (act) {
                    const int p = __builtin_ctz(act);
                    act &= act - 1u;
                    const uint32_t last_p = (uint32_t)__builtin_amdgcn_readlane((int)vLast, p);
                    const float T_in = lane_value(vT, p), P_in = lane_value(vP, p);
                    const float4 c0 = pixc[wave][p][0], c1 = pixc[wave][p][1], c2 = pixc[wave][p][2];
                    const float dx = ex - c0.x, dy = ey - c0.y;
                    const float power = falloff_power(eA, eB, eC, dx, dy);
                    const float Gx = exp_nonpositive(power);
                    const float alpha = fminf(0.99f, eop * Gx);
                    const bool live = (epos < last_p) && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);      // :653-680
                    const float a_eff = live ? alpha : 0.f;
                    const float G = live ? Gx : 0.f;
                    const float om = 1.f - a_eff;
                    const float d_cur = edepth - (dx * epa + dy * epb);
                    // u = (this entry's colour / normal / depth) . (upstream gradient of the pixel)
                    const float u = er * c0.z + eg * c0.w + eb * c1.x + enx * c1.y + eny * c1.z + enz * c1.w + d_cur * c2.y;
                    float m = om, b = a_eff * u;
#ifndef SOAR_EXP_NO_SCAN
                    affine_scan(m, b);
#endif
                    // m = product of (1 - alpha) over this entry and everything behind it in the batch: T in front of the entry
                    // is T_in / m (:683); b + m P_in = P in front of the entry, its neighbour's = P behind this one
                    const float P_front = __builtin_fmaf(m, P_in, b);
                    const float T_mine = T_in * __builtin_amdgcn_rcpf(m);
                    const float P_mine = dpp_or<DPP_WAVE_SHR1, 0xf>(P_in, P_front);
                    const float r_om = __builtin_amdgcn_rcpf(om);
                    const float wgt = a_eff * T_mine;                                            // dchannel_dcolor
                    acc[6] = __builtin_fmaf(wgt, c0.z, acc[6]); acc[7] = __builtin_fmaf(wgt, c0.w, acc[7]);      // :711
                    acc[8] = __builtin_fmaf(wgt, c1.x, acc[8]);
                    acc[9] = __builtin_fmaf(wgt, c1.y, acc[9]); acc[10] = __builtin_fmaf(wgt, c1.z, acc[10]);    // :727 (x10 at the end)
                    acc[11] = __builtin_fmaf(wgt, c1.w, acc[11]);
                    acc[12] = __builtin_fmaf(wgt, c2.y, acc[12]);                                 // :782
                    // ((u - P) + k / om / T) * T + tail / om  (:706,:723,:773,:776,:788,:791-802), divisions folded as above
                    const float dL_dalpha = __builtin_fmaf(u - P_mine, T_mine, c2.z * r_om);
                    const float dL_ddist = dL_dalpha * (oh * G);                                 // :823; 0 for a dead pair (G = 0)
                    acc[0] = __builtin_fmaf(dL_ddist, eA * dx + eB * dy, acc[0]);               // :828 (x 2 ddelx_dx at the end)
                    acc[1] = __builtin_fmaf(dL_ddist, eC * dy + eB * dx, acc[1]);               // :829
                    acc[2] = __builtin_fmaf(dL_ddist, dx * dx, acc[2]);                          // :831-835
                    acc[3] = __builtin_fmaf(dL_ddist, dx * dy, acc[3]);
                    acc[4] = __builtin_fmaf(dL_ddist, dy * dy, acc[4]);
                    acc[5] = __builtin_fmaf(G, dL_dalpha, acc[5]);                               // :854
                    sdD += live ? c2.x : 0.f;                                                    // :839-840: - dL_dpixD plane_(a, b)
                    any_live = any_live || live;
                    // the pixel's state in front of the batch, back into lane p (no v_writelane builtin in this compiler; the empty
                    // asm keeps the two v_readlane in front of the select instead of inside a branch around it)
                    float T_out = lane_value(T_mine, 63), P_out = lane_value(P_front, 63);
                    asm volatile("" : "+s"(T_out), "+s"(P_out));
                    const bool mine = lane == p;
                    vT = mine ? T_out : vT;
                    vP = mine ? P_out : vP;
                }
                STAT_T(t_p);
                STAT_ADD(4, t_p - t_f1);
                // the batch's sums -> accumulation rows, 13 consecutive floats per entry
                acc[0] = acc[0] * two_ddelx - sdD * epa;
                acc[1] = acc[1] * two_ddely - sdD * epb;
                acc[9] *= 10.f; acc[10] *= 10.f; acc[11] *= 10.f;
#pragma unroll
                for (int q = 0; q < 13; q++) xpose[wave][lane * 13 + q] = acc[q];
                xgid[wave][lane] = any_live ? egid : 0xFFFFFFFFu;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int k = 0; k < 13; k++) {
                    const int f = k * WAVE + lane;
                    const int e = (f * 20165) >> 18;                 // f / 13 for f < 832
                    const int q = f - 13 * e;
                    const uint32_t g = xgid[wave][e];
                    const float v = xpose[wave][f];
#ifdef SOAR_EXP_NO_ATOMICS
                    if (g == 0xFFFFFFF0u) a.acc[q] = v;
#else
                    if (g != 0xFFFFFFFFu) {
                        if (WIDE) atomicAdd(a.acc64 + (size_t)g * ACC_STRIDE + q, (double)v);
                        else atomicAdd(a.acc + (size_t)g * ACC_STRIDE + q, v);
                    }
#endif
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                cnt = 0;
                epos = 0xFFFFFFFFu;
                STAT_T(t_x);
                STAT_ADD(5, t_x - t_p);
                if (consumed == n_ring) break;
            }
        }
        STAT_T(t_e0);
        lds_barrier();                                // LDS arrays are overwritten by the next chunk
        STAT_T(t_e1);
        STAT_ADD(1, t_e1 - t_e0);
    }
#ifdef SOAR_BWD_TIMELINE
    tl[3] = wall_clock64();
    if (lane == 0 && a.stats) {
        unsigned long long *slot = a.stats + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 8;
        if (slot[0] == 0)
            for (int k = 0; k < 8; k++) slot[k] = tl[k];
    }
#endif
#ifdef SOAR_BWD_STATS
    st[0] = (unsigned long long)(clock64() - t_begin);
    if (lane == 0 && a.stats) {
        // one slot per wavefront of the grid (plain read-modify-write: shared counters would serialise the launch in the L2)
        unsigned long long *slot = a.stats + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 12;
        for (int k = 0; k < 12; k++) slot[k] += st[k];
    }
#endif
}


// ================================================================================================
// the kernel, third form (round 3, default): one independent wavefront per 4x4 block, lane = list entry
// ================================================================================================
// The entry-lane arithmetic of backward_quad_entries above, without the workgroup around it.  What the profile of that form showed
// (scripts/bwd_timeline.py): a wavefront of an average tile lives 30 us -- 10 us of them in four dependent round trips to memory
// before its first useful instruction (tile -> pixels -> list ids -> records, the last two behind a workgroup barrier) --, executes
// ~2000 vector instructions, and waits at two barriers per chunk for the slowest of its three siblings.  Here:
//   * the block masks (rast_blockmask.hip) already say which entries of the list concern this block: no staging of the tile's
//     whole list, no test, no barrier; a wavefront that has nothing to do leaves without waiting for its neighbours;
//   * the mask words of the block (64 groups = 4096 list positions per load) are asked for together with the pixel's planes; the
//     set bits are compacted, deepest first, into batches of 64 list positions; lists ids and then records are gathered straight into
//     the lanes' registers, TWO and ONE batch ahead of the one being worked on;
//   * the sums of batch n leave through LDS (13 row-contiguous atomic wave-instructions) at the start of batch n + 1, behind the
//     gathers of that step: every wait for a gather then finds the atomics in front of it a whole pixel loop old.
struct BlockWalk {           // descending walk over the 64-position groups of a block's mask words
    uint32_t x0, top;        // first list position of the tile, first position behind the deepest contributor (absolute)
    uint32_t g_top;          // group of position top - 1: the walk visits g_top, g_top - 1, ... x0 >> 6
    int n_groups;
    int s, s_base;           // next group of the sequence, first group of the word window
    unsigned long long rem;  // bits of group s - 1 not yet consumed
    uint32_t base;           // list position of bit 0 of `rem`
};

template <bool WIDE>
__device__ __forceinline__ void backward_block(const BwdArgs &a, const int rank, const int blk, float4 (*pixc)[3], uint32_t *ring,
                                               float *xpose, uint32_t *xgid)
{
    const int lane = threadIdx.x & 63;
    const uint32_t tile_u = a.tile_order[rank];
    if (tile_u == 0xFFFFFFFFu) return;
    const int tile = (int)tile_u;
    const uint2 range = a.ranges[tile];
    if (range.x == range.y) return;
    const int tx = tile % a.gx, ty = tile / a.gx;
    const int bx0 = tx * TILE + ((blk >> 2) & 1) * 8 + (blk & 1) * 4, by0 = ty * TILE + (blk >> 3) * 8 + ((blk >> 1) & 1) * 4;
    const int p_own = lane & 15;                         // lanes 0..15 own the block's pixels (the others hold copies)
    const int px = bx0 + (p_own & 3), py = by0 + (p_own >> 2);
    const bool inside = px < a.W && py < a.H;

    PixelConsts c;
    PixelState st;
    load_pixel_at_once(a, px, py, inside, c, st);
    float vT = st.T, vP = 0.f;                   // lane p: transmittance behind / blend of everything behind . upstream gradient, pixel p
    const uint32_t vLast = c.last;
    const uint32_t deepest = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_max_u32(vLast));
    if (deepest == 0u) return;
    set_wave_priority_by_length(deepest);
    if (lane < 16) {
        pixc[lane][0] = make_float4(c.fx, c.fy, c.dC0, c.dC1);
        pixc[lane][1] = make_float4(c.dC2, c.dN0, c.dN1, c.dN2);
        pixc[lane][2] = make_float4(c.dD, c.dD_ch, c.norm_depth_k + c.tail, 0.f);
    }
    const float two_ddelx = 2.f * c.ddelx_dx, two_ddely = 2.f * c.ddely_dy;

    BlockWalk w;
    w.x0 = range.x; w.top = range.x + deepest;
    w.g_top = (w.top - 1u) >> 6;
    w.n_groups = (int)(w.g_top - (w.x0 >> 6)) + 1;
    w.s = 0; w.s_base = 0; w.rem = 0ull; w.base = 0u;
    uint32_t w_lo = 0u, w_hi = 0u;               // lane j: word of group g_top - (s_base + j)
    auto load_window = [&]() {
        const int sj = w.s_base + lane;
        const unsigned long long word = sj < w.n_groups ? a.masks[(size_t)blk * a.mask_plane + (w.g_top - (uint32_t)sj)] : 0ull;
        w_lo = (uint32_t)word; w_hi = (uint32_t)(word >> 32);
    };
    load_window();

    // next batch of up to 64 list positions, deepest first -> ring[0 .. count)
    auto assemble = [&]() -> int {
        int cnt = 0;
        for (;;) {
            if (w.rem == 0ull) {
                if (w.s >= w.n_groups) break;
                if (w.s - w.s_base >= WAVE) { w.s_base = w.s; load_window(); }
                const int j = w.s - w.s_base;
                unsigned long long word = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)w_lo, j) |
                                          ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)w_hi, j) << 32);
                w.base = (w.g_top - (uint32_t)w.s) << 6;
                // positions at and behind `top` are not walked (nothing was blended there for this block's pixels); positions in
                // front of x0 belong to the tile before
                const uint32_t nvalid = w.top - w.base;                  // > 0
                if (nvalid < 64u) word &= (1ull << nvalid) - 1ull;
                if (w.x0 > w.base) word &= ~0ull << (w.x0 - w.base);     // (only in the last group of the walk: x0 - base < 64)
                w.rem = word;
                w.s++;
                continue;
            }
            const int pop = (int)__builtin_popcountll(w.rem);
            const int room = WAVE - cnt;
            const bool set = (w.rem >> lane) & 1ull;
            const int below = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(w.rem >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)w.rem, 0u));
            const int above = pop - 1 - below;                           // set bits above mine: they go first
            if (set && above < room) ring[cnt + above] = w.base + (uint32_t)lane;
            if (pop <= room) { cnt += pop; w.rem = 0ull; }
            else { cnt = WAVE; w.rem = __ballot(set && above >= room); }
            if (cnt == WAVE) break;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        return cnt;
    };

    // this lane's entry of the batch being worked on (lane 0 = deepest), and of the batch after it
    float ex = 0.f, ey = 0.f, eA = 0.f, eB = 0.f, eC = 0.f, eop = 0.f, edepth = 0.f, epa = 0.f, epb = 0.f;
    float er = 0.f, eg = 0.f, eb = 0.f, enx = 0.f, eny = 0.f, enz = 0.f;
    uint32_t egid = 0u, epos = 0xFFFFFFFFu;      // position relative to the start of the list; 0xFFFFFFFF: no entry in this lane
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0, r3 = r0;
    uint32_t rgid = 0u, rpos = 0xFFFFFFFFu;
    uint32_t ngid = 0u, npos = 0xFFFFFFFFu;      // ids of the batch after that
    int cnt_e = 0, cnt_r = 0, cnt_n = 0;

    auto fetch_ids = [&](int cnt, uint32_t &gid, uint32_t &pos) {
        pos = 0xFFFFFFFFu;
        if (lane < cnt) {
            const uint32_t at = ring[lane];
            gid = a.point_list[at];
            pos = at - w.x0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // the ring is refilled by the next assemble
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto gather = [&]() {                                             // records of (rgid, rpos)
        if (rpos != 0xFFFFFFFFu) {
            const float4 *src = reinterpret_cast<const float4 *>(a.rec + rgid);
            r0 = src[0]; r1 = src[1]; r2 = src[2]; r3 = src[3];
        }
    };
    auto flush_atomics = [&]() {
#pragma unroll
        for (int k = 0; k < 13; k++) {
            const int f = k * WAVE + lane;
            const int e = (f * 20165) >> 18;                 // f / 13 for f < 832
            const int q = f - 13 * e;
            const uint32_t g = xgid[e];
            const float v = xpose[f];
#ifdef SOAR_EXP_NO_ATOMICS
            if (g == 0xFFFFFFF0u) a.acc[q] = v;
#else
            if (g != 0xFFFFFFFFu) {
                if (WIDE) atomicAdd(a.acc64 + (size_t)g * ACC_STRIDE + q, (double)v);
                else atomicAdd(a.acc + (size_t)g * ACC_STRIDE + q, v);
            }
#endif
        }
    };

    // fill the pipeline: batch 0 in (r), batch 1's ids in (n)
    cnt_r = assemble();
    fetch_ids(cnt_r, rgid, rpos);
    gather();
    cnt_n = assemble();
    fetch_ids(cnt_n, ngid, npos);
    bool pending = false;
    for (;;) {
        // (r) -> (e): the batch to work on
        ex = r0.x; ey = r0.y; eA = r0.z; eB = r0.w; eC = r1.x; eop = r1.y; edepth = r1.z; epa = r1.w;
        epb = r2.x; er = r2.y; eg = r2.z; eb = r2.w; enx = r3.x; eny = r3.y; enz = r3.z;
        egid = rgid; epos = rpos; cnt_e = cnt_r;
        if (cnt_e == 0) break;
        // the sums of the batch before leave now, in front of this step's gathers
        if (pending) flush_atomics();
        // (n) -> (r): its records are asked for now, the ids of the batch behind it next
        rgid = ngid; rpos = npos; cnt_r = cnt_n;
        gather();
        cnt_n = assemble();
        fetch_ids(cnt_n, ngid, npos);

        // ---- one batch: cnt_e entries x the pixels that reach it
        const uint32_t nearest = (uint32_t)__builtin_amdgcn_readlane((int)epos, cnt_e - 1);
        uint32_t act = (uint32_t)__ballot(vLast > nearest) & 0xFFFFu;
        float acc[13];
#pragma unroll
        for (int q = 0; q < 13; q++) acc[q] = 0.f;
        float sdD = 0.f;
        float any_live = 0.f;                    // > 0: some pair of this lane's entry was live
        const float oh = -0.5f * eop;
        while (act) {
            const int p = __builtin_ctz(act);
            act &= act - 1u;
            const uint32_t last_p = (uint32_t)__builtin_amdgcn_readlane((int)vLast, p);
            const float T_in = lane_value(vT, p), P_in = lane_value(vP, p);
            const float4 c0 = pixc[p][0], c1 = pixc[p][1], c2 = pixc[p][2];
            const float dx = ex - c0.x, dy = ey - c0.y;
            const float power = falloff_power(eA, eB, eC, dx, dy);
            const float Gx = exp_nonpositive(power);
            const float alpha = fminf(0.99f, eop * Gx);
            // the skip rules (:653-680) as three selects on the number, not one predicate: combining the comparisons first costs
            // scalar mask instructions
            float a_eff = (power > 0.0f) ? 0.f : alpha;
            a_eff = (alpha < 1.0f / 255.0f) ? 0.f : a_eff;
            a_eff = (epos < last_p) ? a_eff : 0.f;
            const bool live = a_eff != 0.f;
            const float G = live ? Gx : 0.f;
            const float d_cur = edepth - (dx * epa + dy * epb);
            const float u = er * c0.z + eg * c0.w + eb * c1.x + enx * c1.y + eny * c1.z + enz * c1.w + d_cur * c2.y;
            float m = 1.f - a_eff, b = a_eff * u;
#if SOAR_BWD_DIV
            const float om = m;
#endif
            const PairProducts pp = affine_scan_with_products(m, b, dx, dy, eA, eB, eC);
            const float P_front = __builtin_fmaf(m, P_in, b);
#if SOAR_BWD_DIV == 2
            // the reference divides (backward.cu:683, :791): IEEE quotients instead of v_rcp_f32 (1 ulp) x multiply
            const float T_mine = T_in / m;
            const float r_om = 1.f / om;
#elif SOAR_BWD_DIV == 1
            // v_rcp_f32 + one residual step each: the quotient T_in / m and the reciprocal 1 / (1 - alpha) to the last bit or next to it
            const float r_m = __builtin_amdgcn_rcpf(m);
            const float q0 = T_in * r_m;
            const float T_mine = __builtin_fmaf(__builtin_fmaf(-m, q0, T_in), r_m, q0);
            const float r_om = __builtin_fmaf(__builtin_fmaf(-om, pp.r_om, 1.f), pp.r_om, pp.r_om);
#else
            const float T_mine = T_in * __builtin_amdgcn_rcpf(m);
            const float r_om = pp.r_om;
#endif
            const float P_mine = dpp_or<DPP_WAVE_SHR1, 0xf>(P_in, P_front);
            const float wgt = a_eff * T_mine;                                            // dchannel_dcolor
            acc[6] = __builtin_fmaf(wgt, c0.z, acc[6]); acc[7] = __builtin_fmaf(wgt, c0.w, acc[7]);      // :711
            acc[8] = __builtin_fmaf(wgt, c1.x, acc[8]);
            acc[9] = __builtin_fmaf(wgt, c1.y, acc[9]); acc[10] = __builtin_fmaf(wgt, c1.z, acc[10]);    // :727 (x10 at the end)
            acc[11] = __builtin_fmaf(wgt, c1.w, acc[11]);
            acc[12] = __builtin_fmaf(wgt, c2.y, acc[12]);                                 // :782
            const float dL_dalpha = __builtin_fmaf(u - P_mine, T_mine, c2.z * r_om);
            const float dL_ddist = dL_dalpha * (oh * G);                                 // :823; 0 for a dead pair (G = 0)
            acc[0] = __builtin_fmaf(dL_ddist, pp.gA, acc[0]);                            // :828 (x 2 ddelx_dx at the end)
            acc[1] = __builtin_fmaf(dL_ddist, pp.gC, acc[1]);                            // :829
            acc[2] = __builtin_fmaf(dL_ddist, pp.dxdx, acc[2]);                          // :831-835
            acc[3] = __builtin_fmaf(dL_ddist, pp.dxdy, acc[3]);
            acc[4] = __builtin_fmaf(dL_ddist, pp.dydy, acc[4]);
            acc[5] = __builtin_fmaf(G, dL_dalpha, acc[5]);                               // :854
            sdD += live ? c2.x : 0.f;                                                    // :839-840: - dL_dpixD plane_(a, b)
            any_live = fmaxf(any_live, a_eff);
#ifdef SOAR_EXP_PAD_VALU
            {   // timing experiment: extra vector instructions (results never used for real)
                float pz = wgt;
#pragma unroll
                for (int k = 0; k < SOAR_EXP_PAD_VALU; k++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(pz) : "v"(dx), "v"(dy));
                if (pz == 123456.f) sdD += 1.f;
            }
#endif
#ifdef SOAR_EXP_PAD_SALU
            {
                int sz = p;
#pragma unroll
                for (int k = 0; k < SOAR_EXP_PAD_SALU; k++) asm volatile("s_add_i32 %0, %0, 1" : "+s"(sz));
                if (sz == -12345) sdD += 1.f;
            }
#endif
            float T_out = lane_value(T_mine, 63), P_out = lane_value(P_front, 63);
            asm volatile("" : "+s"(T_out), "+s"(P_out));
            const bool mine = lane == p;
            vT = mine ? T_out : vT;
            vP = mine ? P_out : vP;
        }
        // the batch's sums -> LDS, [entry][13]; they leave at the start of the next step
        acc[0] = acc[0] * two_ddelx - sdD * epa;
        acc[1] = acc[1] * two_ddely - sdD * epb;
        acc[9] *= 10.f; acc[10] *= 10.f; acc[11] *= 10.f;
#pragma unroll
        for (int q = 0; q < 13; q++) xpose[lane * 13 + q] = acc[q];
        xgid[lane] = any_live != 0.f ? egid : 0xFFFFFFFFu;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        pending = true;
    }
    if (pending) flush_atomics();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");             // the next tile of this wavefront reuses the LDS arrays
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#ifndef SOAR_BWD_BLK_WPE
#define SOAR_BWD_BLK_WPE 4
#endif
// one wavefront per workgroup: a finished block frees its slot at once.  Grid = 16 x ranks; the 16 blocks of a tile are
// consecutive workgroups of one XCD (its L2 holds the tile's records), ranks dealt round-robin to the XCDs.
template <bool WIDE>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SOAR_BWD_BLK_WPE, 8))) render_backward_blocks_kernel(Batch<BwdArgs> batch)
{
    __shared__ float4 pixc[16][3];                       // per pixel: {fx, fy, dC0, dC1 | dC2, dN0, dN1, dN2 | dD, dD_ch, tail terms, -}
    __shared__ uint32_t ring[WAVE];
    __shared__ float xpose[WAVE * 13];                   // a batch's sums, [entry][13]
    __shared__ uint32_t xgid[WAVE];
    int frame, bx;
    batch_interleave(frame, bx);
    const BwdArgs &a = batch.v[frame];
    const int xcd = bx & 7, kth = bx >> 3;
    const int rank0 = (kth >> 4) * 8 + xcd, blk = kth & 15;
    const int stride = (int)(gridDim.x >> 4);                // ranks per pass of the grid (a multiple of 8)
    const int n_work = (int)a.tile_order[(a.ntiles + 7) / 8 * 8];
    for (int rank = rank0; rank < n_work; rank += stride) backward_block<WIDE>(a, rank, blk, pixc, ring, xpose, xgid);
}

#ifdef SOAR_BWD_WPE
constexpr int BWD_WPE_ONE = SOAR_BWD_WPE, BWD_WPE_BATCH = SOAR_BWD_WPE;
#else
constexpr int BWD_WPE_ONE = 5, BWD_WPE_BATCH = 6;
#endif
// Same small grid and rank-stride walk of the longest-first tile order as the forward kernel (render_forward_kernel,
// rast_render_fwd.hip): the tiles behind the first n_work ranks have nothing to differentiate and are never visited.
// WPE = wavefronts per SIMD the register allocation aims at: 5 (95 VGPRs, no scratch) when one frame is launched -- alone or on a
// stream of its own next to other frames' chains --, 6 (80 VGPRs, 12 bytes of scratch per lane) when the frames of a step share
// the launch: measured +1.7 % on the step in that form and -0.7 % in the other (development override: -DSOAR_BWD_WPE=n).
template <bool WIDE, int WPE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, 8))) render_backward_slots_kernel(Batch<BwdArgs> batch)
{
    int frame, bx;
    batch_interleave(frame, bx);
    const BwdArgs &a = batch.v[frame];
    const int xcd = bx & 7, kth = bx >> 3;
    const int rank0 = (kth >> 2) * 8 + xcd, quad = kth & 3;
    const int stride = (int)(gridDim.x >> 2);                // ranks per pass of the grid (a multiple of 8)
    const int n_work = (int)a.tile_order[(a.ntiles + 7) / 8 * 8];
    for (int rank = rank0; rank < n_work; rank += stride) {
        backward_quad<WIDE>(a, rank, quad);
        lds_barrier();                                       // the next item's staging overwrites this one's LDS image
    }
}

// the entry-lane form: 36 KB of LDS per workgroup -> four workgroups (16 wavefronts) per CU, up to 128 VGPRs
#ifndef SOAR_BWD_ENT_WPE
#define SOAR_BWD_ENT_WPE 4
#endif
template <bool WIDE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SOAR_BWD_ENT_WPE, 8))) render_backward_entries_kernel(Batch<BwdArgs> batch)
{
    int frame, bx;
    batch_interleave(frame, bx);
    const BwdArgs &a = batch.v[frame];
    const int xcd = bx & 7, kth = bx >> 3;
    const int rank0 = (kth >> 2) * 8 + xcd, quad = kth & 3;
    const int stride = (int)(gridDim.x >> 2);
    const int n_work = (int)a.tile_order[(a.ntiles + 7) / 8 * 8];
    for (int rank = rank0; rank < n_work; rank += stride) {
        backward_quad_entries<WIDE>(a, rank, quad);
        lds_barrier();
    }
}

}  // namespace

namespace {
// order-insensitive mode: the float64 sums rounded once into the float32 rows the geometry backward reads
struct NarrowArgs { size_t n; const double *wide; float *narrow; };
__global__ void narrow_rows_kernel(Batch<NarrowArgs> batch)
{
    const NarrowArgs &a = batch.v[blockIdx.y];
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < a.n) a.narrow[i] = (float)a.wide[i];
}
// development switch: SOAR_BWD_FORM=slots / entries selects the (pixel, slot) form of rounds 1-2 / the workgroup-staged entry-lane
// form (A/B on one box); default: independent wavefronts on the block masks
int bwd_form()
{
    static const int v = !getenv("SOAR_BWD_FORM") ? 0 : getenv("SOAR_BWD_FORM")[0] == 's' ? 1 : getenv("SOAR_BWD_FORM")[0] == 'e' ? 2 : 0;
    return v;
}
}  // namespace

int launch_render_backward(const SoarRastParams &prm, const GeomBuf &g, const BinBuf &b, const ImageBuf &img,
                           const float *dL_dcolor, const float *dL_dnormal, const float *dL_ddepth, const float *dL_dopac,
                           const float *grad_scale, float *acc, double *acc64, bool blend, hipStream_t stream)
{
    BwdArgs a;
    a.grad_scale = grad_scale;
    a.W = prm.W; a.H = prm.H;
    a.gx = (prm.W + TILE - 1) / TILE; a.gy = (prm.H + TILE - 1) / TILE;
    a.ntiles = a.gx * a.gy;
    a.normalize_depth = prm.cfg_normalize_depth;
    a.ranges = img.ranges; a.tile_order = img.tile_order; a.point_list = b.vals_sorted; a.rec = g.rec; a.bg = prm.bg_dev;
    a.final_T = img.final_T; a.final_D = img.final_D; a.n_contrib = img.n_contrib;
    a.dL_dcolor = dL_dcolor; a.dL_dnormal = dL_dnormal; a.dL_ddepth = dL_ddepth; a.dL_dopac = dL_dopac;
    a.acc = acc; a.acc64 = acc64;
    a.stats = nullptr;
#ifdef SOAR_BWD_TIMELINE
    static unsigned long long *tl_dev = nullptr;
    constexpr size_t TL_SLOTS = 8 * 4 * 4096 * 4;
    if (!tl_dev) { SOAR_HIP_OK(hipMalloc(&tl_dev, TL_SLOTS * 8 * sizeof(unsigned long long))); SOAR_HIP_OK(hipMemset(tl_dev, 0, TL_SLOTS * 8 * sizeof(unsigned long long))); }
    a.stats = tl_dev;
#endif
#ifdef SOAR_BWD_STATS
    static unsigned long long *stats_dev = nullptr;
    constexpr size_t STAT_SLOTS = 8 * 4 * 4096 * 4;
    if (!stats_dev) { SOAR_HIP_OK(hipMalloc(&stats_dev, STAT_SLOTS * 12 * sizeof(unsigned long long))); SOAR_HIP_OK(hipMemset(stats_dev, 0, STAT_SLOTS * 12 * sizeof(unsigned long long))); }
    a.stats = stats_dev;
#endif
    StageTimer timer(ST_RENDER_BWD, stream);
    const int grid_ranks = blend_grid_ranks(a.ntiles);
    const dim3 grid(4 * min((a.ntiles + 7) / 8 * 8, grid_ranks));
    const int form = bwd_form();
    const bool slots = form == 1;
    a.masks = b.block_masks; a.mask_plane = b.mask_plane;
    const dim3 grid_blocks(16 * min((a.ntiles + 7) / 8 * 8, grid_ranks));
    // (one launch site per form: each keeps its own pending argument blocks)
    if (acc64) {
        if (blend) {
            if (slots) SOAR_LAUNCH_BATCHED((render_backward_slots_kernel<true, BWD_WPE_ONE>), grid, dim3(256), 0, stream, a);
            else if (form == 2) SOAR_LAUNCH_BATCHED((render_backward_entries_kernel<true>), grid, dim3(256), 0, stream, a);
            else SOAR_LAUNCH_BATCHED((render_backward_blocks_kernel<true>), grid_blocks, dim3(64), 0, stream, a);
        }
        // in a batch this launches with the last frame like the blend in front of it (it used to launch per call, i.e. for
        // the frames 0 .. n-2 BEFORE their rows had been accumulated)
        NarrowArgs na;
        na.n = (size_t)prm.P * ACC_STRIDE; na.wide = acc64; na.narrow = acc;
        SOAR_LAUNCH_BATCHED(narrow_rows_kernel, dim3((unsigned)((na.n + 255) / 256)), dim3(256), 0, stream, na);
    } else if (slots) {
        if (batch_ctx().n > 1) SOAR_LAUNCH_BATCHED((render_backward_slots_kernel<false, BWD_WPE_BATCH>), grid, dim3(256), 0, stream, a);
        else SOAR_LAUNCH_BATCHED((render_backward_slots_kernel<false, BWD_WPE_ONE>), grid, dim3(256), 0, stream, a);
    } else if (form == 2) {
        SOAR_LAUNCH_BATCHED((render_backward_entries_kernel<false>), grid, dim3(256), 0, stream, a);
    } else {
        SOAR_LAUNCH_BATCHED((render_backward_blocks_kernel<false>), grid_blocks, dim3(64), 0, stream, a);
    }
    SOAR_LAUNCH_OK("render_backward", stream, prm.debug & 1);
#ifdef SOAR_BWD_TIMELINE
    if (batch_ctx().n > 1 && batch_ctx().f == batch_ctx().n - 1) {
        static int tl_launches = 0;
        ++tl_launches;
        if (tl_launches == 8) SOAR_HIP_OK(hipMemsetAsync(tl_dev, 0, TL_SLOTS * 8 * sizeof(unsigned long long), stream));   // in front of launch 9
        if (tl_launches == 9 && getenv("SOAR_BWD_TIMELINE_FILE")) {
            SOAR_HIP_OK(hipStreamSynchronize(stream));
            unsigned long long *all = (unsigned long long *)malloc(TL_SLOTS * 8 * sizeof(unsigned long long));
            SOAR_HIP_OK(hipMemcpy(all, tl_dev, TL_SLOTS * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            FILE *f = fopen(getenv("SOAR_BWD_TIMELINE_FILE"), "wb");
            if (f) { fwrite(all, 1, TL_SLOTS * 8 * sizeof(unsigned long long), f); fclose(f); }
            free(all);
        }
    }
#endif
#ifdef SOAR_BWD_STATS
    if (!batch_ctx().n || batch_ctx().f == batch_ctx().n - 1) {
        static int launches = 0;
        if (++launches % 10 == 0) {
            unsigned long long h[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            SOAR_HIP_OK(hipStreamSynchronize(stream));
            unsigned long long *all = (unsigned long long *)malloc(STAT_SLOTS * 12 * sizeof(unsigned long long));
            SOAR_HIP_OK(hipMemcpy(all, stats_dev, STAT_SLOTS * 12 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            SOAR_HIP_OK(hipMemset(stats_dev, 0, STAT_SLOTS * 12 * sizeof(unsigned long long)));
            unsigned long long longest = 0;
            for (size_t i = 0; i < STAT_SLOTS; i++) {
                for (int k = 0; k < 12; k++) h[k] += all[i * 12 + k];
                if (all[i * 12] > longest) longest = all[i * 12];
            }
            free(all);
            fprintf(stderr, "longest-lived wavefront slot: %.3f Mcycles over the 10 launches\n", longest / 1e6);
            fprintf(stderr, "bwd stats (10 launches, Mcycles summed over wavefronts): total %.1f barrier-wait %.1f phaseA %.1f fill %.1f pixels %.1f flush %.1f | "
                    "batches %llu pixel-iterations %llu filled lanes %llu chunk-iterations %llu survivors %llu\n", h[0] / 1e6, h[1] / 1e6, h[2] / 1e6, h[3] / 1e6,
                    h[4] / 1e6, h[5] / 1e6, h[6], h[7], h[8], h[10], h[11]);
        }
    }
#endif
    return 0;
}

}  // namespace soar

namespace soar {
namespace {
// out[i] = exp_nonpositive(x[i]), ref[i] = expf(x[i]) (the device math library)
__global__ void selftest_exp_kernel(const float *x, int n, float *out, float *ref)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { out[i] = exp_nonpositive(x[i]); ref[i] = expf(x[i]); }
}
}  // namespace
}  // namespace soar

extern "C" int soar_selftest_exp(const float *x_dev, int32_t n, float *out_dev, float *expf_dev, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n <= 0 || !x_dev || !out_dev || !expf_dev) { soar::set_error("soar_selftest_exp: bad arguments"); return 1; }
    hipLaunchKernelGGL(soar::selftest_exp_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, x_dev, n, out_dev, expf_dev);
    SOAR_LAUNCH_OK("selftest_exp", stream, 1);
    return 0;
}

extern "C" int soar_selftest_affine_scan(const float *m64_dev, const float *b64_dev, float *out192_dev, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!m64_dev || !b64_dev || !out192_dev) { soar::set_error("soar_selftest_affine_scan: NULL"); return 1; }
    hipLaunchKernelGGL(soar::selftest_affine_scan_kernel, dim3(1), dim3(64), 0, stream, m64_dev, b64_dev, out192_dev);
    SOAR_LAUNCH_OK("selftest_affine_scan", stream, 1);
    return 0;
}

extern "C" int soar_selftest_wave_reduce(float *out128_dev, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!out128_dev) { soar::set_error("soar_selftest_wave_reduce: NULL"); return 1; }
    hipLaunchKernelGGL(soar::selftest_wave_reduce_kernel, dim3(1), dim3(64), 0, stream, out128_dev);
    SOAR_LAUNCH_OK("selftest_wave_reduce", stream, 1);
    return 0;
}
