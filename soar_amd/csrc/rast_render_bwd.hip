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

}  // namespace

namespace {
// order-insensitive mode: the float64 sums rounded once into the float32 rows the geometry backward reads
__global__ void narrow_rows_kernel(size_t n, const double *__restrict__ wide, float *__restrict__ narrow)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) narrow[i] = (float)wide[i];
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
    StageTimer timer(ST_RENDER_BWD, stream);
    const int grid_ranks = blend_grid_ranks(a.ntiles);
    const dim3 grid(4 * min((a.ntiles + 7) / 8 * 8, grid_ranks));
    if (acc64) {
        if (blend) SOAR_LAUNCH_BATCHED((render_backward_slots_kernel<true, BWD_WPE_ONE>), grid, dim3(256), 0, stream, a);
        const size_t n = (size_t)prm.P * ACC_STRIDE;
        hipLaunchKernelGGL(narrow_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, n, acc64, acc);
    } else {
        // (one launch site per form: each keeps its own pending argument blocks)
        if (batch_ctx().n > 1) SOAR_LAUNCH_BATCHED((render_backward_slots_kernel<false, BWD_WPE_BATCH>), grid, dim3(256), 0, stream, a);
        else SOAR_LAUNCH_BATCHED((render_backward_slots_kernel<false, BWD_WPE_ONE>), grid, dim3(256), 0, stream, a);
    }
    SOAR_LAUNCH_OK("render_backward", stream, prm.debug & 1);
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

extern "C" int soar_selftest_wave_reduce(float *out128_dev, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!out128_dev) { soar::set_error("soar_selftest_wave_reduce: NULL"); return 1; }
    hipLaunchKernelGGL(soar::selftest_wave_reduce_kernel, dim3(1), dim3(64), 0, stream, out128_dev);
    SOAR_LAUNCH_OK("selftest_wave_reduce", stream, 1);
    return 0;
}
