// rast_render_bwd.hip -- back-to-front gradient blend for gfx950.
//
// Replaces renderCUDA<3> backward (DGR/cuda_rasterizer/backward.cu:529-858).
//
// The reference issues 13 global atomicAdd per contributing (pixel, Gaussian) pair, all 256 pixels of a tile hitting the
// same addresses, and every thread walks the whole tile list from its end.  Here one independent wavefront owns a 4x4 pixel
// block of a tile and walks, back to front from the deepest contributor of its pixels, only the list entries the block masks
// (rast_blockmask.hip) name for it, 64 at a time with lane = entry: the back-to-front recurrences of a pixel become a 64-lane
// DPP scan, the 13 gradient terms of an entry accumulate over the block's pixels in the lane's own registers and leave as
// global_atomic_add_f32 wave-instructions of four WHOLE 64-byte accumulation rows acc[gaussian][16] each (this hardware executes
// them at the memory side, one request per row and instruction: the launch's bound until the rows stopped straddling instructions,
// round 5), which geometry_backward_kernel (rast_geom_bwd.hip) consumes.  Same small fixed grid with a rank-stride walk of the longest-first
// tile order as the forward kernel (rast_render_fwd.hip).  Details at the kernel below.
//
// Earlier layouts, measured and retired (profiles/README.md, negative results): per-lane entry pointers with ds_add_f32
// accumulation rows (2190 us), one wavefront per 8x8 quad walking the entries uniformly (1045 us), lane = (pixel, slot) with a
// transpose-reduce (rounds 1-2: 395 us per 4-frame launch) and lane = entry inside a workgroup that stages the tile's list
// (352 us); this file holds the form that replaced them (298-306 us).
#include "soar_common.h"

#include <type_traits>

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
    const uint4 *order_rec;
    const uint32_t *point_list;
    const GaussRec *rec;
    const float *bg;
    const float *final_T;
    const float *final_D;
    const uint32_t *n_contrib;
    const float *dL_dcolor, *dL_dnormal, *dL_ddepth, *dL_dopac;
    const float *grad_scale;         // optional device scalar the four image gradients are multiplied by
    const float *normal_scale;       // optional device scalar for the normal image's gradient alone (soar_rast_backward_occ)
    float *acc;
    double *acc64;                   // order-insensitive mode: float64 accumulation rows (same layout)
    const uint64_t *masks;           // BinBuf::block_masks (rast_blockmask.hip)
    size_t mask_plane;
    // the fused occlusion chain walked back to front in the same pass (OCC): upstream gradient of the occlusion image [3,H,W], the
    // entries' camera-facing flags, what the forward left of the chain per pixel, and the per-Gaussian sums (zeroed by the caller)
    const float *dL_docc_img;
    int occ_planes;                  // 3: [3,H,W]; 1: the three channels' gradients already summed, [1,H,W]
    const float *front;
    const float *final_To;
    const uint32_t *n_contrib_o;
    float *g_values;
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

// Every load issued at once (waiting for n_contrib first and only then asking for the eight gradient planes is two round trips to
// memory in a row at the start of every wavefront).  What a pixel without contributors holds in the
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
    const float gn = a.normal_scale ? *a.normal_scale : 1.f;
    c.dN0 = on ? gs * (gn * n0) : 0.f; c.dN1 = on ? gs * (gn * n1) : 0.f; c.dN2 = on ? gs * (gn * n2) : 0.f;
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

// ================================================================================================
// the arithmetic: lane = list entry, loop over the pixels of the wavefront's block
// ================================================================================================
// Rounds 1-2 blended backward with lane = (pixel, slot) like the forward kernel; per step of 4 entries x 16 pixels that form pays
// for things that are not arithmetic of the gradient: five LDS reads of the entries' records per lane, sixteen quad broadcasts for
// the two recurrences, and a 36-instruction transpose-reduce that turns 64 per-pair terms into per-entry sums (~160 vector + ~28
// scalar instructions per 64 pairs; the form and a workgroup-staged lane = entry form were retired in round 4, profiles/README.md).
// Turned around -- lane = entry, the wavefront walks the (up to 16) pixels of its 4x4 block one after the other -- none of that
// is needed:
//   * a lane keeps its entry's record in registers for the whole batch of 64 surviving entries (read from LDS once);
//   * the pixel's constants AND its running state (T, P; the occlusion chain's T) are wave-uniform broadcast LDS reads; lane 63 writes
//     the state back at the end of the pixel's turn (round 4; it lived in lane `pixel` of two registers before: five register-to-scalar
//     moves, a compare and two selects per step);
//   * the two back-to-front recurrences of a pixel (T = T / (1 - alpha), backward.cu:683; "colour behind me", :701-:766, folded
//     into the scalar P' = alpha u + (1 - alpha) P as above) become ONE inclusive scan over the lanes of the affine maps
//     P -> (1 - alpha_i) P + alpha_i u_i (12 DPP-fused instructions): its multiplier IS the product of the (1 - alpha) behind
//     and at the entry, i.e. T_in / T in front of the entry;
//   * the 13 sums of an entry over the pixels accumulate in the lane's own registers: no cross-lane reduction at all;
//   * pixels that have nothing in the batch (their deepest contributor lies in front of it, or they are outside the image) are
//     skipped by a scalar loop over the set bits of a ballot -- in the other form their lanes idle.
// Survivors of the conservative block test are collected across chunk boundaries (a carried, partly filled batch keeps its
// records in registers) so that batches are full except the last one of a block.  A batch's sums leave through a transposition
// in LDS as 16 atomic wave-instructions of four whole rows each (rows of 64 different Gaussians straight from the lanes would be 64
// separate 4-byte requests per instruction; rounds 3-4: 13 instructions over the consecutive floats of the 52-byte row segments,
// 1.2 requests per row).
// About 100 vector instructions per (pixel, 64 entries) against 160 + 28; T in front of an entry is T_in / (product) with one
// reciprocal instead of a chain of divisions -- inside the gradient tolerance like the shared reciprocal of the other form
// (the forward's T, n_contrib and final_T are not touched by any of this).
#ifdef SOAR_BWD_HIST
__device__ unsigned long long g_bwd_hist[16];
#endif
constexpr int DPP_WAVE_SHR1 = 0x138;
// REGION = the 4 x 4 blocks a wavefront takes: 1, 2 (side by side: 8 x 4 pixels) or the 4 of a quad.  A pair leaves 40 % fewer
// accumulation rows and record gathers for 20 % more pixel steps: at C3 242 -> 219 us per 4-frame launch (a quad: 300; on small
// images -- C2, 250 tiles with work per frame -- halving the number of wavefronts costs more than it saves: 82 -> 113 us).  NOT the
// default: with pairs the worst element of dL_drotations of the C3-size surfel scene lands at 1.07e-4 / 1.16e-4 of the reference's
// in six runs of eight (7.0e-5 in the others; single blocks: 4.9-6.6e-5 in sixteen) -- a pixel's entries are spread over 1.4 x as
// many batches, and although every step of that is within an ulp the strict 1e-4 bar of tests/test_reference_build_gpu.py has no
// room for it.  -DSOAR_BWD_REGION=2 builds it.
#ifndef SOAR_BWD_REGION
#define SOAR_BWD_REGION 1
#endif
#ifndef SOAR_BWD_PACKED
#define SOAR_BWD_PACKED 1
#endif
static_assert(!(SOAR_BWD_PACKED && SOAR_BWD_DIV), "the packed pixel steps only carry the default division form");
#ifndef SOAR_BWD_UNIT_GROUPS
#define SOAR_BWD_UNIT_GROUPS 4     // (mask words compacted at a time: 1.3 KB of list; with the 32 pixels' rows 7.7 KB of LDS per wavefront: 21 per CU, the registers allow 20)
#endif
constexpr int UNIT_GROUPS = SOAR_BWD_UNIT_GROUPS;    // mask words (64 list positions each) the block walk compacts at a time

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
// Segmented forms (round 5): the scan over SEGMENTS of 16 or 32 lanes -- rows of 16 never look across their boundary in the first four
// steps, the fifth step joins the rows of a half, the sixth the halves; leaving the last one / two steps out is the whole difference.
template <int S>
__device__ __forceinline__ PairProducts affine_scan_with_products_seg(float &m, float &b, float dx, float dy, float A, float B, float C)
{
    static_assert(S == 16 || S == 32, "segments of rows");
    PairProducts o;
    if constexpr (S == 16) {
        asm volatile(
            "v_mul_f32 %2, %8, %8\n\t"
            "v_mul_f32 %3, %8, %9\n\t"
            "v_rcp_f32 %7, %1\n\t"
            "v_fmac_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32 %4, %9, %9\n\t"
            "v_fmac_f32_dpp %0, %0, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32 %5, %10, %8\n\t"
            "v_fmac_f32_dpp %0, %0, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32 %6, %12, %9\n\t"
            "v_fmac_f32_dpp %0, %0, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32 %5, %11, %9\n\t"
            "v_fmac_f32 %6, %11, %8"
            : "+v"(b), "+v"(m), "=&v"(o.dxdx), "=&v"(o.dxdy), "=&v"(o.dydy), "=&v"(o.gA), "=&v"(o.gC), "=&v"(o.r_om)
            : "v"(dx), "v"(dy), "v"(A), "v"(B), "v"(C));
    } else {
        asm volatile(
            "v_mul_f32 %2, %8, %8\n\t"
            "v_mul_f32 %3, %8, %9\n\t"
            "v_rcp_f32 %7, %1\n\t"
            "v_fmac_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32 %4, %9, %9\n\t"
            "v_fmac_f32_dpp %0, %0, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32 %5, %10, %8\n\t"
            "v_fmac_f32_dpp %0, %0, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32 %6, %12, %9\n\t"
            "v_fmac_f32_dpp %0, %0, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32 %5, %11, %9\n\t"
            "v_fmac_f32_dpp %0, %0, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
            "v_mul_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
            "v_fmac_f32 %6, %11, %8"
            : "+v"(b), "+v"(m), "=&v"(o.dxdx), "=&v"(o.dxdy), "=&v"(o.dydy), "=&v"(o.gA), "=&v"(o.gC), "=&v"(o.r_om)
            : "v"(dx), "v"(dy), "v"(A), "v"(B), "v"(C));
    }
    return o;
}
template <int S>
__device__ __forceinline__ PairProducts affine_scan_with_products_occ_seg(float &m, float &b, float &mo, float dx, float dy, float A, float B, float C)
{
    static_assert(S == 16 || S == 32, "segments of rows");
    PairProducts o;
    if constexpr (S == 16) {
        asm volatile(
            "v_mul_f32 %2, %9, %9\n\t"
            "v_mul_f32 %3, %9, %10\n\t"
            "v_rcp_f32 %7, %1\n\t"
            "v_fmac_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32_dpp %8, %8, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32 %4, %10, %10\n\t"
            "v_fmac_f32_dpp %0, %0, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32_dpp %8, %8, %8 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32 %5, %11, %9\n\t"
            "v_fmac_f32_dpp %0, %0, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32_dpp %8, %8, %8 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32 %6, %13, %10\n\t"
            "v_fmac_f32_dpp %0, %0, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32_dpp %8, %8, %8 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32 %5, %12, %10\n\t"
            "v_fmac_f32 %6, %12, %9"
            : "+v"(b), "+v"(m), "=&v"(o.dxdx), "=&v"(o.dxdy), "=&v"(o.dydy), "=&v"(o.gA), "=&v"(o.gC), "=&v"(o.r_om), "+v"(mo)
            : "v"(dx), "v"(dy), "v"(A), "v"(B), "v"(C));
    } else {
        asm volatile(
            "v_mul_f32 %2, %9, %9\n\t"
            "v_mul_f32 %3, %9, %10\n\t"
            "v_rcp_f32 %7, %1\n\t"
            "v_fmac_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32_dpp %8, %8, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32 %4, %10, %10\n\t"
            "v_fmac_f32_dpp %0, %0, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32_dpp %8, %8, %8 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32 %5, %11, %9\n\t"
            "v_fmac_f32_dpp %0, %0, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32_dpp %8, %8, %8 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32 %6, %13, %10\n\t"
            "v_fmac_f32_dpp %0, %0, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32_dpp %8, %8, %8 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32 %5, %12, %10\n\t"
            "v_fmac_f32_dpp %0, %0, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
            "v_mul_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
            "v_mul_f32_dpp %8, %8, %8 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
            "v_fmac_f32 %6, %12, %9"
            : "+v"(b), "+v"(m), "=&v"(o.dxdx), "=&v"(o.dxdy), "=&v"(o.dydy), "=&v"(o.gA), "=&v"(o.gC), "=&v"(o.r_om), "+v"(mo)
            : "v"(dx), "v"(dy), "v"(A), "v"(B), "v"(C));
    }
    return o;
}
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
// ... and with the running product of a second chain's factors (the occlusion chain: mo) carried through the same six steps
__device__ __forceinline__ PairProducts affine_scan_with_products_occ(float &m, float &b, float &mo, float dx, float dy, float A, float B, float C)
{
    PairProducts o;
    asm volatile(
        "v_mul_f32 %2, %9, %9\n\t"
        "v_mul_f32 %3, %9, %10\n\t"
        "v_rcp_f32 %7, %1\n\t"
        "v_fmac_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %8, %8, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32 %4, %10, %10\n\t"
        "v_fmac_f32_dpp %0, %0, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %8, %8, %8 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32 %5, %11, %9\n\t"
        "v_fmac_f32_dpp %0, %0, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %8, %8, %8 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32 %6, %13, %10\n\t"
        "v_fmac_f32_dpp %0, %0, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %8, %8, %8 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32 %5, %12, %10\n\t"
        "v_fmac_f32_dpp %0, %0, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_mul_f32_dpp %8, %8, %8 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_fmac_f32 %6, %12, %9\n\t"
        "v_fmac_f32_dpp %0, %0, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "v_mul_f32_dpp %8, %8, %8 row_bcast:31 row_mask:0xc bank_mask:0xf"
        : "+v"(b), "+v"(m), "=&v"(o.dxdx), "=&v"(o.dxdy), "=&v"(o.dydy), "=&v"(o.gA), "=&v"(o.gC), "=&v"(o.r_om), "+v"(mo)
        : "v"(dx), "v"(dy), "v"(A), "v"(B), "v"(C));
    return o;
}
// The transmittance a pixel carries from one batch to the next: T_in / (the batch's product).  The lanes' own T_mine is T_in x v_rcp_f32
// (1 ulp, used once); what is CARRIED takes one residual step -- the quotient to the last bit or next to it -- so that the error does
// not grow with the number of batches a pixel's list is cut into (round 5: pairs of blocks cut it into 1.4 x as many).
#ifndef SOAR_BWD_STATE_REFINE
#define SOAR_BWD_STATE_REFINE 1
#endif
#if SOAR_BWD_STATE_REFINE
#define SOAR_T_STATE(T_in_, m_, q0_) __builtin_fmaf(__builtin_fmaf(-(m_), (q0_), (T_in_)), __builtin_amdgcn_rcpf(m_), (q0_))
#else
#define SOAR_T_STATE(T_in_, m_, q0_) (q0_)
#endif
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

// ================================================================================================
// the kernel: one independent wavefront per 4x4 block, lane = list entry
// ================================================================================================
// The entry-lane arithmetic above without a workgroup around it.  What the profile of the workgroup-staged form of round 3 showed:
// a wavefront of an average tile lives 30 us -- 10 us of them in four dependent round trips to memory
// before its first useful instruction (tile -> pixels -> list ids -> records, the last two behind a workgroup barrier) --, executes
// ~2000 vector instructions, and waits at two barriers per chunk for the slowest of its three siblings.  Here:
//   * the block masks (rast_blockmask.hip) already say which entries of the list concern this block: no staging of the tile's
//     whole list, no test, no barrier; a wavefront that has nothing to do leaves without waiting for its neighbours;
//   * the mask words of the block (64 groups = 4096 list positions per load) are asked for together with the pixel's planes; the
//     set bits are compacted, deepest first, into batches of 64 list positions; list ids are gathered straight into the lanes'
//     registers one batch ahead, the records at the top of the batch's own turn (round 5: holding the next batch's records across the
//     pixel loop cost sixteen registers and with them 36 bytes of scratch per lane -- more than the hidden round trip was worth);
//   * the sums of batch n leave through LDS (16 atomic wave-instructions of four whole rows) at the start of batch n + 1, BEHIND that
//     batch's gather: the wait for the gather of batch n + 2 then finds the atomics in front of it a whole pixel loop old.
template <bool WIDE, bool OCC, int REGION>
__device__ __forceinline__ void backward_block(const BwdArgs &a, const int rank, const int blk, float4 (*pixc)[5], uint32_t *ring,
                                               uint32_t *list, float *xpose, uint32_t *xgid)
{
    const int lane = threadIdx.x & 63;
    constexpr int NC = OCC ? 14 : 13, XS = OCC ? 15 : 13;       // components of a row that leave; stride of a row in `xpose`
    constexpr int NPIX = 16 * REGION;
    static_assert(REGION == 1 || REGION == 2 || REGION == 4, "one block, a pair side by side, or the four of a quad");
    // tile and list range in ONE load (ImageBuf::order_rec; ranks below n_work are tiles with work)
    const uint4 orec = a.order_rec[rank];
    const int tile = (int)orec.x;
    const uint2 range = make_uint2(orec.y, orec.z);
    if (range.x == range.y) return;
    const int tx = tile % a.gx, ty = tile / a.gx;
    // (REGION = 2: the wavefront takes two blocks side by side -- 8 x 4 pixels --, REGION = 4: the four of a quad; `blk` names the first)
    const int bx0 = tx * TILE + ((blk >> 2) & 1) * 8 + (blk & 1) * 4, by0 = ty * TILE + (blk >> 3) * 8 + ((blk >> 1) & 1) * 4;
    const int p_own = lane & (NPIX - 1);                 // lanes 0..NPIX-1 own the region's pixels (the others hold copies)
    const int px = bx0 + ((p_own >> 4) & 1) * 4 + (p_own & 3), py = by0 + (p_own >> 5) * 4 + ((p_own >> 2) & 3);
    const bool inside = px < a.W && py < a.H;

    // The block's mask words: lane j of the window holds the word of group wb + j.  A list of up to 64 groups (4096 entries: all
    // but a few tiles of a frame) fits one window anchored at its first group -- asked for HERE, together with the pixels' planes,
    // one round trip earlier than a window that hangs from the deepest contributor (which is only known once n_contrib is back)
    const uint32_t g_lo = range.x >> 6;
    const bool early = ((range.y - 1u) >> 6) - g_lo < (uint32_t)WAVE;
    uint32_t wb = g_lo, g_end = (range.y - 1u) >> 6;     // window base; last group the window may read
    uint32_t w_lo = 0u, w_hi = 0u;
    auto load_window = [&]() {
        const uint32_t g = wb + (uint32_t)lane;
        unsigned long long word = 0ull;
        if (g <= g_end) {
#pragma unroll
            for (int b = 0; b < REGION; b++) word |= a.masks[(size_t)(blk + b) * a.mask_plane + g];
        }
        w_lo = (uint32_t)word; w_hi = (uint32_t)(word >> 32);
    };
    if (early) load_window();

    PixelConsts c;
    PixelState st;
    // OCC: the occlusion chain of the pixel -- what is left of its transmittance, its last contributor, the upstream gradient of its
    // three (equal) channels; asked for together with the rest
    float To_final = 0.f, G_occ = 0.f;
    uint32_t last_o = 0u;
    if (OCC) {
        const size_t hw = (size_t)a.H * a.W, pix = inside ? (size_t)a.W * py + px : 0;
        const bool three = a.occ_planes != 1;
        const float g0 = a.dL_docc_img[pix], g1 = three ? a.dL_docc_img[hw + pix] : 0.f, g2 = three ? a.dL_docc_img[2 * hw + pix] : 0.f;
        const float tf = a.final_To[pix];
        const uint32_t lo = a.n_contrib_o[pix];
        last_o = inside ? lo : 0u;
        To_final = tf;
        G_occ = last_o != 0u ? (three ? (g0 + g1) + g2 : g0) : 0.f;
    }
    load_pixel_at_once(a, px, py, inside, c, st);
    const uint32_t vLast = OCC ? max(c.last, last_o) : c.last;       // how deep the pixel's walk starts (the occlusion chain skips the
    const uint32_t deepest = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_max_u32(vLast));     // back-facing entries: it may outlive the main one)
    if (deepest == 0u) return;
    set_wave_priority_by_length(deepest);
    if (lane < NPIX) {
        pixc[lane][0] = make_float4(c.fx, c.fy, c.dC0, c.dC1);
        pixc[lane][1] = make_float4(c.dC2, c.dN0, c.dN1, c.dN2);
        pixc[lane][2] = make_float4(c.dD, c.dD_ch, c.norm_depth_k + c.tail, __uint_as_float(c.last));
        // the pixel's running state: transmittance behind / blend of everything behind . upstream gradient -- read back by every lane
        // when the pixel's turn comes in a batch, rewritten by lane 63 at the end of it (four register-to-scalar moves, a compare and
        // two selects per step when it lived in the lanes' registers)
        pixc[lane][3] = make_float4(st.T, 0.f, To_final, __uint_as_float(last_o));
        if (OCC) pixc[lane][4] = make_float4(G_occ, 0.f, 0.f, 0.f);
    }
    const float two_ddelx = 2.f * c.ddelx_dx, two_ddely = 2.f * c.ddely_dy;

    // ---- the walk: the set bits of the block's mask words, deepest list position first.  UNIT_GROUPS groups (512 positions) are
    //      compacted at a time by all 64 lanes (eight lanes per word, 8 bits each) into `list`; a batch is the next 64 entries of it.
    //      (Rounds 3-4a walked the words one at a time -- two register-to-scalar moves, a ballot, two prefix counts and a dozen scalar
    //      instructions per 64 positions, of which a block keeps seven: a quarter of the launch's instructions made batches.)
    const uint32_t x0 = range.x, top = range.x + deepest;       // positions at and behind `top` are not walked (nothing was blended there
    const uint32_t g_top = (top - 1u) >> 6;                      // for this block's pixels); positions in front of x0 belong to the tile before
    g_end = g_top;
    if (!early) {                                // a long list: the window hangs from the deepest contributor's group
        wb = g_top - g_lo >= (uint32_t)WAVE ? g_top - (uint32_t)(WAVE - 1) : g_lo;
        load_window();
    }
    int g_next = (int)g_top;                     // highest group not compacted yet; the walk ends below g_lo
    int n_list = 0, l_pos = 0;                   // list[l_pos .. n_list): positions not handed out yet (wave-uniform)
    auto wave_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto compact_unit = [&]() {                  // groups g_next, g_next - 1, ... g_next - (UNIT_GROUPS - 1) (those at or above g_lo) -> list[n_list ..)
        constexpr int LPW = WAVE / UNIT_GROUPS, BITS = 64 / LPW;           // lanes per mask word, bits per lane
        const int g_hi = g_next, g_low = max(g_hi - (UNIT_GROUPS - 1), (int)g_lo);
        if ((uint32_t)g_low < wb) { wb = (uint32_t)g_hi - g_lo >= (uint32_t)WAVE ? (uint32_t)g_hi - (uint32_t)(WAVE - 1) : g_lo; load_window(); }
        // lane = (word: 0 = the highest group, quarter: highest bits first)
        const int g = g_hi - lane / LPW, q = LPW - 1 - (lane % LPW);
        const int src = (g - (int)wb) & 63;                      // the window's lane that holds the word
        const uint32_t word_lo = (uint32_t)__shfl((int)w_lo, src), word_hi = (uint32_t)__shfl((int)w_hi, src);
        const uint32_t base = ((uint32_t)g << 6) + (uint32_t)(BITS * q);   // list position of bit 0 of this lane's bits
        uint32_t bits = 0u;
        if (g >= g_low) {
            constexpr uint32_t FULL = (1u << BITS) - 1u;
            bits = ((BITS * q >= 32 ? word_hi : word_lo) >> ((BITS * q) & 31)) & FULL;
            if (base + (uint32_t)BITS > top) bits &= base >= top ? 0u : (1u << (top - base)) - 1u;
            if (base < x0) bits &= x0 - base >= (uint32_t)BITS ? 0u : (FULL << (x0 - base)) & FULL;
        }
        const int cnt = __builtin_popcount(bits);
        // inclusive sum over the lanes, lane 0 first (DPP: a lane without a source lane adds 0)
        int incl = cnt;
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, false);      // row_shr:1
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, false);      // row_shr:2
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, false);      // row_shr:4
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, false);      // row_shr:8
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xa, 0xf, false);      // row_bcast:15 -> rows 1, 3
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xc, 0xf, false);      // row_bcast:31 -> rows 2, 3
        const int total = __builtin_amdgcn_readlane(incl, WAVE - 1);
        int off = n_list + incl - cnt;
        while (__ballot(bits != 0u)) {           // highest bit first: at most BITS trips, usually two or three
            if (bits) {
                const int b = 31 - __builtin_clz(bits);
                list[off++] = base + (uint32_t)b;
                bits &= ~(1u << b);
            }
        }
        n_list = __builtin_amdgcn_readfirstlane(n_list + total);       // (wave-uniform by construction: kept in scalar registers)
        g_next = __builtin_amdgcn_readfirstlane(g_low - 1);
    };

    // next batch of up to 64 list positions, deepest first -> ring[0 .. count)
    auto assemble = [&]() -> int {
        while (n_list - l_pos < WAVE && g_next >= (int)g_lo) {
            if (l_pos > 0) {                     // what has not been handed out (fewer than 64 entries) moves to the front
                const int left = n_list - l_pos;
                uint32_t v = 0u;
                if (lane < left) v = list[l_pos + lane];
                wave_sync();
                if (lane < left) list[lane] = v;
                n_list = __builtin_amdgcn_readfirstlane(left); l_pos = 0;
            }
            wave_sync();
            compact_unit();
            wave_sync();
        }
        const int cnt = __builtin_amdgcn_readfirstlane(min(WAVE, n_list - l_pos));
        if (lane < cnt) ring[lane] = list[l_pos + lane];
        l_pos = __builtin_amdgcn_readfirstlane(l_pos + cnt);
        wave_sync();
        return cnt;
    };

    // this lane's entry of the batch being worked on (lane 0 = deepest), and the list id of its entry in the batch after it
    float ex = 0.f, ey = 0.f, eA = 0.f, eB = 0.f, eC = 0.f, eop = 0.f, edepth = 0.f, epa = 0.f, epb = 0.f;
    float er = 0.f, eg = 0.f, eb = 0.f, enx = 0.f, eny = 0.f, enz = 0.f;
    uint32_t egid = 0u, epos = 0xFFFFFFFFu;      // position relative to the start of the list; 0xFFFFFFFF: no entry in this lane
    float efront = 0.f, rfront = 0.f;            // OCC: 1 = camera-facing (an entry of the occlusion chain)
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0, r3 = r0;
    uint32_t rgid = 0u, rpos = 0xFFFFFFFFu;
    int cnt_e = 0, cnt_r = 0;

    // A batch of at most 16 (32) entries is worked on FOUR (TWO) pixels at a time: every segment of 16 (32) lanes holds a copy of the
    // batch -- lane = (pixel of the group, entry) -- and the scan runs inside the segments (SOAR_BWD_PACKED, round 5: a pixel step costs
    // the same ~90 vector instructions whether 5 or 64 of its lanes hold an entry, and 26 % of all pixel steps belong to batches of
    // at most 32 entries: the last batch of a block's list, or its only one)
    auto seg_of = [](int cnt) { return SOAR_BWD_PACKED && cnt <= 16 ? 16 : SOAR_BWD_PACKED && cnt <= 32 ? 32 : WAVE; };
    auto fetch_ids = [&](int cnt, uint32_t &gid, uint32_t &pos) {
        pos = 0xFFFFFFFFu;
        const int li = lane & (seg_of(cnt) - 1);
        if (li < cnt) {
            const uint32_t at = ring[li];
            gid = a.point_list[at];
            pos = at - x0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // the ring is refilled by the next assemble
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto gather = [&]() {                                             // records of (rgid, rpos)
        if (rpos != 0xFFFFFFFFu) {
            const float4 *src = reinterpret_cast<const float4 *>(a.rec + rgid);
            r0 = src[0]; r1 = src[1]; r2 = src[2]; r3 = src[3];
            if (OCC) rfront = a.front[rgid];
        }
    };
    int pending_seg = WAVE, pending_cnt = 0;      // how the batch whose sums wait in LDS was laid out (wave-uniform)
    // a packed batch: every segment holds partial sums of the same (at most S) entries -- added here, segment 0 first
    auto flush_packed = [&](auto s_tag) {
        constexpr int S = decltype(s_tag)::value, NS = WAVE / S, KMAX = S / 4;       // four whole rows per instruction, like the other form
        const int r4 = lane >> 4, q = lane & 15;
        const bool has_q = q < NC;
        const float *xp = xpose + (has_q ? XS * r4 + q : 0);
        const uint32_t *xg = xgid + r4;
        uint32_t g[KMAX];
        float v[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; k++) {
            g[k] = 0xFFFFFFFFu; v[k] = 0.f;
            if (4 * k < pending_cnt) {                       // (wave-uniform: rows 4 k .. 4 k + 3 hold entries)
                uint32_t gg[NS];
                float vv[NS];
#pragma unroll
                for (int t = 0; t < NS; t++) { gg[t] = xg[4 * k + t * S]; vv[t] = xp[4 * XS * k + t * S * XS]; }
                g[k] = gg[0]; v[k] = vv[0];
#pragma unroll
                for (int t = 1; t < NS; t++) { g[k] = min(g[k], gg[t]); v[k] += vv[t]; }      // (an entry dead in a segment: id 0xFFFFFFFF, sums 0)
            }
        }
#pragma unroll
        for (int k = 0; k < KMAX; k++) {
            if (has_q && g[k] != 0xFFFFFFFFu) {
                if (WIDE) atomicAdd(a.acc64 + (size_t)g[k] * ACC_STRIDE + q, (double)v[k]);
#ifdef SOAR_EXP_NO_ATOMICS
                else if (g[k] == 0xFFFFFFFEu) a.acc[q] = v[k];
#else
                else atomicAdd(a.acc + (size_t)g[k] * ACC_STRIDE + q, v[k]);
#endif
            }
        }
    };
    auto flush_atomics = [&]() {
        if (pending_seg == 16) { flush_packed(std::integral_constant<int, 16>()); return; }
        if (pending_seg == 32) { flush_packed(std::integral_constant<int, 32>()); return; }
        // Every instruction adds WHOLE rows: lane = (one of the four rows of the instruction, component), 13 of 16 lanes at work,
        // 16 instructions for the 64 entries.  (Until late in round 5: 13 instructions over the 832 floats as they lie -- a row that
        // straddled two instructions was two requests at the L2, 1.2 per row, and the device-wide float atomics of this path are
        // not combined there: the L2 counters show every one of them forwarded to the memory side (TCC_EA0_ATOMIC = TCC_ATOMIC =
        // TCC_PROBE = 3.9 M per 4-frame launch), which is what the launch waits for.  The lane's (row, component) also is two
        // lane constants now instead of 13 (entry, component) pairs.)
        static_assert(ACC_STRIDE == 16 && WAVE == 64, "four rows of 16 floats per instruction");
        const int r4 = lane >> 4, q = lane & 15;
        const bool has_q = q < NC;
        const float *xp = xpose + (has_q ? XS * r4 + q : 0);
        const uint32_t *xg = xgid + r4;
        uint32_t g[16];
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            g[k] = xg[4 * k];
            v[k] = xp[4 * XS * k];
        }
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (has_q && g[k] != 0xFFFFFFFFu) {
                if (WIDE) atomicAdd(a.acc64 + (size_t)g[k] * ACC_STRIDE + q, (double)v[k]);
#ifdef SOAR_EXP_NO_ATOMICS
                else if (g[k] == 0xFFFFFFFEu) a.acc[q] = v[k];       // (development, results wrong by construction: what the launch costs without them)
#else
                else atomicAdd(a.acc + (size_t)g[k] * ACC_STRIDE + q, v[k]);
#endif
            }
        }
    };

    // fill the pipeline: batch 0's ids in (r)
    cnt_r = assemble();
    fetch_ids(cnt_r, rgid, rpos);
    bool pending = false;
    for (;;) {
        gather();
        // (r) -> (e): the batch to work on
        ex = r0.x; ey = r0.y; eA = r0.z; eB = r0.w; eC = r1.x; eop = r1.y; edepth = r1.z; epa = r1.w;
        epb = r2.x; er = r2.y; eg = r2.z; eb = r2.w; enx = r3.x; eny = r3.y; enz = r3.z;
        egid = rgid; epos = rpos; cnt_e = cnt_r;
        if (OCC) efront = rfront;
        if (cnt_e == 0) break;
        // the sums of the batch before leave now
        if (pending) flush_atomics();
        // the ids of the batch behind this one are asked for now; ITS records only at the top of its own turn (rounds 3-4 kept them in
        // sixteen registers across this batch's pixel loop: 137 registers wanted, 128 allowed at four waves per SIMD, 36 bytes of
        // scratch per lane.  Without them the kernel needs 120 and no scratch: 297-305 -> 290 us per 4-frame launch, round 5)
        cnt_r = assemble();
        fetch_ids(cnt_r, rgid, rpos);

        // ---- one batch: cnt_e entries x the pixels that reach it
        const uint32_t nearest = (uint32_t)__builtin_amdgcn_readlane((int)epos, cnt_e - 1);
        unsigned long long act = __ballot(vLast > nearest) & (NPIX == 64 ? ~0ull : (1ull << (NPIX & 63)) - 1ull);
#ifdef SOAR_BWD_HIST
        if (lane == 0) {     // development (scripts/bwd_hist.py): pixel steps / batches by the batch's entry count
            const int bk = cnt_e <= 8 ? 0 : cnt_e <= 16 ? 1 : cnt_e <= 32 ? 2 : cnt_e < 64 ? 3 : 4;
            atomicAdd(&g_bwd_hist[bk], (unsigned long long)__builtin_popcountll(act));
            atomicAdd(&g_bwd_hist[8 + bk], 1ull);
        }
#endif
        float acc[13];
#pragma unroll
        for (int q = 0; q < 13; q++) acc[q] = 0.f;
        float sdD = 0.f;
        float any_live = 0.f;                    // > 0: some pair of this lane's entry was live
        float acc_o = 0.f;                       // OCC: sum over the pixels of weight x upstream gradient of this lane's entry
        const float oh = -0.5f * eop;
        const int seg = seg_of(cnt_e);           // (wave-uniform)
        // ---- packed: NS = 64 / S pixels per step, lane = (pixel s = lane / S of the group, entry lane % S)
        auto packed_pixels = [&](auto s_tag) {
            constexpr int S = decltype(s_tag)::value, NS = WAVE / S;
            const int sub = lane / S;
            const bool seg_first = (lane & (S - 1)) == 0, seg_last = (lane & (S - 1)) == S - 1;
            while (act) {
                int pp[NS], np = 1;
                pp[0] = (int)__builtin_ctzll(act);
                act &= act - 1ull;
#pragma unroll
                for (int k = 1; k < NS; k++) {
                    pp[k] = pp[0];
                    if (act) { pp[k] = (int)__builtin_ctzll(act); act &= act - 1ull; np = k + 1; }
                }
                int p = pp[0];
#pragma unroll
                for (int k = 1; k < NS; k++) p = sub == k ? pp[k] : p;
                const bool has_pixel = sub < np;             // (a group may end short: its idle segments compute on pixel pp[0] with every pair dead)
                const float4 c0 = pixc[p][0], c1 = pixc[p][1], c2 = pixc[p][2];
                const float4 tp = pixc[p][3];
                const uint32_t last_p = has_pixel ? __float_as_uint(c2.w) : 0u;
                const float T_in = tp.x, P_in = tp.y;
                const float dx = ex - c0.x, dy = ey - c0.y;
                const float power = falloff_power(eA, eB, eC, dx, dy);
                const float Gx = exp_nonpositive(power);
                const float alpha = fminf(0.99f, eop * Gx);
                float a_eff = (power > 0.0f) ? 0.f : alpha;
                a_eff = (alpha < 1.0f / 255.0f) ? 0.f : a_eff;
                float a_o = 0.f, m_o = 1.f;
                if (OCC) {
                    const uint32_t last_occ = has_pixel ? __float_as_uint(tp.w) : 0u;
                    a_o = (epos < last_occ) ? a_eff * efront : 0.f;
                    m_o = 1.f - a_o;
                }
                a_eff = (epos < last_p) ? a_eff : 0.f;
                const bool live = a_eff != 0.f;
                const float G = live ? Gx : 0.f;
                const float d_cur = edepth - (dx * epa + dy * epb);
                const float u = er * c0.z + eg * c0.w + eb * c1.x + enx * c1.y + eny * c1.z + enz * c1.w + d_cur * c2.y;
                float m = 1.f - a_eff, b = a_eff * u;
                const PairProducts pp_ = OCC ? affine_scan_with_products_occ_seg<S>(m, b, m_o, dx, dy, eA, eB, eC)
                                             : affine_scan_with_products_seg<S>(m, b, dx, dy, eA, eB, eC);
                const float P_front = __builtin_fmaf(m, P_in, b);
                const float T_mine = T_in * __builtin_amdgcn_rcpf(m);
                const float r_om = pp_.r_om;
                float P_mine = dpp_or<DPP_WAVE_SHR1, 0xf>(P_in, P_front);
                P_mine = seg_first ? P_in : P_mine;          // (the lane in front belongs to another pixel)
                const float wgt = a_eff * T_mine;
                acc[6] = __builtin_fmaf(wgt, c0.z, acc[6]); acc[7] = __builtin_fmaf(wgt, c0.w, acc[7]);
                acc[8] = __builtin_fmaf(wgt, c1.x, acc[8]);
                acc[9] = __builtin_fmaf(wgt, c1.y, acc[9]); acc[10] = __builtin_fmaf(wgt, c1.z, acc[10]);
                acc[11] = __builtin_fmaf(wgt, c1.w, acc[11]);
                acc[12] = __builtin_fmaf(wgt, c2.y, acc[12]);
                const float dL_dalpha = __builtin_fmaf(u - P_mine, T_mine, c2.z * r_om);
                const float dL_ddist = dL_dalpha * (oh * G);
                acc[0] = __builtin_fmaf(dL_ddist, pp_.gA, acc[0]);
                acc[1] = __builtin_fmaf(dL_ddist, pp_.gC, acc[1]);
                acc[2] = __builtin_fmaf(dL_ddist, pp_.dxdx, acc[2]);
                acc[3] = __builtin_fmaf(dL_ddist, pp_.dxdy, acc[3]);
                acc[4] = __builtin_fmaf(dL_ddist, pp_.dydy, acc[4]);
                acc[5] = __builtin_fmaf(G, dL_dalpha, acc[5]);
                sdD += live ? c2.x : 0.f;
                any_live += a_eff;
                if (OCC) {
                    const float To_mine = tp.z * __builtin_amdgcn_rcpf(m_o);
                    acc_o = __builtin_fmaf(a_o * To_mine, pixc[p][4].x, acc_o);
                    if (seg_last && has_pixel) *reinterpret_cast<float4 *>(&pixc[p][3]) = make_float4(SOAR_T_STATE(T_in, m, T_mine), P_front, SOAR_T_STATE(tp.z, m_o, To_mine), tp.w);
                } else {
                    if (seg_last && has_pixel) *reinterpret_cast<float2 *>(&pixc[p][3]) = make_float2(SOAR_T_STATE(T_in, m, T_mine), P_front);
                }
            }
        };
        if (seg == 16) packed_pixels(std::integral_constant<int, 16>());
        else if (seg == 32) packed_pixels(std::integral_constant<int, 32>());
        while (act) {
            const int p = (int)__builtin_ctzll(act);
            act &= act - 1ull;
            const float4 c0 = pixc[p][0], c1 = pixc[p][1], c2 = pixc[p][2];
            const float4 tp = pixc[p][3];
            const uint32_t last_p = __float_as_uint(c2.w);
            const float T_in = tp.x, P_in = tp.y;
            const float dx = ex - c0.x, dy = ey - c0.y;
            const float power = falloff_power(eA, eB, eC, dx, dy);
            const float Gx = exp_nonpositive(power);
            const float alpha = fminf(0.99f, eop * Gx);
            // the skip rules (:653-680) as three selects on the number, not one predicate: combining the comparisons first costs
            // scalar mask instructions
            float a_eff = (power > 0.0f) ? 0.f : alpha;
            a_eff = (alpha < 1.0f / 255.0f) ? 0.f : a_eff;
            // OCC: the occlusion chain sees the camera-facing entries in front of ITS last contributor
            float a_o = 0.f, m_o = 1.f;
            if (OCC) {
                a_o = (epos < __float_as_uint(tp.w)) ? a_eff * efront : 0.f;
                m_o = 1.f - a_o;
            }
            a_eff = (epos < last_p) ? a_eff : 0.f;
            const bool live = a_eff != 0.f;
            const float G = live ? Gx : 0.f;
            const float d_cur = edepth - (dx * epa + dy * epb);
            const float u = er * c0.z + eg * c0.w + eb * c1.x + enx * c1.y + eny * c1.z + enz * c1.w + d_cur * c2.y;
            float m = 1.f - a_eff, b = a_eff * u;
#if SOAR_BWD_DIV
            const float om = m;
#endif
            const PairProducts pp = OCC ? affine_scan_with_products_occ(m, b, m_o, dx, dy, eA, eB, eC) : affine_scan_with_products(m, b, dx, dy, eA, eB, eC);
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
            any_live += a_eff;                                                           // (a_eff >= 0: > 0 iff some pair was live)
            if (OCC) {
                // transmittance of the occlusion chain in front of my entry: T_in / (the factors behind and at it), like T_mine
                const float To_mine = tp.z * __builtin_amdgcn_rcpf(m_o);
                acc_o = __builtin_fmaf(a_o * To_mine, pixc[p][4].x, acc_o);
                if (lane == 63) *reinterpret_cast<float4 *>(&pixc[p][3]) = make_float4(SOAR_T_STATE(T_in, m, T_mine), P_front, SOAR_T_STATE(tp.z, m_o, To_mine), tp.w);
            } else {
                if (lane == 63) *reinterpret_cast<float2 *>(&pixc[p][3]) = make_float2(SOAR_T_STATE(T_in, m, T_mine), P_front);
            }
        }
        // the batch's sums -> LDS, [entry][13]; they leave at the start of the next step
        acc[0] = acc[0] * two_ddelx - sdD * epa;
        acc[1] = acc[1] * two_ddely - sdD * epb;
        acc[9] *= 10.f; acc[10] *= 10.f; acc[11] *= 10.f;
#pragma unroll
        for (int q = 0; q < 13; q++) xpose[lane * XS + q] = acc[q];
        if (OCC) xpose[lane * XS + 13] = acc_o;          // (slot 13 of the Gaussian's row: leaves with the same request as the others)
        xgid[lane] = (any_live != 0.f || (OCC && acc_o != 0.f)) ? egid : 0xFFFFFFFFu;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        pending = true;
        pending_seg = seg;
        pending_cnt = cnt_e;
    }
    if (pending) flush_atomics();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");             // the next tile of this wavefront reuses the LDS arrays
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#ifndef SOAR_BWD_BLK_WPE
#define SOAR_BWD_BLK_WPE 5     // (88 / 96 VGPRs since the sums' hand-over works out its indices in place: round 5)
#endif
// one wavefront per workgroup: a finished block frees its slot at once.  Grid = 16 x ranks; the 16 blocks of a tile are
// consecutive workgroups of one XCD (its L2 holds the tile's records), ranks dealt round-robin to the XCDs.
template <bool WIDE, bool OCC, int REGION>
__device__ __forceinline__ void backward_blocks(const BwdArgs &a, int bx)
{
    constexpr int NPIX = 16 * REGION;
    __shared__ float4 pixc[NPIX][5];                       // per pixel: {fx, fy, dC0, dC1 | dC2, dN0, dN1, dN2 | dD, dD_ch, tail terms, last |
                                                         //             T, P, T_occ, last_occ | upstream gradient of the occlusion image, -, -, -}
    __shared__ uint32_t ring[WAVE];
    __shared__ uint32_t list[UNIT_GROUPS * WAVE + WAVE]; // compacted list positions of up to UNIT_GROUPS mask words + a batch's worth carried over
    __shared__ float xpose[WAVE * (OCC ? 15 : 13)];      // a batch's sums, [entry][13] ([entry][15] with the occlusion value's gradient as
    __shared__ uint32_t xgid[WAVE];                      // the 14th: an odd stride keeps the lanes' writes on different banks)
    const int xcd = bx & 7, kth = bx >> 3;
    constexpr int PER_TILE = 16 / REGION;                    // wavefronts per tile
    const int rank0 = (kth / PER_TILE) * 8 + xcd, blk = (kth % PER_TILE) * REGION;
    const int stride = (int)(gridDim.x / PER_TILE);          // ranks per pass of the grid (a multiple of 8)
    const int n_work = (int)a.tile_order[(a.ntiles + 7) / 8 * 8];
    for (int rank = rank0; rank < n_work; rank += stride) backward_block<WIDE, OCC, REGION>(a, rank, blk, pixc, ring, list, xpose, xgid);
}
template <bool WIDE, int REGION>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SOAR_BWD_BLK_WPE, 8))) render_backward_blocks_kernel(Batch<BwdArgs> batch)
{
    int frame, bx;
    batch_interleave(frame, bx);
    backward_blocks<WIDE, false, REGION>(batch.v[frame], bx);
}
// ... with the fused occlusion chain walked in the same pass (soar_rast_backward_occ): a few registers more than the 128 that four
// wavefronts per SIMD allow
template <bool WIDE, int REGION>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 8))) render_backward_blocks_occ_kernel(Batch<BwdArgs> batch)
{
    int frame, bx;
    batch_interleave(frame, bx);
    backward_blocks<WIDE, true, REGION>(batch.v[frame], bx);
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
}  // namespace

int launch_render_backward(const SoarRastParams &prm, const GeomBuf &g, const BinBuf &b, const ImageBuf &img,
                           const float *dL_dcolor, const float *dL_dnormal, const float *dL_ddepth, const float *dL_dopac,
                           const float *grad_scale, float *acc, double *acc64, bool blend, const float *dL_dout_occ, float *dL_docc,
                           const float *normal_scale, int occ_planes, hipStream_t stream)
{
    BwdArgs a;
    a.normal_scale = normal_scale;
    a.occ_planes = occ_planes;
    a.dL_docc_img = dL_dout_occ; a.front = g.front; a.final_To = img.final_To; a.n_contrib_o = img.n_contrib_o; a.g_values = dL_docc;
    const bool occ = dL_dout_occ != nullptr;
    a.grad_scale = grad_scale;
    a.W = prm.W; a.H = prm.H;
    a.gx = (prm.W + TILE - 1) / TILE; a.gy = (prm.H + TILE - 1) / TILE;
    a.ntiles = a.gx * a.gy;
    a.normalize_depth = prm.cfg_normalize_depth;
    a.ranges = img.ranges; a.tile_order = img.tile_order; a.order_rec = img.order_rec; a.point_list = b.vals_sorted; a.rec = g.rec; a.bg = prm.bg_dev;
    a.final_T = img.final_T; a.final_D = img.final_D; a.n_contrib = img.n_contrib;
    a.dL_dcolor = dL_dcolor; a.dL_dnormal = dL_dnormal; a.dL_ddepth = dL_ddepth; a.dL_dopac = dL_dopac;
    a.acc = acc; a.acc64 = acc64;
    StageTimer timer(ST_RENDER_BWD, stream);
    const int grid_ranks = blend_grid_ranks(a.ntiles);
    a.masks = b.block_masks; a.mask_plane = b.mask_plane;
    constexpr int region = SOAR_BWD_REGION;
    const dim3 grid_blocks((16 / region) * min((a.ntiles + 7) / 8 * 8, grid_ranks));
    if (acc64) {
        if (blend && occ) SOAR_LAUNCH_BATCHED((render_backward_blocks_occ_kernel<true, region>), grid_blocks, dim3(64), 0, stream, a);
        else if (blend) SOAR_LAUNCH_BATCHED((render_backward_blocks_kernel<true, region>), grid_blocks, dim3(64), 0, stream, a);
        // in a batch this launches with the last frame like the blend in front of it (it used to launch per call, i.e. for
        // the frames 0 .. n-2 BEFORE their rows had been accumulated)
        NarrowArgs na;
        na.n = (size_t)prm.P * ACC_STRIDE; na.wide = acc64; na.narrow = acc;
        SOAR_LAUNCH_BATCHED(narrow_rows_kernel, dim3((unsigned)((na.n + 255) / 256)), dim3(256), 0, stream, na);
    } else {
        if (occ) SOAR_LAUNCH_BATCHED((render_backward_blocks_occ_kernel<false, region>), grid_blocks, dim3(64), 0, stream, a);
        else SOAR_LAUNCH_BATCHED((render_backward_blocks_kernel<false, region>), grid_blocks, dim3(64), 0, stream, a);
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

extern "C" int soar_selftest_affine_scan(const float *m64_dev, const float *b64_dev, float *out192_dev, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!m64_dev || !b64_dev || !out192_dev) { soar::set_error("soar_selftest_affine_scan: NULL"); return 1; }
    hipLaunchKernelGGL(soar::selftest_affine_scan_kernel, dim3(1), dim3(64), 0, stream, m64_dev, b64_dev, out192_dev);
    SOAR_LAUNCH_OK("selftest_affine_scan", stream, 1);
    return 0;
}

#ifdef SOAR_BWD_HIST
extern "C" int soar_debug_bwd_hist(unsigned long long *out16)
{
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(soar::g_bwd_hist), 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#endif
