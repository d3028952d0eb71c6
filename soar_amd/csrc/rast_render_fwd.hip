// rast_render_fwd.hip -- per-tile front-to-back alpha blend for gfx950.
//
// Replaces renderCUDA<3> forward (DGR/cuda_rasterizer/forward.cu:390-692) and depth_differencing
// (auxiliary.h:390-397).
//
// The reference runs one thread per pixel and one entry of the tile list per step: on a tile whose list has
// thousands of entries that is thousands of dependent steps, and on MI355X (256 CUs) only ~1000 tiles of a 1080p
// frame have any work -- the launch ends as a long tail of single wavefronts.  This kernel spreads the SAME sequential
// semantics over 4x more wavefronts and 4x shorter dependent chains:
//
//  * workgroup = one 8x8 pixel quad of a 16x16 tile (4 workgroups per tile), wavefront = one 4x4 pixel block;
//  * lane = (pixel p = lane >> 2, slot s = lane & 3): the four lanes of a pixel take FOUR consecutive surviving
//    entries of the list at once.  Each lane evaluates its own alpha (falloff + exp: the long part), then the
//    running transmittance runs through the four slots as a quad scan (three fused v_mul_f32_dpp, lane k <- lane k - 1), in
//    exactly the reference's order of multiplications -- including "stop before the entry that would drop T below
//    1e-4" -- so T, n_contrib and the set of blended entries are those of the sequential loop.  Colour / normal /
//    depth sums are kept per slot and folded once per pixel at the end;
//  * the list is staged per workgroup in chunks of 256 entries: thread t gathers the 64-byte record of entry t
//    (software-pipelined one chunk ahead) into LDS as four float4 arrays; every wavefront then tests its 64-entry
//    sub-chunks against ITS 4x4 block (lanes = entries, conservative bound splat_may_touch_rect) and compacts the
//    survivors' LDS slots, in list order, into a ring of its own (ballot-prefix ranks) that the steps read;
//  * scalar instructions are kept out of the loops: one SIMD of gfx950 issues a scalar instruction only every ~4 cycles
//    whatever the number of resident wavefronts (tests/tools/issue_model).  Liveness of a pixel is a 0 / 1 number folded
//    into alpha, not a predicate (a per-lane bool that lives across the loop is a 64-bit scalar mask merged at every join);
//  * a small grid of 4 x blend_grid_ranks() workgroups (1024 ranks at 1080p, 4096 at 4K) walks the longest-list-first tile order of tile_order_block (soar_common.h) with
//    a rank stride, ranks dealt round-robin to the XCDs with the four quads of a tile on one XCD: the long tiles start
//    first, everywhere, and the ~29 000 workgroups of tiles without work (a slot, two loads and ~100 instructions each) are
//    never launched.  The
//    tiles no Gaussian touches are filled with their background values by all workgroups at the end -- unless they
//    already hold them (SoarRastParams.debug bit 2: same outputs as the previous call).
#include "soar_common.h"

#include <cstdio>
#include <cstdlib>

namespace soar {

namespace {

#ifndef SOAR_FWD_CHUNK
#define SOAR_FWD_CHUNK 256
#endif
constexpr int CHUNK = SOAR_FWD_CHUNK;   // list entries staged per workgroup iteration (one per thread)

struct FwdArgs {
    int W, H, gx, gy, ntiles;
    int normalize_depth;
    const uint2 *ranges;
    const uint32_t *tile_order;
    const uint4 *order_rec;
    const uint32_t *point_list;
    const GaussRec *rec;
    const float *bg;
    float *final_T;
    float *final_D;
    uint32_t *n_contrib;
    float *out_color, *out_normal, *out_depth, *out_opac;
    const float *occ_values;         // fused occlusion pass: per-Gaussian value blended over the camera-facing splats only
    const float *front;              // [P] 1 = camera-facing (preprocess)
    float *out_occ;                  // [3,H,W]
    uint32_t *bg_tiles;              // ImageBuf::bg_tiles
    const uint32_t *bg_state;        // ImageBuf::bg_state
    int keep_background;             // SoarRastParams.debug bit 2
    float *final_To;                 // ImageBuf::final_To / n_contrib_o (OCC): what the backward blend needs to walk the occlusion
    uint32_t *n_contrib_o;           // chain back to front (soar_rast_backward_occ)
    unsigned long long *wave_log;    // diagnostic build only (SOAR_WAVE_LOG): per wave {t_start, t_end, list length, iterations}
    // BinBuf::block_masks (rast_blockmask.hip describes the layout): phase A's survivor word of every 64 list positions a block's
    // wavefront tests is what the backward blend walks -- left behind here instead of being derived again by a pass of its own
    unsigned long long *masks;       // or null: nothing is emitted
    size_t mask_plane;
};

constexpr int DPP_QUAD_BCAST0 = 0x00, DPP_QUAD_BCAST1 = 0x55, DPP_QUAD_BCAST2 = 0xAA, DPP_QUAD_BCAST3 = 0xFF;
constexpr int DPP_QUAD_XOR1 = 0xB1, DPP_QUAD_XOR2 = 0x4E;
constexpr int DPP_QUAD_SHIFT1 = 0x90;          // quad_perm [0,0,1,2]: lane k reads lane k - 1 of its quad (lane 0 itself)

template <int CTRL>
__device__ __forceinline__ float quad_move(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ uint32_t quad_move_u(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}

// background values of a tile no Gaussian touches (what the blend's epilogue writes for T = 1, forward.cu:618-633)
template <bool OCC>
__device__ __forceinline__ void fill_tile(const FwdArgs &a, int tile, int tid, int nthreads)
{
    const int tx = tile % a.gx, ty = tile / a.gx;
    const size_t hw = (size_t)a.H * a.W;
    const float Tc = (float)(1 - 0.000001);
    const bool vec = (a.W & 3) == 0;
    for (int i = tid; i < 14 * 64; i += nthreads) {
        const int plane = i >> 6, r = i & 63;
        const int x = tx * TILE + (r & 3) * 4, y = ty * TILE + (r >> 2);
        float v = 0.f;
        float *dst = nullptr;
        switch (plane) {
        case 0: dst = a.final_T; v = Tc; break;
        case 1: dst = reinterpret_cast<float *>(a.n_contrib); v = 0.f; break;        // bits of 0u
        case 2: case 3: case 4: dst = a.out_color + (plane - 2) * hw; v = 0.f + Tc * a.bg[plane - 2]; break;
        case 5: case 6: case 7: dst = a.out_normal + (plane - 5) * hw; v = 0.f; break;
        case 8: dst = a.out_depth; v = a.normalize_depth ? 0.f / (1.f - Tc) : 0.f + Tc * 10.f; break;
        case 9: dst = a.out_opac; v = 1.f - Tc; break;
        case 10: if (a.normalize_depth) { dst = a.final_D; v = 0.f; } break;
        case 11: case 12: case 13: if (OCC) { dst = a.out_occ + (plane - 11) * hw; v = 0.f + Tc * a.bg[plane - 11]; } break;
        default: break;
        }
        if (!dst || y >= a.H || x >= a.W) continue;
        float *p = dst + (size_t)a.W * y + x;
        if (vec) {
            *reinterpret_cast<float4 *>(p) = make_float4(v, v, v, v);
        } else {
            for (int k = 0; k < 4 && x + k < a.W; k++) p[k] = v;
        }
    }
}

// Running products of one pixel's four slots, EXCLUSIVE: T is replicated in the quad, m_front is the factor of the lane in front of
// this one (1 in slot 0); lane k returns ((T m_0) m_1 ...) m_{k-1} -- the transmittance in FRONT of its entry, every product a plain
// rounded multiply in list order.  Pass k makes lane k final (lane 0 keeps T through the passes with the factor 1); lanes behind hold
// partial values until their pass.  (The inclusive form of rounds 2-4a needed a shifted copy and a select to get at this value.)
__device__ __forceinline__ float quad_scan_front(float T, float m_front)
{
    float e = T;
    e = mul_keep(quad_move<DPP_QUAD_SHIFT1>(e), m_front);
    e = mul_keep(quad_move<DPP_QUAD_SHIFT1>(e), m_front);
    e = mul_keep(quad_move<DPP_QUAD_SHIFT1>(e), m_front);
    return e;
}

// the same for two chains at once, their steps side by side (a DPP operand written by the instruction in front costs two idle cycles)
__device__ __forceinline__ void quad_scan_front2(float T, float m_front, float U, float n_front, float &e, float &f)
{
    e = T;
    f = U;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        e = mul_keep(quad_move<DPP_QUAD_SHIFT1>(e), m_front);
        f = mul_keep(quad_move<DPP_QUAD_SHIFT1>(f), n_front);
    }
}

// the element at byte offset 16 j of an LDS array of 16-byte elements (the ring holds 16 j)
__device__ __forceinline__ float4 lds_at(const float4 *arr, uint32_t j16)
{
    return *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(arr) + j16);
}

// OCC = true additionally blends, in the same walk of the list, what a second rasterization with render_front = 1 and
// colours = occ_values would produce (TS/renderer/diff_gaussian_rasterizer.py:281-291): the per-pixel sequence of
// camera-facing entries is the same subsequence of this list (preprocess differs between the two passes only by the
// back-face cull, forward.cu:262-266), so a second transmittance chain that ignores the back-facing entries reproduces
// that pass without a second preprocess / sort / blend.
template <bool LOG, bool OCC>
__device__ __forceinline__ void blend_quad(const FwdArgs &a, const int rank, const int quad)
{
    __shared__ float4 sq0[CHUNK + 1], sq1[CHUNK + 1], sq2[CHUNK + 1], sq3[CHUNK + 1];   // +1: an all-zero record
    __shared__ float4 sq4[OCC ? CHUNK + 1 : 1];                                          // {occ value, camera-facing, -, -}: the stride of the others
    __shared__ int wave_alive[2][4];
    __shared__ unsigned short todo_ring[4][CHUNK + 4];                                    // per wavefront: 16 x LDS slot (= byte offset) of a chunk's relevant entries
    __shared__ unsigned long long wmask[4][CHUNK / WAVE];                                 // per wavefront: phase A's survivor words of a chunk
    unsigned long long t_start = 0, t_ready = 0, t_blended = 0, n_iter = 0, n_useful = 0;
    if (LOG) t_start = wall_clock64();

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // tile and list range in ONE load (ImageBuf::order_rec; the ranks walked here are tiles with work)
    if (LOG && a.tile_order[rank] == 0xFFFFFFFFu) return;        // (the logging build walks the padded order)
    const uint4 orec = a.order_rec[rank];
    const int tile = (int)orec.x, seq = rank * 4 + quad;
    const int tx = tile % a.gx, ty = tile / a.gx;
    // 4x4 pixel block of this wave inside the 8x8 quad of this workgroup inside the 16x16 tile
    const int bx0 = tx * TILE + (quad & 1) * 8 + (wave & 1) * 4, by0 = ty * TILE + (quad >> 1) * 8 + (wave >> 1) * 4;
    const int pxl = lane >> 2, slot = lane & 3;
    const int px = bx0 + (pxl & 3), py = by0 + (pxl >> 2);
    const bool inside = px < a.W && py < a.H;
    const float not_first = slot == 0 ? 0.f : 1.f;
    float fx = (float)px, fy = (float)py;
    // (opaque to the compiler: it would rather convert the integers again in every step of the blend loop than hold two registers --
    // a conversion costs a full-rate multiply-add twice over, profiles/r04d_valu_issue_rate.txt)
    asm volatile("" : "+v"(fx), "+v"(fy));

    const uint2 range = make_uint2(orec.y, orec.z);
    set_wave_priority_by_length(range.y - range.x);

    float T = 1.0f;                                  // replicated in the four lanes of a pixel
    float C0 = 0.f, C1 = 0.f, C2 = 0.f, N0 = 0.f, N1 = 0.f, N2 = 0.f, D = 0.f;   // per-slot partial sums
    uint32_t last_contributor = 0;                   // per-slot, folded with max at the end
    uint32_t last_contributor_o = 0;                 // the occlusion chain's
    // 1 while the pixel blends, 0 once it has stopped (replicated in its four lanes).  Kept as a number, not a predicate:
    // a per-lane bool that lives across the loop becomes a 64-bit mask in scalar registers, merged with three scalar
    // instructions at every join -- and one SIMD issues a scalar instruction only every ~4 cycles, half the vector rate
    float alive = inside ? 1.f : 0.f;
    float T_o = 1.0f, Co = 0.f;                      // occlusion pass: transmittance (replicated), per-slot sum
    float alive_o = (inside && OCC) ? 1.f : 0.f;
    bool wave_done = (__ballot(alive != 0.f) == 0ull);

    if (tid == 0) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        sq0[CHUNK] = z; sq1[CHUNK] = z; sq2[CHUNK] = z; sq3[CHUNK] = z;
        if (OCC) sq4[CHUNK] = z;
    }

    // The list ids travel one chunk ahead; a chunk's RECORDS are gathered at the top of its iteration and staged at once.  (Rounds 1-4
    // kept the next chunk's records in 18 registers across the blend of this one -- the gather's round trip off the critical path of
    // the long tiles.  Those registers were the difference between 95 and 71: at the 80 that six waves per SIMD allow the kernel
    // spilled 24 bytes per lane, and without them it runs SEVEN waves per SIMD with no scratch at all -- the other wavefronts cover the
    // gather better than the prefetch did: 244 -> 221 us per 4-frame launch, round 5.)
    uint32_t id_next = 0;                            // list id of this thread's entry in the next chunk
    if (range.x + tid < range.y) id_next = a.point_list[range.x + tid];
    int parity = 0;
    // Phase A's survivor word of every 64 list positions this wavefront tests IS the block mask the backward blend walks
    // (BinBuf::block_masks: bit l of word g of plane `block` = list position 64 g + l; a superset of what the backward needs: an
    // entry dropped here touches no pixel that still blends).  The words of a chunk wait in LDS and leave at the top of the NEXT
    // iteration -- behind the wait for that chunk's records, in front of the gathers issued there: memory operations complete in
    // order, and an atomic issued right in front of a wait for older loads is waited for as well (+15 us on the launch when the
    // words left where they are made).  A word may straddle two words of the plane and share them with the tile in front or
    // behind: OR, into words that tile_order_binned_kernel cleared one launch ago.
    int emit_n = 0;                                  // survivor words waiting in wmask[wave][..] (wave-uniform)
    uint32_t emit_base = 0;                          // list position of bit 0 of the first one
    auto emit_masks = [&]() {
#ifdef SOAR_EXP_FWD_NO_MASK_EMIT       // (development, the backward's input wrong by construction: what the forward costs without leaving the
        emit_n = 0;                    // masks behind -- 209.5-214 us against 237 per 4-frame launch at C3, profiles/README.md round 6)
        return;
#endif
        if (a.masks && lane < emit_n) {
            const unsigned long long wd = wmask[wave][lane];
            if (wd != 0ull) {
                const uint32_t p0 = emit_base + (uint32_t)lane * WAVE, sh = p0 & 63u;
                unsigned long long *row = a.masks + (size_t)(quad * 4 + wave) * a.mask_plane + (p0 >> 6);
                __hip_atomic_fetch_or(row, wd << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (sh) __hip_atomic_fetch_or(row + 1, wd >> (64u - sh), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        emit_n = 0;
    };
    for (uint32_t base = range.x; base < range.y; base += CHUNK, parity ^= 1) {
        const int n = min((uint32_t)CHUNK, range.y - base);
        emit_masks();                                // (the chunk before)
        emit_base = base;
        if (tid < n) {
            const uint32_t id = id_next;
            const float4 *src = reinterpret_cast<const float4 *>(a.rec + id);
            sq0[tid] = src[0]; sq1[tid] = src[1]; sq2[tid] = src[2]; sq3[tid] = src[3];
            if (OCC) *reinterpret_cast<float2 *>(&sq4[tid]) = make_float2(a.occ_values[id], a.front[id]);
        }
        if (base + CHUNK + tid < range.y) id_next = a.point_list[base + CHUNK + tid];
        lds_barrier();
        if (LOG && base == range.x) t_ready = wall_clock64();

        if (!wave_done) {
            // The entries only matter for the pixels that are still blending: the test rectangle of phase A is the bounding
            // box of those pixels inside the 4x4 block (a block on the silhouette keeps walking the list for its uncovered
            // pixels long after the covered ones have saturated; entries that touch only finished pixels are dropped).
            float rx0 = (float)bx0, ry0 = (float)by0, rex = 3.f, rey = 3.f;
            {
                const unsigned long long am = __ballot(alive + alive_o != 0.f);           // 4 lanes per pixel, pixel = lane >> 2
                // active columns / rows of the block: pixel pp owns the nibble 4 pp of the ballot, a row of pixels 16 bits
                const uint32_t lo = (uint32_t)am, hi = (uint32_t)(am >> 32);
                const uint32_t rows = ((lo & 0xFFFFu) ? 1u : 0u) | ((lo >> 16) ? 2u : 0u) | ((hi & 0xFFFFu) ? 4u : 0u) | ((hi >> 16) ? 8u : 0u);
                uint32_t fold = lo | hi;
                fold |= fold >> 16;                                                        // nibble c = column c of any row
                const uint32_t cols = (fold & 1u) | ((fold >> 3) & 2u) | ((fold >> 6) & 4u) | ((fold >> 9) & 8u);
                if (cols) {
                    const int c0 = __builtin_ctz(cols), c1 = 31 - __builtin_clz(cols), r0 = __builtin_ctz(rows), r1 = 31 - __builtin_clz(rows);
                    rx0 = (float)(bx0 + c0); ry0 = (float)(by0 + r0); rex = (float)(c1 - c0); rey = (float)(r1 - r0);
                }
            }
            // phase A -- lanes = entries, 64 at a time over the whole chunk: conservative test against that rectangle.  The relevant
            // entries' LDS slots are compacted in list order into the wavefront's ring (ballot-prefix ranks) -- ONE ring for the chunk
            // (round 4: it was one per 64 entries; a block keeps 4-5 of 64 on average, and every ring's last step ran with one to three
            // of its four slots empty: a fifth of all steps' slots); three pad entries behind the last one point at the zero record
            int n_todo = 0;
            for (int sub = 0; sub < n; sub += WAVE) {
                bool relevant = false;
                if (sub + lane < n) {
                    const float4 e0 = sq0[sub + lane], e1 = sq1[sub + lane];
                    relevant = splat_may_touch_rect(e0.x, e0.y, e0.z, e0.w, e1.x, sq3[sub + lane].w, rx0, ry0, rex, rey);
                }
                const unsigned long long todo = __ballot(relevant);
                if (relevant) todo_ring[wave][n_todo + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(todo >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)todo, 0u))] = (unsigned short)(16 * (sub + lane));
                n_todo += (int)__builtin_popcountll(todo);
                if (lane == 0) wmask[wave][sub / WAVE] = todo;                  // left behind for the backward blend: emit_masks below
            }
            emit_n = (n + WAVE - 1) / WAVE;
            {
                if (lane < 3) todo_ring[wave][n_todo + lane] = (unsigned short)(16 * CHUNK);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                // 16 x (list position + 1) of LDS slot 0, in a vector register (the positions are kept x 16 -- what the ring holds --
                // until the epilogue: no shift in the loop; an operand from a scalar register halves the add's rate)
                uint32_t contrib16 = 16u * (base - range.x + 1u);
                asm volatile("" : "+v"(contrib16));

                // phase B -- lanes = (pixel, slot): four surviving entries per step, in list order
                for (int it = 0; it < n_todo; it += 4) {
                    if (LOG) n_iter++;
                    const uint32_t j16 = todo_ring[wave][it + slot];
                    const float4 q0 = lds_at(sq0, j16), q1 = lds_at(sq1, j16), q2 = lds_at(sq2, j16), q3 = lds_at(sq3, j16);
                    // x,y,A,B | C,opacity,depth,plane_a | plane_b,r,g,b | nx,ny,nz,-
                    const float dx = q0.x - fx, dy = q0.y - fy;
                    const float power = falloff_power(q0.z, q0.w, q1.x, dx, dy);          // forward.cu:507-508
                    const float alpha = fminf(0.99f, q1.y * exp_nonpositive(power));
                    // skip rules (:512, :545) zero the effective alpha of this lane's entry
                    float a_live = (power > 0.0f) ? 0.f : alpha;
                    a_live = (alpha < 1.0f / 255.0f) ? 0.f : a_live;
                    if (LOG) n_useful += (unsigned long long)__builtin_popcountll(__ballot(a_live * alive > 0.f));
                    const float a_eff = a_live * alive;                                   // x 1 or x 0: exact
                    // running transmittance through the four slots, reference order (:548-553, :602).
                    // Invariant: T >= 1e-4 in every lane, so "T*(1-a) < 1e-4" can only fire on a live entry.
                    const float om = 1.f - a_eff;
                    // OCC: the same chain over the camera-facing entries only, with its own transmittance and stop
                    float2 e4 = make_float2(0.f, 0.f);
                    float a_o = 0.f, mo = 1.f;
                    if (OCC) {
                        e4 = *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(sq4) + j16);      // (8 of the slot's 16 bytes)
                        a_o = a_live * e4.y * alive_o;                                    // camera-facing flag and liveness are 0 / 1
                        mo = 1.f - a_o;
                    }
                    // unclamped running products first: transmittances only shrink, so some slot stops the pixel in
                    // this step iff the last product is below the threshold -- and in most steps no pixel of the
                    // wavefront stops: then the clamping selects and the per-slot "stopped" predicate are not needed.
                    // The products run through the quad as a scan (lane k <- lane k - 1, three fused DPP multiplies): lane k
                    // ends with ((T om_0) om_1 ...) om_{k-1}, the transmittance in front of its entry in the reference's
                    // order of roundings.  The two chains' scans are issued step by step side by side.
                    // (the factor of the lane in front: 1 in slot 0, else 1 - its alpha -- 1 - a x {0, 1}: the same rounding as its own om)
                    const float om_front = __builtin_fmaf(-quad_move<DPP_QUAD_SHIFT1>(a_eff), not_first, 1.f);
                    float T_front, U_front = 1.f;
                    if (OCC) quad_scan_front2(T, om_front, T_o, __builtin_fmaf(-quad_move<DPP_QUAD_SHIFT1>(a_o), not_first, 1.f), T_front, U_front);
                    else T_front = quad_scan_front(T, om_front);
                    const float x = mul_keep(T_front, om), y = OCC ? mul_keep(U_front, mo) : 1.f;     // the products behind my entry
                    const float p3 = quad_move<DPP_QUAD_BCAST3>(x);
                    const float v3 = OCC ? quad_move<DPP_QUAD_BCAST3>(y) : 1.f;
                    float w, w_o = 0.f;
                    bool some_stop = false;                                               // wave-uniform
                    if (__ballot(p3 < 0.0001f) == 0ull) {
                        w = a_eff * T_front;
                        T = p3;
                    } else {
                        // (the general form: the same products, with the clamping selects)
                        const float om0 = quad_move<DPP_QUAD_BCAST0>(om), om1 = quad_move<DPP_QUAD_BCAST1>(om),
                                    om2 = quad_move<DPP_QUAD_BCAST2>(om), om3 = quad_move<DPP_QUAD_BCAST3>(om);
                        const float t0 = mul_keep(T, om0);
                        const bool s0 = t0 < 0.0001f;
                        const float T1 = s0 ? T : t0;
                        const float t1 = mul_keep(T1, om1);
                        const bool s1 = s0 || (t1 < 0.0001f);
                        const float T2 = s1 ? T1 : t1;
                        const float t2 = mul_keep(T2, om2);
                        const bool s2 = s1 || (t2 < 0.0001f);
                        const float T3 = s2 ? T2 : t2;
                        const float t3 = mul_keep(T3, om3);
                        const bool s3 = s2 || (t3 < 0.0001f);
                        // transmittance in front of MY entry, and whether the pixel had stopped at or before it
                        const float T_mine = slot == 0 ? T : slot == 1 ? T1 : slot == 2 ? T2 : T3;
                        const bool stopped = slot == 0 ? s0 : slot == 1 ? s1 : slot == 2 ? s2 : s3;
                        w = stopped ? 0.f : a_eff * T_mine;
                        T = s3 ? T3 : t3;
                        alive = s3 ? 0.f : alive;
                        some_stop = true;
                    }
                    if (OCC) {
                        if (__ballot(v3 < 0.0001f) == 0ull) {
                            w_o = a_o * U_front;
                            T_o = v3;
                        } else {
                            const float mo0 = quad_move<DPP_QUAD_BCAST0>(mo), mo1 = quad_move<DPP_QUAD_BCAST1>(mo),
                                        mo2 = quad_move<DPP_QUAD_BCAST2>(mo), mo3 = quad_move<DPP_QUAD_BCAST3>(mo);
                            const float u0 = mul_keep(T_o, mo0);
                            const bool z0 = u0 < 0.0001f;
                            const float U1 = z0 ? T_o : u0;
                            const float u1 = mul_keep(U1, mo1);
                            const bool z1 = z0 || (u1 < 0.0001f);
                            const float U2 = z1 ? U1 : u1;
                            const float u2 = mul_keep(U2, mo2);
                            const bool z2 = z1 || (u2 < 0.0001f);
                            const float U3 = z2 ? U2 : u2;
                            const float u3 = mul_keep(U3, mo3);
                            const bool z3 = z2 || (u3 < 0.0001f);
                            const float U_mine = slot == 0 ? T_o : slot == 1 ? U1 : slot == 2 ? U2 : U3;
                            const bool stopped_o = slot == 0 ? z0 : slot == 1 ? z1 : slot == 2 ? z2 : z3;
                            w_o = stopped_o ? 0.f : a_o * U_mine;
                            T_o = z3 ? U3 : u3;
                            alive_o = z3 ? 0.f : alive_o;
                            some_stop = true;
                        }
                    }
                    const bool blend = w != 0.f;          // w == 0 adds exactly nothing to the sums below (records are finite)
                    const float depth = q1.z - (dx * q1.w + dy * q2.x);                   // depth on the surfel plane
                    D = __builtin_fmaf(depth, w, D);
                    C0 = __builtin_fmaf(q2.y, w, C0);
                    C1 = __builtin_fmaf(q2.z, w, C1);
                    C2 = __builtin_fmaf(q2.w, w, C2);
                    N0 = __builtin_fmaf(q3.x, w, N0);
                    N1 = __builtin_fmaf(q3.y, w, N1);
                    N2 = __builtin_fmaf(q3.z, w, N2);
                    last_contributor = blend ? contrib16 + j16 : last_contributor;        // (x 16: see contrib16)
                    if (OCC) {
                        Co = __builtin_fmaf(e4.x, w_o, Co);
                        last_contributor_o = (w_o != 0.f) ? contrib16 + j16 : last_contributor_o;
                    }
                    if (some_stop && __ballot(alive + alive_o != 0.f) == 0ull) { wave_done = true; break; }
                }
            }
        }
        // "is any wavefront of the workgroup still blending?" -- also the barrier that protects the LDS arrays before
        // the next chunk overwrites them
        if (lane == 0) wave_alive[parity][wave] = wave_done ? 0 : 1;
        lds_barrier();
        if ((wave_alive[parity][0] | wave_alive[parity][1] | wave_alive[parity][2] | wave_alive[parity][3]) == 0) break;
    }
    emit_masks();                                    // (the last chunk this wavefront tested)

    if (LOG) t_blended = wall_clock64();
    // fold the four slots of every pixel
    D += quad_move<DPP_QUAD_XOR1>(D); D += quad_move<DPP_QUAD_XOR2>(D);
    C0 += quad_move<DPP_QUAD_XOR1>(C0); C0 += quad_move<DPP_QUAD_XOR2>(C0);
    C1 += quad_move<DPP_QUAD_XOR1>(C1); C1 += quad_move<DPP_QUAD_XOR2>(C1);
    C2 += quad_move<DPP_QUAD_XOR1>(C2); C2 += quad_move<DPP_QUAD_XOR2>(C2);
    N0 += quad_move<DPP_QUAD_XOR1>(N0); N0 += quad_move<DPP_QUAD_XOR2>(N0);
    N1 += quad_move<DPP_QUAD_XOR1>(N1); N1 += quad_move<DPP_QUAD_XOR2>(N1);
    N2 += quad_move<DPP_QUAD_XOR1>(N2); N2 += quad_move<DPP_QUAD_XOR2>(N2);
    last_contributor = max(last_contributor, quad_move_u<DPP_QUAD_XOR1>(last_contributor));
    last_contributor = max(last_contributor, quad_move_u<DPP_QUAD_XOR2>(last_contributor));
    last_contributor >>= 4;                          // (kept x 16 in the loop)
    if (OCC) {
        Co += quad_move<DPP_QUAD_XOR1>(Co); Co += quad_move<DPP_QUAD_XOR2>(Co);
        last_contributor_o = max(last_contributor_o, quad_move_u<DPP_QUAD_XOR1>(last_contributor_o));
        last_contributor_o = max(last_contributor_o, quad_move_u<DPP_QUAD_XOR2>(last_contributor_o));
        last_contributor_o >>= 4;
    }

    if (inside && slot == 0) {
        // epilogue, forward.cu:618-633
        T = fminf((float)(1 - 0.000001), T);
        const size_t pix = (size_t)a.W * py + px;
        const size_t hw = (size_t)a.H * a.W;
        a.final_T[pix] = T;
        a.n_contrib[pix] = last_contributor;
        a.out_color[pix] = C0 + T * a.bg[0];
        a.out_color[hw + pix] = C1 + T * a.bg[1];
        a.out_color[2 * hw + pix] = C2 + T * a.bg[2];
        a.out_normal[pix] = N0;
        a.out_normal[hw + pix] = N1;
        a.out_normal[2 * hw + pix] = N2;
        a.out_depth[pix] = a.normalize_depth ? D / (1.f - T) : D + T * 10.f;
        a.out_opac[pix] = 1.f - T;
        if (a.normalize_depth) a.final_D[pix] = D;
        if (OCC) {
            T_o = fminf((float)(1 - 0.000001), T_o);
            a.final_To[pix] = T_o;
            a.n_contrib_o[pix] = last_contributor_o;
            a.out_occ[pix] = Co + T_o * a.bg[0];
            a.out_occ[hw + pix] = Co + T_o * a.bg[1];
            a.out_occ[2 * hw + pix] = Co + T_o * a.bg[2];
        }
    }
    if (LOG && lane == 0) {
        unsigned long long *w = a.wave_log + ((size_t)seq * 4 + wave) * 4;
        // (w[2]: list length | time to the first staged chunk << 24 | time from the end of the blending to here << 44, in 10 ns)
        const unsigned long long t_end = wall_clock64();
        w[0] = t_start; w[1] = t_end; w[2] = (unsigned long long)(range.y - range.x) | (min(t_ready - t_start, 0xFFFFFull) << 24) | (min(t_end - t_blended, 0xFFFFFull) << 44);
        w[3] = n_iter | (n_useful << 24);
    }
}

// The launch covers the first gridDim.x / 4 ranks of the longest-first tile order, four workgroups (quads) per tile; the
// ranks are dealt round-robin over the 8 XCDs with the four quads of a tile on one XCD (one L2 serves the tile's records).
// The grid is much smaller than the tile count: only ~10 % of the tiles of a 1080p frame of one person have any work, and
// even a workgroup that exits at once costs ~0.6 ns all told (slot, loads, ~100 instructions) -- four frames in flight paid
// ~100 us per step for empty workgroups.  Tiles with work beyond the grid (a denser scene) are reached by the rank-stride loop; the tiles no
// Gaussian touches sit behind the first n_work ranks and are filled with their background values, whole tiles, by all
// workgroups once their blending is done.
#ifndef SOAR_FWD_WPE
#define SOAR_FWD_WPE 7       // (71 VGPRs, 22.8 KB of LDS per workgroup: seven workgroups per CU)
#endif
template <bool LOG, bool OCC>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SOAR_FWD_WPE, 8))) render_forward_kernel(Batch<FwdArgs> batch)
{
    int frame, bx;
    batch_interleave(frame, bx);
    const FwdArgs &a = batch.v[frame];
    const int xcd = bx & 7, kth = bx >> 3;
    const int rank0 = (kth >> 2) * 8 + xcd, quad = kth & 3;
    const int stride = (int)(gridDim.x >> 2);                // ranks per pass of the grid (a multiple of 8)
    const int Tpad = (a.ntiles + 7) / 8 * 8;
    const int n_work = LOG ? Tpad : (int)a.tile_order[Tpad];
    for (int rank = rank0; rank < n_work; rank += stride) {
        if (quad == 0 && threadIdx.x == 0 && a.tile_order[rank] != 0xFFFFFFFFu)
            __hip_atomic_store(a.bg_tiles + a.tile_order[rank], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        blend_quad<LOG, OCC>(a, rank, quad);
        lds_barrier();                                       // the next item's staging overwrites this one's LDS image
    }
    // A tile that was empty in the previous forward blend into the SAME output planes with the same background still holds its
    // values (the caller says so: SoarRastParams.debug bit 2): 85 % of the output bytes of a 1080p frame of one person, which
    // would otherwise be written again -- and written back from the L2s when the kernel ends
    // (the caller's promise covers the buffers; that the background VALUES are those of the previous forward is checked on the
    // device: tile_order_block compared them one launch ago)
    const bool keep = a.keep_background && a.bg_state[4] == 0u;
    if (!LOG)
        for (int rank = n_work + bx; rank < a.ntiles; rank += (int)gridDim.x) {
            const int tile = (int)a.tile_order[rank];
            const uint32_t holds_background = __hip_atomic_load(a.bg_tiles + tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lds_barrier();                                   // every wavefront has read the flag before the first one may set it below
            if (keep && holds_background == 1u) continue;
            fill_tile<OCC>(a, tile, (int)threadIdx.x, 256);
            if (threadIdx.x == 0) __hip_atomic_store(a.bg_tiles + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
}


// ------------------------------------------------------------------------------------------------------------------------------
// Gradient of the fused occlusion image w.r.t. the per-Gaussian occlusion values (the reference trains them: loss_occ,
// TS/system/gaussian_surfel_mvdream.py:412-417, through an occlusion pass whose colours are occ.repeat(1, 3) and whose geometry is
// detached, TS/renderer/diff_gaussian_rasterizer.py:281-291):
//     dL/docc_i = sum over pixels of (g_0 + g_1 + g_2)(pixel) * alpha_i(pixel) * T_occ,i(pixel)
// The weights alpha * T_occ are those of the occlusion chain of blend_quad<.., OCC = true>; this kernel walks the same lists
// with the same arithmetic for that chain only (same roundings, same stops) and, instead of blending a colour, adds up
// weight x upstream gradient over the 16 pixels of a wavefront's 4x4 block per surviving entry -- one atomic per (wavefront,
// entry) that contributed.  Pixels whose upstream gradient is zero never start (loss_occ masks the person's pixels).
struct OccGradArgs {
    int W, H, gx, gy, ntiles;
    const uint2 *ranges;
    const uint32_t *tile_order;
    const uint4 *order_rec;
    const uint32_t *point_list;
    const GaussRec *rec;
    const float *front;              // [P] 1 = camera-facing (preprocess)
    const float *g_occ;              // [3,H,W] upstream gradient of the occlusion image
    float *g_values;                 // [P] out, zero-filled before the launch
};

__device__ __forceinline__ void occ_grad_quad(const OccGradArgs &a, const int rank, const int quad)
{
    __shared__ float4 sq0[CHUNK + 1];                                                    // x, y, A, B
    __shared__ float4 sq1[CHUNK + 1];                                                    // C, opacity, cull threshold, camera-facing
    __shared__ uint32_t sid[CHUNK + 1];
    __shared__ int wave_alive[2][4];
    __shared__ unsigned short todo_ring[4][WAVE + 4];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t tile_u = a.tile_order[rank];
    if (tile_u == 0xFFFFFFFFu) return;
    const int tile = (int)tile_u;
    const int tx = tile % a.gx, ty = tile / a.gx;
    const int bx0 = tx * TILE + (quad & 1) * 8 + (wave & 1) * 4, by0 = ty * TILE + (quad >> 1) * 8 + (wave >> 1) * 4;
    const int pxl = lane >> 2, slot = lane & 3;
    const int px = bx0 + (pxl & 3), py = by0 + (pxl >> 2);
    const bool inside = px < a.W && py < a.H;
    const float not_first = slot == 0 ? 0.f : 1.f;
    float fx = (float)px, fy = (float)py;
    asm volatile("" : "+v"(fx), "+v"(fy));           // (kept as floats: see blend_quad)
    const uint2 range = a.ranges[tile];

    float G = 0.f;
    if (inside) {
        const size_t pix = (size_t)a.W * py + px, hw = (size_t)a.H * a.W;
        G = a.g_occ[pix] + a.g_occ[hw + pix] + a.g_occ[2 * hw + pix];
    }
    float T_o = 1.0f;
    float alive_o = (inside && G != 0.f) ? 1.f : 0.f;
    bool wave_done = (__ballot(alive_o != 0.f) == 0ull);

    if (tid == 0) {
        sq0[CHUNK] = make_float4(0.f, 0.f, 0.f, 0.f);
        sq1[CHUNK] = make_float4(0.f, 0.f, 0.f, 0.f);
        sid[CHUNK] = 0u;
    }
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0;
    uint32_t rid = 0, id_next = 0;
    if (range.x + tid < range.y) {
        rid = a.point_list[range.x + tid];
        const float4 *src = reinterpret_cast<const float4 *>(a.rec + rid);
        r0 = src[0];
        const float4 t1 = src[1];
        r1 = make_float4(t1.x, t1.y, src[3].w, a.front[rid]);
    }
    if (range.x + CHUNK + tid < range.y) id_next = a.point_list[range.x + CHUNK + tid];
    int parity = 0;
    for (uint32_t base = range.x; base < range.y; base += CHUNK, parity ^= 1) {
        const int n = min((uint32_t)CHUNK, range.y - base);
        if (tid < n) { sq0[tid] = r0; sq1[tid] = r1; sid[tid] = rid; }
        if (base + CHUNK + tid < range.y) {
            rid = id_next;
            const float4 *src = reinterpret_cast<const float4 *>(a.rec + rid);
            r0 = src[0];
            const float4 t1 = src[1];
            r1 = make_float4(t1.x, t1.y, src[3].w, a.front[rid]);
        }
        if (base + 2 * CHUNK + tid < range.y) id_next = a.point_list[base + 2 * CHUNK + tid];
        lds_barrier();

        if (!wave_done) {
            float rx0 = (float)bx0, ry0 = (float)by0, rex = 3.f, rey = 3.f;             // bounding box of the pixels still walking
            {
                const unsigned long long am = __ballot(alive_o != 0.f);
                const uint32_t lo = (uint32_t)am, hi = (uint32_t)(am >> 32);
                const uint32_t rows = ((lo & 0xFFFFu) ? 1u : 0u) | ((lo >> 16) ? 2u : 0u) | ((hi & 0xFFFFu) ? 4u : 0u) | ((hi >> 16) ? 8u : 0u);
                uint32_t fold = lo | hi;
                fold |= fold >> 16;                                                        // nibble c = column c of any row
                const uint32_t cols = (fold & 1u) | ((fold >> 3) & 2u) | ((fold >> 6) & 4u) | ((fold >> 9) & 8u);
                if (cols) {
                    const int c0 = __builtin_ctz(cols), c1 = 31 - __builtin_clz(cols), q0 = __builtin_ctz(rows), q1 = 31 - __builtin_clz(rows);
                    rx0 = (float)(bx0 + c0); ry0 = (float)(by0 + q0); rex = (float)(c1 - c0); rey = (float)(q1 - q0);
                }
            }
            for (int sub = 0; sub < n; sub += WAVE) {
                bool relevant = false;
                if (sub + lane < n) {
                    const float4 e0 = sq0[sub + lane], e1 = sq1[sub + lane];
                    relevant = e1.w != 0.f && splat_may_touch_rect(e0.x, e0.y, e0.z, e0.w, e1.x, e1.z, rx0, ry0, rex, rey);
                }
                const unsigned long long todo = __ballot(relevant);
                const int n_todo = (int)__builtin_popcountll(todo);
                if (relevant) todo_ring[wave][__builtin_amdgcn_mbcnt_hi((uint32_t)(todo >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)todo, 0u))] = (unsigned short)(sub + lane);
                if (lane < 3) todo_ring[wave][n_todo + lane] = (unsigned short)CHUNK;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

                for (int it = 0; it < n_todo; it += 4) {
                    const int j = todo_ring[wave][it + slot];
                    const float4 q0 = sq0[j], q1 = sq1[j];
                    const float dx = q0.x - fx, dy = q0.y - fy;
                    const float power = falloff_power(q0.z, q0.w, q1.x, dx, dy);
                    const float alpha = fminf(0.99f, q1.y * exp_nonpositive(power));
                    float a_live = (power > 0.0f) ? 0.f : alpha;
                    a_live = (alpha < 1.0f / 255.0f) ? 0.f : a_live;
                    const float a_o = a_live * q1.w * alive_o;                           // (back-facing entries never enter the ring; the
                    const float mo = 1.f - a_o;                                          //  pad record has flag 0)
                    const float U_front = quad_scan_front(T_o, __builtin_fmaf(-quad_move<DPP_QUAD_SHIFT1>(a_o), not_first, 1.f));
                    const float v3 = quad_move<DPP_QUAD_BCAST3>(mul_keep(U_front, mo));
                    float w_o;
                    bool some_stop = false;
                    if (__ballot(v3 < 0.0001f) == 0ull) {
                        w_o = a_o * U_front;
                        T_o = v3;
                    } else {
                        const float mo0 = quad_move<DPP_QUAD_BCAST0>(mo), mo1 = quad_move<DPP_QUAD_BCAST1>(mo),
                                    mo2 = quad_move<DPP_QUAD_BCAST2>(mo), mo3 = quad_move<DPP_QUAD_BCAST3>(mo);
                        const float u0 = mul_keep(T_o, mo0);
                        const bool z0 = u0 < 0.0001f;
                        const float U1 = z0 ? T_o : u0;
                        const float u1 = mul_keep(U1, mo1);
                        const bool z1 = z0 || (u1 < 0.0001f);
                        const float U2 = z1 ? U1 : u1;
                        const float u2 = mul_keep(U2, mo2);
                        const bool z2 = z1 || (u2 < 0.0001f);
                        const float U3 = z2 ? U2 : u2;
                        const float u3 = mul_keep(U3, mo3);
                        const bool z3 = z2 || (u3 < 0.0001f);
                        const float U_mine = slot == 0 ? T_o : slot == 1 ? U1 : slot == 2 ? U2 : U3;
                        const bool stopped_o = slot == 0 ? z0 : slot == 1 ? z1 : slot == 2 ? z2 : z3;
                        w_o = stopped_o ? 0.f : a_o * U_mine;
                        T_o = z3 ? U3 : u3;
                        alive_o = z3 ? 0.f : alive_o;
                        some_stop = true;
                    }
                    if (__ballot(w_o != 0.f) != 0ull) {
                        // sum over the 16 pixels of the block, per slot (lanes with the same lane & 3)
                        float c = w_o * G;
                        c += __shfl_xor(c, 4);
                        c += __shfl_xor(c, 8);
                        c += __shfl_xor(c, 16);
                        c += __shfl_xor(c, 32);
                        if (lane < 4 && c != 0.f) atomicAdd(a.g_values + sid[j], c);
                    }
                    if (some_stop && __ballot(alive_o != 0.f) == 0ull) { wave_done = true; break; }
                }
                if (wave_done) break;
            }
        }
        if (lane == 0) wave_alive[parity][wave] = wave_done ? 0 : 1;
        lds_barrier();
        if ((wave_alive[parity][0] | wave_alive[parity][1] | wave_alive[parity][2] | wave_alive[parity][3]) == 0) break;
    }
}

__global__ void __launch_bounds__(256) occ_backward_kernel(Batch<OccGradArgs> batch)
{
    int frame, bx;
    batch_interleave(frame, bx);
    const OccGradArgs &a = batch.v[frame];
    const int xcd = bx & 7, kth = bx >> 3;
    const int rank0 = (kth >> 2) * 8 + xcd, quad = kth & 3;
    const int stride = (int)(gridDim.x >> 2);
    const int Tpad = (a.ntiles + 7) / 8 * 8;
    const int n_work = (int)a.tile_order[Tpad];
    for (int rank = rank0; rank < n_work; rank += stride) {
        occ_grad_quad(a, rank, quad);
        lds_barrier();
    }
}

}  // namespace

int launch_render_forward(const SoarRastParams &prm, const GeomBuf &g, const BinBuf &b, ImageBuf &img,
                          float *out_color, float *out_normal, float *out_depth, float *out_opac,
                          const float *occ_values, float *out_occ, hipStream_t stream)
{
    FwdArgs a;
    a.W = prm.W; a.H = prm.H;
    a.gx = (prm.W + TILE - 1) / TILE; a.gy = (prm.H + TILE - 1) / TILE;
    a.ntiles = a.gx * a.gy;
    a.normalize_depth = prm.cfg_normalize_depth;
    a.ranges = img.ranges; a.tile_order = img.tile_order; a.order_rec = img.order_rec; a.point_list = b.vals_sorted; a.rec = g.rec; a.bg = prm.bg_dev;
    a.final_T = img.final_T; a.final_D = img.final_D; a.n_contrib = img.n_contrib;
    a.out_color = out_color; a.out_normal = out_normal; a.out_depth = out_depth; a.out_opac = out_opac;
    a.occ_values = occ_values; a.front = g.front; a.out_occ = out_occ;
    a.final_To = img.final_To; a.n_contrib_o = img.n_contrib_o;
    a.bg_tiles = img.bg_tiles; a.bg_state = img.bg_state; a.keep_background = (prm.debug & 4) ? 1 : 0;
    a.wave_log = nullptr;
    a.masks = reinterpret_cast<unsigned long long *>(b.block_masks); a.mask_plane = b.mask_plane;
    const char *log_path = getenv("SOAR_WAVE_LOG");          // diagnostic: dump per-wave timelines of ONE launch
    const int grid_ranks = blend_grid_ranks(a.ntiles);
    const int Tpad = (a.ntiles + 7) / 8 * 8;
    const int nblocks = 4 * (log_path ? Tpad : min(Tpad, grid_ranks));
    StageTimer timer(ST_RENDER_FWD, stream);
    static int logged = 0;
    if (log_path && !logged && prm.render_front == 0) {
        logged = 1;
        const size_t nbytes = sizeof(unsigned long long) * 16 * (size_t)nblocks;
        SOAR_HIP_OK(hipMalloc(&a.wave_log, nbytes));
        SOAR_HIP_OK(hipMemsetAsync(a.wave_log, 0, nbytes, stream));
        if (out_occ) SOAR_LAUNCH_BATCHED((render_forward_kernel<true, true>), dim3(nblocks), dim3(256), 0, stream, a);
        else SOAR_LAUNCH_BATCHED((render_forward_kernel<true, false>), dim3(nblocks), dim3(256), 0, stream, a);
        SOAR_HIP_OK(hipStreamSynchronize(stream));
        unsigned long long *host = (unsigned long long *)malloc(nbytes);
        SOAR_HIP_OK(hipMemcpy(host, a.wave_log, nbytes, hipMemcpyDeviceToHost));
        FILE *f = fopen(log_path, "wb");
        if (f) { fwrite(host, 1, nbytes, f); fclose(f); }
        free(host);
        (void)hipFree(a.wave_log);
        return 0;
    }
    if (out_occ) SOAR_LAUNCH_BATCHED((render_forward_kernel<false, true>), dim3(nblocks), dim3(256), 0, stream, a);
    else SOAR_LAUNCH_BATCHED((render_forward_kernel<false, false>), dim3(nblocks), dim3(256), 0, stream, a);
    SOAR_LAUNCH_OK("render_forward", stream, prm.debug);
    return 0;
}

int launch_occ_backward(const SoarRastParams &prm, const GeomBuf &g, const BinBuf &b, const ImageBuf &img, const float *dL_dout_occ,
                        float *dL_docc, hipStream_t stream)
{
    OccGradArgs a;
    a.W = prm.W; a.H = prm.H;
    a.gx = (prm.W + TILE - 1) / TILE; a.gy = (prm.H + TILE - 1) / TILE;
    a.ntiles = a.gx * a.gy;
    a.ranges = img.ranges; a.tile_order = img.tile_order; a.order_rec = img.order_rec; a.point_list = b.vals_sorted; a.rec = g.rec; a.front = g.front;
    a.g_occ = dL_dout_occ; a.g_values = dL_docc;
    const int Tpad = (a.ntiles + 7) / 8 * 8;
    const int nblocks = 4 * min(Tpad, blend_grid_ranks(a.ntiles));
    SOAR_HIP_OK(hipMemsetAsync(dL_docc, 0, sizeof(float) * (size_t)prm.P, stream));
    StageTimer timer(ST_RENDER_BWD, stream);
    SOAR_LAUNCH_BATCHED(occ_backward_kernel, dim3(nblocks), dim3(256), 0, stream, a);
    SOAR_LAUNCH_OK("occ_backward", stream, prm.debug);
    return 0;
}

}  // namespace soar
