// rast_render_fwd.hip -- per-tile front-to-back alpha blend for gfx950.
//
// Replaces renderCUDA<3> forward (DGR/cuda_rasterizer/forward.cu:390-692) and depth_differencing
// (auxiliary.h:390-397).
//
// CDNA4 mapping (not the reference's 256-thread lock-step block):
//  * a 16x16 tile is owned by one 256-thread workgroup, but each of its four wavefronts owns an 8x8
//    pixel quad and walks the tile's depth-sorted list on its own: no workgroup barrier anywhere, a wave
//    leaves as soon as its 64 pixels are saturated (wave-level vote = one s_cbranch on the exec mask);
//  * the list is consumed in chunks of 64: lane l gathers the 64-byte record of entry l (four 16-byte
//    loads, one cache-line half) and parks it in the wave's private 4 KiB LDS slab; the blend loop then
//    reads records with wave-uniform (broadcast) ds_read_b128 -- two for the alpha test, two more only if
//    some lane survives it;
//  * before that loop the SAME 64 lanes, still holding one record each, evaluate a conservative bound of the splat's
//    alpha over the quad (splat_may_touch_quad): a ballot yields the 64-bit set of entries that can matter, and the
//    blend loop visits only those bits (s_ff1) -- entries that every pixel would skip cost nothing;
//  * workgroup -> tile mapping is XCD-aware: consecutive tiles (which share Gaussians) stay on one XCD's L2.
#include "soar_common.h"

#include <cstdio>
#include <cstdlib>

namespace soar {

namespace {

struct FwdArgs {
    int W, H, gx, gy, ntiles;
    int normalize_depth;
    const uint2 *ranges;
    const uint32_t *point_list;
    const GaussRec *rec;
    const float *bg;
    float *final_T;
    float *final_D;
    uint32_t *n_contrib;
    float *out_color, *out_normal, *out_depth, *out_opac;
    unsigned long long *wave_log;    // diagnostic build only (SOAR_WAVE_LOG): per wave {t_start, t_end, list length, iterations}
};

// blocks are dealt round-robin over the 8 XCDs: give every XCD one contiguous run of tiles
__device__ __forceinline__ int xcd_tile(int bid, int n)
{
    const int q = n >> 3, r = n & 7;
    const int xcd = bid & 7, within = bid >> 3;
    return xcd * q + min(xcd, r) + within;
}

template <bool LOG>
__global__ void __launch_bounds__(256) render_forward_kernel(FwdArgs a)
{
    __shared__ GaussRec slab[4][WAVE + 1];             // +1: an all-zero record
    unsigned long long t_start = 0, n_iter = 0;
    if (LOG) t_start = wall_clock64();

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tile = xcd_tile(blockIdx.x, a.ntiles);
    const int tx = tile % a.gx, ty = tile / a.gx;
    const int px = tx * TILE + (wave & 1) * 8 + (lane & 7);
    const int py = ty * TILE + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < a.W && py < a.H;
    const float fx = (float)px, fy = (float)py;

    const uint2 range = a.ranges[tile];

    float T = 1.0f;
    float C0 = 0.f, C1 = 0.f, C2 = 0.f, N0 = 0.f, N1 = 0.f, N2 = 0.f, D = 0.f;
    uint32_t last_contributor = 0;
    bool done = !inside;

    GaussRec *my = slab[wave];
    const float4 *myq = reinterpret_cast<const float4 *>(my);
    if (lane < 4) reinterpret_cast<float4 *>(my + WAVE)[lane] = make_float4(0.f, 0.f, 0.f, 0.f);

    const float quad_x0 = (float)(tx * TILE + (wave & 1) * 8), quad_y0 = (float)(ty * TILE + (wave >> 1) * 8);

    // software pipeline over chunks: the records of chunk k+1 are requested before the blend loop of chunk k runs,
    // so the dependent (list entry -> record) gather latency stays off the critical path of long tile lists
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0, r3 = r0;
    if (range.x + lane < range.y) {
        const float4 *src = reinterpret_cast<const float4 *>(a.rec + a.point_list[range.x + lane]);
        r0 = src[0]; r1 = src[1]; r2 = src[2]; r3 = src[3];
    }
    for (uint32_t base = range.x; base < range.y; base += WAVE) {
        if (__ballot(!done) == 0ull) break;                 // whole quad saturated
        const int n = min((uint32_t)WAVE, range.y - base);
        // phase A -- lanes = list entries: park the record in the slab and vote whether the splat can reach the
        // 1/255 alpha floor anywhere in this quad
        bool relevant = false;
        if (lane < n) {
            float4 *dst = reinterpret_cast<float4 *>(my + lane);
            dst[0] = r0; dst[1] = r1; dst[2] = r2; dst[3] = r3;
            relevant = splat_may_touch_quad(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, quad_x0, quad_y0);
        }
        if (base + WAVE + lane < range.y) {
            const float4 *src = reinterpret_cast<const float4 *>(a.rec + a.point_list[base + WAVE + lane]);
            r0 = src[0]; r1 = src[1]; r2 = src[2]; r3 = src[3];
        }
        unsigned long long todo = __ballot(relevant);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // phase B -- lanes = pixels: walk the surviving entries in list order
        const uint32_t contrib0 = base - range.x;           // entries before this chunk
        // surviving entries are taken FOUR at a time: their 16 LDS reads are issued together and the four alpha
        // evaluations (falloff + exp: the long dependent chains) are independent, so they interleave in the single
        // wave that is left on a SIMD at the tail of the launch; only the short transmittance chain stays serial.
        // Missing group members point at the all-zero record (alpha = 0: never live).
        while (todo != 0ull) {
            if (LOG) n_iter++;
            int jj[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                jj[k] = todo ? __builtin_ctzll(todo) : WAVE;
                todo = todo ? (todo & (todo - 1ull)) : 0ull;
            }
            float4 q0[4], q1[4], q2[4], q3[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                q0[k] = myq[4 * jj[k] + 0]; q1[k] = myq[4 * jj[k] + 1];     // x,y,A,B | C,opacity,depth,plane_a
                q2[k] = myq[4 * jj[k] + 2]; q3[k] = myq[4 * jj[k] + 3];     // plane_b,r,g,b | nx,ny,nz,-
            }
            float dx[4], dy[4], power[4], alpha[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                dx[k] = q0[k].x - fx; dy[k] = q0[k].y - fy;
                power[k] = falloff_power(q0[k].z, q0[k].w, q1[k].x, dx[k], dy[k]);          // forward.cu:507-508
                alpha[k] = fminf(0.99f, q1[k].y * exp_nonpositive(power[k]));
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                // Branch-free (selects only: no exec-mask juggling in the serial part).  Skip rules :512,:545 zero the
                // effective alpha; a pixel saturates BEFORE blending the entry that would drop T below 1e-4 (:549-552).
                // Invariant: T >= 1e-4 for every lane, so test_T < 1e-4 can only happen on a live lane.
                float a_eff = (power[k] > 0.0f) ? 0.f : alpha[k];
                a_eff = (alpha[k] < 1.0f / 255.0f) ? 0.f : a_eff;
                a_eff = done ? 0.f : a_eff;
                const float test_T = mul_one_minus(T, a_eff);
                const bool stop = test_T < 0.0001f;
                done = stop ? true : done;
                const float w = stop ? 0.f : a_eff * T;
                const bool blend = w != 0.f;
                const float depth = q1[k].z - (dx[k] * q1[k].w + dy[k] * q2[k].x);         // depth on the surfel plane
                D = blend ? __builtin_fmaf(depth, w, D) : D;
                C0 = blend ? __builtin_fmaf(q2[k].y, w, C0) : C0;
                C1 = blend ? __builtin_fmaf(q2[k].z, w, C1) : C1;
                C2 = blend ? __builtin_fmaf(q2[k].w, w, C2) : C2;
                N0 = blend ? __builtin_fmaf(q3[k].x, w, N0) : N0;
                N1 = blend ? __builtin_fmaf(q3[k].y, w, N1) : N1;
                N2 = blend ? __builtin_fmaf(q3[k].z, w, N2) : N2;
                T = blend ? test_T : T;
                last_contributor = blend ? contrib0 + (uint32_t)jj[k] + 1u : last_contributor;
            }
            if (__ballot(!done) == 0ull) break;              // wave-uniform: whole quad saturated
        }
        __builtin_amdgcn_wave_barrier();                     // slab is overwritten by the next chunk
    }

    if (inside) {
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
    }
    if (LOG && (threadIdx.x & 63) == 0) {
        unsigned long long *w = a.wave_log + ((size_t)tile * 4 + (threadIdx.x >> 6)) * 4;
        w[0] = t_start; w[1] = wall_clock64(); w[2] = range.y - range.x; w[3] = n_iter;
    }
}

}  // namespace

int launch_render_forward(const SoarRastParams &prm, const GeomBuf &g, const BinBuf &b, ImageBuf &img,
                          float *out_color, float *out_normal, float *out_depth, float *out_opac, hipStream_t stream)
{
    FwdArgs a;
    a.W = prm.W; a.H = prm.H;
    a.gx = (prm.W + TILE - 1) / TILE; a.gy = (prm.H + TILE - 1) / TILE;
    a.ntiles = a.gx * a.gy;
    a.normalize_depth = prm.cfg_normalize_depth;
    a.ranges = img.ranges; a.point_list = b.vals_sorted; a.rec = g.rec; a.bg = prm.bg_dev;
    a.final_T = img.final_T; a.final_D = img.final_D; a.n_contrib = img.n_contrib;
    a.out_color = out_color; a.out_normal = out_normal; a.out_depth = out_depth; a.out_opac = out_opac;
    StageTimer timer(ST_RENDER_FWD, stream);
    a.wave_log = nullptr;
    const char *log_path = getenv("SOAR_WAVE_LOG");          // diagnostic: dump per-wave timelines of ONE launch
    static int logged = 0;
    if (log_path && !logged && prm.render_front == 0) {
        logged = 1;
        const size_t nbytes = sizeof(unsigned long long) * 16 * (size_t)a.ntiles;
        SOAR_HIP_OK(hipMalloc(&a.wave_log, nbytes));
        SOAR_HIP_OK(hipMemsetAsync(a.wave_log, 0, nbytes, stream));
        hipLaunchKernelGGL(render_forward_kernel<true>, dim3(a.ntiles), dim3(256), 0, stream, a);
        SOAR_HIP_OK(hipStreamSynchronize(stream));
        unsigned long long *host = (unsigned long long *)malloc(nbytes);
        SOAR_HIP_OK(hipMemcpy(host, a.wave_log, nbytes, hipMemcpyDeviceToHost));
        FILE *f = fopen(log_path, "wb");
        if (f) { fwrite(host, 1, nbytes, f); fclose(f); }
        free(host);
        (void)hipFree(a.wave_log);
        return 0;
    }
    hipLaunchKernelGGL(render_forward_kernel<false>, dim3(a.ntiles), dim3(256), 0, stream, a);
    SOAR_LAUNCH_OK("render_forward", stream, prm.debug);
    return 0;
}

}  // namespace soar
