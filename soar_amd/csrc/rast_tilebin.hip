// rast_tilebin.hip -- tile binning without a global sort of the (tile | depth) instance keys, gfx950.
//
// Produces exactly what duplicateWithKeys + cub::DeviceRadixSort::SortPairs + identifyTileRanges produce for the
// ascending sort (DGR/cuda_rasterizer/rasterizer_impl.cu:66-124, 266-295): for every tile the list of Gaussians whose
// rectangle covers it, ordered by view depth, ties by Gaussian index (a stable sort of keys emitted in index order),
// laid out tile after tile (`point_list`), plus `ranges`.  The reference sorts R = sum(tiles_touched) 64-bit keys
// (7.6e5 at 100k Gaussians / 1080p); the library sort that does this here is ~20 launch-bound kernels, 180 us.
//
// Same result from the structure of the problem instead:
//   1. bucket_count / bucket_scatter / bucket_sort: depth order of the P (not R) Gaussians -- monotone buckets of the key
//      range (reduced by preprocess), one small sort per bucket; emits the ids and tile rectangles in depth order.
//   2. band_count / band_place: the depth-ordered rectangles are split -- stably, by ballot-prefix compaction, no atomics --
//      into one list per BAND of tile rows (a band = the rows of one super-tile row, or a few of them on very tall images).
//      A super-tile only has to look at its own band's list (~1/10 of the Gaussians at 1080p) instead of all of them.
//   3. bin_count: one workgroup per super-tile (4x4 tiles) walks its band list and counts the instances of its 16 tiles on a
//      5x5 difference grid in LDS; tile_scan: exclusive scan of the tile counts = `ranges` (+ capacity check).
//   4. bin_tiles: same walk; the rectangles that touch the super-tile are kept (in order) with the mask of the tiles they
//      cover and appended to each covered tile's list with ballot-prefix compaction: every list comes out in depth order
//      with no per-tile sort and no atomics.  One extra workgroup of the same launch builds the longest-list-first tile
//      order of the blend kernels (it only needs the tile counts).
// The descending sort (back views) keeps the library path (rast_binning.hip).
#include "soar_common.h"

#include <cstdio>
#include <cstdlib>

namespace soar {

namespace {

__device__ __forceinline__ int prefix_in_mask(unsigned long long m)
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// ---- depth order of the visible Gaussians: monotone buckets + a small sort per bucket ---------------------------------
// (key, index) pairs are unique, so any sort of the 64-bit value (key << 32 | index) is the stable order the reference
// gets from its radix sort.  Keys are spread over B buckets by a monotone map of [kmin, kmax] (both reduced by
// preprocess); a bucket holds a few dozen pairs and is sorted by one wavefront in registers.  Larger buckets are sorted by
// the whole workgroup in LDS, and the (pathological: thousands of equal depths) ones beyond that in global memory.
constexpr int BKT_MAX = 8192;            // buckets (upper bound)
constexpr int BKT_LDS = 2048;            // pairs one workgroup sorts in LDS

__device__ __forceinline__ uint32_t bucket_of(uint32_t key, uint32_t kmin, float scale, int B)
{
    // monotone in key: int->float conversion, a positive multiply and the truncation are all monotone
    return (uint32_t)min(B - 1, (int)((float)(key - kmin) * scale));
}

struct BucketCountArgs {
    int P;
    int B;
    const uint32_t *depth_key;
    const uint32_t *blk_stats;
    int nblk;
    uint32_t *header;
    uint32_t *bucket_cnt;
    uint32_t *slot;
};
__device__ __forceinline__ void bucket_count_kernel_body(int P, int B, const uint32_t *__restrict__ depth_key, const uint32_t *__restrict__ blk_stats, int nblk,
                    uint32_t *__restrict__ header, uint32_t *__restrict__ bucket_cnt, uint32_t *__restrict__ slot)
{
    // every workgroup folds the per-block statistics preprocess left behind (max of: key, ~key, x1, y1, ~x0, ~y0)
    __shared__ uint32_t red[4][BLK_STATS];
    {
        uint32_t v[BLK_STATS];
#pragma unroll
        for (int k = 0; k < BLK_STATS; k++) v[k] = 0u;
        for (int b = threadIdx.x; b < nblk; b += 256)
#pragma unroll
            for (int k = 0; k < BLK_STATS; k++) v[k] = max(v[k], blk_stats[b * BLK_STATS + k]);
#pragma unroll
        for (int k = 0; k < BLK_STATS; k++) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v[k] = max(v[k], (uint32_t)__shfl_xor((int)v[k], off));
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v[k];
        }
        __syncthreads();
        if (threadIdx.x < BLK_STATS) {
            const uint32_t m = max(max(red[0][threadIdx.x], red[1][threadIdx.x]), max(red[2][threadIdx.x], red[3][threadIdx.x]));
            red[0][threadIdx.x] = m;
            if (blockIdx.x == 0) header[H_KMAX + threadIdx.x] = m;      // H_KMAX, H_NOT_KMIN, H_X1, H_Y1, H_NOT_X0, H_NOT_Y0
        }
        __syncthreads();
    }
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const uint32_t key = depth_key[i];
    if (key == 0xFFFFFFFFu) { slot[i] = 0xFFFFFFFFu; return; }
    const uint32_t kmin = ~red[0][1], kmax = red[0][0];
    const float scale = (float)B / ((float)(kmax - kmin) + 1.0f);
    slot[i] = atomicAdd(&bucket_cnt[bucket_of(key, kmin, scale, B)], 1u);
}
__global__ void __launch_bounds__(256) bucket_count_kernel(Batch<BucketCountArgs> batch)
{
    const BucketCountArgs &a = batch.v[blockIdx.y];
    bucket_count_kernel_body(a.P, a.B, a.depth_key, a.blk_stats, a.nblk, a.header, a.bucket_cnt, a.slot);
}


struct BucketScatterArgs {
    int P;
    int B;
    const uint32_t *depth_key;
    uint32_t *header;
    const uint32_t *bucket_cnt;
    uint32_t *bucket_base;
    const uint32_t *slot;
    uint64_t *pairs;
};
__device__ __forceinline__ void bucket_scatter_kernel_body(int P, int B, const uint32_t *__restrict__ depth_key, uint32_t *__restrict__ header,
                      const uint32_t *__restrict__ bucket_cnt, uint32_t *__restrict__ bucket_base,
                      const uint32_t *__restrict__ slot, uint64_t *__restrict__ pairs)
{
    __shared__ uint32_t base[BKT_MAX];
    __shared__ uint32_t part[1024];
    const int tid = threadIdx.x;
    // exclusive prefix of the bucket counts (every workgroup recomputes it: 32 KB from L2)
    const int per = B / 1024 > 0 ? B / 1024 : 1;
    uint32_t s = 0;
    for (int k = 0; k < per; k++) {
        const int b = tid * per + k;
        if (b < B) s += bucket_cnt[b];
    }
    // inclusive scan over the 1024 threads: wavefront shuffles, then the 16 wavefront totals (two barriers instead of twenty)
    uint32_t incl = s;
    {
        const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
            if (lane >= d) incl += up;
        }
        if (lane == WAVE - 1) part[wave] = incl;
        __syncthreads();
        uint32_t wbase = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) wbase += w < wave ? part[w] : 0u;
        incl += wbase;
        __syncthreads();
        part[tid] = incl;                                   // (part[1023] = the total, read below)
        __syncthreads();
    }
    uint32_t run = incl - s;
    for (int k = 0; k < per; k++) {
        const int b = tid * per + k;
        if (b < B) {
            base[b] = run;
            if (blockIdx.x == 0) bucket_base[b] = run;
            run += bucket_cnt[b];
        }
    }
    if (blockIdx.x == 0 && tid == 1023) { bucket_base[B] = part[1023]; header[H_NVIS] = part[1023]; }
    __syncthreads();
    const uint32_t kmin = ~header[H_NOT_KMIN], kmax = header[H_KMAX];
    const float scale = (float)B / ((float)(kmax - kmin) + 1.0f);
    for (int i = blockIdx.x * 1024 + tid; i < P; i += gridDim.x * 1024) {
        const uint32_t key = depth_key[i];
        if (key == 0xFFFFFFFFu) continue;
        pairs[base[bucket_of(key, kmin, scale, B)] + slot[i]] = ((uint64_t)key << 32) | (uint32_t)i;
    }
}
__global__ void __launch_bounds__(1024) bucket_scatter_kernel(Batch<BucketScatterArgs> batch)
{
    const BucketScatterArgs &a = batch.v[blockIdx.y];
    bucket_scatter_kernel_body(a.P, a.B, a.depth_key, a.header, a.bucket_cnt, a.bucket_base, a.slot, a.pairs);
}


__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int mask)
{
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, mask), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), mask);
    return ((uint64_t)hi << 32) | lo;
}

// ascending bitonic sort of one value per lane
__device__ __forceinline__ uint64_t wave_sort64(uint64_t v, int lane)
{
#pragma unroll
    for (int k = 2; k <= WAVE; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const uint64_t o = shfl_xor_u64(v, j);
            const bool up = (lane & k) == 0, lower = (lane & j) == 0;
            const bool take_min = up == lower;
            v = take_min ? (v < o ? v : o) : (v < o ? o : v);
        }
    }
    return v;
}

struct BucketSortArgs {
    int B;
    const uint32_t *bucket_base;
    uint64_t *pairs;
    const uint2 *rect;
    uint32_t *ids_sorted;
    uint2 *rect_sorted;
};

__device__ __forceinline__ void emit_sorted(const BucketSortArgs &a, uint32_t pos, uint64_t pair)
{
    const uint32_t id = (uint32_t)pair;
    const uint2 rc = a.rect[id];
    a.ids_sorted[pos] = id;
    a.rect_sorted[pos] = rc;
}

// workgroup = 4 wavefronts = 4 consecutive buckets
__global__ void __launch_bounds__(256) bucket_sort_kernel(Batch<BucketSortArgs> batch)
{
    const BucketSortArgs &a = batch.v[blockIdx.y];
    __shared__ uint64_t lds[BKT_LDS];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // phase 1: every wavefront sorts its own bucket if it fits one value per lane
    {
        const int b = blockIdx.x * 4 + wave;
        if (b < a.B) {
            const uint32_t lo = a.bucket_base[b], n = a.bucket_base[b + 1] - lo;
            if (n > 0u && n <= (uint32_t)WAVE) {
                uint64_t v = lane < (int)n ? a.pairs[lo + lane] : ~0ull;
                v = wave_sort64(v, lane);
                if (lane < (int)n) emit_sorted(a, lo + lane, v);
            }
        }
    }
    // phase 2: the buckets that did not fit, one after the other, by the whole workgroup: normalised bitonic network
    // (every comparator ascending, first step of a merge mirrored), which sorts any n when comparators whose upper end
    // is >= n are skipped -- in LDS, or in global memory for the (pathological: thousands of equal depths) buckets
    // beyond the LDS capacity
    for (int w = 0; w < 4; w++) {
        const int b = blockIdx.x * 4 + w;
        if (b >= a.B) break;
        const uint32_t lo = a.bucket_base[b], n = a.bucket_base[b + 1] - lo;
        if (n <= (uint32_t)WAVE) continue;
        uint32_t n2 = 1;
        while (n2 < n) n2 <<= 1;
        const bool in_lds = n <= (uint32_t)BKT_LDS;
        uint64_t *p = in_lds ? lds : a.pairs + lo;
        __syncthreads();
        if (in_lds) {
            for (uint32_t i = tid; i < n; i += 256) lds[i] = a.pairs[lo + i];
            __syncthreads();
        }
        for (uint32_t k = 2; k <= n2; k <<= 1) {
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t t = tid; t < n2 / 2; t += 256) {
                    uint32_t i, o;
                    if (j == (k >> 1)) {                   // mirror step
                        const uint32_t blk = t / j, r = t % j;
                        i = blk * k + r;
                        o = blk * k + k - 1 - r;
                    } else {
                        i = 2 * t - (t & (j - 1));
                        o = i + j;
                    }
                    if (o < n) {
                        const uint64_t x = p[i], y = p[o];
                        if (x > y) { p[i] = y; p[o] = x; }
                    }
                }
                if (!in_lds) __threadfence_block();
                __syncthreads();
            }
        }
        for (uint32_t i = tid; i < n; i += 256) emit_sorted(a, lo + i, p[i]);
    }
}

// ---- band lists -----------------------------------------------------------------------------------------------------------
// band b = tile rows [b * band_rows, (b + 1) * band_rows).  band_count: per (band, 1024-chunk of the depth order) the number of
// rectangles that reach into the band; band_place: exclusive scan of that matrix in (band, chunk) order = where every chunk's
// entries of every band go, then the same ballots again to place them.  Both keep the depth order inside a band.
constexpr int BAND_MAX = 64;            // bands (upper bound: one lane per band in the per-wavefront counts)
constexpr int BAND_THREADS = 1024;
constexpr int BAND_WAVES = BAND_THREADS / WAVE;

struct BandArgs {
    int nb, band_rows, nchunk;
    uint32_t capacity;                  // entries the band arrays hold (= instances the caller's binning buffer was sized for)
    uint32_t *header;
    const uint2 *rect_sorted;
    const uint32_t *ids_sorted;
    uint32_t *band_cnt;                 // [nb][nchunk]
    uint32_t *band_info;                // [2 * BAND_MAX]: start, length of every band's list
    uint2 *band_rect;
    uint32_t *band_id;
};

// bands [b0, b1] a rectangle reaches into (b1 < b0: none)
__device__ __forceinline__ void band_span(uint2 rc, int band_rows, int &b0, int &b1)
{
    const int y0 = (int)(rc.y & 0xFFFFu), y1 = (int)(rc.y >> 16);
    b0 = y0 / band_rows;
    b1 = y1 > y0 ? (y1 - 1) / band_rows : -1;
}

// per-wavefront counts of the chunk's rectangles per band -> LDS wcnt[wave][band]
__device__ __forceinline__ void band_wave_counts(int nb, int b0, int b1, int lane, uint32_t (*wcnt)[BAND_MAX], int wave)
{
    uint32_t mine = 0;
    for (int b = 0; b < nb; b++) {
        const uint32_t c = (uint32_t)__builtin_popcountll(__ballot(b0 <= b && b <= b1));
        mine = lane == b ? c : mine;
    }
    if (lane < nb) wcnt[wave][lane] = mine;
}

__global__ void __launch_bounds__(BAND_THREADS) band_count_kernel(Batch<BandArgs> batch)
{
    const BandArgs &a = batch.v[blockIdx.y];
    __shared__ uint32_t wcnt[BAND_WAVES][BAND_MAX];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nvis = (int)a.header[H_NVIS];
    const int k = blockIdx.x * BAND_THREADS + tid;
    int b0 = 0, b1 = -1;
    if (k < nvis) band_span(a.rect_sorted[k], a.band_rows, b0, b1);
    band_wave_counts(a.nb, b0, b1, lane, wcnt, wave);
    __syncthreads();
    if (tid < a.nb) {
        uint32_t t = 0;
#pragma unroll
        for (int w = 0; w < BAND_WAVES; w++) t += wcnt[w][tid];
        a.band_cnt[(size_t)tid * a.nchunk + blockIdx.x] = t;
    }
}

__global__ void __launch_bounds__(BAND_THREADS) band_place_kernel(Batch<BandArgs> batch)
{
    const BandArgs &a = batch.v[blockIdx.y];
    __shared__ uint32_t wcnt[BAND_WAVES][BAND_MAX];
    __shared__ uint32_t part[BAND_THREADS];
    __shared__ uint32_t base[BAND_MAX];         // where this chunk's entries of every band go
    __shared__ uint32_t total_s;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nvis = (int)a.header[H_NVIS];
    const int k = blockIdx.x * BAND_THREADS + tid;
    uint2 rc = make_uint2(0u, 0u);
    int b0 = 0, b1 = -1;
    if (k < nvis) { rc = a.rect_sorted[k]; band_span(rc, a.band_rows, b0, b1); }
    band_wave_counts(a.nb, b0, b1, lane, wcnt, wave);

    // exclusive scan of band_cnt in (band, chunk) order (every workgroup redoes it: nb * nchunk <= a few thousand words from L2)
    const int n = a.nb * a.nchunk, per = (n + BAND_THREADS - 1) / BAND_THREADS;
    const int i0 = tid * per, i1 = min(n, i0 + per);
    uint32_t sum = 0;
    for (int i = i0; i < i1; i++) sum += a.band_cnt[i];
    // block-wide exclusive scan of the per-thread sums: wavefront scan by shuffles, then the 16 wavefront totals
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
        if (lane >= d) incl += up;
    }
    if (lane == WAVE - 1) part[wave] = incl;
    __syncthreads();
    uint32_t wbase = 0;
#pragma unroll
    for (int w = 0; w < BAND_WAVES; w++) wbase += w < wave ? part[w] : 0u;
    if (tid == BAND_THREADS - 1) total_s = wbase + incl;
    uint32_t run = wbase + incl - sum;
    for (int i = i0; i < i1; i++) {
        const int b = i / a.nchunk, c = i - b * a.nchunk;
        if (c == (int)blockIdx.x) base[b] = run;
        if (blockIdx.x == 0 && c == 0) a.band_info[b] = run;                    // start of band b
        run += a.band_cnt[i];
        if (blockIdx.x == 0 && c == a.nchunk - 1) a.band_info[BAND_MAX + b] = run;   // end of band b (turned into a length below)
    }
    __syncthreads();
    const uint32_t total = total_s;
    const bool fits = total <= a.capacity;
    if (blockIdx.x == 0) {
        // a band list that does not fit the caller's buffer: every band is left empty, the overflow is reported like a
        // tile-list overflow (tile_scan_kernel keeps the flag)
        if (tid < a.nb) {
            const uint32_t st = a.band_info[tid], en = a.band_info[BAND_MAX + tid];
            a.band_info[BAND_MAX + tid] = fits ? en - st : 0u;
        }
        if (tid == 0) a.header[H_BAND_OVERFLOW] = fits ? 0u : total;
    }
    if (!fits) return;
    // place: wavefront w's entries of band b follow those of the wavefronts before it
    for (int b = 0; b < a.nb; b++) {
        const bool in = b0 <= b && b <= b1;
        const unsigned long long m = __ballot(in);
        if (m == 0ull) continue;
        uint32_t woff = 0;
        for (int w = 0; w < wave; w++) woff += wcnt[w][b];
        if (in) {
            const uint32_t pos = base[b] + woff + (uint32_t)prefix_in_mask(m);
            a.band_rect[pos] = rc;
            a.band_id[pos] = a.ids_sorted[k];
        }
    }
}

constexpr int BIN_SUPER = 4;            // tiles per side of a workgroup's super-tile: one tile per wavefront
constexpr int BIN_THREADS = 1024;
constexpr int BIN_WAVES = BIN_THREADS / WAVE;
constexpr int BIN_UNROLL = 8;           // rectangles per thread and trip (independent loads in flight)
constexpr int BIN_CHUNK = BIN_THREADS * BIN_UNROLL;
constexpr int BIN_CAP = BIN_CHUNK;      // survivors buffered between two flushes: a whole trip's hits always fit (64 KB of LDS)

struct SuperTile {
    int tx0, ty0, tx1, ty1;
};
__device__ __forceinline__ SuperTile super_tile_of(int block, int gx, int gy)
{
    const int nsx = (gx + BIN_SUPER - 1) / BIN_SUPER;
    SuperTile s;
    s.tx0 = (block % nsx) * BIN_SUPER;
    s.ty0 = (block / nsx) * BIN_SUPER;
    s.tx1 = min(gx, s.tx0 + BIN_SUPER);
    s.ty1 = min(gy, s.ty0 + BIN_SUPER);
    return s;
}
__device__ __forceinline__ bool rect_hits(uint2 rc, const SuperTile &s)
{
    const int x0 = (int)(rc.x & 0xFFFFu), x1 = (int)(rc.x >> 16), y0 = (int)(rc.y & 0xFFFFu), y1 = (int)(rc.y >> 16);
    return x0 < s.tx1 && x1 > s.tx0 && y0 < s.ty1 && y1 > s.ty0;       // empty rectangles are never stored in the sorted list
}

// Pass A: instances per tile.  One workgroup per 4x4 tiles walks the depth-ordered rectangles and adds every rectangle
// that touches its tiles to a 5x5 difference grid in LDS (4 LDS atomics per hit); the grid's 2-D prefix sum is the
// number of instances of each of its 16 tiles.  Workgroups outside the bounding box of all rectangles leave at once.
// band of a super-tile: its row of super-tiles, or several rows per band on very tall images
__device__ __forceinline__ int band_of_block(int block, int gx, int band_rows)
{
    const int nsx = (gx + BIN_SUPER - 1) / BIN_SUPER;
    return ((block / nsx) * BIN_SUPER) / band_rows;
}

struct BinCountArgs {
    const uint32_t *header;
    int gx;
    int gy;
    int band_rows;
    const uint32_t *band_info;
    const uint2 *band_rect;
    uint32_t *tile_count;
};
__device__ __forceinline__ void bin_count_kernel_body(const int bx, const uint32_t *__restrict__ header, int gx, int gy, int band_rows, const uint32_t *__restrict__ band_info,
                 const uint2 *__restrict__ band_rect, uint32_t *__restrict__ tile_count)
{
    __shared__ int diff[BIN_SUPER + 1][BIN_SUPER + 1];
    const int tid = threadIdx.x;
    const SuperTile st = super_tile_of(bx, gx, gy);
    const int bx0 = (int)~header[H_NOT_X0], by0 = (int)~header[H_NOT_Y0], bx1 = (int)header[H_X1], by1 = (int)header[H_Y1];
    const int band = band_of_block(bx, gx, band_rows);
    const uint2 *__restrict__ rect_sorted = band_rect + band_info[band];      // this band's rectangles, depth order
    const int P = header[H_NVIS] ? (int)band_info[BAND_MAX + band] : 0;
    const bool inside = P > 0 && st.tx0 < bx1 && st.tx1 > bx0 && st.ty0 < by1 && st.ty1 > by0;
    if (tid < (BIN_SUPER + 1) * (BIN_SUPER + 1)) (&diff[0][0])[tid] = 0;
    lds_barrier();
    if (inside) {
        uint2 rc[BIN_UNROLL];
#pragma unroll
        for (int j = 0; j < BIN_UNROLL; j++) rc[j] = j * BIN_THREADS + tid < P ? rect_sorted[j * BIN_THREADS + tid] : make_uint2(0u, 0u);
        for (int base = 0; base < P; base += BIN_CHUNK) {
            uint2 nx[BIN_UNROLL];                              // next trip's rectangles are in flight while this one is counted
#pragma unroll
            for (int j = 0; j < BIN_UNROLL; j++) {
                const int k = base + BIN_CHUNK + j * BIN_THREADS + tid;
                nx[j] = k < P ? rect_sorted[k] : make_uint2(0u, 0u);
            }
#pragma unroll
            for (int j = 0; j < BIN_UNROLL; j++) {
                if (rect_hits(rc[j], st)) {
                    const int x0 = max((int)(rc[j].x & 0xFFFFu), st.tx0) - st.tx0, x1 = min((int)(rc[j].x >> 16), st.tx1) - st.tx0;
                    const int y0 = max((int)(rc[j].y & 0xFFFFu), st.ty0) - st.ty0, y1 = min((int)(rc[j].y >> 16), st.ty1) - st.ty0;
                    atomicAdd(&diff[y0][x0], 1);
                    atomicAdd(&diff[y0][x1], -1);
                    atomicAdd(&diff[y1][x0], -1);
                    atomicAdd(&diff[y1][x1], 1);
                }
                rc[j] = nx[j];
            }
        }
    }
    lds_barrier();
    if (tid == 0) {
        for (int y = 0; y < BIN_SUPER; y++)
            for (int x = 1; x < BIN_SUPER; x++) diff[y][x] += diff[y][x - 1];
        for (int x = 0; x < BIN_SUPER; x++)
            for (int y = 1; y < BIN_SUPER; y++) diff[y][x] += diff[y - 1][x];
    }
    lds_barrier();
    if (tid < BIN_SUPER * BIN_SUPER) {
        const int tx = st.tx0 + (tid & 3), ty = st.ty0 + (tid >> 2);
        if (tx < st.tx1 && ty < st.ty1) tile_count[ty * gx + tx] = (uint32_t)diff[tid >> 2][tid & 3];
    }
}
__global__ void __launch_bounds__(BIN_THREADS) bin_count_kernel(Batch<BinCountArgs> batch)
{
    int frame, bx;
    batch_interleave1(frame, bx);
    const BinCountArgs &a = batch.v[frame];
    bin_count_kernel_body(bx, a.header, a.gx, a.gy, a.band_rows, a.band_info, a.band_rect, a.tile_count);
}


// One workgroup: exclusive scan of the tile counts in tile order = ranges (untouched tiles keep (0,0) like
// identifyTileRanges, rasterizer_impl.cu:287-295).
// `capacity` = instances the caller's binning buffer holds: when the lists would not fit, every range is left empty
// (nothing is written or rendered) and header[H_OVERFLOW] reports the number that was needed.
struct TileScanArgs {
    int T;
    const uint32_t *tile_count;
    uint2 *ranges;
    uint32_t capacity;
    uint32_t *header;
};
__device__ __forceinline__ void tile_scan_kernel_body(int T, const uint32_t *__restrict__ tile_count, uint2 *__restrict__ ranges,
                                                         uint32_t capacity, uint32_t *__restrict__ header)
{
    __shared__ uint32_t part[16];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // every thread owns `per` consecutive tiles (a multiple of 4: 16-byte loads, several in flight -- one load per trip made this
    // one-workgroup kernel a chain of load latencies: 51 us for the 32 400 tiles of a 4K frame); the counts stay in registers
    // between the two passes when they fit (per <= 32: images up to 4K)
    const int per = ((T + 1023) / 1024 + 3) / 4 * 4;
    const int t0 = tid * per, t1 = min(T, t0 + per);
    constexpr int KEEP = 32;
    uint32_t cnt[KEEP];
    const bool keep = per <= KEEP && (T & 3) == 0;           // (whole uint4 groups: t1 - t0 is a multiple of 4 as well)
    uint32_t s = 0;
    if (keep) {
#pragma unroll
        for (int k = 0; k < KEEP; k += 4) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (k < per && t0 + k < t1) v = *reinterpret_cast<const uint4 *>(tile_count + t0 + k);
            cnt[k] = v.x; cnt[k + 1] = v.y; cnt[k + 2] = v.z; cnt[k + 3] = v.w;
            s += (v.x + v.y) + (v.z + v.w);
        }
    } else {
        for (int t = t0; t < t1; t++) s += tile_count[t];
    }
    uint32_t incl = s;                         // wavefront scan by shuffles, then the 16 wavefront totals
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
        if (lane >= d) incl += up;
    }
    if (lane == WAVE - 1) part[wave] = incl;
    __syncthreads();
    uint32_t wbase = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) { wbase += w < wave ? part[w] : 0u; total += part[w]; }
    const uint32_t band_over = header[H_BAND_OVERFLOW];
    const bool fits = total <= capacity && band_over == 0u;
    if (tid == 0) { header[H_TOTAL] = total; header[H_OVERFLOW] = fits ? 0u : max(total, band_over); }
    uint32_t run = wbase + incl - s;
    if (keep) {
#pragma unroll
        for (int k = 0; k < KEEP; k += 2) {
            if (k < per && t0 + k < t1) {                     // two ranges = one 16-byte store
                const uint32_t c0 = cnt[k], c1 = cnt[k + 1];
                const uint2 r0 = (c0 && fits) ? make_uint2(run, run + c0) : make_uint2(0u, 0u);
                run += c0;
                const uint2 r1 = (c1 && fits) ? make_uint2(run, run + c1) : make_uint2(0u, 0u);
                run += c1;
                *reinterpret_cast<uint4 *>(ranges + t0 + k) = make_uint4(r0.x, r0.y, r1.x, r1.y);
            }
        }
    } else {
        for (int t = t0; t < t1; t++) {
            const uint32_t c = tile_count[t];
            ranges[t] = (c && fits) ? make_uint2(run, run + c) : make_uint2(0u, 0u);
            run += c;
        }
    }
}
__global__ void __launch_bounds__(1024) tile_scan_kernel(Batch<TileScanArgs> batch)
{
    const TileScanArgs &a = batch.v[blockIdx.y];
    tile_scan_kernel_body(a.T, a.tile_count, a.ranges, a.capacity, a.header);
}


// 16-bit mask of the tiles of the super-tile a rectangle covers (bit 4*y + x)
__device__ __forceinline__ uint32_t cover_mask(uint2 rc, const SuperTile &s)
{
    const int x0 = max((int)(rc.x & 0xFFFFu), s.tx0) - s.tx0, x1 = min((int)(rc.x >> 16), s.tx1) - s.tx0;
    const int y0 = max((int)(rc.y & 0xFFFFu), s.ty0) - s.ty0, y1 = min((int)(rc.y >> 16), s.ty1) - s.ty0;
    const uint32_t row = (1u << x1) - (1u << x0), ym = (1u << y1) - (1u << y0);
    const uint32_t spread = (ym & 1u) | ((ym & 2u) << 3) | ((ym & 4u) << 6) | ((ym & 8u) << 9);
    return row * spread;                       // row < 16: no carries between the nibbles
}

// Pass B: the lists.  Same walk; the rectangles that touch the 4x4 tiles are kept (in order) in LDS with the mask of
// the tiles they cover.  A flush hands the buffered survivors out in slabs of 64 to the wavefronts; per slab and tile
// a ballot gives the number of entries, a scan over the slabs gives every slab its place in each tile's list, and
// the ids are appended with ballot-prefix compaction: every list comes out in depth order, without atomics or sorts.
struct BinTilesArgs {
    const uint32_t *header;
    int gx;
    int gy;
    int band_rows;
    const uint32_t *band_info;
    const uint2 *band_rect;
    const uint32_t *band_id;
    const uint2 *ranges;
    uint32_t *point_list;
    int nblocks_tiles;
    const uint32_t *tile_count;
    uint32_t *tile_order;
    const float *bg;
    int normalize_depth;
    uint32_t *bg_state;
    unsigned long long *dbg;
    uint32_t *tile_xy;       // BinBuf::tile_xy
};
__device__ __forceinline__ void bin_tiles_kernel_body(const int bx, const uint32_t *__restrict__ header, int gx, int gy, int band_rows, const uint32_t *__restrict__ band_info,
                 const uint2 *__restrict__ band_rect, const uint32_t *__restrict__ band_id, const uint2 *__restrict__ ranges,
                 uint32_t *__restrict__ point_list, int nblocks_tiles, const uint32_t *__restrict__ tile_count,
                 uint32_t *__restrict__ tile_order, const float *__restrict__ bg, int normalize_depth,
                 uint32_t *__restrict__ bg_state, unsigned long long *__restrict__ dbg, uint32_t *__restrict__ tile_xy)
{
    const unsigned long long dbg_t0 = dbg ? wall_clock64() : 0ull;
    unsigned long long dbg_flush = 0;
    int dbg_nflush = 0, dbg_hits = 0;
    if (bx == nblocks_tiles) {
        // the extra workgroup: longest-list-first order of the tiles for the blend launches (needs the counts only)
        const int T = gx * gy;
        tile_order_block(T, (T + 7) / 8 * 8, tile_count, ranges, tile_order, bg, normalize_depth, bg_state);
        return;
    }
    constexpr int NT = BIN_SUPER * BIN_SUPER;
    __shared__ uint32_t surv_mask[BIN_CAP], surv_id[BIN_CAP];
    __shared__ uint32_t tile_cursor[NT];
    __shared__ uint32_t wave_cnt[2][BIN_WAVES];
    __shared__ int any_s;
    const int band = band_of_block(bx, gx, band_rows);
    const uint2 *__restrict__ rect_sorted = band_rect + band_info[band];      // this band's rectangles / ids, depth order
    const uint32_t *__restrict__ ids_sorted = band_id + band_info[band];
    const int P = header[H_NVIS] ? (int)band_info[BAND_MAX + band] : 0;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const SuperTile st = super_tile_of(bx, gx, gy);

    if (tid == 0) any_s = 0;
    lds_barrier();
    if (tid < NT) {
        const int tx = st.tx0 + (tid & 3), ty = st.ty0 + (tid >> 2);
        uint2 rg = make_uint2(0u, 0u);
        if (tx < st.tx1 && ty < st.ty1) rg = ranges[ty * gx + tx];
        tile_cursor[tid] = rg.x;
        if (rg.y != rg.x) any_s = 1;
    }
    lds_barrier();
    if (!any_s) return;                           // no Gaussian touches these 16 tiles

    // Flush: wavefront t owns tile t of the super-tile (16 wavefronts, 16 tiles) and walks the buffered survivors slab by
    // slab; the entries that cover its tile are appended to the tile's list by ballot-prefix compaction at the wavefront's own
    // running cursor -- one pass, no counting phase, no barrier between slabs; lists stay in depth order.
    static_assert(BIN_WAVES == NT, "one wavefront per tile of the super-tile");
    int nbuf = 0;
    uint32_t cursor = tile_cursor[wave];          // next free position of this wavefront's tile list
    const uint32_t my_xy = ((uint32_t)(st.ty0 + (wave >> 2)) << 16) | (uint32_t)(st.tx0 + (wave & 3));   // ... and whose list it is (block masks)
    auto flush = [&]() {
        const unsigned long long f0 = dbg ? wall_clock64() : 0ull;
        dbg_nflush++; dbg_hits += nbuf;
        lds_barrier();                                // the survivors of the last trip are in LDS
        const int nslab = (nbuf + WAVE - 1) / WAVE;
        constexpr int FU = 8;                          // slabs per round: their LDS reads are in flight together
        for (int sl = 0; sl < nslab; sl += FU) {
            uint32_t m[FU], id[FU];
#pragma unroll
            for (int u = 0; u < FU; u++) {
                const int e = (sl + u) * WAVE + lane;
                m[u] = e < nbuf ? surv_mask[e] : 0u;
                id[u] = surv_id[min(e, BIN_CAP - 1)];
            }
#pragma unroll
            for (int u = 0; u < FU; u++) {
                const bool h = (m[u] >> wave) & 1u;
                const unsigned long long bal = __ballot(h);
                if (h) {
                    const uint32_t at = cursor + (uint32_t)prefix_in_mask(bal);
                    point_list[at] = id[u];
                    tile_xy[at] = my_xy;
                }
                cursor += (uint32_t)__builtin_popcountll(bal);
            }
        }
        lds_barrier();                                // the buffer may be overwritten
        nbuf = 0;
        if (dbg) dbg_flush += wall_clock64() - f0;
    };

    // wavefront w scans the contiguous slice [base + w*512, base + (w+1)*512) of every trip: survivors stay in depth
    // order when the wavefronts append one after the other.  The next trip's loads are issued before this one is used.
    uint2 rc[BIN_UNROLL];
    uint32_t id[BIN_UNROLL];
    {
        const int k0 = wave * (WAVE * BIN_UNROLL) + lane;
#pragma unroll
        for (int j = 0; j < BIN_UNROLL; j++) {
            const int k = k0 + j * WAVE;
            rc[j] = k < P ? rect_sorted[k] : make_uint2(0u, 0u);
            id[j] = k < P ? ids_sorted[k] : 0u;
        }
    }
    auto append = [&](uint32_t at, const unsigned long long (&hits)[BIN_UNROLL]) {
#pragma unroll
        for (int j = 0; j < BIN_UNROLL; j++) {
            if ((hits[j] >> lane) & 1ull) {
                const uint32_t pos = at + (uint32_t)prefix_in_mask(hits[j]);
                surv_mask[pos] = cover_mask(rc[j], st);
                surv_id[pos] = id[j];
            }
            at += (uint32_t)__builtin_popcountll(hits[j]);
        }
    };
    int parity = 0;
    for (int base = 0; base < P; base += BIN_CHUNK, parity ^= 1) {
        uint2 nrc[BIN_UNROLL];
        uint32_t nid[BIN_UNROLL];
        {
            const int k0 = base + BIN_CHUNK + wave * (WAVE * BIN_UNROLL) + lane;
#pragma unroll
            for (int j = 0; j < BIN_UNROLL; j++) {
                const int k = k0 + j * WAVE;
                nrc[j] = k < P ? rect_sorted[k] : make_uint2(0u, 0u);
                nid[j] = k < P ? ids_sorted[k] : 0u;
            }
        }
        unsigned long long hits[BIN_UNROLL];
        uint32_t mine = 0;
#pragma unroll
        for (int j = 0; j < BIN_UNROLL; j++) {
            hits[j] = __ballot(rect_hits(rc[j], st));
            mine += (uint32_t)__builtin_popcountll(hits[j]);
        }
        if (lane == 0) wave_cnt[parity][wave] = mine;
        lds_barrier();
        uint32_t off = 0, total = 0;
#pragma unroll
        for (int w = 0; w < BIN_WAVES; w++) {
            const uint32_t c = wave_cnt[parity][w];
            off += w < wave ? c : 0u;
            total += c;
        }
        if (total != 0u) {
            if (nbuf + (int)total > BIN_CAP) flush();          // total <= BIN_CHUNK = BIN_CAP: after a flush a trip always fits
            append((uint32_t)nbuf + off, hits);
            nbuf += (int)total;
        }
#pragma unroll
        for (int j = 0; j < BIN_UNROLL; j++) { rc[j] = nrc[j]; id[j] = nid[j]; }
    }
    flush();
    if (dbg && tid == 0) {
        unsigned long long *w = dbg + (size_t)bx * 4;
        w[0] = wall_clock64() - dbg_t0; w[1] = dbg_flush; w[2] = ((unsigned long long)dbg_nflush << 32) | (unsigned)dbg_hits; w[3] = (unsigned long long)P;
    }
}
__global__ void __launch_bounds__(BIN_THREADS) bin_tiles_kernel(Batch<BinTilesArgs> batch)
{
    int frame, bx;
    batch_interleave1(frame, bx);
    const BinTilesArgs &a = batch.v[frame];
    bin_tiles_kernel_body(bx, a.header, a.gx, a.gy, a.band_rows, a.band_info, a.band_rect, a.band_id, a.ranges, a.point_list, a.nblocks_tiles, a.tile_count, a.tile_order, a.bg, a.normalize_depth, a.bg_state, a.dbg, a.tile_xy);
}


}  // namespace

static int bucket_count_for(int32_t P)
{
    int B = 256;
    while (B < BKT_MAX && B * 16 < P) B <<= 1;          // ~16 Gaussians per bucket
    return B;
}

// geometry stage: bucket counts and the scatter of the (key, index) pairs
int launch_depth_buckets(const SoarRastParams &prm, GeomBuf &g, hipStream_t stream)
{
    const int B = bucket_count_for(prm.P);
    const int nblk = (prm.P + 255) / 256;               // = preprocess grid: one statistics row per block
    StageTimer timer(ST_SORT, stream);
    const BucketCountArgs ca = {prm.P, B, g.depth_key, g.blk_stats, nblk, g.header, g.bucket_cnt, g.sort_slot};
    SOAR_LAUNCH_BATCHED(bucket_count_kernel, dim3(nblk), dim3(256), 0, stream, ca);
    const BucketScatterArgs sa = {prm.P, B, g.depth_key, g.header, g.bucket_cnt, g.bucket_base, g.sort_slot, g.sort_pairs};
    SOAR_LAUNCH_BATCHED(bucket_scatter_kernel, dim3(min(64, (prm.P + 1023) / 1024)), dim3(1024), 0, stream, sa);
    SOAR_LAUNCH_OK("depth_buckets", stream, prm.debug);
    return 0;
}

int launch_tile_binning(const SoarRastParams &prm, GeomBuf &g, BinBuf &b, ImageBuf &img, int64_t capacity, hipStream_t stream)
{
    const int gx = (prm.W + TILE - 1) / TILE, gy = (prm.H + TILE - 1) / TILE;
    const int nsx = (gx + BIN_SUPER - 1) / BIN_SUPER, nsy = (gy + BIN_SUPER - 1) / BIN_SUPER;
    {
        StageTimer timer(ST_SORT, stream);
        BucketSortArgs a;
        a.B = bucket_count_for(prm.P);
        a.bucket_base = g.bucket_base; a.pairs = g.sort_pairs; a.rect = g.rect; a.ids_sorted = g.ids_sorted;
        a.rect_sorted = g.rect_sorted;
        SOAR_LAUNCH_BATCHED(bucket_sort_kernel, dim3((a.B + 3) / 4), dim3(256), 0, stream, a);
    }
    SOAR_LAUNCH_OK("bucket_sort", stream, prm.debug);
    // band lists: the band arrays live in the key / value scratch of the binning buffer (only the descending path and the key
    // export use it for keys): at most one entry per (band, Gaussian) <= one per (tile, Gaussian) instance <= capacity
    const int band_rows = BIN_SUPER * ((nsy + BAND_MAX - 1) / BAND_MAX);
    BandArgs ba;
    ba.band_rows = band_rows;
    ba.nb = (gy + band_rows - 1) / band_rows;
    ba.nchunk = (prm.P + BAND_THREADS - 1) / BAND_THREADS;
    ba.capacity = (uint32_t)(capacity > 0xFFFFFFFFll ? 0xFFFFFFFFll : capacity);
    ba.header = g.header; ba.rect_sorted = g.rect_sorted; ba.ids_sorted = g.ids_sorted; ba.band_cnt = g.band_cnt;
    ba.band_info = g.band_info;
    ba.band_rect = reinterpret_cast<uint2 *>(b.keys_unsorted); ba.band_id = b.vals_unsorted;
    {
        StageTimer timer(ST_RANGES, stream);
        SOAR_LAUNCH_BATCHED(band_count_kernel, dim3(ba.nchunk), dim3(BAND_THREADS), 0, stream, ba);
        SOAR_LAUNCH_BATCHED(band_place_kernel, dim3(ba.nchunk), dim3(BAND_THREADS), 0, stream, ba);
        const BinCountArgs bc = {g.header, gx, gy, band_rows, g.band_info, ba.band_rect, img.tile_count};
        SOAR_LAUNCH_BATCHED(bin_count_kernel, dim3(nsx * nsy), dim3(BIN_THREADS), 0, stream, bc);
        const TileScanArgs ts = {gx * gy, img.tile_count, img.ranges, ba.capacity, g.header};
        SOAR_LAUNCH_BATCHED(tile_scan_kernel, dim3(1), dim3(1024), 0, stream, ts);
    }
    SOAR_LAUNCH_OK("tile_ranges", stream, prm.debug);
    {
        StageTimer timer(ST_EMIT_KEYS, stream);
        unsigned long long *dbg = nullptr;
        static int dbg_left = getenv("SOAR_BIN_LOG") ? 1 : 0;          // diagnostic: per-workgroup timings of ONE launch
        if (dbg_left > 0 && prm.W >= 1920) {
            dbg_left = 0;
            const size_t nw = (size_t)(nsx * nsy + 1) * 4;
            SOAR_HIP_OK(hipMalloc(&dbg, 8 * nw));
            SOAR_HIP_OK(hipMemsetAsync(dbg, 0, 8 * nw, stream));
            const BinTilesArgs bt = {g.header, gx, gy, band_rows, g.band_info, ba.band_rect, ba.band_id, img.ranges, b.vals_sorted, nsx * nsy,
                                     img.tile_count, img.tile_order, prm.bg_dev, prm.cfg_normalize_depth, img.bg_state, dbg, b.tile_xy};
            SOAR_LAUNCH_BATCHED(bin_tiles_kernel, dim3(nsx * nsy + 1), dim3(BIN_THREADS), 0, stream, bt);
            SOAR_HIP_OK(hipStreamSynchronize(stream));
            unsigned long long *h = (unsigned long long *)malloc(8 * nw);
            SOAR_HIP_OK(hipMemcpy(h, dbg, 8 * nw, hipMemcpyDeviceToHost));
            (void)hipFree(dbg);
            // the five slowest workgroups
            for (int rep = 0; rep < 5; rep++) {
                int best = -1;
                for (int i = 0; i < nsx * nsy; i++) if (best < 0 || h[i * 4] > h[best * 4]) best = i;
                fprintf(stderr, "[bin_tiles] WG %d: %.1f us total, %.1f us in %llu flushes, %llu survivors of %llu band entries\n", best,
                        h[best * 4] / 100.0, h[best * 4 + 1] / 100.0, h[best * 4 + 2] >> 32, h[best * 4 + 2] & 0xFFFFFFFFull, h[best * 4 + 3]);
                h[best * 4] = 0;
            }
            free(h);
            return 0;
        }
        const BinTilesArgs bt = {g.header, gx, gy, band_rows, g.band_info, ba.band_rect, ba.band_id, img.ranges, b.vals_sorted, nsx * nsy,
                                 img.tile_count, img.tile_order, prm.bg_dev, prm.cfg_normalize_depth, img.bg_state, dbg, b.tile_xy};
        SOAR_LAUNCH_BATCHED(bin_tiles_kernel, dim3(nsx * nsy + 1), dim3(BIN_THREADS), 0, stream, bt);
    }
    SOAR_LAUNCH_OK("bin_tiles", stream, prm.debug);
    return 0;
}

}  // namespace soar
