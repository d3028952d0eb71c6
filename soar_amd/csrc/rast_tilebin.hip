// rast_tilebin.hip -- tile binning without a global sort of the (tile | depth) instance keys, gfx950.
//
// Produces exactly what duplicateWithKeys + cub::DeviceRadixSort::SortPairs + identifyTileRanges produce for the
// ascending sort (DGR/cuda_rasterizer/rasterizer_impl.cu:66-124, 266-295): for every tile the list of Gaussians whose
// rectangle covers it, ordered by view depth, ties by Gaussian index (a stable sort of keys emitted in index order),
// plus `ranges`.  (The reference lays the lists out tile after tile; here every tile's list is contiguous and `ranges` says where,
// but the tiles follow each other in the order their workgroups reserved room -- soar_rast_export_state re-packs for inspection.)
// The reference sorts R = sum(tiles_touched) 64-bit keys (7.6e5 at 100k Gaussians / 1080p); the library sort that does this here
// is ~20 launch-bound kernels, 180 us.
//
// Same result from the structure of the problem instead:
//   1. bucket_count / bucket_scatter / bucket_sort: depth order of the P (not R) Gaussians -- monotone buckets of the key
//      range (reduced by preprocess), one small sort per bucket; emits the ids and tile rectangles in depth order.
//   2. band_count / band_place: the depth-ordered rectangles are split -- stably, by ballot-prefix compaction, no atomics --
//      into one list per BAND of tile rows (a band = the rows of one super-tile row, or a few of them on very tall images).
//      A super-tile only has to look at its own band's list (~1/10 of the Gaussians at 1080p) instead of all of them.
//   3. bin_tiles: one workgroup per super-tile (4x4 tiles) walks its band list: every wavefront a contiguous slice of it, keeping
//      the rectangles that touch the super-tile (in order) in a ring of its own and counting them per tile; ONE atomic add reserves room
//      for the 16 lists (ranges, capacity check); then every wavefront appends its kept entries to each covered tile's list with
//      ballot-prefix compaction: every list comes out in depth order with no per-tile sort and no atomics on the lists.
//      (Rounds 1-2: a counting launch on an LDS difference grid, a one-workgroup scan of the tile counts, then the placing launch.)
//   The longest-list-first tile order of the blend kernels needs every tile's count: one workgroup of the NEXT launch builds it
//   (block_mask_kernel, rast_blockmask.hip).
// Back views (descending sort) take the same path with flipped keys; the 64-bit key sort (rast_binning.hip) only serves the key export
// and the empty case.
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
constexpr int BKT_LDS = 2048;            // pairs one workgroup sorts in LDS
constexpr int BAND_MAX = 64;            // bands of tile rows (upper bound: one lane per band in the per-wavefront counts)
constexpr int BIN_SUPER = 4;            // tiles per side of a super-tile, the unit of bin_tiles_kernel's work

__device__ __forceinline__ uint32_t bucket_of(uint32_t key, uint32_t kmin, float scale, int B)
{
    // monotone in key: int->float conversion, a positive multiply and the truncation are all monotone
    return (uint32_t)min(B - 1, (int)((float)(key - kmin) * scale));
}

// Places in the buckets without global atomics.  A workgroup takes BKT_CHUNK consecutive Gaussians and counts them into the B
// bucket counters in LDS (32 KB): the value an LDS atomic returns is the Gaussian's place among the workgroup's members of its
// bucket.  The counters go to row `w` of a [workgroups][B] matrix; the workgroup that finishes last turns the matrix into
// where every workgroup's members of every bucket start (exclusive scan over the buckets of the column sums, then down the
// columns), so that the scatter is one gather per Gaussian: pairs[start[w][bucket] + place].  (Rounds 1-2: one returning
// global atomic per Gaussian on 8192 counters -- 31 us per 4-frame step at C3 -- and every workgroup of the scatter recomputing
// the scan of the counters.)
constexpr int BKT_CHUNK = 16384;         // Gaussians per counting workgroup
constexpr int BKT_GROUPS = BKT_MAX / 1024; // groups of 1024 buckets: one scanning workgroup each
// Back views (SoarRastParams.sort_descending: the reference's SortPairsDescending, rasterizer_impl.cu:277-285) are the same order of
// the flipped keys: ~key ascending = key descending, and pairs of equal depth still come out in index order -- what a stable
// descending sort of the (tile | depth) keys leaves inside a tile.  Only the two kernels that look at a key know about it.
struct BucketCountArgs {
    int P;
    int B;
    int descending;
    const uint32_t *depth_key;
    const uint32_t *blk_stats;
    int nblk;
    uint32_t *header;
    uint32_t *bucket_mat;    // [ceil(P / BKT_CHUNK)][B]: counts, then starts
    uint32_t *bucket_base;   // [B + 1]: start of every bucket
    uint32_t *slot;          // [P] place among the workgroup's members of the bucket
    uint32_t *band_info;     // (the scan launch resets the bands' column extents for band_count)
};
__global__ void __launch_bounds__(1024) bucket_count_kernel(Batch<BucketCountArgs> batch)
{
    const BucketCountArgs &a = batch.v[blockIdx.y];
    __shared__ uint32_t cnt[BKT_MAX];
    __shared__ uint32_t red[16][BLK_STATS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, B = a.B;
    for (int b = tid; b < B; b += 1024) cnt[b] = 0u;
    // every workgroup folds the per-block statistics preprocess left behind (max of: key, ~key, x1, y1, ~x0, ~y0)
    {
        uint32_t v[BLK_STATS];
#pragma unroll
        for (int k = 0; k < BLK_STATS; k++) v[k] = 0u;
        for (int b = tid; b < a.nblk; b += 1024)
#pragma unroll
            for (int k = 0; k < BLK_STATS; k++) v[k] = max(v[k], a.blk_stats[b * BLK_STATS + k]);
#pragma unroll
        for (int k = 0; k < BLK_STATS; k++) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v[k] = max(v[k], (uint32_t)__shfl_xor((int)v[k], off));
            if (lane == 0) red[wave][k] = v[k];
        }
        __syncthreads();
        if (tid < BLK_STATS) {
            uint32_t m = 0u;
#pragma unroll
            for (int w = 0; w < 16; w++) m = max(m, red[w][tid]);
            red[0][tid] = m;
            if (blockIdx.x == 0) a.header[H_KMAX + tid] = m;      // H_KMAX, H_NOT_KMIN, H_X1, H_Y1, H_NOT_X0, H_NOT_Y0
        }
        __syncthreads();
    }
    // (statistics: max key, max ~key.  Of the flipped keys the largest is max ~key and the smallest ~(max key))
    const uint32_t kmin = a.descending ? ~red[0][0] : ~red[0][1], kmax = a.descending ? red[0][1] : red[0][0];
    const float scale = (float)B / ((float)(kmax - kmin) + 1.0f);
    const int i0 = (int)blockIdx.x * BKT_CHUNK;
    constexpr int PER = BKT_CHUNK / 1024;
    uint32_t key[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int i = i0 + k * 1024 + tid;
        key[k] = a.depth_key[min(i, a.P - 1)];
        if (i >= a.P) key[k] = 0xFFFFFFFFu;
    }
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int i = i0 + k * 1024 + tid;
        if (i < a.P) a.slot[i] = key[k] == 0xFFFFFFFFu ? 0xFFFFFFFFu : atomicAdd(&cnt[bucket_of(a.descending ? ~key[k] : key[k], kmin, scale, B)], 1u);
    }
    __syncthreads();
    uint32_t *row = a.bucket_mat + (size_t)blockIdx.x * B;
    uint32_t *gtot = a.bucket_mat + (size_t)gridDim.x * B + (size_t)blockIdx.x * BKT_GROUPS;   // members per group of 1024 buckets
    static_assert(BKT_GROUPS <= 16 * BLK_STATS, "the group totals reuse the statistics' LDS");
    if (tid < BKT_GROUPS) (&red[0][0])[tid] = 0u;  // (the statistics are not needed any more)
    __syncthreads();
    for (int b = tid, grp = 0; b - tid < B; b += 1024, grp++) {
        uint32_t c = b < B ? cnt[b] : 0u;
        if (b < B) row[b] = c;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) c += (uint32_t)__shfl_xor((int)c, off);
        if (lane == 0 && c) atomicAdd(&(&red[0][0])[grp], c);
    }
    __syncthreads();
    if (tid < BKT_GROUPS) gtot[tid] = (&red[0][0])[tid];
}

// The [workgroups][B] counts become starts: one workgroup per 1024 consecutive buckets, one bucket per thread (consecutive threads
// read consecutive cells of a row).  What lies before the group comes from the group totals the counting workgroups left behind.
// (A launch of its own rather than "the last counting workgroup does it": handing data from one workgroup to another inside a
// launch takes a device-scope release and a chain of acquire loads.)
__global__ void __launch_bounds__(1024) bucket_scan_kernel(Batch<BucketCountArgs> batch)
{
    const BucketCountArgs &a = batch.v[blockIdx.y];
    __shared__ uint32_t part[16];
    __shared__ uint32_t before_s, all_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, B = a.B, NW = (a.P + BKT_CHUNK - 1) / BKT_CHUNK;
    const int grp = (int)blockIdx.x, b = grp * 1024 + tid;
    if (grp * 1024 >= B) return;
    if (tid == 0) { before_s = 0u; all_s = 0u; }
    uint32_t s = 0;
    if (b < B)
        for (int w = 0; w < NW; w++) s += a.bucket_mat[(size_t)w * B + b];
    __syncthreads();
    {   // members of the groups before this one (and of all groups: the number of visible Gaussians)
        const uint32_t *gtot = a.bucket_mat + (size_t)NW * B;
        uint32_t before = 0, all = 0;
        for (int k = tid; k < NW * BKT_GROUPS; k += 1024) {
            const uint32_t c = gtot[k];
            all += c;
            before += (k % BKT_GROUPS) < grp ? c : 0u;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            before += (uint32_t)__shfl_xor((int)before, off);
            all += (uint32_t)__shfl_xor((int)all, off);
        }
        if (lane == 0 && all) { atomicAdd(&before_s, before); atomicAdd(&all_s, all); }
    }
    uint32_t incl = s;                        // inclusive scan over the 1024 threads: wavefront shuffles, then the 16 wavefront totals
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
    uint32_t run = before_s + wbase + incl - s;
    if (b < B) {
        a.bucket_base[b] = run;
        for (int w0 = 0; w0 < NW; w0 += 8) {
            uint32_t c[8];
#pragma unroll
            for (int u = 0; u < 8; u++) c[u] = w0 + u < NW ? a.bucket_mat[(size_t)(w0 + u) * B + b] : 0u;
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (w0 + u < NW) a.bucket_mat[(size_t)(w0 + u) * B + b] = run;
                run += c[u];
            }
        }
    }
    if (grp == 0 && tid == 0) { a.bucket_base[B] = all_s; a.header[H_NVIS] = all_s; }
    if (grp == 0 && tid < BAND_MAX) { a.band_info[2 * BAND_MAX + tid] = 0xFFFFFFFFu; a.band_info[3 * BAND_MAX + tid] = 0u; }
}

struct BucketScatterArgs {
    int P;
    int B;
    int descending;
    const uint32_t *depth_key;
    const uint32_t *header;
    const uint32_t *bucket_mat;
    const uint32_t *slot;
    uint64_t *pairs;
};
__global__ void __launch_bounds__(256) bucket_scatter_kernel(Batch<BucketScatterArgs> batch)
{
    const BucketScatterArgs &a = batch.v[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.P) return;
    const uint32_t key_raw = a.depth_key[i], place = a.slot[i];
    if (key_raw == 0xFFFFFFFFu) return;
    const uint32_t key = a.descending ? ~key_raw : key_raw;
    const uint32_t kmin = a.descending ? ~a.header[H_KMAX] : ~a.header[H_NOT_KMIN], kmax = a.descending ? a.header[H_NOT_KMIN] : a.header[H_KMAX];
    const float scale = (float)a.B / ((float)(kmax - kmin) + 1.0f);
    const uint32_t start = a.bucket_mat[(size_t)(i / BKT_CHUNK) * a.B + bucket_of(key, kmin, scale, a.B)];
    a.pairs[start + place] = ((uint64_t)key << 32) | (uint32_t)i;
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

// workgroup = 4 wavefronts, wavefront = BKT_RUN consecutive buckets
#ifndef SOAR_BKT_RUN
#define SOAR_BKT_RUN 2       // (1 / 2 / 4 / 8 buckets per wavefront: 46.8 / 41.7 / 45.4 / 53.6 us for the depth order of a 4-frame step at C3)
#endif
constexpr int BKT_RUN = SOAR_BKT_RUN;
__global__ void __launch_bounds__(256) bucket_sort_kernel(Batch<BucketSortArgs> batch)
{
    const BucketSortArgs &a = batch.v[blockIdx.y];
    __shared__ uint64_t lds[BKT_LDS];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // phase 1: consecutive buckets are consecutive in `pairs` and ordered among each other (the bucket map is monotone): a run of
    // buckets that fits one value per lane is sorted as ONE sequence -- with ~12 pairs per bucket runs of two nearly always do.
    // Otherwise bucket by bucket, whichever of them fit.
    {
        const int b0 = (blockIdx.x * 4 + wave) * BKT_RUN, b1 = min(a.B, b0 + BKT_RUN);
        if (b0 < a.B) {
            uint32_t edge[BKT_RUN + 1];
#pragma unroll
            for (int k = 0; k <= BKT_RUN; k++) edge[k] = a.bucket_base[min(b0 + k, b1)];
            const uint32_t lo = edge[0], n = edge[BKT_RUN] - lo;
            if (n > 0u && n <= (uint32_t)WAVE) {
                uint64_t v = lane < (int)n ? a.pairs[lo + lane] : ~0ull;
                v = wave_sort64(v, lane);
                if (lane < (int)n) emit_sorted(a, lo + lane, v);
            } else if (n > 0u) {
#pragma unroll
                for (int k = 0; k < BKT_RUN; k++) {
                    const uint32_t l = edge[k], m = edge[k + 1] - l;
                    if (m > 0u && m <= (uint32_t)WAVE) {
                        uint64_t v = lane < (int)m ? a.pairs[l + lane] : ~0ull;
                        v = wave_sort64(v, lane);
                        if (lane < (int)m) emit_sorted(a, l + lane, v);
                    }
                }
            }
        }
    }
    // phase 2: the buckets that did not fit, one after the other, by the whole workgroup: normalised bitonic network
    // (every comparator ascending, first step of a merge mirrored), which sorts any n when comparators whose upper end
    // is >= n are skipped -- in LDS, or in global memory for the (pathological: thousands of equal depths) buckets
    // beyond the LDS capacity
    // (the workgroup's bucket boundaries in one round trip, not one per trip of the loop)
    __shared__ uint32_t edges[4 * BKT_RUN + 1];
    if (tid <= 4 * BKT_RUN) edges[tid] = a.bucket_base[min((int)blockIdx.x * 4 * BKT_RUN + tid, a.B)];
    __syncthreads();
    for (int w = 0; w < 4 * BKT_RUN; w++) {
        const int b = blockIdx.x * 4 * BKT_RUN + w;
        if (b >= a.B) break;
        const uint32_t lo = edges[w], n = edges[w + 1] - lo;
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
constexpr int BAND_THREADS = 1024;
constexpr int BAND_WAVES = BAND_THREADS / WAVE;

struct BandArgs {
    int nb, band_rows, nchunk;
    uint32_t capacity;                  // entries the band arrays hold (= instances the caller's binning buffer was sized for)
    uint32_t *header;
    const uint2 *rect_sorted;
    const uint32_t *ids_sorted;
    uint32_t *band_cnt;                 // [nb][nchunk]
    uint32_t *band_info;                // [4 * BAND_MAX]: start, length of every band's list; first / one past the last tile column it reaches
    uint2 *band_rect;
    uint32_t *band_id;
    // what bin_tiles_kernel starts from: every tile's range and count cleared, and the list of the super-tiles with work
    int T, nsx, nsy, split_at;
    uint2 *ranges;
    uint32_t *tile_count;
    uint32_t *work;
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
    // (and the tile columns every band's rectangles reach: a super-tile outside its band's extent leaves bin_tiles at once)
    __shared__ uint32_t col0[BAND_MAX], col1[BAND_MAX];
    if (tid < BAND_MAX) { col0[tid] = 0xFFFFFFFFu; col1[tid] = 0u; }
    int b0 = 0, b1 = -1;
    uint2 rc = make_uint2(0u, 0u);
    if (k < nvis) { rc = a.rect_sorted[k]; band_span(rc, a.band_rows, b0, b1); }
    band_wave_counts(a.nb, b0, b1, lane, wcnt, wave);
    __syncthreads();
    for (int b = b0; b <= b1; b++) { atomicMin(&col0[b], rc.x & 0xFFFFu); atomicMax(&col1[b], rc.x >> 16); }
    if (tid < a.nb) {
        uint32_t t = 0;
#pragma unroll
        for (int w = 0; w < BAND_WAVES; w++) t += wcnt[w][tid];
        a.band_cnt[(size_t)tid * a.nchunk + blockIdx.x] = t;
    }
    __syncthreads();
    if (tid < a.nb && col1[tid] > col0[tid]) {
        atomicMin(&a.band_info[2 * BAND_MAX + tid], col0[tid]);
        atomicMax(&a.band_info[3 * BAND_MAX + tid], col1[tid]);
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
    // (bin_tiles only visits the super-tiles with work: every tile starts empty)
    for (int t = (int)blockIdx.x * BAND_THREADS + tid; t < a.T; t += (int)gridDim.x * BAND_THREADS) {
        a.tile_count[t] = 0u;
        a.ranges[t] = make_uint2(0u, 0u);
    }
    if (blockIdx.x == 0) {
        // a band list that does not fit the caller's buffer: every band is left empty, the overflow is reported like a
        // tile-list overflow (tile_scan_kernel keeps the flag)
        uint32_t len = 0u;                                  // lane b of wavefront 0: entries of band b
        if (tid < a.nb) {
            const uint32_t st = a.band_info[tid], en = a.band_info[BAND_MAX + tid];
            len = fits ? en - st : 0u;
            a.band_info[BAND_MAX + tid] = len;
        }
        if (tid == 0) {
            a.header[H_BAND_OVERFLOW] = fits ? 0u : total;
            a.header[H_TOTAL] = 0u;                         // bin_tiles adds up the instances ...
            a.header[H_OVERFLOW] = fits ? 0u : total;       // ... and raises this when the lists do not fit either
        }
        if (wave == 0) {
            // The super-tiles bin_tiles has work for, the longest bands' first (they hold its heaviest workgroups; the launch is a
            // grid of at most SOAR_BIN_GRID workgroups per frame that takes the list in turns: a workgroup of sixteen wavefronts and
            // 64 KB of LDS that only finds out that it has nothing to do still has to wait for a slot of its size behind the busy
            // ones -- 2040 launched for ~700 with work cost the launch a quarter of its time).  Of a band: the columns its
            // rectangles reach, or every column when it is long enough for the others to help (bin_tiles_body).
            const int b = lane, rows_per_band = a.band_rows / BIN_SUPER, srow0 = b * rows_per_band;
            const int nrows = b < a.nb ? max(0, min(rows_per_band, a.nsy - srow0)) : 0;
            int c0 = 0, ncols = 0;
            if (len > 0u) {
                const int ex0 = (int)(a.band_info[2 * BAND_MAX + b] & 0xFFFFu), ex1 = (int)a.band_info[3 * BAND_MAX + b];
                const int se0 = ex0 / BIN_SUPER, se1 = min(a.nsx, (ex1 + BIN_SUPER - 1) / BIN_SUPER);
                const bool split = len > (uint32_t)a.split_at;
                c0 = split ? 0 : se0;
                ncols = split ? a.nsx : max(0, se1 - se0);
            }
            const uint32_t items = (uint32_t)(ncols * nrows);
            uint32_t before = 0u, all = 0u;
            for (int o = 0; o < a.nb; o++) {
                const uint32_t len_o = (uint32_t)__shfl((int)len, o), items_o = (uint32_t)__shfl((int)items, o);
                before += (len_o > len || (len_o == len && o < b)) ? items_o : 0u;
                all += items_o;
            }
            uint32_t at = before;
            for (int r = 0; r < nrows; r++)
                for (int c = 0; c < ncols; c++) a.work[at++] = (uint32_t)((srow0 + r) * a.nsx + c0 + c);
            if (lane == 0) a.header[H_BIN_WORK] = all;
        }
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
// band of a super-tile: its row of super-tiles, or several rows per band on very tall images
__device__ __forceinline__ int band_of_block(int block, int gx, int band_rows)
{
    const int nsx = (gx + BIN_SUPER - 1) / BIN_SUPER;
    return ((block / nsx) * BIN_SUPER) / band_rows;
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

// The lists.  One workgroup per 4x4 tiles walks its band's depth-ordered rectangles and keeps those that touch its 16 tiles (in
// order) with the mask of the tiles they cover; it counts them per tile, reserves room for its 16 lists behind whatever has been
// reserved so far (ONE atomic add on header[H_TOTAL] per workgroup: the list of a tile is contiguous and in depth order, the tiles of
// a frame lie in the order the workgroups got there -- `ranges` says where; nothing downstream assumes tile order), and appends
// every tile's entries by ballot-prefix compaction: no per-tile sort, no atomics on the lists.  (Rounds 1-2 had a counting launch, a
// one-workgroup scan of the tile counts and this launch, 100 us per 4-frame step at C3; rounds 3-5 one launch with the kept entries
// in row buffers shared by the workgroup, a barrier per trip of 4096 rectangles, 66 us.)
// When the lists do not fit the caller's buffer (`capacity`) the workgroup writes no list; header[H_OVERFLOW] ends up as the number
// of instances needed, and the workgroup that builds the tile order one launch later (block_mask_kernel) empties every range.
struct BinTilesArgs {
    uint32_t *header;
    int gx;
    int gy;
    int band_rows;
    const uint32_t *band_info;
    const uint2 *band_rect;
    const uint32_t *band_id;
    uint2 *ranges;
    uint32_t *point_list;
    uint32_t *tile_count;
    uint32_t capacity;
    unsigned long long *dbg;
    uint32_t *tile_xy;       // BinBuf::tile_xy
    int split_at;            // rectangles in a band from which idle columns help (SOAR_BIN_SPLIT_AT)
    const uint32_t *work;    // ImageBuf::bin_work: header[H_BIN_WORK] super-tiles
};
// Round 6 (profiles/r06_ab_bin_tiles.txt: 67 -> 47 us per 4-frame launch at C3): no barrier inside a walk, no shared buffers.
// Wavefront w takes the CONTIGUOUS slice [w * len, (w + 1) * len) of the band's depth order: its walk
// leaves the sizes of the 16 lists per wavefront, an exclusive scan of those over the wavefronts (per tile) says where every
// wavefront's entries of every tile start -- slice after slice = depth order -- and every wavefront then appends straight to the 16
// lists at 16 running cursors in scalar registers.
//   * These launches are bound by vector instructions: a wavefront instruction occupies its SIMD for four cycles, and the log
//     (SOAR_BIN_LOG) of every variant tried fits  time = vector instructions of the workgroup's wavefronts / 2400 per us  + ~9 us of
//     fixed latencies (first loads, three barriers, the reservation's atomic).  A slab of 64 rectangles none of which touches the
//     super-tile costs its loads and ~12 instructions (most slabs of most workgroups: 30 super-tiles walk every band; the test itself is
//     two packed 16-bit multiply-adds and a sign mask), one with a hit ~45 more (cover mask, compaction).
//   * The rectangles that do touch it are compacted -- (id, 16-bit cover mask), in order -- into a ring of CAP entries PRIVATE to the
//     wavefront (no barrier: the LDS executes one wavefront's operations in order), and both the counting and the placing work on DENSE
//     slabs of 64 kept entries: one ballot-prefix compaction per (dense slab, tile) -- 6 vector instructions and two stores; the ballot
//     itself is the stores' lane mask.
//   * A wavefront whose kept entries all fit the ring never walks the band a second time: the first walk captured them.
//     The others walk their slice again, streaming through the ring.
//   * The launch is a grid of at most SOAR_BIN_GRID workgroups per frame over the LIST of super-tiles with work that band_place_kernel
//     leaves (the columns every band's rectangles reach -- band_count's extents --, the longest bands first): a sixteen-wavefront
//     workgroup with 64 KB of LDS that only finds out that it has nothing to do still queues for a slot of its size behind the busy ones.
//   * In a band of more than SPLIT_AT rectangles the list holds EVERY column, and the ones outside the extent HELP: the h-th of them
//     takes the lower two rows of tiles of the h-th column inside, whose own workgroup keeps the upper two (the launch waits for its
//     heaviest workgroup: 72k list entries from a band of 26k, 49 us alone; halved 40 us).
//   Tried and dropped, all with `point_list` bit-exact (same file): placing straight from the walk's slabs (74 us: 16 compactions per
//   slab with ANY hit); a pool of kept-entry blocks in LDS that wavefront t goes through for tile t (balanced whichever wavefronts found
//   the entries -- a patch of surface is a narrow range of depths, so a super-tile's hits cluster in a few slices -- but 16 wavefronts x
//   ~30 instructions per block: 65 us), the same with a tile's entries merged across blocks by a cross-lane permute into full 64-lane
//   stores (71 us: the stores were not what the placing waited for); two workgroups per super-tile in the grid (+13 us for DISPATCHING
//   2040 more idle 16-wavefront workgroups, whatever they then do); four helpers' rows instead of two (55 us: three more walks of the
//   band per heavy super-tile); workgroups of 8 wavefronts (58-62 us); sizes counted by 16 ballots per slab (57 us).
#ifndef SOAR_BIN_UNROLL
#define SOAR_BIN_UNROLL 4
#endif
#ifndef SOAR_BIN_SPLIT_AT
#define SOAR_BIN_SPLIT_AT 12288      // rectangles in a band from which its super-tiles are shared by two workgroups
#endif
#ifndef SOAR_BIN_WAVES
#define SOAR_BIN_WAVES 16     // (8: 58-62 us against 51)
#endif
constexpr int BIN_WAVES = SOAR_BIN_WAVES, BIN_THREADS = BIN_WAVES * WAVE;
#ifndef SOAR_BIN_GRID
#define SOAR_BIN_GRID 512           // workgroups of a frame's launch (the slots of the chip): they take the list of super-tiles with work in turns
                                    // (128 / 192 / 256 / 384 / 512 at C3, four frames per launch: 60.3 / 52.2 / 47.3 / 47.2 / 47.3 us)
#endif
#ifndef SOAR_BIN_MAX_PARTS
#define SOAR_BIN_MAX_PARTS 2         // (4: a row of tiles per workgroup where three helpers are free -- 55 us against 52)
#endif
#ifndef SOAR_BIN_CAPTURE
#define SOAR_BIN_CAPTURE 512         // kept entries a wavefront's ring holds (power of two >= 128): 4 KB per wavefront, 64 KB per workgroup
                                     // (128 / 256 / 512: 57.0 / 52.4 / 50.9 us)
#endif
__device__ __forceinline__ void wave_lds_order()
{
    // the wavefront's own LDS writes before its own LDS reads: ordering for the compiler only (the hardware keeps one wavefront's
    // LDS operations in order)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// does the rectangle touch the super-tile?  x0 < tx1 && x1 > tx0 as the signs of (tx1 - 1 - x0, x1 - tx0 - 1), two 16-bit halves of one
// packed multiply-add (coordinates are tile indices: < 2^15), likewise y
struct SuperTileTest {
    uint32_t cx, cy;
};
__device__ __forceinline__ SuperTileTest super_tile_test(const SuperTile &s)
{
    SuperTileTest t;
    t.cx = ((uint32_t)(s.tx1 - 1) & 0xFFFFu) | ((uint32_t)(-s.tx0 - 1) << 16);
    t.cy = ((uint32_t)(s.ty1 - 1) & 0xFFFFu) | ((uint32_t)(-s.ty0 - 1) << 16);
    return t;
}
typedef short short2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bool rect_hits_packed(uint2 rc, const SuperTileTest &t)
{
    // low half: -x0 + (tx1 - 1), high half: +x1 + (-tx0 - 1); both >= 0 <=> hit
    const uint32_t sign = 0x0001FFFFu;                 // (-1, +1) as two int16: low half first
    short2v a, b, c;
    uint32_t dx, dy;
    __builtin_memcpy(&a, &rc.x, 4); __builtin_memcpy(&b, &sign, 4); __builtin_memcpy(&c, &t.cx, 4);
    short2v rx = a * b + c;
    __builtin_memcpy(&a, &rc.y, 4); __builtin_memcpy(&c, &t.cy, 4);
    short2v ry = a * b + c;
    __builtin_memcpy(&dx, &rx, 4); __builtin_memcpy(&dy, &ry, 4);
    return ((dx | dy) & 0x80008000u) == 0u;
}
__device__ __forceinline__ void bin_tiles_body(const int bx, const BinTilesArgs &a)
{
    const unsigned long long dbg_t0 = a.dbg ? wall_clock64() : 0ull;
    constexpr int NT = BIN_SUPER * BIN_SUPER, U = SOAR_BIN_UNROLL, CAP = SOAR_BIN_CAPTURE;
    static_assert(BIN_SUPER == 4 && BIN_WAVES * WAVE >= NT, "16 tiles: a cover mask has 16 bits");
    static_assert(CAP >= 2 * WAVE && (CAP & (CAP - 1)) == 0, "a ring holds a dense slab plus what one slab of the band can add");
    __shared__ uint2 ring[BIN_WAVES][CAP];             // kept entries of one wavefront: (id, cover mask)
    __shared__ uint32_t wave_tile[BIN_WAVES][NT];      // sizes per (wavefront, tile); then: entries of the wavefronts before it
    __shared__ uint32_t tile_cnt[NT], tile_base[NT];
    __shared__ int fits_s;
    const uint32_t *__restrict__ header = a.header;
    const int gx = a.gx, gy = a.gy;
    const int band = band_of_block(bx, gx, a.band_rows);
    const uint2 *__restrict__ rect_sorted = a.band_rect + a.band_info[band];
    const uint32_t *__restrict__ ids_sorted = a.band_id + a.band_info[band];
    const int P = header[H_NVIS] ? (int)a.band_info[BAND_MAX + band] : 0;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;   // (scalar: uniform loop bounds)
    char *__restrict__ point_list = reinterpret_cast<char *>(a.point_list);
    char *__restrict__ tile_xy = reinterpret_cast<char *>(a.tile_xy);
    // The super-tile columns the band's rectangles reach (band_count).  In a band of more than SPLIT_AT rectangles the other columns are
    // on the work list too and HELP: the h-th of them takes the lower two rows of tiles of the h-th column inside the extent (its own
    // walk, its own reservation), whose own workgroup then only takes the upper two -- the launch waits for its heaviest workgroup
    // (72k list entries from a band of 26k at C3: 49 us alone).
    const int nsx = (gx + BIN_SUPER - 1) / BIN_SUPER, scol = bx % nsx;
    const int ex0 = (int)(a.band_info[2 * BAND_MAX + band] & 0xFFFFu), ex1 = (int)a.band_info[3 * BAND_MAX + band];
    const int se0 = ex0 / BIN_SUPER, se1 = (ex1 + BIN_SUPER - 1) / BIN_SUPER, width = se1 - se0;
    const bool inside = P > 0 && scol >= se0 && scol < se1, split = P > a.split_at;
    SuperTile st = super_tile_of(bx, gx, gy);
    // the idle columns are dealt to the columns inside the extent in turn: column j gets the helpers j, j + width, j + 2 width, ...
    // and is shared by 2 workgroups (two rows of tiles each) with one helper, by 4 (a row each) with three
    const int helpers = nsx - width;
    auto parts_of = [&](int j) {
        const int q = helpers / width + (j < helpers % width ? 1 : 0);
        return !split ? 1 : (q >= 3 && SOAR_BIN_MAX_PARTS >= 4) ? 4 : q >= 1 ? 2 : 1;
    };
    int sub = 0, parts = 1;
    if (!inside) {
        const int h = scol < se0 ? scol : scol - width;                 // its number among the columns outside the extent
        const int j = width > 0 ? h % width : 0;
        if (split && width > 0) { parts = parts_of(j); sub = h / width + 1; }
        if (sub == 0 || sub >= parts) {
            if (a.dbg && tid == 0) { a.dbg[(size_t)bx * 4] = 1ull; a.dbg[(size_t)bx * 4 + 1] = dbg_t0; }
            return;
        }
        st = super_tile_of(bx - scol + se0 + j, gx, gy);
    } else {
        parts = parts_of(scol - se0);
    }
    {
        const int rows = BIN_SUPER / parts;
        st.ty0 += sub * rows;
        st.ty1 = min(st.ty1, st.ty0 + rows);
        if (st.ty0 >= st.ty1) return;                  // (a last row of super-tiles with fewer rows of tiles)
    }
    const int my_tx = st.tx0 + (tid & 3), my_ty = st.ty0 + ((tid >> 2) & 3);          // thread t < 16 reports tile t
    const bool my_tile = tid < NT && my_tx < st.tx1 && my_ty < st.ty1;
    // this wavefront's slabs (64 rectangles each) of the band
    const int nslab = (P + WAVE - 1) / WAVE, per = (nslab + BIN_WAVES - 1) / BIN_WAVES;
    const int s0 = wave * per, s1 = min(nslab, s0 + per);
    uint2 *__restrict__ my_ring = ring[wave];
    const SuperTileTest stt = super_tile_test(st);

    // The walk of the slice: `keep(hit ballot, hit, id, mask)` for every slab with a rectangle that touches the super-tile.
    // (loads behind the end of the list are clamped, not skipped: a load under a branch makes the compiler wait for ALL loads in
    // flight at the next use of any of them, and the prefetch of the next trip hides nothing)
    const char *__restrict__ rect_bytes = reinterpret_cast<const char *>(rect_sorted);
    const char *__restrict__ id_bytes = reinterpret_cast<const char *>(ids_sorted);
    const int in_slice = min(P, s1 * WAVE);            // rectangles behind this are not this wavefront's
    auto walk = [&](auto &&keep) {
        uint2 rc[U];
        uint32_t id[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t k = (uint32_t)min((s0 + u) * WAVE + lane, P - 1);       // (32-bit byte offsets: P < 2^28)
            rc[u] = *reinterpret_cast<const uint2 *>(rect_bytes + (k << 3));
            id[u] = *reinterpret_cast<const uint32_t *>(id_bytes + (k << 2));
        }
        for (int s = s0; s < s1; s += U) {
            uint2 nrc[U];
            uint32_t nid[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t k = (uint32_t)min((s + U + u) * WAVE + lane, P - 1);
                nrc[u] = *reinterpret_cast<const uint2 *>(rect_bytes + (k << 3));
                nid[u] = *reinterpret_cast<const uint32_t *>(id_bytes + (k << 2));
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const bool hit = rect_hits_packed(rc[u], stt) & ((s + u) * WAVE + lane < in_slice);
                const unsigned long long hb = __builtin_amdgcn_ballot_w64(hit);
                if (hb != 0ull) keep(hb, hit, id[u], hit ? cover_mask(rc[u], st) : 0u);
                rc[u] = nrc[u];
                id[u] = nid[u];
            }
        }
    };

    // walk 1: the kept entries into the ring while they fit, and the sizes of the 16 lists: 16 byte-wide counters per lane (the cover
    // mask spread over four registers), folded over the wavefront before a byte could overflow and at the end -- lane t < 16 then holds
    // the wavefront's number of entries of tile t.  The masks are counted from the ring's DENSE slabs of 64 kept entries (16 instructions
    // per dense slab instead of per slab of the band with a hit in it); from the slab that no longer fits, the ring's content is
    // counted and the rest slab by slab.
    uint32_t mine = 0u;
    int nkept = 0;                                     // kept entries of this wavefront (scalar)
    {
        uint32_t acc[4] = {0u, 0u, 0u, 0u};
        int since = 0;
        auto fold = [&]() {
            uint32_t part[8];
#pragma unroll
            for (int d = 0; d < 4; d++) { part[2 * d] = acc[d] & 0x00FF00FFu; part[2 * d + 1] = (acc[d] >> 8) & 0x00FF00FFu; acc[d] = 0u; }
#pragma unroll
            for (int q = 0; q < 8; q++)
#pragma unroll
                for (int d = 1; d < WAVE; d <<= 1) part[q] += (uint32_t)__shfl_xor((int)part[q], d);
            // part[2d] = tiles 4d (low half) and 4d+2 (high half), part[2d+1] = tiles 4d+1 and 4d+3
            const int d = (lane >> 2) & 3, k = lane & 3;
            const uint32_t w = (k & 1) ? (d == 0 ? part[1] : d == 1 ? part[3] : d == 2 ? part[5] : part[7])
                                       : (d == 0 ? part[0] : d == 1 ? part[2] : d == 2 ? part[4] : part[6]);
            mine += (k & 2) ? w >> 16 : w & 0xFFFFu;
            since = 0;
        };
        auto count_masks = [&](uint32_t m) {
#pragma unroll
            for (int d = 0; d < 4; d++) acc[d] += (((m >> (4 * d)) & 15u) * 0x00204081u) & 0x01010101u;
            if (++since == 255) fold();
        };
        auto count_ring = [&](int n) {
            wave_lds_order();
            for (int h = 0; h < n; h += WAVE) count_masks(h + lane < n ? my_ring[h + lane].y : 0u);
        };
        bool full = false;
        walk([&](unsigned long long hb, bool hit, uint32_t id, uint32_t m) {
            const int n = __builtin_popcountll(hb);
            if (!full && nkept + n > CAP) { count_ring(nkept); full = true; }
            if (full) count_masks(m);
            else if (hit) my_ring[nkept + prefix_in_mask(hb)] = make_uint2(id, m);
            nkept += n;
        });
        if (!full) count_ring(nkept);
        if (since) fold();
    }
    if (lane < NT) wave_tile[wave][lane] = mine;
    lds_barrier();
    const unsigned long long dbg_t1 = a.dbg ? wall_clock64() : 0ull;
    {   // thread (w, t): the entries of tile t in the wavefronts before w, and in all of them
        const int w = (tid >> 4) & (BIN_WAVES - 1), t = tid & (NT - 1);
        uint32_t before = 0u, all = 0u;
        if (tid < BIN_WAVES * NT) {
#pragma unroll
            for (int v = 0; v < BIN_WAVES; v++) {
                const uint32_t c = wave_tile[v][t];
                before += v < w ? c : 0u;
                all += c;
            }
        }
        lds_barrier();
        if (tid < BIN_WAVES * NT) wave_tile[w][t] = before;
        if (tid < NT) tile_cnt[tid] = all;
    }
    lds_barrier();
    if (tid == 0) {
        uint32_t sum = 0;
#pragma unroll
        for (int t = 0; t < NT; t++) { tile_base[t] = sum; sum += tile_cnt[t]; }
        uint32_t start = 0;
        bool fits = true;
        if (sum) {
            start = atomicAdd(&a.header[H_TOTAL], sum);
            fits = (uint64_t)start + sum <= (uint64_t)a.capacity;
            if (!fits) atomicMax(&a.header[H_OVERFLOW], start + sum);     // the last one to get here leaves the number needed
        }
#pragma unroll
        for (int t = 0; t < NT; t++) tile_base[t] += start;
        fits_s = fits ? 1 : 0;
    }
    lds_barrier();
    const bool fits = fits_s != 0;
    {
        const uint32_t cnt = tile_cnt[tid & (NT - 1)], first = tile_base[tid & (NT - 1)];
        if (my_tile) {
            a.tile_count[my_ty * gx + my_tx] = cnt;
            a.ranges[my_ty * gx + my_tx] = (cnt && fits) ? make_uint2(first, first + cnt) : make_uint2(0u, 0u);
        }
    }
    // placing: the wavefronts that kept something append their entries, dense slab by dense slab; cursors in scalar registers
    if (fits && nkept > 0) {
        const uint32_t curv = lane < NT ? tile_base[lane] + wave_tile[wave][lane] : 0u;
        uint32_t cur[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) cur[t] = (uint32_t)__builtin_amdgcn_readlane((int)curv, t);
        const uint32_t xy0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(((uint32_t)st.ty0 << 16) | (uint32_t)st.tx0));
        auto place = [&](int head, int cnt) {          // the ring's entries [head, head + cnt), cnt <= 64
            wave_lds_order();
            const uint2 e = my_ring[(head + lane) & (CAP - 1)];
            const uint32_t m = lane < cnt ? e.y : 0u;
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const unsigned long long bal = __builtin_amdgcn_ballot_w64((m & (1u << t)) != 0u);
                if (bal != 0ull) {
                    if (__builtin_amdgcn_inverse_ballot_w64(bal)) {     // (the ballot IS the stores' lane mask: no second compare)
                        const uint32_t at = (cur[t] + (uint32_t)prefix_in_mask(bal)) << 2;        // byte offset: capacity < 2^30 entries
                        *reinterpret_cast<uint32_t *>(point_list + at) = e.x;
                        uint32_t xy = xy0;                 // (a scalar, and kept out of 16 loop-invariant vector registers: two
                        asm volatile("" : "+s"(xy));       //  workgroups per CU need <= 64)
                        *reinterpret_cast<uint32_t *>(tile_xy + at) = xy + (((uint32_t)(t >> 2) << 16) | (uint32_t)(t & 3));
                    }
                    cur[t] += (uint32_t)__builtin_popcountll(bal);
                }
            }
            wave_lds_order();                           // (the slots are free for the walk's next entries only now)
        };
        if (nkept <= CAP) {
            for (int h = 0; h < nkept; h += WAVE) place(h, min(WAVE, nkept - h));
        } else {
            int head = 0, n = 0;
            walk([&](unsigned long long hb, bool hit, uint32_t id, uint32_t m) {
                if (hit) my_ring[(head + n + prefix_in_mask(hb)) & (CAP - 1)] = make_uint2(id, m);
                n += __builtin_popcountll(hb);
                if (n >= WAVE) { place(head, WAVE); head = (head + WAVE) & (CAP - 1); n -= WAVE; }
            });
            if (n) place(head, n);
        }
    }
    if (a.dbg) lds_barrier();                          // (the log's end of the workgroup: its last wavefront's)
    if (a.dbg && tid == 0) {
        unsigned long long *w = a.dbg + (size_t)bx * 4;
        uint32_t kept = 0;
        for (int t = 0; t < NT; t++) kept += tile_cnt[t];
        w[0] = wall_clock64() - dbg_t0; w[1] = dbg_t0; w[2] = (unsigned long long)kept;
        w[3] = ((dbg_t1 - dbg_t0) << 32) | (unsigned long long)P;
    }
}

#ifndef SOAR_BIN_WPE
#define SOAR_BIN_WPE 8        // two workgroups per CU
#endif
__global__ void __launch_bounds__(BIN_THREADS) __attribute__((amdgpu_waves_per_eu(SOAR_BIN_WPE, 8))) bin_tiles_kernel(Batch<BinTilesArgs> batch)
{
    int frame, bx;
    batch_interleave1(frame, bx);
    const BinTilesArgs &a = batch.v[frame];
    const uint32_t n_work = a.header[H_BIN_WORK];
    for (uint32_t i = (uint32_t)bx; i < n_work; i += gridDim.x) {
        bin_tiles_body((int)a.work[i], a);
        lds_barrier();                                      // (the next super-tile's rings overwrite this one's)
    }
}


}  // namespace

static int bucket_count_for(int32_t P)
{
    int B = 256;
    while (B < BKT_MAX && B * 16 < P) B <<= 1;          // ~16 Gaussians per bucket
    return B;
}

// geometry stage: places in the depth buckets and the scatter of the (key, index) pairs
int launch_depth_buckets(const SoarRastParams &prm, GeomBuf &g, hipStream_t stream)
{
    const int B = bucket_count_for(prm.P);
    const int nblk = (prm.P + 63) / 64;                 // one statistics row per wavefront of preprocess (64 Gaussians)
    const int nw = (prm.P + BKT_CHUNK - 1) / BKT_CHUNK;
    StageTimer timer(ST_SORT, stream);
    const BucketCountArgs ca = {prm.P, B, prm.sort_descending ? 1 : 0, g.depth_key, g.blk_stats, nblk, g.header, g.bucket_mat, g.bucket_base, g.sort_slot, g.band_info};
    SOAR_LAUNCH_BATCHED(bucket_count_kernel, dim3(nw), dim3(1024), 0, stream, ca);
    SOAR_LAUNCH_BATCHED(bucket_scan_kernel, dim3((B + 1023) / 1024), dim3(1024), 0, stream, ca);
    const BucketScatterArgs sa = {prm.P, B, prm.sort_descending ? 1 : 0, g.depth_key, g.header, g.bucket_mat, g.sort_slot, g.sort_pairs};
    SOAR_LAUNCH_BATCHED(bucket_scatter_kernel, dim3((prm.P + 255) / 256), dim3(256), 0, stream, sa);
    SOAR_LAUNCH_OK("depth_buckets", stream, prm.debug);
    return 0;
}

int launch_tile_binning(const SoarRastParams &prm, GeomBuf &g, BinBuf &b, ImageBuf &img, int64_t capacity, hipStream_t stream)
{
    if (prm.P >= (1 << 28)) { set_error("tile binning packs a Gaussian's index into 28 bits: P = %d is too large", prm.P); return 1; }
    if (capacity >= (1ll << 30)) { set_error("tile binning addresses the lists with 32-bit byte offsets: a capacity of %lld instances is too large", (long long)capacity); return 1; }
    const int gx = (prm.W + TILE - 1) / TILE, gy = (prm.H + TILE - 1) / TILE;
    const int nsx = (gx + BIN_SUPER - 1) / BIN_SUPER, nsy = (gy + BIN_SUPER - 1) / BIN_SUPER;
    {
        StageTimer timer(ST_SORT, stream);
        BucketSortArgs a;
        a.B = bucket_count_for(prm.P);
        a.bucket_base = g.bucket_base; a.pairs = g.sort_pairs; a.rect = g.rect; a.ids_sorted = g.ids_sorted;
        a.rect_sorted = g.rect_sorted;
        SOAR_LAUNCH_BATCHED(bucket_sort_kernel, dim3((a.B + 4 * BKT_RUN - 1) / (4 * BKT_RUN)), dim3(256), 0, stream, a);
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
    // (the environment variable of the same name: a test's way to reach the helpers' code on scenes the CPU oracle finishes in seconds)
    static const int split_at = getenv("SOAR_BIN_SPLIT_AT") ? atoi(getenv("SOAR_BIN_SPLIT_AT")) : SOAR_BIN_SPLIT_AT;
    ba.T = gx * gy; ba.nsx = nsx; ba.nsy = nsy; ba.split_at = split_at;
    ba.ranges = img.ranges; ba.tile_count = img.tile_count; ba.work = img.bin_work;
    {
        StageTimer timer(ST_RANGES, stream);
        SOAR_LAUNCH_BATCHED(band_count_kernel, dim3(ba.nchunk), dim3(BAND_THREADS), 0, stream, ba);
        SOAR_LAUNCH_BATCHED(band_place_kernel, dim3(ba.nchunk), dim3(BAND_THREADS), 0, stream, ba);
    }
    SOAR_LAUNCH_OK("band_lists", stream, prm.debug);
    {
        StageTimer timer(ST_EMIT_KEYS, stream);
        unsigned long long *dbg = nullptr;
        static int dbg_left = getenv("SOAR_BIN_LOG") ? 1 : 0;          // diagnostic: per-workgroup timings of ONE launch
        const bool log_now = dbg_left > 0 && prm.W >= 1920 && !batch_ctx().n;
        if (log_now) {
            dbg_left = 0;
            SOAR_HIP_OK(hipMalloc(&dbg, 8 * (size_t)(nsx * nsy) * 4));
            SOAR_HIP_OK(hipMemsetAsync(dbg, 0, 8 * (size_t)(nsx * nsy) * 4, stream));
        }
        const BinTilesArgs bt = {g.header, gx, gy, band_rows, g.band_info, ba.band_rect, ba.band_id, img.ranges, b.vals_sorted,
                                 img.tile_count, ba.capacity, dbg, b.tile_xy, split_at, img.bin_work};
        SOAR_LAUNCH_BATCHED(bin_tiles_kernel, dim3(min(nsx * nsy, SOAR_BIN_GRID)), dim3(BIN_THREADS), 0, stream, bt);
        if (log_now) {
            const int nwg = nsx * nsy;
            const size_t nw = (size_t)nwg * 4;
            SOAR_HIP_OK(hipStreamSynchronize(stream));
            unsigned long long *h = (unsigned long long *)malloc(8 * nw);
            SOAR_HIP_OK(hipMemcpy(h, dbg, 8 * nw, hipMemcpyDeviceToHost));
            (void)hipFree(dbg);
            {   // when the workgroups started and ended, relative to the first one's start (100 MHz clock); the busy ones one by one
                unsigned long long first = ~0ull, last = 0ull;
                for (int i = 0; i < nwg; i++) if (h[i * 4]) { first = h[i * 4 + 1] < first ? h[i * 4 + 1] : first; }
                for (int i = 0; i < nwg; i++) if (h[i * 4]) { const unsigned long long e = h[i * 4 + 1] + h[i * 4]; last = e > last ? e : last; }
                fprintf(stderr, "[bin_tiles] %d workgroups: first start to last end %.1f us\n", nwg, (last - first) / 100.0);
                for (int i = 0; i < nwg; i++)
                    if (h[i * 4] > 1500)
                        fprintf(stderr, "[bin_tiles]   WG %d: start %.1f, %.1f us (walk %.1f), %llu list entries from a band of %llu\n", i,
                                (h[i * 4 + 1] - first) / 100.0, h[i * 4] / 100.0, (h[i * 4 + 3] >> 32) / 100.0, h[i * 4 + 2] & 0xFFFFFFFFull,
                                h[i * 4 + 3] & 0xFFFFFFFFull);
            }
            free(h);
        }
    }
    SOAR_LAUNCH_OK("bin_tiles", stream, prm.debug);
    return 0;
}

}  // namespace soar
