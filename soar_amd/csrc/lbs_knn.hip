// lbs_knn.hip -- exact K nearest canonical SMPL-X vertices + inverse-distance skinning weights, gfx950.
//
// Replaces SMPL_Guidance.query_weights_smpl (TS/utils/smpl.py:618-637): pytorch3d's brute-force knn_points (K = 30 over
// V = 10475 vertices for every Gaussian, 1e9 distance evaluations per call) followed by a gather of [30,55] rows.
//
// Here the vertex set is bucketed into a uniform grid (cell ids -> stable radix sort -> per-cell ranges), the queries
// are sorted by cell too, and one wavefront takes 64 consecutive sorted queries (lane = query).  All lanes of a cell
// share the candidate set -- the box of cells around their cell -- so the walk over candidates is UNIFORM: candidate
// positions and skinning rows are wave-uniform (scalar) loads, there is no divergence and no per-lane gather.
//   pass 1: every lane keeps the K smallest squared distances in registers (median-of-3 insertion chain, no
//           indices); the four wavefronts of a workgroup each scan a quarter of the LDS-staged candidates and merge
//           their lists (two sorted lists -> min against the reversed other = a bitonic sequence of the 32 smallest, five
//           compare-exchange stages; the last merge only needs the K-th value); the box grows until, for every lane, the K-th distance is provably final (distance to the
//           nearest open face of the box) -- the search is exact;
//   pass 2: one query at a time per wavefront: lanes = candidates pick those below the query's K-th distance (ties at
//           the K-th place in list order until K are reached), then lanes = joints blend their skinning rows, so row
//           reads and the 220-byte result row are coalesced.
// A generic one-thread-per-query kernel (any K <= 32, any J) is kept for non-default K / J.
#include "soar_common.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

namespace soar {

namespace {

constexpr int KNN_THREADS = 128;
constexpr int KNN_MAXK = 32;
constexpr int GRID_MAX = 64;             // cells per axis (upper bound)
constexpr int GRID_RES = 28;             // cells along the longest extent: a 3x3x3 box holds the K = 30 nearest for SMPL-X density
constexpr int KNN_K = 30;                // the only K the path uses (smpl.py:627: K=30)
constexpr int KNN_JMAX = 56;             // joints held in registers by the fast kernel (J = 55)
constexpr int GRID_CELLS = GRID_MAX * GRID_MAX * GRID_MAX;

struct GridMeta {
    float minx, miny, minz, h, inv_h;
    int nx, ny, nz;
};

__global__ void __launch_bounds__(256) grid_meta_kernel(const float *__restrict__ verts, int V, GridMeta *meta)
{
    __shared__ float lo[3][256], hi[3][256];
    const int tid = threadIdx.x;
    float l[3] = {3.0e38f, 3.0e38f, 3.0e38f}, h[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int v = tid; v < V; v += 256)
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float c = verts[3 * v + k];
            l[k] = fminf(l[k], c);
            h[k] = fmaxf(h[k], c);
        }
#pragma unroll
    for (int k = 0; k < 3; k++) { lo[k][tid] = l[k]; hi[k][tid] = h[k]; }
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s)
#pragma unroll
            for (int k = 0; k < 3; k++) {
                lo[k][tid] = fminf(lo[k][tid], lo[k][tid + s]);
                hi[k][tid] = fmaxf(hi[k][tid], hi[k][tid + s]);
            }
        __syncthreads();
    }
    if (tid == 0) {
        const float ex = hi[0][0] - lo[0][0], ey = hi[1][0] - lo[1][0], ez = hi[2][0] - lo[2][0];
        const float ext = fmaxf(fmaxf(ex, ey), fmaxf(ez, 1e-6f));
        GridMeta m;
        m.h = ext / GRID_RES;
        m.inv_h = 1.0f / m.h;
        m.minx = lo[0][0]; m.miny = lo[1][0]; m.minz = lo[2][0];
        m.nx = min(GRID_MAX, (int)(ex * m.inv_h) + 1);
        m.ny = min(GRID_MAX, (int)(ey * m.inv_h) + 1);
        m.nz = min(GRID_MAX, (int)(ez * m.inv_h) + 1);
        *meta = m;
    }
}

__device__ __forceinline__ int cell_coord(float v, float lo, float inv_h, int n)
{
    return min(n - 1, max(0, (int)floorf((v - lo) * inv_h)));
}

__global__ void __launch_bounds__(256)
grid_cells_kernel(const float *__restrict__ verts, int V, const GridMeta *__restrict__ meta, uint32_t *__restrict__ keys,
                  uint32_t *__restrict__ vals)
{
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    const GridMeta m = *meta;
    const int cx = cell_coord(verts[3 * v], m.minx, m.inv_h, m.nx);
    const int cy = cell_coord(verts[3 * v + 1], m.miny, m.inv_h, m.ny);
    const int cz = cell_coord(verts[3 * v + 2], m.minz, m.inv_h, m.nz);
    keys[v] = (uint32_t)((cz * GRID_MAX + cy) * GRID_MAX + cx);
    vals[v] = (uint32_t)v;
}

__global__ void __launch_bounds__(256)
grid_ranges_kernel(const float *__restrict__ verts, int V, const uint32_t *__restrict__ keys_sorted,
                   const uint32_t *__restrict__ vals_sorted, uint2 *__restrict__ cell_range, float4 *__restrict__ sorted_verts)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= V) return;
    const uint32_t c = keys_sorted[i];
    if (i == 0 || keys_sorted[i - 1] != c) cell_range[c].x = (uint32_t)i;
    if (i == V - 1 || keys_sorted[i + 1] != c) cell_range[c].y = (uint32_t)(i + 1);
    const uint32_t v = vals_sorted[i];
    sorted_verts[i] = make_float4(verts[3 * v], verts[3 * v + 1], verts[3 * v + 2], __uint_as_float(v));
}

__global__ void __launch_bounds__(KNN_THREADS)
knn_grid_kernel(const float *__restrict__ xyz, int P, int V, const GridMeta *__restrict__ meta,
                const uint2 *__restrict__ cell_range, const float4 *__restrict__ sorted_verts,
                const float *__restrict__ vert_weights, int J, int K, float *__restrict__ weights_out,
                int32_t *__restrict__ knn_idx_out)
{
    __shared__ float best_d[KNN_MAXK][KNN_THREADS];     // [k][thread]
    __shared__ int best_i[KNN_MAXK][KNN_THREADS];

    const int tid = threadIdx.x;
    const int p = blockIdx.x * KNN_THREADS + tid;
    if (p >= P) return;
    const GridMeta m = *meta;
    const float x = xyz[3 * p], y = xyz[3 * p + 1], z = xyz[3 * p + 2];
    const int cx = cell_coord(x, m.minx, m.inv_h, m.nx), cy = cell_coord(y, m.miny, m.inv_h, m.ny),
              cz = cell_coord(z, m.minz, m.inv_h, m.nz);

    for (int k = 0; k < K; k++) { best_d[k][tid] = 3.0e38f; best_i[k][tid] = -1; }
    float worst = 3.0e38f;
    int worst_slot = 0;

    for (int r = 0; r < GRID_MAX; r++) {
        // visit the cells of the cube shell with Chebyshev radius r around (cx,cy,cz)
        for (int dz = -r; dz <= r; dz++) {
            const int gz = cz + dz;
            if (gz < 0 || gz >= m.nz) continue;
            for (int dy = -r; dy <= r; dy++) {
                const int gy = cy + dy;
                if (gy < 0 || gy >= m.ny) continue;
                const bool face = (abs(dz) == r) || (abs(dy) == r);
                const int step = face ? 1 : max(2 * r, 1);               // interior rows: only dx = -r and dx = +r
                for (int dx = -r; dx <= r; dx += step) {
                    const int gx = cx + dx;
                    if (gx < 0 || gx >= m.nx) continue;
                    const uint2 rg = cell_range[(gz * GRID_MAX + gy) * GRID_MAX + gx];
                    for (uint32_t i = rg.x; i < rg.y; i++) {
                        const float4 v = sorted_verts[i];
                        const float ddx = x - v.x, ddy = y - v.y, ddz = z - v.z;
                        const float d2 = ddx * ddx + ddy * ddy + ddz * ddz;
                        if (d2 < worst) {
                            best_d[worst_slot][tid] = d2;
                            best_i[worst_slot][tid] = (int)__float_as_uint(v.w);
                            float w = -1.f;
                            int ws = 0;
                            for (int k = 0; k < K; k++) {
                                const float dk = best_d[k][tid];
                                if (dk > w) { w = dk; ws = k; }
                            }
                            worst = w;
                            worst_slot = ws;
                        }
                    }
                }
            }
        }
        // every vertex not visited yet lies outside the box of searched cells (or outside the grid, where there is
        // none): the distance from the query to the nearest still-open face bounds them from below
        float bound = 3.0e38f;
        bool open = false;
        const float qx = x - m.minx, qy = y - m.miny, qz = z - m.minz;
        if (cx - r > 0) { open = true; bound = fminf(bound, qx - (cx - r) * m.h); }
        if (cx + r < m.nx - 1) { open = true; bound = fminf(bound, (cx + r + 1) * m.h - qx); }
        if (cy - r > 0) { open = true; bound = fminf(bound, qy - (cy - r) * m.h); }
        if (cy + r < m.ny - 1) { open = true; bound = fminf(bound, (cy + r + 1) * m.h - qy); }
        if (cz - r > 0) { open = true; bound = fminf(bound, qz - (cz - r) * m.h); }
        if (cz + r < m.nz - 1) { open = true; bound = fminf(bound, (cz + r + 1) * m.h - qz); }
        if (!open) break;                                   // whole grid searched
        bound = fmaxf(bound, 0.f) * 0.9999f;                // rounding slack on the face positions
        if (worst <= bound * bound) break;                  // K-th best is final
    }

    // order the K hits by (distance, index)
    for (int i = 1; i < K; i++) {
        const float d = best_d[i][tid];
        const int id = best_i[i][tid];
        int j = i - 1;
        while (j >= 0 && (best_d[j][tid] > d || (best_d[j][tid] == d && best_i[j][tid] > id))) {
            best_d[j + 1][tid] = best_d[j][tid];
            best_i[j + 1][tid] = best_i[j][tid];
            j--;
        }
        best_d[j + 1][tid] = d;
        best_i[j + 1][tid] = id;
    }

    // ws = (1/d) / sum(1/d), d = clamp(sqrt(d2), 1e-4, 1)   (smpl.py:630-634)
    float norm = 0.f;
    for (int k = 0; k < K; k++) {
        const float d = fminf(fmaxf(sqrtf(best_d[k][tid]), 0.0001f), 1.0f);
        const float w = 1.0f / d;
        best_d[k][tid] = w;
        norm += w;
    }
    if (knn_idx_out)
        for (int k = 0; k < K; k++) knn_idx_out[(size_t)p * K + k] = best_i[k][tid];

    // weights[p, :] = sum_k ws_k * vert_weights[idx_k, :]   (smpl.py:632-635)
    float *out = weights_out + (size_t)p * J;
    for (int j0 = 0; j0 < J; j0 += 8) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < K; k++) {
            const int id = best_i[k][tid];
            if (id < 0) continue;
            const float w = best_d[k][tid] / norm;
            const float *row = vert_weights + (size_t)id * J + j0;
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (j0 + u < J) acc[u] += w * row[u];
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (j0 + u < J) out[j0 + u] = acc[u];
    }
}


// cell key of every query (clamped into the grid) for the query sort
__global__ void __launch_bounds__(256)
query_cells_kernel(const float *__restrict__ xyz, int P, const GridMeta *__restrict__ meta, uint32_t *__restrict__ keys,
                   uint32_t *__restrict__ vals)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const GridMeta m = *meta;
    const int cx = cell_coord(xyz[3 * p], m.minx, m.inv_h, m.nx);
    const int cy = cell_coord(xyz[3 * p + 1], m.miny, m.inv_h, m.ny);
    const int cz = cell_coord(xyz[3 * p + 2], m.minz, m.inv_h, m.nz);
    keys[p] = (uint32_t)((cz * GRID_MAX + cy) * GRID_MAX + cx);
    vals[p] = (uint32_t)p;
}

// cell keys of the queries taken in a given order (an earlier sort's permutation: positions move little between
// optimizer steps, so the order still groups most queries by cell; correctness never depends on it)
__global__ void __launch_bounds__(256)
query_cells_ordered_kernel(const float *__restrict__ xyz, int P, const GridMeta *__restrict__ meta,
                           const uint32_t *__restrict__ order, uint32_t *__restrict__ keys)
{
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= P) return;
    const GridMeta m = *meta;
    const uint32_t p = order[q];
    const int cx = cell_coord(xyz[3 * p], m.minx, m.inv_h, m.nx);
    const int cy = cell_coord(xyz[3 * p + 1], m.miny, m.inv_h, m.ny);
    const int cz = cell_coord(xyz[3 * p + 2], m.minz, m.inv_h, m.nz);
    keys[q] = (uint32_t)((cz * GRID_MAX + cy) * GRID_MAX + cx);
}

__device__ __forceinline__ float dist2_exact(float x, float y, float z, float4 v)
{
#pragma clang fp contract(off)            // both passes must see the same bits
    const float dx = x - v.x, dy = y - v.y, dz = z - v.z;
    return (dx * dx + dy * dy) + dz * dz;
}

// skinning rows in sorted-vertex order, padded to KNN_JMAX floats (16-byte aligned rows -> wide uniform loads)
__global__ void __launch_bounds__(256)
pad_rows_kernel(const float *__restrict__ vert_weights, int V, int J, const float4 *__restrict__ sorted_verts,
                float *__restrict__ rows)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= V * KNN_JMAX) return;
    const int i = e / KNN_JMAX, j = e % KNN_JMAX;
    const uint32_t v = __float_as_uint(sorted_verts[i].w);
    rows[e] = j < J ? vert_weights[(size_t)v * J + j] : 0.f;
}

constexpr int KNN_SLOTS = 4;             // wavefronts that share the cell segments of one 64-query chunk
constexpr int KNN_CAND = 768;            // candidates staged in LDS at a time (12 KB)
constexpr int KNN_RMAX = 3;              // beyond this box radius the search falls back to all vertices
constexpr int KNN_ROWS = (2 * KNN_RMAX + 1) * (2 * KNN_RMAX + 1);

constexpr int KNN_WAVES = 4;             // wavefronts per workgroup: they split the candidates (pass 1) and the queries (pass 2)
constexpr int KNN_STATE_STRIDE = 32;     // neighbour state: the 32 nearest vertices of a query (positions in the vertex grid's order)
// per-query state of soar_lbs_knn_refresh (layout: carve_knn_state)
struct KnnStatePtrs {        // (every array in the queries' sorted order: entry q belongs to query order[q])
    uint32_t *nbr;           // [P][32] the 32 nearest vertices, ascending grid position
    float4 *ref30, *ref32;   // [P] {position the 30-subset / the 32-set was determined at, half the gap behind it (0: unknown)}
    uint32_t *in30;          // [P] which of the 32 are the K = 30 nearest
    uint32_t *work;          // the work lists of the seeded search (layout: knn_work_* below)
    float *d2;               // [P][32] squared distances to the 32 stored vertices at this refresh's positions (< 0: the query is on the work list)
};

// Work lists: KNN_WORK_LISTS of them, each with its own counter in its own 256-byte line -- every query that fails its certificates
// takes a slot with a returning atomic, and ~3000 of those on ONE address cost the certificate launch 11 of its 22 us (they are
// served one after the other); 32 consecutive queries share a list, the lists' counters sit in different memory channels.
// Layout in words: [KNN_WORK_LISTS][64] {entries, searchers done, ...} (one line each) | [KNN_WORK_LISTS][cap] {query, bits of its search radius squared}
constexpr int KNN_WORK_LISTS = 64, KNN_WORK_LINE = 64;
__host__ __device__ __forceinline__ uint32_t knn_work_cap(int P) { return (uint32_t)(P > 0 ? P : 1) / KNN_WORK_LISTS + 64u; }   // >= 32 ceil(ceil(P / 32) / 64)
__host__ __device__ __forceinline__ size_t knn_work_words(int P) { return (size_t)KNN_WORK_LISTS * KNN_WORK_LINE + 2 * (size_t)KNN_WORK_LISTS * knn_work_cap(P); }
__device__ __forceinline__ uint32_t knn_work_list_of(uint32_t q) { return (q >> 5) & (KNN_WORK_LISTS - 1); }
__device__ __forceinline__ uint32_t *knn_work_item(uint32_t *work, int P, uint32_t list, uint32_t at)
{
    return work + KNN_WORK_LISTS * KNN_WORK_LINE + 2 * ((size_t)list * knn_work_cap(P) + at);
}

// First wavefront of the workgroup: list the contiguous sorted-vertex ranges covered by the box of radius r around
// (cx,cy,cz) (the cells of one grid row are adjacent keys) as prefix offsets; info[0] = number of candidates,
// info[1] = number of rows.
__device__ __forceinline__ void box_rows(const uint2 *__restrict__ cell_range, const GridMeta &m, int cx, int cy, int cz,
                                         int r, int V, uint32_t *row_start, uint32_t *row_prefix, int *info, int lane)
{
    if (r > KNN_RMAX) {                                      // brute force over all vertices
        if (lane == 0) { row_start[0] = 0u; row_prefix[0] = 0u; row_prefix[1] = (uint32_t)V; info[0] = V; info[1] = 1; }
        return;
    }
    const int side = 2 * r + 1;
    const int n_rows = side * side;                          // <= 49 <= 64: one row per lane
    uint32_t s = 0, e = 0;
    if (lane < n_rows) {
        const int gz = cz + lane / side - r, gy = cy + lane % side - r;
        if (gz >= 0 && gz < m.nz && gy >= 0 && gy < m.ny) {
            const int base = (gz * GRID_MAX + gy) * GRID_MAX;
            bool any = false;
            for (int gx = max(cx - r, 0); gx <= min(cx + r, m.nx - 1); gx++) {
                const uint2 rg = cell_range[base + gx];
                if (rg.y > rg.x) {
                    if (!any) s = rg.x;
                    e = rg.y;
                    any = true;
                }
            }
        }
    }
    // inclusive scan of the row lengths over the lanes
    uint32_t incl = e - s;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
        if (lane >= d) incl += up;
    }
    if (lane < n_rows) { row_start[lane] = s; row_prefix[lane + 1] = incl; }
    if (lane == 0) { row_prefix[0] = 0u; info[1] = n_rows; }
    if (lane == n_rows - 1) info[0] = (int)incl;
}

// Workgroup-cooperative: stage candidates [base, base + n) of the flattened box list into LDS (one coalesced 16-byte
// load per thread).  Caller synchronises.
__device__ __forceinline__ void stage_candidates(const float4 *__restrict__ sorted_verts, const uint32_t *row_start,
                                                 const uint32_t *row_prefix, int n_rows, int base, int n, float4 *cand)
{
    for (int t = threadIdx.x; t < n; t += KNN_WAVES * WAVE) {
        const uint32_t g = (uint32_t)(base + t);
        int row = 0;
        while (row + 1 < n_rows && row_prefix[row + 1] <= g) row++;
        const uint32_t pos = row_start[row] + (g - row_prefix[row]);
        float4 v = sorted_verts[pos];
        v.w = __uint_as_float(pos);
        cand[t] = v;
    }
}

template <int K>
__device__ __forceinline__ void chain_insert(float (&best)[K], float d)
{
    // sorted insertion, one op per slot: new b[k] = median(b[k-1], d, b[k])
#pragma unroll
    for (int k = K - 1; k > 0; k--) best[k] = __builtin_amdgcn_fmed3f(best[k - 1], d, best[k]);
    best[0] = fminf(best[0], d);
}

// Two ascending lists of K distances -> the K smallest of their union, ascending.  Elementwise min of one list against the other
// reversed (both padded to 32 with +inf) is a bitonic sequence that holds the 32 smallest; five compare-exchange stages sort it
// (80 min/max pairs instead of the 900 median steps of K chain insertions).  `other`: the second list, element k at other[k * stride].
template <int K>
struct PaddedList {                                  // best[0 .. K-1] followed by two more values: 32 slots without a second array
    float (&b)[K];
    float &x, &y;
    __device__ __forceinline__ float &operator[](int i) { return i < K ? b[i] : (i == K ? x : y); }
};
template <int K>
__device__ __forceinline__ void merge_sorted(float (&best)[K], const float *other, int stride)
{
    static_assert(K == 30, "two padding slots");
    // m[k] = min(best_padded[k], other_padded[31 - k]) in place: slots 30, 31 meet other[1], other[0]; slots 0, 1 meet the padding
    float x = other[1 * stride], y = other[0];
#pragma unroll
    for (int k = 2; k < K; k++) best[k] = fminf(best[k], other[(31 - k) * stride]);
    PaddedList<K> m{best, x, y};
#pragma unroll
    for (int j = 16; j >= 1; j >>= 1)
#pragma unroll
        for (int i = 0; i < 32; i++)
            if ((i & j) == 0) {
                const float lo = fminf(m[i], m[i + j]), hi = fmaxf(m[i], m[i + j]);
                m[i] = lo; m[i + j] = hi;
            }
}
// The same union when only its K-th smallest value and the number of elements strictly below it are wanted (the last merge):
// no sort at all.  Returns tau; need = how many elements AT tau belong to the K.
template <int K>
__device__ __forceinline__ float merge_sorted_kth(const float (&best)[K], const float *other, int stride, int &need)
{
    float v[K];
    float tau = 0.f;                                   // squared distances are >= 0
#pragma unroll
    for (int k = 0; k < K; k++) {
        v[k] = fminf(best[k], other[(K - 1 - k) * stride]);
        tau = fmaxf(tau, v[k]);
    }
    need = K;
#pragma unroll
    for (int k = 0; k < K; k++) need -= (v[k] < tau) ? 1 : 0;
    return tau;
}

// Heaviest-first order of the (chunk, slot) work items of the launch below.  An item's time grows with the vertices in the 3x3x3
// cell box of its segments' cells (pass 1) and with its number of queries (pass 2): at C3 126 of the 2217 workgroups with work
// scan 450+ candidates and run twice as long as the median, and a tenth of them used to start at 75 us of a 158 us launch (the
// second segment of a chunk sat in gridDim.y, behind every first segment).  Items without a segment go last, in one block: an
// empty workgroup in every second or fourth position would leave the CUs the dispatcher deals them to idle.
constexpr uint32_t KNN_ORDER_STAMP = 0x534F4152u;
// one wavefront per chunk, lane = query: cost of the chunk's (up to) KNN_SLOTS items, 0 = no segment
__global__ void __launch_bounds__(WAVE)
item_cost_kernel(const uint32_t *__restrict__ q_keys, int P, const GridMeta *__restrict__ meta, const uint2 *__restrict__ cell_range,
                 uint32_t *__restrict__ cost)
{
    const int lane = threadIdx.x, chunk = blockIdx.x, q = chunk * WAVE + lane;
    const bool valid = q < P;
    const uint32_t key = valid ? q_keys[q] : 0xFFFFFFFFu;
    const GridMeta m = *meta;
    uint32_t item[KNN_SLOTS];
#pragma unroll
    for (int k = 0; k < KNN_SLOTS; k++) item[k] = 0u;
    unsigned long long remaining = __ballot(valid);
    for (int seg = 0; remaining != 0ull; seg++) {                       // the same segments as knn_cell_kernel
        const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)key, (int)__builtin_ctzll(remaining));
        const unsigned long long mine = __ballot(valid && key == c);
        remaining &= ~mine;
        const int cx = (int)(c % GRID_MAX), cy = (int)((c / GRID_MAX) % GRID_MAX), cz = (int)(c / (GRID_MAX * GRID_MAX));
        uint32_t n = 0;                                                  // lane < 27: one cell of the box
        if (lane < 27) {
            const int gx = cx + lane % 3 - 1, gy = cy + (lane / 3) % 3 - 1, gz = cz + lane / 9 - 1;
            if (gx >= 0 && gx < m.nx && gy >= 0 && gy < m.ny && gz >= 0 && gz < m.nz) {
                const uint2 rg = cell_range[(gz * GRID_MAX + gy) * GRID_MAX + gx];
                n = rg.y - rg.x;
            }
        }
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) n += (uint32_t)__shfl_xor((int)n, off);
        n = (uint32_t)__builtin_amdgcn_readfirstlane((int)n);
        if (n < 2u * KNN_K) {                                            // a sparse box will grow: count the 5x5x5 one
            uint32_t n5 = 0;
            for (int t = lane; t < 125; t += WAVE) {
                const int gx = cx + t % 5 - 2, gy = cy + (t / 5) % 5 - 2, gz = cz + t / 25 - 2;
                if (gx >= 0 && gx < m.nx && gy >= 0 && gy < m.ny && gz >= 0 && gz < m.nz) {
                    const uint2 rg = cell_range[(gz * GRID_MAX + gy) * GRID_MAX + gx];
                    n5 += rg.y - rg.x;
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) n5 += (uint32_t)__shfl_xor((int)n5, off);
            n += (uint32_t)__builtin_amdgcn_readfirstlane((int)n5);      // both boxes are scanned
        }
        // measured: ~0.08 us per candidate + ~0.7 us per query + 12 us per segment (scripts/knn_log.py)
        const uint32_t us100 = 8u * n + 70u * (uint32_t)__builtin_popcountll(mine) + 1200u;
#pragma unroll
        for (int k = 0; k < KNN_SLOTS; k++) item[k] += (seg % KNN_SLOTS) == k ? us100 : 0u;
    }
    if (lane < KNN_SLOTS) cost[chunk * KNN_SLOTS + lane] = lane == 0 ? item[0] : lane == 1 ? item[1] : lane == 2 ? item[2] : item[3];
}
// counting sort of the items into 32 cost classes, heaviest first, items without work last (one workgroup; the order inside a
// class is whatever the LDS atomics give: it only schedules)
__global__ void __launch_bounds__(1024) item_order_kernel(const uint32_t *__restrict__ cost, int n_items, uint32_t *__restrict__ order)
{
    __shared__ uint32_t count[33], cursor[33];
    const int tid = threadIdx.x;
    if (tid < 33) count[tid] = 0u;
    __syncthreads();
    auto class_of = [&](uint32_t c) -> int { return c == 0u ? 32 : 31 - (int)min(31u, c / 400u); };     // 4 us per class
    for (int i = tid; i < n_items; i += 1024) atomicAdd(&count[class_of(cost[i])], 1u);
    __syncthreads();
    if (tid == 0) {
        uint32_t acc = 0;
        for (int k = 0; k < 33; k++) { cursor[k] = acc; acc += count[k]; }
    }
    __syncthreads();
    for (int i = tid; i < n_items; i += 1024) order[atomicAdd(&cursor[class_of(cost[i])], 1u)] = (uint32_t)i;
    if (tid == 0) order[n_items] = KNN_ORDER_STAMP ^ (uint32_t)n_items;      // "this workspace holds an order for n_items items"
}

template <bool WITH_IDX, bool LOG = false>
// five wavefronts per SIMD (96 VGPRs, 16 bytes of scratch per lane) against four at 110: 139 -> 132 us
__global__ void __launch_bounds__(KNN_WAVES *WAVE) __attribute__((amdgpu_waves_per_eu(5, 8)))
knn_cell_kernel(const float *__restrict__ xyz, int P, int V, const GridMeta *__restrict__ meta,
                const uint2 *__restrict__ cell_range, const float4 *__restrict__ sorted_verts,
                const uint32_t *__restrict__ q_keys, const uint32_t *__restrict__ q_ids,
                const float *__restrict__ rows_padded, int J, float *__restrict__ weights_out,
                int32_t *__restrict__ knn_idx_out, const uint32_t *__restrict__ item_order,
                KnnStatePtrs state,
                unsigned long long *__restrict__ wave_log = nullptr)
{
    constexpr int K = KNN_K;
    __shared__ float4 cand[KNN_CAND];                      // {x, y, z, sorted position}
    __shared__ uint32_t row_start[KNN_ROWS], row_prefix[KNN_ROWS + 1];
    __shared__ int info[4];                                // candidates, rows, "box is final"
    __shared__ float merge[2][K][WAVE];                    // top-K lists handed between wavefronts
    __shared__ float tau_s[WAVE];
    __shared__ int need_s[WAVE];
    __shared__ uint32_t list_pos[KNN_WAVES][WAVE];         // the (<= K) neighbours of the query a wavefront is blending
    __shared__ float list_w[KNN_WAVES][WAVE];
    unsigned long long t_start = 0, n_cand = 0, n_blend = 0, n_pairs = 0;
    if (LOG) t_start = wall_clock64();

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // heaviest items first -- when the workspace holds an order for this launch (a call that keeps a stored query order on a
    // workspace no sort has been through finds no stamp and takes the items in place)
    const uint32_t n_items = gridDim.x;
    const bool ordered = item_order && item_order[n_items] == (KNN_ORDER_STAMP ^ n_items);
    // (in place = every chunk's first segment, then every second one, ...: never an idle workgroup in a periodic position)
    const uint32_t n_chunks = n_items / KNN_SLOTS;
    const uint32_t item = ordered ? min(item_order[blockIdx.x], n_items - 1u) : (blockIdx.x % n_chunks) * KNN_SLOTS + blockIdx.x / n_chunks;
    const int chunk = (int)(item / KNN_SLOTS), slot = (int)(item % KNN_SLOTS);
    const int q = chunk * WAVE + lane;
    const bool valid = q < P;
    const uint32_t key = valid ? q_keys[q] : 0xFFFFFFFFu;
    // a segment = ALL the lanes of the chunk that share one cell (adjacent when the query order is fresh; with a reused order the
    // same cell can come back later in the chunk -- one segment all the same, or two workgroups would blend the same queries);
    // this workgroup takes segments slot, slot + KNN_SLOTS, ... so that a chunk that straddles many sparse cells is shared by
    // several workgroups
    unsigned long long remaining = __ballot(valid);
    if (remaining == 0ull) return;

    const int p = valid ? (int)q_ids[q] : 0;
    const float x = xyz[3 * p], y = xyz[3 * p + 1], z = xyz[3 * p + 2];
    const GridMeta m = *meta;
    const float qx = x - m.minx, qy = y - m.miny, qz = z - m.minz;

    for (int seg = 0; remaining != 0ull; seg++) {
        const int head = (int)__builtin_ctzll(remaining);
        const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)key, head);
        const bool active = valid && key == c;
        remaining &= ~__ballot(active);
        if ((seg % KNN_SLOTS) != slot) continue;
        const int cx = (int)(c % GRID_MAX), cy = (int)((c / GRID_MAX) % GRID_MAX), cz = (int)(c / (GRID_MAX * GRID_MAX));
        if (LOG) n_pairs++;

        // ---- pass 1 (lane = query): the K smallest squared distances of every query; each wavefront scans a quarter
        //      of the candidates, the four sorted lists are merged through LDS; the box grows until every query is
        //      certified (its K-th distance is below the distance to the nearest open face of the box)
        int r = 1, n_rows = 0, N = 0;
        for (;; r++) {
            __syncthreads();
            if (wave == 0) box_rows(cell_range, m, cx, cy, cz, r, V, row_start, row_prefix, info, lane);
            __syncthreads();
            N = info[0];
            n_rows = info[1];
            if (LOG) n_cand += N;
            float best[K];
#pragma unroll
            for (int k = 0; k < K; k++) best[k] = 3.0e38f;
            for (int base = 0; base < N; base += KNN_CAND) {
                const int n = min(KNN_CAND, N - base);
                if (base > 0) __syncthreads();
                stage_candidates(sorted_verts, row_start, row_prefix, n_rows, base, n, cand);
                __syncthreads();
#pragma unroll 2
                for (int i = wave; i < n; i += KNN_WAVES) chain_insert<K>(best, dist2_exact(x, y, z, cand[i]));
            }
            // merge: waves 2,3 -> waves 0,1 ; wave 1 -> wave 0
            if (wave >= 2) {
#pragma unroll
                for (int k = 0; k < K; k++) merge[wave - 2][k][lane] = best[k];
            }
            __syncthreads();
            if (wave < 2) merge_sorted<K>(best, &merge[wave][0][lane], WAVE);
            __syncthreads();
            if (wave == 1) {
#pragma unroll
                for (int k = 0; k < K; k++) merge[0][k][lane] = best[k];
            }
            __syncthreads();
            if (wave == 0) {
                int need;                                             // how many candidates AT tau the query takes
                const float tau = merge_sorted_kth<K>(best, &merge[0][0][lane], WAVE, need);
                tau_s[lane] = tau;
                need_s[lane] = need;
                // vertices not visited yet lie beyond an open face of the box: the distance to the nearest one bounds them
                float bound = 3.0e38f;
                bool open = false;
                if (cx - r > 0) { open = true; bound = fminf(bound, qx - (cx - r) * m.h); }
                if (cx + r < m.nx - 1) { open = true; bound = fminf(bound, (cx + r + 1) * m.h - qx); }
                if (cy - r > 0) { open = true; bound = fminf(bound, qy - (cy - r) * m.h); }
                if (cy + r < m.ny - 1) { open = true; bound = fminf(bound, (cy + r + 1) * m.h - qy); }
                if (cz - r > 0) { open = true; bound = fminf(bound, qz - (cz - r) * m.h); }
                if (cz + r < m.nz - 1) { open = true; bound = fminf(bound, (cz + r + 1) * m.h - qz); }
                bound = fmaxf(bound, 0.f) * 0.9999f;                  // rounding slack on the face positions
                const bool certified = !open || r > KNN_RMAX || tau <= bound * bound;
                const bool final_box = __ballot(active && !certified) == 0ull;
                if (lane == 0) info[2] = final_box ? 1 : 0;
            }
            __syncthreads();
            if (info[2]) break;
        }
        const float tau = tau_s[lane];
        const int need = need_s[lane];

        // ---- pass 2 (one query at a time per wavefront; lanes = candidates, then lanes = joints): the neighbours of
        //      a query are the candidates below its K-th distance (ties at the K-th place in list order until K are
        //      reached); their skinning rows are blended with lanes = joints: row reads and the result row coalesce.
        const bool single = N <= KNN_CAND;                            // candidates still staged from pass 1
        float norm_lane = 0.f;                                        // lane = query: sum of 1/d over all chunks
        int cnt_lane = 0;
        int need_left = need;                                         // lane = query: candidates AT tau still to take (the
                                                                      // count runs on through the chunks of a large box)
        uint32_t *lpos = list_pos[wave];
        float *lw = list_w[wave];
        for (int base = 0; base < N; base += KNN_CAND) {
            const int n = min(KNN_CAND, N - base);
            if (!single) {
                __syncthreads();
                stage_candidates(sorted_verts, row_start, row_prefix, n_rows, base, n, cand);
                __syncthreads();
            }
            int ordinal = 0;
            for (unsigned long long todo = __ballot(active); todo != 0ull; todo &= todo - 1ull, ordinal++) {
                if ((ordinal % KNN_WAVES) != wave) continue;
                const int l = (int)__builtin_ctzll(todo);
                const float ux = __shfl(x, l), uy = __shfl(y, l), uz = __shfl(z, l), utau = __shfl(tau, l);
                const int pl = __builtin_amdgcn_readlane(p, l);
                int uneed = __builtin_amdgcn_readlane(need_left, l);
                const int taken0 = __builtin_amdgcn_readlane(cnt_lane, l);
                int cnt = 0;
                __builtin_amdgcn_wave_barrier();
                for (int t0 = 0; t0 < n; t0 += WAVE) {
                    const int g = t0 + lane;
                    const float4 v = cand[min(g, n - 1)];
                    const float d = dist2_exact(ux, uy, uz, v);
                    const bool lt = g < n && d < utau, eq = g < n && d == utau;
                    const unsigned long long eqm = __ballot(eq);
                    const int eq_rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(eqm >> 32),
                                                                       __builtin_amdgcn_mbcnt_lo((uint32_t)eqm, 0u));
                    const bool take = lt || (eq && eq_rank < uneed);
                    uneed = max(0, uneed - (int)__builtin_popcountll(eqm));
                    const unsigned long long tm = __ballot(take);
                    if (tm == 0ull) continue;
                    const int at = cnt + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(tm >> 32),
                                                                       __builtin_amdgcn_mbcnt_lo((uint32_t)tm, 0u));
                    if (take && at < WAVE) { lpos[at] = __float_as_uint(v.w); lw[at] = d; }
                    cnt += (int)__builtin_popcountll(tm);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                cnt = min(cnt, WAVE);
                if (LOG) n_blend += cnt;
                // ws = (1/d) / sum(1/d), d = clamp(sqrt(d2), 1e-4, 1)   (smpl.py:630-634); normalised at the end
                if (lane < cnt) lw[lane] = 1.0f / fminf(fmaxf(sqrtf(lw[lane]), 0.0001f), 1.0f);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                // blend: lanes = joints.  ONE chain of fused multiply-adds over the neighbours in vertex-grid order, whatever the
                // number of candidate chunks the box needed: the result is a function of the query and its neighbour set alone
                // (soar_lbs_knn_refresh below reproduces it from a stored set, bit for bit)
                const int jl = min(lane, KNN_JMAX - 1);
                float accj = (base == 0 || lane >= J) ? 0.f : weights_out[(size_t)pl * J + lane];
                float norm = base == 0 ? 0.f : __shfl(norm_lane, l);
#pragma unroll 15
                for (int k = 0; k < cnt; k++) {
                    const float wk = lw[k];
                    accj = __builtin_fmaf(wk, rows_padded[(size_t)lpos[k] * KNN_JMAX + jl], accj);
                    norm += wk;
                }
                if (WITH_IDX && lane < cnt && taken0 + lane < K)
                    knn_idx_out[(size_t)pl * K + taken0 + lane] = (int)__float_as_uint(sorted_verts[lpos[lane]].w);
                if (state.nbr) {
                    // neighbour state for soar_lbs_knn_refresh: the K nearest (positions in the vertex grid's order, ascending; the two
                    // spare slots repeat the last one) and gaps of 0 -- the first refresh searches, seeded by these, and measures them
                    // (the state is kept in the queries' SORTED order, like q_ids: the refresh walks it front to back)
                    const size_t ql = (size_t)chunk * WAVE + l;
                    if (lane < cnt && taken0 + lane < K) state.nbr[ql * KNN_STATE_STRIDE + taken0 + lane] = lpos[lane];
                    if (lane < 2 && cnt > 0 && taken0 + cnt >= K) state.nbr[ql * KNN_STATE_STRIDE + K + lane] = lpos[cnt - 1];
                    if (lane == 0) {
                        state.ref30[ql] = make_float4(ux, uy, uz, 0.f);
                        state.ref32[ql] = make_float4(ux, uy, uz, 0.f);
                        state.in30[ql] = 0x3FFFFFFFu;
                    }
                }
                if (lane == l) { norm_lane = norm; cnt_lane += cnt; need_left = uneed; }
                if (lane < J) {
                    float *out = weights_out + (size_t)pl * J + lane;
                    *out = single ? accj / norm : accj;
                }
            }
        }
        if (!single) {                                                // normalise the rows accumulated over several chunks
            int ordinal = 0;
            for (unsigned long long todo = __ballot(active); todo != 0ull; todo &= todo - 1ull, ordinal++) {
                if ((ordinal % KNN_WAVES) != wave) continue;
                const int l = (int)__builtin_ctzll(todo);
                const int pl = __builtin_amdgcn_readlane(p, l);
                const float nl = __shfl(norm_lane, l);
                if (lane < J) weights_out[(size_t)pl * J + lane] /= nl;
            }
        }
    }
    if (LOG && lane == 0) {
        unsigned long long *w = wave_log + ((size_t)blockIdx.x * KNN_WAVES + wave) * 8;
        w[0] = t_start; w[1] = wall_clock64(); w[2] = n_pairs; w[3] = n_cand; w[4] = n_blend;
    }
}


// ---- neighbour sets that follow the queries ------------------------------------------------------------------------------------
// The canonical vertices never move (TS/utils/smpl.py:508-511) and a query moves by an optimizer step: the K = 30 nearest vertices
// of almost every query are those of the step before.  For every vertex |d(x, v) - d(x_ref, v)| <= |x - x_ref|: while a query has
// moved less than half the gap between its 30th and 31st distance at x_ref (minus rounding slack), its 30 stored vertices are still
// strictly closer than every other one -- the set the full search would return, certified without searching.  About one query in
// fifteen has a gap smaller than an optimizer step and would fail that every time; so the state holds the 32 nearest vertices and
// a second gap, behind the 32nd: while THAT certificate holds, the 30 nearest are among the stored 32 and are found by ranking 32
// distances in registers.  Only a query that fails both (under 1 % of them per step at the reference's learning rate) is searched
// again, exactly, seeded by its old set: every vertex it can want lies within the largest new distance to an old neighbour, so only
// the grid cells that ball touches are scanned.
// Whatever the tier, the K distances, the inverse-distance weights and the blend of the K skinning rows are then recomputed with the
// expressions and in the (vertex-grid) order of knn_cell_kernel: the weights are the full search's bit for bit.
#ifndef SOAR_KNN_SLOT_UNROLL
#define SOAR_KNN_SLOT_UNROLL 5
#endif
#ifndef SOAR_KNN_SEARCH_UNROLL
#define SOAR_KNN_SEARCH_UNROLL 8
#endif
constexpr int KNN_KEEP = KNN_STATE_STRIDE;   // neighbours kept per query
constexpr int RF_CAP = 192;              // candidates the seeded search holds between two selections (KNN_KEEP + a chunk of 64 fit)

__device__ __forceinline__ float half_gap(float near_d2, float far_d2)
{
    const float far = sqrtf(far_d2);
    return 0.5f * (far - sqrtf(near_d2)) - 4.0e-6f * far - 1.0e-12f;          // rounding of the distances, generously
}

// lanes 0..31 hold (pos ascending, d2) of the 32 stored vertices: the mask of the K nearest among them ((distance, grid position)
// order: the full search's tie rule) and half the gap behind the K-th
__device__ __forceinline__ uint32_t rank_32(float d2, int lane, float &h30)
{
    int rank = 0;
#pragma unroll
    for (int j = 0; j < KNN_KEEP; j++) {
        const float dj = __shfl(d2, j);
        rank += (dj < d2 || (dj == d2 && j < lane)) ? 1 : 0;
    }
    const bool mine = lane < KNN_KEEP;
    const bool in = mine && rank < KNN_K;
    float d_in = in ? d2 : 0.f, d_out = (mine && !in) ? d2 : 3.0e38f;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        d_in = fmaxf(d_in, __shfl_xor(d_in, off));
        d_out = fminf(d_out, __shfl_xor(d_out, off));
    }
    h30 = half_gap(d_in, d_out);
    return (uint32_t)__ballot(in);
}

// The blend of the K in-set neighbours' skinning rows, ascending grid position, FOUR queries per trip.  One query's blend with lanes =
// joints is a chain of 30 (LDS read, row load, multiply-add) steps that keeps 55 lanes busy with one float each -- the kernel is bound by the number of instructions it issues, not by the
// rows' bytes (profiles/README.md, round 4).  Here a slot holds a query's list {row byte offset, weight} x 30; sixteen lanes take a
// slot, lane t of them the joints 4t .. 4t+3 as ONE 16-byte load per row (rows are 224 bytes: 14 lanes of the sixteen), so a
// trip issues a quarter of the loads and LDS reads per query and no more multiply-adds.  Every (query, joint) sum is the same chain of
// fused multiply-adds over the neighbours in ascending grid position, the norm the same chain of additions as in knn_cell_kernel's
// blend: bit for bit the full search's weights.
constexpr int KNN_FSLOTS = 4;
struct KnnSlots { uint2 list[KNN_FSLOTS][KNN_KEEP]; int p[KNN_FSLOTS]; };
// lanes 0..31 / 32..63 hold (pos ascending, d2) of the 32 stored vertices of the queries of slot_lo / slot_lo + 1 (p < 0: no query)
__device__ __forceinline__ void slots_fill_pair(KnnSlots &sl, int slot_lo, int p, uint32_t mask, uint32_t pos, float d2, int lane)
{
    const int slot = slot_lo + (lane >> 5), l5 = lane & (KNN_KEEP - 1);
    if (p >= 0 && ((mask >> l5) & 1u)) {
        const int at = __builtin_popcount(mask & ((1u << l5) - 1u));
        // ws = (1/d) / sum(1/d), d = clamp(sqrt(d2), 1e-4, 1) (smpl.py:630-634); the row's BYTE offset (V x 224 bytes stay far below 2^32)
        sl.list[slot][at] = make_uint2(pos * (uint32_t)(KNN_JMAX * sizeof(float)), __float_as_uint(1.0f / fminf(fmaxf(sqrtf(d2), 0.0001f), 1.0f)));
    }
    if (l5 == 0) sl.p[slot] = p;
}
template <int UNROLL>
__device__ __forceinline__ void slots_blend(KnnSlots &sl, int lane, const float *__restrict__ rows_padded, int J,
                                            float *__restrict__ weights_out)
{
    static_assert(KNN_K % 2 == 0 && KNN_JMAX % 4 == 0 && KNN_JMAX / 4 <= 16 && KNN_FSLOTS * 16 == WAVE && KNN_KEEP * 2 == WAVE &&
                  KNN_JMAX * sizeof(float) <= KNN_KEEP * sizeof(uint2) && KNN_JMAX <= WAVE, "slot layout");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int g = lane >> 4, t = lane & 15;
    const int p = sl.p[g];
    if (p >= 0) {
        const uint32_t t16 = (uint32_t)min(t, KNN_JMAX / 4 - 1) * 16u;
        const char *rows_bytes = reinterpret_cast<const char *>(rows_padded);
        const uint4 *mine = reinterpret_cast<const uint4 *>(sl.list[g]);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        float norm = 0.f;
#pragma unroll UNROLL
        for (int k2 = 0; k2 < KNN_K / 2; k2++) {
            const uint4 e = mine[k2];
            const float4 r0 = *reinterpret_cast<const float4 *>(rows_bytes + (e.x + t16));
            const float4 r1 = *reinterpret_cast<const float4 *>(rows_bytes + (e.z + t16));
            const float w0 = __uint_as_float(e.y), w1 = __uint_as_float(e.w);
            acc.x = __builtin_fmaf(w0, r0.x, acc.x); acc.y = __builtin_fmaf(w0, r0.y, acc.y);
            acc.z = __builtin_fmaf(w0, r0.z, acc.z); acc.w = __builtin_fmaf(w0, r0.w, acc.w);
            norm += w0;
            acc.x = __builtin_fmaf(w1, r1.x, acc.x); acc.y = __builtin_fmaf(w1, r1.y, acc.y);
            acc.z = __builtin_fmaf(w1, r1.z, acc.z); acc.w = __builtin_fmaf(w1, r1.w, acc.w);
            norm += w1;
        }
        // (the lists are read: the slot's row of results goes where its list was -- [KNN_JMAX] floats fit in [KNN_KEEP] pairs)
        float4 *mine_out = reinterpret_cast<float4 *>(sl.list[g]);
        if (t < KNN_JMAX / 4) mine_out[t] = make_float4(acc.x / norm, acc.y / norm, acc.z / norm, acc.w / norm);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // one row per store: lane j writes joint j (220 contiguous bytes; with lanes = (slot, four joints) a store touched all four rows
    // in 4-byte pieces 16 bytes apart)
#pragma unroll
    for (int r = 0; r < KNN_FSLOTS; r++) {
        const int pr = sl.p[r];                                        // (wave-uniform)
        if (pr >= 0 && lane < J) weights_out[(size_t)pr * J + lane] = reinterpret_cast<const float *>(sl.list[r])[lane];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// the same for TWO queries at once: lanes 0..31 hold the 32 stored vertices of one query, lanes 32..63 those of another; each half
// gets its own mask and gap (the distances of a half travel by ds_bpermute instead of v_readlane)
__device__ __forceinline__ uint32_t rank_32_halves(float d2, int lane, float &h30)
{
    const int l5 = lane & (KNN_KEEP - 1), base = lane & KNN_KEEP;
    int rank = 0;
#pragma unroll
    for (int j = 0; j < KNN_KEEP; j++) {
        const float dj = __shfl(d2, base + j);
        rank += (dj < d2 || (dj == d2 && j < l5)) ? 1 : 0;
    }
    const bool in = rank < KNN_K;
    float d_in = in ? d2 : 0.f, d_out = in ? 3.0e38f : d2;
#pragma unroll
    for (int off = KNN_KEEP / 2; off > 0; off >>= 1) {
        d_in = fmaxf(d_in, __shfl_xor(d_in, off));
        d_out = fminf(d_out, __shfl_xor(d_out, off));
    }
    h30 = half_gap(d_in, d_out);
    const unsigned long long b = __ballot(in);
    return (uint32_t)(base ? (b >> 32) : b);
}

// First launch of a refresh -- tiers 1 and 2: which queries keep their neighbour sets.  Half a wavefront per query (lane = stored
// vertex; what belongs to the query itself -- id, position, reference points -- is loaded by all 32 lanes from one address: one
// request), four pairs of queries per wavefront with every load of the four in flight before the first is used: a query is a chain
// of two dependent reads (its id -> its position; its neighbour list -> their coordinates) around very little arithmetic.  Leaves the
// 32 distances of a certified query for the blend (st.d2), -1 for a query that goes on the work list.
#ifndef SOAR_KNN_CERT_PAIRS
#define SOAR_KNN_CERT_PAIRS 2
#endif
constexpr int KNN_CERT_PAIRS = SOAR_KNN_CERT_PAIRS;
__global__ void __launch_bounds__(KNN_WAVES *WAVE)
knn_certify_kernel(const float *__restrict__ xyz, int P, const float4 *__restrict__ sorted_verts, const uint32_t *__restrict__ order,
                   KnnStatePtrs st)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l5 = lane & (KNN_KEEP - 1);
    const int q_base = ((int)blockIdx.x * KNN_WAVES + wave) * (2 * KNN_CERT_PAIRS) + (lane >> 5);
    if (q_base - (lane >> 5) >= P) return;
    int qs[KNN_CERT_PAIRS];
    float4 ra[KNN_CERT_PAIRS], rb[KNN_CERT_PAIRS], vv[KNN_CERT_PAIRS];
    float xs[KNN_CERT_PAIRS], ys[KNN_CERT_PAIRS], zs[KNN_CERT_PAIRS];
#pragma unroll
    for (int i = 0; i < KNN_CERT_PAIRS; i++) {
        qs[i] = q_base + 2 * i;
        const int qc = min(qs[i], P - 1);
        const int p = order ? (int)order[qc] : qc;
        const uint32_t pos = st.nbr[(size_t)qc * KNN_STATE_STRIDE + l5];
        ra[i] = st.ref30[qc]; rb[i] = st.ref32[qc];
        xs[i] = xyz[3 * p]; ys[i] = xyz[3 * p + 1]; zs[i] = xyz[3 * p + 2];
        vv[i] = sorted_verts[pos];
    }
#pragma unroll
    for (int i = 0; i < KNN_CERT_PAIRS; i++) {
        const int q = qs[i];
        const bool valid = q < P;
        const float x = xs[i], y = ys[i], z = zs[i];
        const float d2 = dist2_exact(x, y, z, vv[i]);
        const float ax = x - ra[i].x, ay = y - ra[i].y, az = z - ra[i].z, bx = x - rb[i].x, by = y - rb[i].y, bz = z - rb[i].z;
        // (an upper bound of the displacement is all a certificate needs -- v_sqrt_f32 is within one ulp, the factor covers 1e-4 --;
        // the correctly rounded square roots were a sixth of this launch's instructions.  Which tier a query takes may differ by
        // that; its weights never do)
        const float moved_a = __builtin_amdgcn_sqrtf((ax * ax + ay * ay) + az * az) * 1.0001f + 1.0e-12f;
        const float moved_b = __builtin_amdgcn_sqrtf((bx * bx + by * by) + bz * bz) * 1.0001f + 1.0e-12f;
        const bool ok_a = ra[i].w > 0.f && moved_a < ra[i].w, ok_b = rb[i].w > 0.f && moved_b < rb[i].w;
        const bool tier3 = valid && !ok_a && !ok_b, tier2 = valid && !ok_a && ok_b;
        if (__ballot(tier3)) {
            // neither certificate holds: the seeded search (knn_blend_search_kernel) takes the query, with the largest new distance to a
            // stored neighbour -- the radius of its search
            float far = d2;
#pragma unroll
            for (int off = KNN_KEEP / 2; off > 0; off >>= 1) far = fmaxf(far, __shfl_xor(far, off));
            if (tier3 && l5 == 0) {
                const uint32_t list = knn_work_list_of((uint32_t)q);
                uint32_t *item = knn_work_item(st.work, P, list, atomicAdd(st.work + list * KNN_WORK_LINE, 1u));
                item[0] = (uint32_t)q;
                item[1] = __float_as_uint(far);
            }
        }
        if (__ballot(tier2)) {
            // the 32 stored vertices still are the 32 nearest: the K nearest among them, and the gap behind the K-th, from here
            float h30;
            const uint32_t mask = rank_32_halves(d2, lane, h30);
            if (tier2 && l5 == 0) { st.ref30[q] = make_float4(x, y, z, h30); st.in30[q] = mask; }
        }
        if (valid) st.d2[(size_t)q * KNN_STATE_STRIDE + l5] = tier3 ? -1.f : d2;
    }
}

// The second launch of a refresh.  Its first `search_blocks` workgroups take the work list (tier 3: one wavefront per query; a seeded
// search is ~20 us of dependent steps whatever the list's length), the others blend the certified queries, four per wavefront, from
// what knn_certify_kernel left (distances, in-set masks): the searches run UNDER the blends instead of behind them.
#ifndef SOAR_KNN_SEARCH_BLOCKS
#define SOAR_KNN_SEARCH_BLOCKS 768
#endif
#ifndef SOAR_KNN_BS_WPE
#define SOAR_KNN_BS_WPE 4
#endif
__global__ void __launch_bounds__(KNN_WAVES *WAVE) __attribute__((amdgpu_waves_per_eu(SOAR_KNN_BS_WPE, 8)))
knn_blend_search_kernel(const float *__restrict__ xyz, int P, int search_blocks, const GridMeta *__restrict__ meta,
                        const uint2 *__restrict__ cell_range, const float4 *__restrict__ sorted_verts, const float *__restrict__ rows_padded,
                        int J, const uint32_t *__restrict__ order, KnnStatePtrs st, float *__restrict__ weights_out,
                        uint32_t *__restrict__ counters)
{
    constexpr int KEEP = KNN_KEEP;
    __shared__ uint32_t c_pos[KNN_WAVES][RF_CAP];
    __shared__ float c_d[KNN_WAVES][RF_CAP];
    __shared__ uint32_t row_first[KNN_WAVES][WAVE], row_end[KNN_WAVES][WAVE];
    __shared__ __attribute__((aligned(16))) KnnSlots slots[KNN_WAVES];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if ((int)blockIdx.x >= search_blocks) {
        // ---- certified queries: consecutive ones of the (cell-sorted) order share most neighbours -- the rows mostly come from the CU's L1
        const int q0 = (((int)blockIdx.x - search_blocks) * KNN_WAVES + wave) * KNN_FSLOTS;
        if (q0 >= P) return;
        const int l5 = lane & (KNN_KEEP - 1);
        float dd[2];
        uint32_t pp[2], mm[2];
        int pq[2];
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int q = q0 + 2 * r + (lane >> 5);
            const size_t at = (size_t)min(q, P - 1) * KNN_STATE_STRIDE + l5;
            dd[r] = st.d2[at];
            pp[r] = st.nbr[at];
            mm[r] = st.in30[min(q, P - 1)];
            pq[r] = order ? (int)order[min(q, P - 1)] : min(q, P - 1);
            if (q >= P) dd[r] = -1.f;
        }
#pragma unroll
        for (int r = 0; r < 2; r++) slots_fill_pair(slots[wave], 2 * r, dd[r] >= 0.f ? pq[r] : -1, mm[r], pp[r], dd[r], lane);
        slots_blend<SOAR_KNN_SLOT_UNROLL>(slots[wave], lane, rows_padded, J, weights_out);
        return;
    }
    uint32_t *cp = c_pos[wave];
    float *cd = c_d[wave];
    const GridMeta m = *meta;
    // wavefront g of the searchers takes list g % 64 (the four of a workgroup different ones), items g / 64, g / 64 + searchers / 64, ...
    const uint32_t searcher = blockIdx.x * KNN_WAVES + wave, my_list = searcher % KNN_WORK_LISTS;
    const uint32_t n_mine = min(st.work[my_list * KNN_WORK_LINE], knn_work_cap(P));
    for (uint32_t w = searcher / KNN_WORK_LISTS; w < n_mine; w += (uint32_t)search_blocks * KNN_WAVES / KNN_WORK_LISTS) {
        const uint32_t *item = knn_work_item(st.work, P, my_list, w);
        const size_t q = item[0];
#ifdef SOAR_KNN_SEARCH_LOG
        const unsigned long long lt0 = wall_clock64();
        unsigned long long lt1 = lt0, lt2 = lt0;
        int l_sel = 0, l_cand = 0, l_rows = 0, l_nin = 0;
#endif
        // everything within the largest new distance to an old neighbour (there are at least K vertices that close) ...
        const float tau_ub = __uint_as_float(item[1]);
        const int p = order ? (int)order[q] : (int)q;
        const float x = xyz[3 * p], y = xyz[3 * p + 1], z = xyz[3 * p + 2];
        int n_in = 0;                                                 // candidates held in (cp, cd), ascending grid position
        float next_d2 = 3.0e38f, ball2 = 0.f;
        // A query that is not a number (a parameter that diverged under the optimizer) compares false with every radius: its ball
        // would never hold a vertex and the loop below would never end.  It gets NaN weights (what the full search's arithmetic
        // gives it) and certificates that fail, so that it comes back here until it is a number again.
        bool sane = (x - x == 0.f) && (y - y == 0.f) && (z - z == 0.f) && (tau_ub >= 0.f);
        int growths = 0;
        // ... and a fifteenth further: what the search proves about the vertices it does NOT return is "at least the ball's radius
        // away" -- with the bare radius the gap behind the set would always be measured as zero (on a surface the K-th and
        // (K+1)-th distances differ by ~1/60 of the K-th on average).  Enlarged again should the ball hold fewer than KEEP vertices
        // (the state of a fresh full search only knows K of them)
        for (float grow = 1.0f + 1.0f / 15.0f; sane; grow *= 1.3f) {
            // (V >= KEEP is checked by the host: a ball that holds every vertex ends the loop; the count bounds it whatever happens:
            // 1.3^96 is 1e11 times the first radius)
            if (++growths > 96) { sane = false; break; }
            const float ball = sqrtf(tau_ub) * grow + 1.0e-6f;
            ball2 = ball * ball;
            const float rho = ball * 1.0001f + 1.0e-7f;
            const int cx0 = cell_coord(x - rho, m.minx, m.inv_h, m.nx), cx1 = cell_coord(x + rho, m.minx, m.inv_h, m.nx);
            const int cy0 = cell_coord(y - rho, m.miny, m.inv_h, m.ny), cy1 = cell_coord(y + rho, m.miny, m.inv_h, m.ny);
            const int cz0 = cell_coord(z - rho, m.minz, m.inv_h, m.nz), cz1 = cell_coord(z + rho, m.minz, m.inv_h, m.nz);
            const int ny_b = cy1 - cy0 + 1, n_rows = ny_b * (cz1 - cz0 + 1);
            n_in = 0;
            next_d2 = 3.0e38f;
            // keep the KEEP best of the n_in held candidates (ties in list = grid order), remember the best one dropped
            auto select_keep = [&]() {
#ifdef SOAR_KNN_SEARCH_LOG
                l_sel++;
#endif
                if (n_in <= WAVE) {
                    // the usual case (a ball seeded by the old set holds ~38 vertices): one candidate per lane, the others' distances by
                    // v_readlane instead of an LDS read per candidate and three slots of bookkeeping
                    const bool have = lane < n_in;
                    const uint32_t pos0 = have ? cp[lane] : 0u;
                    const float d0 = have ? cd[lane] : 3.0e38f;
                    int rk = 0;
                    for (int j = 0; j < n_in; j++) {
                        const float dj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d0), j));
                        rk += (dj < d0 || (dj == d0 && j < lane)) ? 1 : 0;
                    }
                    const bool keep = have && rk < KEEP;
                    const unsigned long long km = __ballot(keep);
                    if (keep) {
                        const int at = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0u));
                        cp[at] = pos0;
                        cd[at] = d0;
                    }
                    float dropped = (have && !keep) ? d0 : 3.0e38f;
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) dropped = fminf(dropped, __shfl_xor(dropped, off));
                    next_d2 = fminf(next_d2, dropped);
                    n_in = (int)__builtin_popcountll(km);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    return;
                }
                uint32_t e_pos[RF_CAP / WAVE];
                float e_d[RF_CAP / WAVE];
                int rank[RF_CAP / WAVE];
#pragma unroll
                for (int u = 0; u < RF_CAP / WAVE; u++) {
                    const int i = u * WAVE + lane;
                    e_pos[u] = i < n_in ? cp[i] : 0u;
                    e_d[u] = i < n_in ? cd[i] : 3.0e38f;
                    rank[u] = 0;
                }
                for (int j = 0; j < n_in; j++) {
                    const float dj = cd[j];                           // (wave-uniform address)
#pragma unroll
                    for (int u = 0; u < RF_CAP / WAVE; u++)
                        rank[u] += (dj < e_d[u] || (dj == e_d[u] && j < u * WAVE + lane)) ? 1 : 0;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                int kept = 0;
#pragma unroll
                for (int u = 0; u < RF_CAP / WAVE; u++) {
                    const bool have = u * WAVE + lane < n_in;
                    const bool keep = have && rank[u] < KEEP;
                    const unsigned long long km = __ballot(keep);
                    if (keep) {
                        const int at = kept + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0u));
                        cp[at] = e_pos[u];
                        cd[at] = e_d[u];
                    }
                    kept += (int)__builtin_popcountll(km);
                    float dropped = (have && !keep) ? e_d[u] : 3.0e38f;
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) dropped = fminf(dropped, __shfl_xor(dropped, off));
                    next_d2 = fminf(next_d2, dropped);
                }
                n_in = kept;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            };
            float tau_cur = ball2;                                    // candidates above it are not collected
            for (int r0 = 0; r0 < n_rows; r0 += WAVE) {
                // lane = one (gz, gy) row of the box: the cells of a grid row are adjacent keys = one range of grid positions
                uint32_t rs = 0u, re = 0u;
                if (r0 + lane < n_rows) {
                    const int gz = cz0 + (r0 + lane) / ny_b, gy = cy0 + (r0 + lane) % ny_b;
                    const int base = (gz * GRID_MAX + gy) * GRID_MAX;
                    bool any = false;
                    for (int gx = cx0; gx <= cx1; gx++) {
                        const uint2 rg = cell_range[base + gx];
                        if (rg.y > rg.x) {
                            if (!any) rs = rg.x;
                            re = rg.y;
                            any = true;
                        }
                    }
                }
                // the rows' candidates as one list (ascending grid position): lanes = candidates, 64 at a time
                uint32_t incl = re - rs;
#pragma unroll
                for (int d = 1; d < WAVE; d <<= 1) {
                    const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
                    if (lane >= d) incl += up;
                }
                const int n_cand = (int)__builtin_amdgcn_readlane((int)incl, WAVE - 1);
#ifdef SOAR_KNN_SEARCH_LOG
                l_cand += n_cand; l_rows = n_rows;
#endif
                row_first[wave][lane] = rs;
                row_end[wave][lane] = incl;                           // candidates of the rows 0 .. lane
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const int rows_here = min(WAVE, n_rows - r0);
                constexpr int TRIPS = 4;                              // candidate gathers in flight together
                for (int g00 = 0; g00 < n_cand; g00 += TRIPS * WAVE) {
                uint32_t tt[TRIPS];
                float4 vv[TRIPS];
#pragma unroll
                for (int u = 0; u < TRIPS; u++) {
                    const int g = g00 + u * WAVE + lane;
                    int row = 0;
                    while (row + 1 < rows_here && row_end[wave][row] <= (uint32_t)g) row++;
                    tt[u] = row_first[wave][row] + ((uint32_t)g - (row ? row_end[wave][row - 1] : 0u));
                    vv[u] = sorted_verts[g < n_cand ? tt[u] : 0u];
                }
#pragma unroll
                for (int u = 0; u < TRIPS; u++) {
                    const int g0 = g00 + u * WAVE;
                    if (g0 >= n_cand) break;
                    const int g = g0 + lane;
                    const uint32_t t = tt[u];
                    const bool have = g < n_cand;
                    const float d = have ? dist2_exact(x, y, z, vv[u]) : 3.0e38f;
                    const bool in = d <= tau_cur;
                    // outside the ball but inside the box: a lower bound of the distances behind the set all the same
                    float out_d = (have && !in) ? d : 3.0e38f;
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) out_d = fminf(out_d, __shfl_xor(out_d, off));
                    next_d2 = fminf(next_d2, out_d);
                    const unsigned long long im = __ballot(in);
                    if (im == 0ull) continue;
                    if (n_in + (int)__builtin_popcountll(im) > RF_CAP) {
                        select_keep();                                // KEEP stay (KEEP + 64 <= RF_CAP): nothing above their largest
                        float worst = lane < KEEP ? cd[lane] : 0.f;   // distance can be among the KEEP best any more
#pragma unroll
                        for (int off = 32; off > 0; off >>= 1) worst = fmaxf(worst, __shfl_xor(worst, off));
                        tau_cur = worst;
                    }
                    const bool in2 = d <= tau_cur;
                    const unsigned long long im2 = __ballot(in2);
                    if (in2) {
                        const int at = n_in + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(im2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)im2, 0u));
                        cp[at] = t;
                        cd[at] = d;
                    }
                    float out2 = (in && !in2) ? d : 3.0e38f;
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) out2 = fminf(out2, __shfl_xor(out2, off));
                    next_d2 = fminf(next_d2, out2);
                    n_in += (int)__builtin_popcountll(im2);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // row_first / row_end are rewritten by the next 64 rows
                __builtin_amdgcn_wave_barrier();
            }
#ifdef SOAR_KNN_SEARCH_LOG
            lt1 = wall_clock64(); l_nin = n_in;
#endif
            if (n_in >= KEEP) { select_keep(); break; }               // n_in == KEEP now
        }
        if (!sane) {                                                  // (wave-uniform)
            const float nan = __uint_as_float(0x7FC00000u);
            if (lane < J) weights_out[(size_t)p * J + lane] = nan;
            if (lane == 0) {
                st.ref30[q] = make_float4(x, y, z, -1.f);
                st.ref32[q] = make_float4(x, y, z, -1.f);
            }
            continue;
        }
#ifdef SOAR_KNN_SEARCH_LOG
        lt2 = wall_clock64();
#endif
        const uint32_t pos = lane < KEEP ? cp[lane] : 0u;
        const float d2 = lane < KEEP ? cd[lane] : 3.0e38f;
        float far = lane < KEEP ? d2 : 0.f;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) far = fmaxf(far, __shfl_xor(far, off));
        // every vertex that is not among the KEEP is at least this far: the best dropped candidate, or the ball's radius
        const float h32 = half_gap(far, fminf(next_d2, ball2));
        float h30;
        const uint32_t mask = rank_32(lane < KEEP ? d2 : 3.0e38f, lane, h30);
        if (lane < KEEP) st.nbr[q * KNN_STATE_STRIDE + lane] = pos;
        if (lane == 0) {
            st.ref30[q] = make_float4(x, y, z, h30);
            st.ref32[q] = make_float4(x, y, z, h32);
            st.in30[q] = mask;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#ifdef SOAR_KNN_SEARCH_LOG
        const unsigned long long lt3 = wall_clock64();
#endif
        // (as one slot of the four-query blend: its row loads travel ten at a time -- the one-query chain of 30 (LDS read, row load,
        // multiply-add) steps came out of the compiler as 30 round trips one after the other here, 11 of a search's 26 us)
        slots_fill_pair(slots[wave], 0, lane < KNN_KEEP ? p : -1, mask, pos, d2, lane);
        if (lane == 0) { slots[wave].p[2] = -1; slots[wave].p[3] = -1; }
        slots_blend<SOAR_KNN_SEARCH_UNROLL>(slots[wave], lane, rows_padded, J, weights_out);
#ifdef SOAR_KNN_SEARCH_LOG
        if (lane == 0) {
            const unsigned long long lt4 = wall_clock64();
            float *lg = st.d2 + q * KNN_STATE_STRIDE;
            auto put = [&](int k, float v) { lg[k] = -(1.f + v); };
            put(0, (float)(lt4 - lt0)); put(1, (float)(lt1 - lt0)); put(2, (float)(lt2 - lt1)); put(3, (float)(lt3 - lt2)); put(4, (float)(lt4 - lt3));
            put(5, (float)growths); put(6, (float)l_sel); put(7, (float)l_cand); put(8, (float)l_rows); put(9, (float)l_nin);
        }
#endif
    }
    // The last of a list's searchers to get here empties the list for the next refresh (a memset launch in front of the certificate
    // kernel was 4.7 us of a 64 us refresh) and adds its length to the statistics: every searcher has read the length by then.
    if (lane == 0) {
        uint32_t *line = st.work + my_list * KNN_WORK_LINE;
        if (atomicAdd(line + 1, 1u) == (uint32_t)search_blocks * KNN_WAVES / KNN_WORK_LISTS - 1u) {
            const uint32_t n = line[0];
            if (counters && n) atomicAdd(counters, n);
            line[0] = 0u;
            line[1] = 0u;
        }
    }
}

}  // namespace

}  // namespace soar

using namespace soar;

namespace {

// the vertex grid: everything that depends on the (static) canonical vertices and their skinning rows only
struct KnnGrid {
    GridMeta *meta;
    uint2 *cell_range;       // [GRID_CELLS]
    float4 *sorted_verts;    // [V] {x, y, z, vertex id}
    float *rows;             // [V][KNN_JMAX] skinning rows in sorted order, zero padded
    uint32_t *k0, *k1, *v0, *v1;   // build temporaries
    void *sort_tmp;
    size_t sort_bytes;
    size_t total;
};

int carve_knn_grid(void *base, int32_t V, KnnGrid *g, hipStream_t stream)
{
    size_t sort_bytes = 0;
    (void)rocprim::radix_sort_pairs((void *)nullptr, sort_bytes, (uint32_t *)nullptr, (uint32_t *)nullptr,
                                    (uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)V, 0u, 18u, stream);     // size query only
    char *b = static_cast<char *>(base);
    size_t off = 0;
    auto carve = [&](size_t n) { size_t o = off; off = align_up(off + n); return b + o; };
    g->meta = reinterpret_cast<GridMeta *>(carve(sizeof(GridMeta)));
    g->cell_range = reinterpret_cast<uint2 *>(carve(sizeof(uint2) * (size_t)GRID_CELLS));
    g->sorted_verts = reinterpret_cast<float4 *>(carve(sizeof(float4) * (size_t)V));
    g->rows = reinterpret_cast<float *>(carve(sizeof(float) * (size_t)V * KNN_JMAX));
    g->k0 = reinterpret_cast<uint32_t *>(carve(4 * (size_t)V));
    g->k1 = reinterpret_cast<uint32_t *>(carve(4 * (size_t)V));
    g->v0 = reinterpret_cast<uint32_t *>(carve(4 * (size_t)V));
    g->v1 = reinterpret_cast<uint32_t *>(carve(4 * (size_t)V));
    g->sort_tmp = carve(sort_bytes);
    g->sort_bytes = sort_bytes;
    g->total = off + ALIGN;
    return 0;
}

int knn_build(const float *verts, int32_t V, const float *vert_weights, int32_t J, const KnnGrid &g, hipStream_t stream)
{
    hipLaunchKernelGGL(grid_meta_kernel, dim3(1), dim3(256), 0, stream, verts, V, g.meta);
    hipLaunchKernelGGL(grid_cells_kernel, dim3((V + 255) / 256), dim3(256), 0, stream, verts, V, g.meta, g.k0, g.v0);
    size_t sort_bytes = g.sort_bytes;
    SOAR_HIP_OK(rocprim::radix_sort_pairs(g.sort_tmp, sort_bytes, g.k0, g.k1, g.v0, g.v1, (size_t)V, 0u, 18u, stream));
    SOAR_HIP_OK(hipMemsetAsync(g.cell_range, 0, sizeof(uint2) * (size_t)GRID_CELLS, stream));
    hipLaunchKernelGGL(grid_ranges_kernel, dim3((V + 255) / 256), dim3(256), 0, stream, verts, V, g.k1, g.v1, g.cell_range,
                       g.sorted_verts);
    if (J <= KNN_JMAX)
        hipLaunchKernelGGL(pad_rows_kernel, dim3((V * KNN_JMAX + 255) / 256), dim3(256), 0, stream, vert_weights, V, J,
                           g.sorted_verts, g.rows);
    SOAR_LAUNCH_OK("lbs_knn_build_grid", stream, 0);
    return 0;
}

// order / resort: optional caller-owned query order [P].  resort != 0 (or order == NULL): the queries are sorted by cell
// and the order is stored; resort == 0: the stored order is reused and only the cell keys are recomputed.
struct QueryWs {
    size_t sort_bytes, total;
    uint32_t *qk0, *qk1, *qv0, *qv1;
    uint32_t *item_cost, *item_order; // [KNN_SLOTS * ceil(P / 64) (+ 1)] cost and heaviest-first order of the (chunk, slot) work items
                                      // (kept between re-sorts)
};
// layout of the caller-owned query workspace: [rocPRIM sort temp | qk0 | qk1 | qv0 | qv1 | item costs | item order], each 256-byte aligned
int carve_query_ws(void *base, int32_t P, QueryWs *out, hipStream_t stream)
{
    size_t qsort_bytes = 0;
    // size query only (the return code is ignored like in scan_temp_bytes / sort_temp_bytes: it must also work on a box
    // without a GPU, where the sizing entry points are part of the CPU test suite)
    (void)rocprim::radix_sort_pairs((void *)nullptr, qsort_bytes, (uint32_t *)nullptr, (uint32_t *)nullptr,
                                    (uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)(P > 0 ? P : 1), 0u, 18u, stream);
    size_t off = 0;
    auto carve = [&](size_t n) { size_t o = off; off = align_up(off + n); return o; };
    const size_t o_tmp = carve(qsort_bytes), o_qk0 = carve(4 * (size_t)P), o_qk1 = carve(4 * (size_t)P),
                 o_qv0 = carve(4 * (size_t)P), o_qv1 = carve(4 * (size_t)P), o_cost = carve(4 * KNN_SLOTS * (((size_t)P + WAVE - 1) / WAVE)),
                 o_ord = carve(4 * (KNN_SLOTS * (((size_t)P + WAVE - 1) / WAVE) + 1));
    (void)o_tmp;
    char *b = static_cast<char *>(base);
    out->sort_bytes = qsort_bytes;
    out->total = off;
    out->qk0 = reinterpret_cast<uint32_t *>(b + o_qk0); out->qk1 = reinterpret_cast<uint32_t *>(b + o_qk1);
    out->qv0 = reinterpret_cast<uint32_t *>(b + o_qv0); out->qv1 = reinterpret_cast<uint32_t *>(b + o_qv1);
    out->item_cost = reinterpret_cast<uint32_t *>(b + o_cost); out->item_order = reinterpret_cast<uint32_t *>(b + o_ord);
    return 0;
}

// neighbour state of soar_lbs_knn_refresh: [P][32] grid positions | [P] ref30 | [P] ref32 | [P] in30 | work lists | [P][32] distances
struct KnnState { KnnStatePtrs p; size_t total; };
int carve_knn_state(void *base, int32_t P, KnnState *out)
{
    char *b = static_cast<char *>(base);
    const size_t n = (size_t)(P > 0 ? P : 1);
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes); return b + o; };
    out->p.nbr = reinterpret_cast<uint32_t *>(carve(sizeof(uint32_t) * KNN_STATE_STRIDE * n));
    out->p.ref30 = reinterpret_cast<float4 *>(carve(sizeof(float4) * n));
    out->p.ref32 = reinterpret_cast<float4 *>(carve(sizeof(float4) * n));
    out->p.in30 = reinterpret_cast<uint32_t *>(carve(sizeof(uint32_t) * n));
    out->p.work = reinterpret_cast<uint32_t *>(carve(sizeof(uint32_t) * knn_work_words(P)));
    out->p.d2 = reinterpret_cast<float *>(carve(sizeof(float) * KNN_STATE_STRIDE * n));
    out->total = off;
    return 0;
}

int knn_query(const KnnGrid &g, int32_t V, const float *vert_weights, int32_t J, const float *xyz, int32_t P, int32_t K,
              float *weights_out, int32_t *knn_idx_out, uint32_t *order, int resort, void *query_ws, size_t query_ws_bytes,
              hipStream_t stream, const KnnState *state = nullptr)
{
    const bool fast = (K == KNN_K) && (J <= KNN_JMAX);
    KnnStatePtrs st_none = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const KnnStatePtrs st = state ? state->p : st_none;
    if (!fast && state) { set_error("soar_lbs_knn: the neighbour state needs K = %d and J <= %d", KNN_K, KNN_JMAX); return 1; }
    if (!fast) {
        hipLaunchKernelGGL(knn_grid_kernel, dim3((P + KNN_THREADS - 1) / KNN_THREADS), dim3(KNN_THREADS), 0, stream, xyz, P, V,
                           g.meta, g.cell_range, g.sorted_verts, vert_weights, J, K, weights_out, knn_idx_out);
        SOAR_LAUNCH_OK("lbs_knn_weights", stream, 0);
        return 0;
    }
    // query-side workspace (keys / ids of the cell sort): caller-owned, carved here
    QueryWs ws;
    if (carve_query_ws(query_ws, P, &ws, stream)) return 1;
    if (ws.total > query_ws_bytes) {
        set_error("soar_lbs_knn_query: query workspace too small (%zu bytes, need %zu for P=%d)", query_ws_bytes, ws.total, P);
        return 1;
    }
    size_t qsort_bytes = ws.sort_bytes;
    char *b = static_cast<char *>(query_ws);
    const size_t o_tmp = 0;
    uint32_t *qk0 = ws.qk0, *qk1 = ws.qk1, *qv0 = ws.qv0, *qv1 = ws.qv1;
    if (order && !resort) {
        hipLaunchKernelGGL(query_cells_ordered_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, xyz, P, g.meta, order, qk1);
        qv1 = order;
    } else {
        hipLaunchKernelGGL(query_cells_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, xyz, P, g.meta, qk0, qv0);
        SOAR_HIP_OK(rocprim::radix_sort_pairs(b + o_tmp, qsort_bytes, qk0, qk1, qv0, order ? order : qv1, (size_t)P, 0u, 18u,
                                              stream));
        if (order) qv1 = order;
        // the item order is rebuilt with every sort and reused by the calls that keep the stored query order
        const int nchunks = (P + WAVE - 1) / WAVE;
        hipLaunchKernelGGL(item_cost_kernel, dim3(nchunks), dim3(WAVE), 0, stream, qk1, P, g.meta, g.cell_range, ws.item_cost);
        hipLaunchKernelGGL(item_order_kernel, dim3(1), dim3(1024), 0, stream, ws.item_cost, nchunks * KNN_SLOTS, ws.item_order);
    }
    const dim3 grid(((P + WAVE - 1) / WAVE) * KNN_SLOTS);
    const uint32_t *chunk_order = getenv("SOAR_KNN_NO_ORDER") ? nullptr : ws.item_order;         // development switch
    const char *log_path = getenv("SOAR_KNN_LOG");            // diagnostic: per-wave timeline of one launch
    if (log_path && !knn_idx_out) {
        unsigned long long *log_dev = nullptr;
        const size_t nbytes = sizeof(unsigned long long) * 8 * (size_t)grid.x * KNN_WAVES;
        SOAR_HIP_OK(hipMalloc(&log_dev, nbytes));
        SOAR_HIP_OK(hipMemsetAsync(log_dev, 0, nbytes, stream));
        hipLaunchKernelGGL((knn_cell_kernel<false, true>), grid, dim3(KNN_WAVES * WAVE), 0, stream, xyz, P, V, g.meta, g.cell_range,
                           g.sorted_verts, qk1, qv1, g.rows, J, weights_out, knn_idx_out, chunk_order, st, log_dev);
        SOAR_HIP_OK(hipStreamSynchronize(stream));
        unsigned long long *host = (unsigned long long *)malloc(nbytes);
        SOAR_HIP_OK(hipMemcpy(host, log_dev, nbytes, hipMemcpyDeviceToHost));
        FILE *f = fopen(log_path, "wb");
        if (f) { fwrite(host, 1, nbytes, f); fclose(f); }
        free(host);
        (void)hipFree(log_dev);
        return 0;
    }
    if (knn_idx_out)
        hipLaunchKernelGGL(knn_cell_kernel<true>, grid, dim3(KNN_WAVES * WAVE), 0, stream, xyz, P, V, g.meta, g.cell_range,
                           g.sorted_verts, qk1, qv1, g.rows, J, weights_out, knn_idx_out, chunk_order, st);
    else
        hipLaunchKernelGGL(knn_cell_kernel<false>, grid, dim3(KNN_WAVES * WAVE), 0, stream, xyz, P, V, g.meta, g.cell_range,
                           g.sorted_verts, qk1, qv1, g.rows, J, weights_out, knn_idx_out, chunk_order, st);
    SOAR_LAUNCH_OK("lbs_knn_weights", stream, 0);
    return 0;
}

int check_knn_sizes(int32_t P, int32_t V, int32_t J, int32_t K)
{
    if (P < 0 || V <= 0 || J <= 0 || K <= 0) { set_error("soar_lbs_knn: bad sizes P=%d V=%d J=%d K=%d", P, V, J, K); return 1; }
    if (K > KNN_MAXK || K > V) { set_error("soar_lbs_knn: K=%d unsupported (max %d, V=%d)", K, KNN_MAXK, V); return 1; }
    return 0;
}

}  // namespace

extern "C" int soar_lbs_knn_grid_bytes(int32_t V, size_t *bytes)
{
    if (V <= 0 || !bytes) { set_error("soar_lbs_knn_grid_bytes: bad arguments"); return 1; }
    KnnGrid g;
    if (carve_knn_grid(nullptr, V, &g, nullptr)) return 1;
    *bytes = g.total;
    return 0;
}

extern "C" int soar_lbs_knn_build_grid(const float *verts, int32_t V, const float *vert_weights, int32_t J, void *grid_buffer,
                                       void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (V <= 0 || J <= 0 || !verts || !vert_weights || !grid_buffer) { set_error("soar_lbs_knn_build_grid: bad arguments"); return 1; }
    KnnGrid g;
    if (carve_knn_grid(grid_buffer, V, &g, stream)) return 1;
    return knn_build(verts, V, vert_weights, J, g, stream);
}

extern "C" int soar_lbs_knn_query_bytes(int32_t P, size_t *bytes)
{
    if (P < 0 || !bytes) { set_error("soar_lbs_knn_query_bytes: bad arguments"); return 1; }
    QueryWs ws;
    if (carve_query_ws(nullptr, P, &ws, nullptr)) return 1;
    *bytes = ws.total;
    return 0;
}

static int check_ws(const char *who, const void *ws, size_t have)
{
    if (!ws) { set_error("%s: NULL workspace (caller-owned, see the *_bytes query)", who); return 1; }
    if (reinterpret_cast<uintptr_t>(ws) % ALIGN) { set_error("%s: workspace is not %zu-byte aligned", who, ALIGN); return 1; }
    (void)have;
    return 0;
}

extern "C" int soar_lbs_knn_query(const void *grid_buffer, int32_t V, const float *vert_weights, int32_t J, const float *xyz,
                                  int32_t P, int32_t K, float *weights_out, int32_t *knn_idx_out, void *query_workspace,
                                  size_t query_workspace_bytes, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (check_knn_sizes(P, V, J, K)) return 1;
    if (P == 0) return 0;
    if (!grid_buffer || !xyz || !vert_weights || !weights_out) { set_error("soar_lbs_knn_query: NULL pointer"); return 1; }
    if (check_ws("soar_lbs_knn_query", query_workspace, query_workspace_bytes)) return 1;
    KnnGrid g;
    if (carve_knn_grid(const_cast<void *>(grid_buffer), V, &g, stream)) return 1;
    StageTimer timer(ST_LBS_KNN, stream);
    return knn_query(g, V, vert_weights, J, xyz, P, K, weights_out, knn_idx_out, nullptr, 1, query_workspace, query_workspace_bytes,
                     stream);
}

extern "C" int soar_lbs_knn_query_ordered(const void *grid_buffer, int32_t V, const float *vert_weights, int32_t J, const float *xyz,
                                          int32_t P, int32_t K, uint32_t *order, int32_t resort, float *weights_out,
                                          int32_t *knn_idx_out, void *query_workspace, size_t query_workspace_bytes, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (check_knn_sizes(P, V, J, K)) return 1;
    if (P == 0) return 0;
    if (!grid_buffer || !xyz || !vert_weights || !weights_out || !order) { set_error("soar_lbs_knn_query_ordered: NULL pointer"); return 1; }
    if (check_ws("soar_lbs_knn_query_ordered", query_workspace, query_workspace_bytes)) return 1;
    KnnGrid g;
    if (carve_knn_grid(const_cast<void *>(grid_buffer), V, &g, stream)) return 1;
    StageTimer timer(ST_LBS_KNN, stream);
    return knn_query(g, V, vert_weights, J, xyz, P, K, weights_out, knn_idx_out, order, resort, query_workspace,
                     query_workspace_bytes, stream);
}

extern "C" int soar_lbs_knn_state_bytes(int32_t P, size_t *bytes)
{
    if (P < 0 || !bytes) { set_error("soar_lbs_knn_state_bytes: bad arguments"); return 1; }
    KnnState st;
    carve_knn_state(nullptr, P, &st);
    *bytes = st.total;
    return 0;
}

extern "C" int soar_lbs_knn_query_state(const void *grid_buffer, int32_t V, const float *vert_weights, int32_t J, const float *xyz,
                                        int32_t P, uint32_t *order, int32_t resort, float *weights_out, void *state_buffer,
                                        void *query_workspace, size_t query_workspace_bytes, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (check_knn_sizes(P, V, J, KNN_K)) return 1;
    if (V < KNN_STATE_STRIDE) { set_error("soar_lbs_knn_query_state: the neighbour state keeps %d vertices per query, V=%d", KNN_STATE_STRIDE, V); return 1; }
    if (P == 0) return 0;
    if (!grid_buffer || !xyz || !vert_weights || !weights_out || !order || !state_buffer) { set_error("soar_lbs_knn_query_state: NULL pointer"); return 1; }
    if (check_ws("soar_lbs_knn_query_state", query_workspace, query_workspace_bytes) || check_ws("soar_lbs_knn_query_state", state_buffer, 0)) return 1;
    KnnGrid g;
    if (carve_knn_grid(const_cast<void *>(grid_buffer), V, &g, stream)) return 1;
    if (V <= KNN_KEEP) { set_error("soar_lbs_knn_query_state: the neighbour state keeps %d vertices per query, V = %d", KNN_KEEP, V); return 1; }
    KnnState st;
    carve_knn_state(state_buffer, P, &st);
    StageTimer timer(ST_LBS_KNN, stream);
    SOAR_HIP_OK(hipMemsetAsync(st.p.work, 0, KNN_WORK_LISTS * KNN_WORK_LINE * sizeof(uint32_t), stream));
    return knn_query(g, V, vert_weights, J, xyz, P, KNN_K, weights_out, nullptr, order, resort, query_workspace, query_workspace_bytes,
                     stream, &st);
}

extern "C" int soar_lbs_knn_refresh(const void *grid_buffer, int32_t V, int32_t J, const float *xyz, int32_t P, const uint32_t *order,
                                    void *state_buffer, float *weights_out, uint32_t *searched_counter_dev, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (check_knn_sizes(P, V, J, KNN_K)) return 1;
    if (J > KNN_JMAX) { set_error("soar_lbs_knn_refresh: J <= %d", KNN_JMAX); return 1; }
    if (V < KNN_STATE_STRIDE) { set_error("soar_lbs_knn_refresh: the neighbour state keeps %d vertices per query, V=%d", KNN_STATE_STRIDE, V); return 1; }
    if (P == 0) return 0;
    if (!grid_buffer || !xyz || !weights_out || !state_buffer || !order) { set_error("soar_lbs_knn_refresh: NULL pointer"); return 1; }
    if (check_ws("soar_lbs_knn_refresh", state_buffer, 0)) return 1;
    KnnGrid g;
    if (carve_knn_grid(const_cast<void *>(grid_buffer), V, &g, stream)) return 1;
    KnnState st;
    carve_knn_state(state_buffer, P, &st);
    StageTimer timer(ST_LBS_KNN, stream);
    const int per_block = KNN_WAVES * 2 * KNN_CERT_PAIRS, nblocks = (P + per_block - 1) / per_block;
    // (the work lists are empty: soar_lbs_knn_query_state cleared them, every refresh leaves them empty again)
    static_assert(SOAR_KNN_SEARCH_BLOCKS * KNN_WAVES % KNN_WORK_LISTS == 0, "searchers per list");
    hipLaunchKernelGGL(knn_certify_kernel, dim3(nblocks), dim3(KNN_WAVES * WAVE), 0, stream, xyz, P, g.sorted_verts, order, st.p);
    const int search_blocks = SOAR_KNN_SEARCH_BLOCKS, blend_blocks = (P + KNN_WAVES * KNN_FSLOTS - 1) / (KNN_WAVES * KNN_FSLOTS);
#ifdef SOAR_KNN_SPLIT_LAUNCH      // diagnostic build: the two halves of the second launch one after the other (what each takes alone)
    hipLaunchKernelGGL(knn_blend_search_kernel, dim3(blend_blocks), dim3(KNN_WAVES * WAVE), 0, stream, xyz, P, 0,
                       g.meta, g.cell_range, g.sorted_verts, g.rows, J, order, st.p, weights_out, searched_counter_dev);
    hipLaunchKernelGGL(knn_blend_search_kernel, dim3(search_blocks), dim3(KNN_WAVES * WAVE), 0, stream, xyz, P, search_blocks,
                       g.meta, g.cell_range, g.sorted_verts, g.rows, J, order, st.p, weights_out, searched_counter_dev);
#else
    hipLaunchKernelGGL(knn_blend_search_kernel, dim3(search_blocks + blend_blocks), dim3(KNN_WAVES * WAVE), 0, stream, xyz, P, search_blocks,
                       g.meta, g.cell_range, g.sorted_verts, g.rows, J, order, st.p, weights_out, searched_counter_dev);
#endif
    SOAR_LAUNCH_OK("lbs_knn_refresh", stream, 0);
    return 0;
}

extern "C" int soar_lbs_knn_weights_bytes(int32_t P, int32_t V, size_t *bytes)
{
    if (P < 0 || V <= 0 || !bytes) { set_error("soar_lbs_knn_weights_bytes: bad arguments"); return 1; }
    KnnGrid g;
    if (carve_knn_grid(nullptr, V, &g, nullptr)) return 1;
    QueryWs ws;
    if (carve_query_ws(nullptr, P, &ws, nullptr)) return 1;
    *bytes = align_up(g.total) + ws.total;
    return 0;
}

extern "C" int soar_lbs_knn_weights(const float *xyz, int32_t P, const float *verts, int32_t V, const float *vert_weights,
                                    int32_t J, int32_t K, float *weights_out, int32_t *knn_idx_out, void *workspace,
                                    size_t workspace_bytes, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (check_knn_sizes(P, V, J, K)) return 1;
    if (P == 0) return 0;
    if (!xyz || !verts || !vert_weights || !weights_out) { set_error("soar_lbs_knn_weights: NULL pointer"); return 1; }
    if (check_ws("soar_lbs_knn_weights", workspace, workspace_bytes)) return 1;
    KnnGrid g;
    if (carve_knn_grid(workspace, V, &g, stream)) return 1;
    QueryWs ws;
    if (carve_query_ws(nullptr, P, &ws, stream)) return 1;
    const size_t grid_part = align_up(g.total);
    if (workspace_bytes < grid_part + ws.total) {
        set_error("soar_lbs_knn_weights: workspace too small (%zu bytes, need %zu)", workspace_bytes, grid_part + ws.total);
        return 1;
    }
    StageTimer timer(ST_LBS_KNN, stream);
    if (knn_build(verts, V, vert_weights, J, g, stream)) return 1;
    return knn_query(g, V, vert_weights, J, xyz, P, K, weights_out, knn_idx_out, nullptr, 1,
                     static_cast<char *>(workspace) + grid_part, workspace_bytes - grid_part, stream);
}
