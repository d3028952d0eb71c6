// rast_binning.hip -- tile binning for gfx950: inclusive scan of tile counts, (tile|depth) key emission,
// stable radix sort restricted to bits [0, 32+bit), per-tile ranges.
//
// Replaces cub::DeviceScan::InclusiveSum (DGR/cuda_rasterizer/rasterizer_impl.cu:242-245),
// duplicateWithKeys (:66-99), cub::DeviceRadixSort::SortPairs[Descending] (:266-285),
// cudaMemset + identifyTileRanges (:104-124, :287-295).
// All outputs are integers and bit-exact with the reference semantics (stable sort: equal keys keep
// emission order, i.e. ascending Gaussian index).
#include "soar_common.h"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace soar {

uint32_t higher_msb(uint32_t n)
{
    uint32_t msb = sizeof(n) * 4;
    uint32_t step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step;
        else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

size_t scan_temp_bytes(int32_t P)
{
    size_t bytes = 0;
    (void)rocprim::inclusive_scan((void *)nullptr, bytes, (uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)(P > 0 ? P : 1),
                                  rocprim::plus<uint32_t>(), (hipStream_t)0);
    return bytes;
}

size_t sort_temp_bytes(int64_t R)
{
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs((void *)nullptr, bytes, (uint64_t *)nullptr, (uint64_t *)nullptr, (uint32_t *)nullptr,
                                    (uint32_t *)nullptr, (size_t)(R > 0 ? R : 1), 0u, 64u, (hipStream_t)0);
    size_t bytes_desc = 0;
    (void)rocprim::radix_sort_pairs_desc((void *)nullptr, bytes_desc, (uint64_t *)nullptr, (uint64_t *)nullptr,
                                         (uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)(R > 0 ? R : 1), 0u, 64u,
                                         (hipStream_t)0);
    return bytes > bytes_desc ? bytes : bytes_desc;
}

int launch_scan(const SoarRastParams &prm, GeomBuf &g, hipStream_t stream)
{
    StageTimer timer(ST_SCAN, stream);
    size_t bytes = g.scan_temp_bytes;
    SOAR_HIP_OK(rocprim::inclusive_scan(g.scan_temp, bytes, g.tiles_touched, g.point_offsets, (size_t)prm.P,
                                        rocprim::plus<uint32_t>(), stream));
    SOAR_LAUNCH_OK("inclusive_scan", stream, prm.debug);
    return 0;
}

namespace {

// One thread per Gaussian walks its tile rectangle row-major (y outer, x inner) and emits
// key = (tile << 32) | bits(view depth), value = Gaussian index.
__global__ void __launch_bounds__(256)
emit_keys_kernel(int P, const uint2 *__restrict__ rect, const uint32_t *__restrict__ depth_key,
                 const uint32_t *__restrict__ offsets, uint64_t *__restrict__ keys, uint32_t *__restrict__ vals, int gx)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P) return;
    const uint2 rc = rect[idx];                       // the tile rectangle preprocess computed (getRect, auxiliary.h:53-63)
    const int x0 = (int)(rc.x & 0xFFFFu), x1 = (int)(rc.x >> 16), y0 = (int)(rc.y & 0xFFFFu), y1 = (int)(rc.y >> 16);
    if (x1 <= x0 || y1 <= y0) return;
    uint32_t off = (idx == 0) ? 0u : offsets[idx - 1];
    const uint64_t depth_bits = (uint64_t)depth_key[idx];
    for (int y = y0; y < y1; y++) {
        for (int x = x0; x < x1; x++) {
            uint64_t key = (uint64_t)(uint32_t)(y * gx + x);
            key <<= 32;
            key |= depth_bits;
            keys[off] = key;
            vals[off] = (uint32_t)idx;
            off++;
        }
    }
}

__global__ void __launch_bounds__(256)
tile_ranges_kernel(int64_t L, const uint64_t *__restrict__ keys, uint2 *__restrict__ ranges, uint32_t *__restrict__ tile_xy, uint32_t gx)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= L) return;
    const uint32_t cur = (uint32_t)(keys[idx] >> 32);
    tile_xy[idx] = ((cur / gx) << 16) | (cur % gx);          // the tile of every list position (block masks)
    if (idx == 0) {
        ranges[cur].x = 0;
    } else {
        const uint32_t prev = (uint32_t)(keys[idx - 1] >> 32);
        if (cur != prev) {
            ranges[prev].y = (uint32_t)idx;
            ranges[cur].x = (uint32_t)idx;
        }
    }
    if (idx == L - 1) ranges[cur].y = (uint32_t)L;
}

__global__ void __launch_bounds__(1024) tile_order_kernel(int T, int Tpad, const uint2 *__restrict__ ranges, uint32_t *__restrict__ order,
                                                          uint4 *__restrict__ order_rec, const float *__restrict__ bg, int normalize_depth,
                                                          uint32_t *__restrict__ bg_state)
{
    tile_order_block(T, Tpad, nullptr, ranges, order, order_rec, bg, normalize_depth, bg_state);
}

}  // namespace

int launch_tile_order(const SoarRastParams &prm, ImageBuf &img, hipStream_t stream)
{
    const int T = ((prm.W + TILE - 1) / TILE) * ((prm.H + TILE - 1) / TILE);
    const int Tpad = (T + 7) / 8 * 8;
    hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, stream, T, Tpad, img.ranges, img.tile_order, img.order_rec,
                       prm.bg_dev, prm.cfg_normalize_depth, img.bg_state);
    SOAR_LAUNCH_OK("tile_order", stream, prm.debug);
    return 0;
}

int launch_binning(const SoarRastParams &prm, GeomBuf &g, BinBuf &b, ImageBuf &img, int64_t R,
                   hipStream_t stream)
{
    const int gx = (prm.W + TILE - 1) / TILE, gy = (prm.H + TILE - 1) / TILE;
    SOAR_HIP_OK(hipMemsetAsync(img.ranges, 0, sizeof(uint2) * (size_t)gx * gy, stream));
    if (R <= 0) return launch_tile_order(prm, img, stream);

    {
    StageTimer timer(ST_EMIT_KEYS, stream);
    hipLaunchKernelGGL(emit_keys_kernel, dim3((prm.P + 255) / 256), dim3(256), 0, stream, prm.P, g.rect, g.depth_key,
                       g.point_offsets, b.keys_unsorted, b.vals_unsorted, gx);
    }
    SOAR_LAUNCH_OK("emit_keys", stream, prm.debug);

    const unsigned end_bit = 32u + higher_msb((uint32_t)(gx * gy));
    size_t bytes = b.sort_temp_bytes;
    {
    StageTimer timer(ST_SORT, stream);
    if (!prm.sort_descending) {
        SOAR_HIP_OK(rocprim::radix_sort_pairs(b.sort_temp, bytes, b.keys_unsorted, b.keys_sorted, b.vals_unsorted,
                                              b.vals_sorted, (size_t)R, 0u, end_bit, stream));
    } else {
        SOAR_HIP_OK(rocprim::radix_sort_pairs_desc(b.sort_temp, bytes, b.keys_unsorted, b.keys_sorted, b.vals_unsorted,
                                                   b.vals_sorted, (size_t)R, 0u, end_bit, stream));
    }
    }
    SOAR_LAUNCH_OK("radix_sort_pairs", stream, prm.debug);

    {
    StageTimer timer(ST_RANGES, stream);
    hipLaunchKernelGGL(tile_ranges_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, stream, R, b.keys_sorted,
                       img.ranges, b.tile_xy, (uint32_t)gx);
    }
    SOAR_LAUNCH_OK("tile_ranges", stream, prm.debug);
    return launch_tile_order(prm, img, stream);
}

}  // namespace soar
