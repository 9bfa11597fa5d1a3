// Tile-binned gridding: the window gridder (grid_mfma.hip) for visibility streams WITHOUT locality.
//
// The window kernel keeps a 32x32 piece of the grid in a wave's accumulators and lives on
// consecutive visibilities landing in (nearly) the same place: true for the baseline-sorted,
// adjacent-merged stream the reference's preprocessor delivers (loader_ms.py:465-467,
// preprocess.cpp:334-397), false for a time-ordered or shuffled stream, where every record
// costs a whole-window flush (8 KB of float atomics) and the kernel degenerates to the atomic rate
// of the per-tap kernel (grid.mako's situation without its bin sort).  This file restores the
// locality on the device:
//   1. key = bin of the footprint origin, bins of (window slack + 1)^2 cells in serpentine row
//      order -- all visibilities of a bin share one window position, neighbouring bins differ by a
//      partial window move;
//   2. stable radix sort of (key, index) pairs (hipCUB / rocPRIM), only the bits the key needs;
//   3. gather uv / w_plane / vis into sorted copies (one pass, 18 + 8 P bytes per visibility);
//   4. the unchanged window kernel over the sorted copies.
// Results equal the direct path's up to the order of float additions.
// kimg_grid_jumps counts the records that would force a window jump, so that callers can choose.
#include "kimg_common.h"
#include <hipcub/hipcub.hpp>

int kimg_grid_mfma(void *grid, int64_t grid_row_stride, int64_t grid_pol_stride, int grid_size,
                   int P, const float *weights_grid, int64_t wg_row_stride, int64_t wg_pol_stride,
                   const int16_t *uv, const int16_t *w_plane, const void *vis, int64_t num_vis,
                   const void *convolve_kernel, int w_planes, int oversample, int kernel_width,
                   void *workspace, size_t workspace_bytes, int arith, hipStream_t stream);
bool kimg_grid_mfma_supported(int P, int w_planes, int oversample, int kernel_width);
size_t kimg_grid_mfma_workspace_bytes(int P, int w_planes, int oversample, int kernel_width);
int kimg_degrid_mfma(const void *grid, int64_t grid_row_stride, int64_t grid_pol_stride,
                     int grid_size, int P, const int16_t *uv, const int16_t *w_plane,
                     const float *weights, void *vis, int64_t num_vis, const void *convolve_kernel,
                     int w_planes, int oversample, int kernel_width, void *workspace,
                     size_t workspace_bytes, int arith, hipStream_t stream);
size_t kimg_degrid_mfma_workspace_bytes(int P, int w_planes, int oversample, int kernel_width);
bool kimg_degrid_mfma_supported(int P, int w_planes, int oversample, int kernel_width);

namespace {

constexpr int WIN = 32;

// Window slack (cells a footprint origin may move without a flush) of the launches that grid a
// K-tap kernel: K itself up to 32 taps, (K+1)/2-tap blocks above.
__host__ __device__ inline int window_slack(int K)
{
    const int taps = K > WIN ? (K + 1) / 2 : K;
    return WIN - taps;
}

struct bin_geometry {
    int half;           // grid_size / 2: u + half is a non-negative cell index
    int bin;            // cells per bin along either axis
    int nbu;            // bins per row
    unsigned last_key;  // key of records outside the grid (sorted to the end; the kernel skips them)
};

__global__ __launch_bounds__(256) void bin_key_kernel(
    const int2 *__restrict__ uv, int64_t n, bin_geometry g, int grid_size,
    unsigned *__restrict__ keys, unsigned *__restrict__ index)
{
    const int64_t i = blockIdx.x * (int64_t) blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const int2 r = uv[i];
    const int u = (short) (r.x & 0xffff) + g.half, v = (short) (r.x >> 16) + g.half;
    unsigned key = g.last_key;
    if ((unsigned) u < (unsigned) grid_size && (unsigned) v < (unsigned) grid_size) {
        const int bv = v / g.bin;
        int bu = u / g.bin;
        if (bv & 1)
            bu = g.nbu - 1 - bu;        // serpentine: the end of a bin row is next to the start of the next
        key = (unsigned) bv * (unsigned) g.nbu + (unsigned) bu;
    }
    keys[i] = key;
    index[i] = (unsigned) i;
}

template <int P>
__global__ __launch_bounds__(256) void gather_kernel(
    const unsigned *__restrict__ index, int64_t n, const int2 *__restrict__ uv,
    const int16_t *__restrict__ w_plane, const float2 *__restrict__ vis,
    int2 *__restrict__ uv_out, int16_t *__restrict__ wp_out, float2 *__restrict__ vis_out)
{
    const int64_t i = blockIdx.x * (int64_t) blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const unsigned src = index[i];
    uv_out[i] = uv[src];
    wp_out[i] = w_plane[src];
#pragma unroll
    for (int p = 0; p < P; p++)
        vis_out[i * P + p] = vis[(int64_t) src * P + p];
}

// the degridder's inputs in tile order (it also needs the statistical weights) ...
template <int P>
__global__ __launch_bounds__(256) void gather_degrid_kernel(
    const unsigned *__restrict__ index, int64_t n, const int2 *__restrict__ uv,
    const int16_t *__restrict__ w_plane, const float *__restrict__ weights,
    const float2 *__restrict__ vis, int2 *__restrict__ uv_out, int16_t *__restrict__ wp_out,
    float *__restrict__ w_out, float2 *__restrict__ vis_out)
{
    const int64_t i = blockIdx.x * (int64_t) blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const int64_t src = index[i];
    uv_out[i] = uv[src];
    wp_out[i] = w_plane[src];
#pragma unroll
    for (int p = 0; p < P; p++) {
        w_out[i * P + p] = weights[src * P + p];
        vis_out[i * P + p] = vis[src * P + p];
    }
}

// ... and its results back in the caller's order
template <int P>
__global__ __launch_bounds__(256) void scatter_vis_kernel(
    const unsigned *__restrict__ index, int64_t n, const float2 *__restrict__ vis_sorted,
    float2 *__restrict__ vis)
{
    const int64_t i = blockIdx.x * (int64_t) blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const int64_t dst = index[i];
#pragma unroll
    for (int p = 0; p < P; p++)
        vis[dst * P + p] = vis_sorted[i * P + p];
}

__global__ __launch_bounds__(256) void jump_count_kernel(
    const int2 *__restrict__ uv, int64_t n, int slack, unsigned *__restrict__ count)
{
    unsigned local = 0;
    for (int64_t i = blockIdx.x * (int64_t) blockDim.x + threadIdx.x + 1; i < n;
         i += (int64_t) gridDim.x * blockDim.x) {
        const int2 a = uv[i - 1], b = uv[i];
        const int du = (short) (b.x & 0xffff) - (short) (a.x & 0xffff);
        const int dv = (short) (b.x >> 16) - (short) (a.x >> 16);
        local += (abs(du) > slack || abs(dv) > slack) ? 1u : 0u;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        local += __shfl_xor(local, off, WAVE);
    if ((threadIdx.x & 63) == 0 && local)
        atomicAdd(count, local);
}

struct binned_ws {
    size_t table;       // the window kernel's own scratch (padded table copy), first
    size_t keys[2], index[2], uv, wp, vis, weights, cub, cub_bytes, total;
};

size_t align256(size_t x) { return (x + 255) / 256 * 256; }

hipError_t layout(int64_t n, int P, int W, int OV, int K, binned_ws &ws, bool degrid = false)
{
    size_t off = align256(degrid ? kimg_degrid_mfma_workspace_bytes(P, W, OV, K)
                                 : kimg_grid_mfma_workspace_bytes(P, W, OV, K));
    ws.table = 0;
    for (int i = 0; i < 2; i++) {
        ws.keys[i] = off;
        off += align256((size_t) n * sizeof(unsigned));
        ws.index[i] = off;
        off += align256((size_t) n * sizeof(unsigned));
    }
    ws.uv = off;
    off += align256((size_t) n * sizeof(int2));
    ws.wp = off;
    off += align256((size_t) n * sizeof(int16_t));
    ws.vis = off;
    off += align256((size_t) n * P * sizeof(float2));
    ws.weights = off;
    if (degrid)
        off += align256((size_t) n * P * sizeof(float));
    ws.cub_bytes = 0;
    hipcub::DoubleBuffer<unsigned> k(nullptr, nullptr), v(nullptr, nullptr);
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(nullptr, ws.cub_bytes, k, v, (int) n, 0, 32,
                                                      (hipStream_t) 0);
    ws.cub = off;
    off += align256(ws.cub_bytes);
    ws.total = off;
    return e;
}

} // namespace

size_t kimg_grid_binned_workspace_bytes_impl(int64_t n, int P, int W, int OV, int K)
{
    if (n <= 0 || n >= ((int64_t) 1 << 31) || !kimg_grid_mfma_supported(P, W, OV, K))
        return 0;
    binned_ws ws;
    if (layout(n, P, W, OV, K, ws) != hipSuccess)
        return 0;
    return ws.total;
}

// keys of the footprint-origin bins, sorted; index.Current() = the caller's record of sorted place i
static int sort_by_tile(const int16_t *uv, int64_t num_vis, int grid_size, int kernel_width,
                        unsigned char *base, const binned_ws &ws,
                        hipcub::DoubleBuffer<unsigned> &keys, hipcub::DoubleBuffer<unsigned> &index,
                        hipStream_t stream)
{
    bin_geometry g;
    g.half = grid_size / 2;
    g.bin = window_slack(kernel_width) + 1;
    g.nbu = (grid_size + g.bin - 1) / g.bin;
    const int nbv = g.nbu;
    g.last_key = (unsigned) g.nbu * (unsigned) nbv;
    int bits = 1;
    while (bits < 32 && (g.last_key >> bits) != 0)
        bits++;
    bin_key_kernel<<<kimg_divup(num_vis, 256), 256, 0, stream>>>(
        reinterpret_cast<const int2 *>(uv), num_vis, g, grid_size, keys.Current(), index.Current());
    size_t cub_bytes = ws.cub_bytes;
    KIMG_HIP(hipcub::DeviceRadixSort::SortPairs(base + ws.cub, cub_bytes, keys, index, (int) num_vis,
                                                0, bits, stream));
    return kimg_launch_status();
}

int kimg_degrid_binned(const void *grid, int64_t grid_row_stride, int64_t grid_pol_stride,
                       int grid_size, int P, const int16_t *uv, const int16_t *w_plane,
                       const float *weights, void *vis, int64_t num_vis, const void *convolve_kernel,
                       int w_planes, int oversample, int kernel_width, void *workspace,
                       size_t workspace_bytes, int arith, hipStream_t stream)
{
    if (num_vis >= ((int64_t) 1 << 31))
        return KIMG_EUNSUPPORTED;
    binned_ws ws;
    hipError_t e = layout(num_vis, P, w_planes, oversample, kernel_width, ws, true);
    if (e != hipSuccess)
        return -(int) e;
    if (workspace == nullptr || workspace_bytes < ws.total)
        return KIMG_EWORKSPACE;
    unsigned char *base = static_cast<unsigned char *>(workspace);
    hipcub::DoubleBuffer<unsigned> keys(reinterpret_cast<unsigned *>(base + ws.keys[0]),
                                        reinterpret_cast<unsigned *>(base + ws.keys[1]));
    hipcub::DoubleBuffer<unsigned> index(reinterpret_cast<unsigned *>(base + ws.index[0]),
                                         reinterpret_cast<unsigned *>(base + ws.index[1]));
    int rc = sort_by_tile(uv, num_vis, grid_size, kernel_width, base, ws, keys, index, stream);
    if (rc)
        return rc;
    const int blocks = kimg_divup(num_vis, 256);
    int2 *uv_s = reinterpret_cast<int2 *>(base + ws.uv);
    int16_t *wp_s = reinterpret_cast<int16_t *>(base + ws.wp);
    float *w_s = reinterpret_cast<float *>(base + ws.weights);
    float2 *vis_s = reinterpret_cast<float2 *>(base + ws.vis);
#define GATHER(PP) gather_degrid_kernel<PP><<<blocks, 256, 0, stream>>>(index.Current(), num_vis, \
        reinterpret_cast<const int2 *>(uv), w_plane, weights, static_cast<const float2 *>(vis), uv_s, wp_s, \
        w_s, vis_s)
    switch (P) {
    case 1: GATHER(1); break;
    case 2: GATHER(2); break;
    case 3: GATHER(3); break;
    default: GATHER(4); break;
    }
#undef GATHER
    rc = kimg_launch_status();
    if (rc)
        return rc;
    rc = kimg_degrid_mfma(grid, grid_row_stride, grid_pol_stride, grid_size, P,
                          reinterpret_cast<const int16_t *>(uv_s), wp_s, w_s, vis_s, num_vis,
                          convolve_kernel, w_planes, oversample, kernel_width, base + ws.table,
                          ws.keys[0], arith, stream);
    if (rc)
        return rc;
#define SCATTER(PP) scatter_vis_kernel<PP><<<blocks, 256, 0, stream>>>(index.Current(), num_vis, vis_s, \
        static_cast<float2 *>(vis))
    switch (P) {
    case 1: SCATTER(1); break;
    case 2: SCATTER(2); break;
    case 3: SCATTER(3); break;
    default: SCATTER(4); break;
    }
#undef SCATTER
    return kimg_launch_status();
}

int kimg_grid_binned(void *grid, int64_t grid_row_stride, int64_t grid_pol_stride, int grid_size,
                     int P, const float *weights_grid, int64_t wg_row_stride, int64_t wg_pol_stride,
                     const int16_t *uv, const int16_t *w_plane, const void *vis, int64_t num_vis,
                     const void *convolve_kernel, int w_planes, int oversample, int kernel_width,
                     void *workspace, size_t workspace_bytes, int arith, hipStream_t stream)
{
    if (num_vis >= ((int64_t) 1 << 31))
        return KIMG_EUNSUPPORTED;
    binned_ws ws;
    hipError_t e = layout(num_vis, P, w_planes, oversample, kernel_width, ws);
    if (e != hipSuccess)
        return -(int) e;
    if (workspace == nullptr || workspace_bytes < ws.total)
        return KIMG_EWORKSPACE;
    unsigned char *base = static_cast<unsigned char *>(workspace);
    hipcub::DoubleBuffer<unsigned> keys(reinterpret_cast<unsigned *>(base + ws.keys[0]),
                                        reinterpret_cast<unsigned *>(base + ws.keys[1]));
    hipcub::DoubleBuffer<unsigned> index(reinterpret_cast<unsigned *>(base + ws.index[0]),
                                         reinterpret_cast<unsigned *>(base + ws.index[1]));
    int rc = sort_by_tile(uv, num_vis, grid_size, kernel_width, base, ws, keys, index, stream);
    if (rc)
        return rc;
    const int blocks = kimg_divup(num_vis, 256);
    int2 *uv_s = reinterpret_cast<int2 *>(base + ws.uv);
    int16_t *wp_s = reinterpret_cast<int16_t *>(base + ws.wp);
    float2 *vis_s = reinterpret_cast<float2 *>(base + ws.vis);
#define GATHER(PP) gather_kernel<PP><<<blocks, 256, 0, stream>>>(index.Current(), num_vis, \
        reinterpret_cast<const int2 *>(uv), w_plane, static_cast<const float2 *>(vis), uv_s, wp_s, vis_s)
    switch (P) {
    case 1: GATHER(1); break;
    case 2: GATHER(2); break;
    case 3: GATHER(3); break;
    default: GATHER(4); break;
    }
#undef GATHER
    rc = kimg_launch_status();
    if (rc)
        return rc;
    return kimg_grid_mfma(grid, grid_row_stride, grid_pol_stride, grid_size, P, weights_grid,
                          wg_row_stride, wg_pol_stride, reinterpret_cast<const int16_t *>(uv_s), wp_s,
                          vis_s, num_vis, convolve_kernel, w_planes, oversample, kernel_width,
                          base + ws.table, ws.keys[0], arith, stream);
}

extern "C" size_t kimg_grid_binned_workspace_bytes(int64_t max_vis, int num_polarizations,
                                                  int w_planes, int oversample, int kernel_width)
{
    return kimg_grid_binned_workspace_bytes_impl(max_vis, num_polarizations, w_planes, oversample,
                                                 kernel_width);
}

extern "C" size_t kimg_degrid_binned_workspace_bytes(int64_t max_vis, int num_polarizations,
                                                    int w_planes, int oversample, int kernel_width)
{
    if (max_vis <= 0 || max_vis >= ((int64_t) 1 << 31)
        || !kimg_degrid_mfma_supported(num_polarizations, w_planes, oversample, kernel_width))
        return 0;
    binned_ws ws;
    if (layout(max_vis, num_polarizations, w_planes, oversample, kernel_width, ws, true) != hipSuccess)
        return 0;
    return ws.total;
}

extern "C" int kimg_grid_jumps(const int16_t *uv, int64_t num_vis, int kernel_width,
                               uint32_t *count, void *stream)
{
    KIMG_CHECK_ARG(uv && count && num_vis >= 0 && kernel_width >= 1);
    hipStream_t s = (hipStream_t) stream;
    KIMG_HIP(hipMemsetAsync(count, 0, sizeof(uint32_t), s));
    if (num_vis < 2)
        return 0;
    int blocks = kimg_divup(num_vis, 256 * 16);
    if (blocks > 2048)
        blocks = 2048;
    const int slack = kernel_width > 2 * WIN ? 0 : window_slack(kernel_width);
    jump_count_kernel<<<blocks, 256, 0, s>>>(reinterpret_cast<const int2 *>(uv), num_vis, slack, count);
    return kimg_launch_status();
}

// (kimg_preload, api.hip)
KIMG_PRELOAD_THIS_UNIT(bin_key_kernel)
