// Once-per-channel re-ordering (and re-compression) of a stored W-slice.
//
// No counterpart launch site in the reference: its preprocessor emits the records of a slice in
// arrival order -- baseline-sorted load blocks, adjacent-merged (loader_ms.py:465-468,
// preprocess.cpp:334-397) -- and its GPU gridder copes with that order on every pass (bin sort
// inside grid.mako, grid.py:436-463 get_bin_size).  Here the visibilities of a channel stay in HBM
// for all passes (weights, PSF, image, degrid + regrid per major cycle), so the store is put ONCE
// into the order the window kernels (grid_mfma.hip, degrid_mfma.hip) run fastest on:
//
//   strips of (window slack + 1) grid COLUMNS, each strip swept along v, odd strips backwards
//   (serpentine), i.e. key = strip(u) : v' [: u in strip : sub_v : sub_u : w_plane].
//
// Inside a strip every footprint fits the 32-cell window along u, and the window only ever moves
// along v: a row move flushes (gridder) or reloads (degridder) whole 256-byte rows with two wave
// instructions, where a column move takes sixteen with a few lanes each, and a baseline change in
// the arrival order takes a whole 8 KB window.  Measured (C2 geometry, 7.2 M merged records of a
// 50 M-visibility channel): gridder 3.0 -> 9.8 G records/s, degridder 4.0 -> 8.8.
//
// With `merge` the sort covers the whole quantised coordinate, so that ALL records of a slice with
// equal (u, v, sub_u, sub_v, w_plane) become neighbours -- the reference only merges records that
// arrive next to each other -- and each run is summed left to right in float32 in arrival order
// (the sort is stable), exactly like compress() sums an arrival run (preprocess.cpp:334-372).
// Gridding, weights and degridding are linear in (vis, weights) for equal coordinates, so the
// results differ from the unmerged store's only by the order of float additions.
#include "kimg_common.h"
#include <hipcub/hipcub.hpp>

namespace {

constexpr int WIN = 32;

__host__ __device__ inline int strip_width(int K)
{
    const int taps = K > WIN ? (K + 1) / 2 : K;
    return WIN - taps + 1;              // window slack + 1
}

inline int bits_for(unsigned values)      // bits needed for 0 .. values - 1
{
    int b = 0;
    while (b < 32 && ((uint64_t) 1 << b) < values)
        b++;
    return b;
}

struct key_layout {
    int width;              // columns per strip
    int wp_bits, sub_bits, in_bits;
    int total_bits;
};

key_layout make_layout(int kernel_width, int oversample, int w_planes, bool merge)
{
    key_layout k;
    k.width = strip_width(kernel_width);
    k.wp_bits = merge ? bits_for((unsigned) w_planes) : 0;
    k.sub_bits = merge ? bits_for((unsigned) oversample) : 0;
    k.in_bits = merge ? bits_for((unsigned) k.width) : 0;
    const int strips = 65536 / k.width + 1;
    k.total_bits = k.wp_bits + 2 * k.sub_bits + k.in_bits + 16 + bits_for((unsigned) strips);
    return k;
}

__global__ __launch_bounds__(256) void strip_key_kernel(
    const int2 *__restrict__ uv, const int16_t *__restrict__ w_plane, int64_t n, key_layout k,
    unsigned long long *__restrict__ keys, unsigned *__restrict__ index)
{
    const int64_t i = blockIdx.x * (int64_t) blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const int2 r = uv[i];
    const unsigned u = (unsigned) ((short) (r.x & 0xffff) + 32768);
    const unsigned v = (unsigned) ((short) (r.x >> 16) + 32768);
    const unsigned strip = u / (unsigned) k.width;
    const unsigned vv = (strip & 1u) ? 65535u - v : v;      // serpentine
    unsigned long long key = ((unsigned long long) strip << 16) | vv;
    if (k.in_bits | k.sub_bits | k.wp_bits) {
        const unsigned su = (unsigned) (r.y & 0xffff), sv = (unsigned) ((unsigned) r.y >> 16);
        key = (key << k.in_bits) | (u - strip * (unsigned) k.width);
        key = (key << k.sub_bits) | (sv & ((1u << k.sub_bits) - 1u));
        key = (key << k.sub_bits) | (su & ((1u << k.sub_bits) - 1u));
        key = (key << k.wp_bits) | ((unsigned) (unsigned short) w_plane[i] & ((1u << k.wp_bits) - 1u));
    }
    keys[i] = key;
    index[i] = (unsigned) i;
}

template <int P>
__global__ __launch_bounds__(256) void store_gather_kernel(
    const unsigned *__restrict__ index, int64_t n, const int2 *__restrict__ uv,
    const int16_t *__restrict__ w_plane, const float *__restrict__ weights,
    const float2 *__restrict__ vis, int2 *__restrict__ uv_out, int16_t *__restrict__ wp_out,
    float *__restrict__ w_out, float2 *__restrict__ vis_out, unsigned long long *__restrict__ count)
{
    const int64_t i = blockIdx.x * (int64_t) blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    if (i == 0)
        *count = (unsigned long long) n;
    const int64_t src = index[i];
    uv_out[i] = uv[src];
    wp_out[i] = w_plane[src];
#pragma unroll
    for (int p = 0; p < P; p++) {
        w_out[i * P + p] = weights[src * P + p];
        vis_out[i * P + p] = vis[src * P + p];
    }
}

// head[i] = 1 when sorted record i starts a run of equal coordinates.  Out-of-range sub-pixel or
// plane indices were masked into the key; the comparison is on the records themselves, so two
// such records are only merged when they are truly equal.
// A run also ends at every multiple of STORE_MAX_RUN sorted positions: one thread sums a run left
// to right (float32, arrival order: that order IS the result), so a run of a million equal
// coordinates -- autocorrelations or zero-length baselines at uv = 0, a snapshot's PSF, synthetic
// data with constant uv -- would be one thread working for seconds while the device idles.  Cut
// runs are simply several records at the same coordinates: gridding, weights and degridding are
// linear in them.
constexpr int64_t STORE_MAX_RUN = 4096;
struct head_flag {
    const unsigned *index;
    const int2 *uv;
    const int16_t *w_plane;
    __device__ unsigned operator()(int64_t i) const
    {
        if ((i & (STORE_MAX_RUN - 1)) == 0)
            return 1u;
        const unsigned a = index[i - 1], b = index[i];
        const int2 ra = uv[a], rb = uv[b];
        return (ra.x != rb.x || ra.y != rb.y || w_plane[a] != w_plane[b]) ? 1u : 0u;
    }
};

// One thread per sorted record; the thread of a run's head walks the run (arrival order: the sort
// is stable) and sums weights and visibilities left to right in float32.
template <int P>
__global__ __launch_bounds__(256) void store_merge_kernel(
    const unsigned *__restrict__ index, const unsigned *__restrict__ run_end /* inclusive scan of heads */,
    int64_t n, const int2 *__restrict__ uv, const int16_t *__restrict__ w_plane,
    const float *__restrict__ weights, const float2 *__restrict__ vis, int2 *__restrict__ uv_out,
    int16_t *__restrict__ wp_out, float *__restrict__ w_out, float2 *__restrict__ vis_out,
    unsigned long long *__restrict__ count)
{
    const int64_t i = blockIdx.x * (int64_t) blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const unsigned run = run_end[i];                // 1-based number of the run record i belongs to
    if (i == n - 1)
        *count = run;
    if (i > 0 && run_end[i - 1] == run)
        return;                                     // not a head
    const int64_t first = index[i];
    float w[P];
    float2 s[P];
#pragma unroll
    for (int p = 0; p < P; p++) {
        w[p] = weights[first * P + p];
        s[p] = vis[first * P + p];
    }
    for (int64_t j = i + 1; j < n && run_end[j] == run; j++) {
        const int64_t src = index[j];
#pragma unroll
        for (int p = 0; p < P; p++) {
            const float2 x = vis[src * P + p];
            w[p] += weights[src * P + p];
            s[p].x += x.x;
            s[p].y += x.y;
        }
    }
    const int64_t o = (int64_t) run - 1;
    uv_out[o] = uv[first];
    wp_out[o] = w_plane[first];
#pragma unroll
    for (int p = 0; p < P; p++) {
        w_out[o * P + p] = w[p];
        vis_out[o * P + p] = s[p];
    }
}

struct reorder_ws {
    size_t keys[2], index[2], runs, cub, cub_bytes, total;
};

size_t align256(size_t x) { return (x + 255) / 256 * 256; }

hipError_t layout(int64_t n, reorder_ws &ws)
{
    size_t off = 0;
    for (int i = 0; i < 2; i++) {
        ws.keys[i] = off;
        off += align256((size_t) n * sizeof(unsigned long long));
        ws.index[i] = off;
        off += align256((size_t) n * sizeof(unsigned));
    }
    ws.runs = off;
    off += align256((size_t) n * sizeof(unsigned));
    hipcub::DoubleBuffer<unsigned long long> k(nullptr, nullptr);
    hipcub::DoubleBuffer<unsigned> v(nullptr, nullptr);
    size_t sort_bytes = 0, scan_bytes = 0;
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, k, v, (int) n, 0, 64,
                                                      (hipStream_t) 0);
    if (e != hipSuccess)
        return e;
    head_flag hf{nullptr, nullptr, nullptr};
    hipcub::CountingInputIterator<int64_t> counting(0);
    hipcub::TransformInputIterator<unsigned, head_flag, hipcub::CountingInputIterator<int64_t>> heads(
        counting, hf);
    e = hipcub::DeviceScan::InclusiveSum(nullptr, scan_bytes, heads, (unsigned *) nullptr, (int) n,
                                         (hipStream_t) 0);
    ws.cub_bytes = sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
    ws.cub = off;
    off += align256(ws.cub_bytes);
    ws.total = off;
    return e;
}

} // namespace

extern "C" size_t kimg_store_reorder_workspace_bytes(int64_t num_vis)
{
    if (num_vis <= 0 || num_vis >= ((int64_t) 1 << 31))
        return 0;
    reorder_ws ws;
    if (layout(num_vis, ws) != hipSuccess)
        return 0;
    return ws.total;
}

extern "C" int kimg_store_reorder(int num_polarizations, int64_t num_vis, int kernel_width,
                                  int oversample, int w_planes, int merge, const int16_t *uv,
                                  const int16_t *w_plane, const float *weights, const void *vis,
                                  int16_t *out_uv, int16_t *out_w_plane, float *out_weights,
                                  void *out_vis, uint64_t *out_count, void *workspace,
                                  size_t workspace_bytes, void *stream)
{
    KIMG_CHECK_ARG(num_vis >= 0 && kernel_width >= 1 && oversample >= 1 && w_planes >= 1 && out_count);
    if (num_polarizations < 1 || num_polarizations > 4 || num_vis >= ((int64_t) 1 << 31))
        return KIMG_EUNSUPPORTED;
    hipStream_t s = (hipStream_t) stream;
    if (num_vis == 0) {
        KIMG_HIP(hipMemsetAsync(out_count, 0, sizeof(uint64_t), s));
        return 0;
    }
    KIMG_CHECK_ARG(uv && w_plane && weights && vis && out_uv && out_w_plane && out_weights && out_vis);
    reorder_ws ws;
    hipError_t e = layout(num_vis, ws);
    if (e != hipSuccess)
        return -(int) e;
    if (workspace == nullptr || workspace_bytes < ws.total)
        return KIMG_EWORKSPACE;
    const key_layout k = make_layout(kernel_width, oversample, w_planes, merge != 0);
    if (k.total_bits > 64)
        return KIMG_EUNSUPPORTED;
    unsigned char *base = static_cast<unsigned char *>(workspace);
    hipcub::DoubleBuffer<unsigned long long> keys(
        reinterpret_cast<unsigned long long *>(base + ws.keys[0]),
        reinterpret_cast<unsigned long long *>(base + ws.keys[1]));
    hipcub::DoubleBuffer<unsigned> index(reinterpret_cast<unsigned *>(base + ws.index[0]),
                                         reinterpret_cast<unsigned *>(base + ws.index[1]));
    const int blocks = kimg_divup(num_vis, 256);
    const int2 *uv2 = reinterpret_cast<const int2 *>(uv);
    strip_key_kernel<<<blocks, 256, 0, s>>>(uv2, w_plane, num_vis, k, keys.Current(), index.Current());
    size_t cub_bytes = ws.cub_bytes;
    KIMG_HIP(hipcub::DeviceRadixSort::SortPairs(base + ws.cub, cub_bytes, keys, index, (int) num_vis,
                                                0, k.total_bits, s));
    int2 *uv_o = reinterpret_cast<int2 *>(out_uv);
    const float2 *vis2 = static_cast<const float2 *>(vis);
    float2 *vis_o = static_cast<float2 *>(out_vis);
    if (!merge) {
#define GATHER(PP) store_gather_kernel<PP><<<blocks, 256, 0, s>>>(index.Current(), num_vis, uv2, w_plane, \
        weights, vis2, uv_o, out_w_plane, out_weights, vis_o, reinterpret_cast<unsigned long long *>(out_count))
        switch (num_polarizations) {
        case 1: GATHER(1); break;
        case 2: GATHER(2); break;
        case 3: GATHER(3); break;
        default: GATHER(4); break;
        }
#undef GATHER
        return kimg_launch_status();
    }
    unsigned *runs = reinterpret_cast<unsigned *>(base + ws.runs);
    head_flag hf{index.Current(), uv2, w_plane};
    hipcub::CountingInputIterator<int64_t> counting(0);
    hipcub::TransformInputIterator<unsigned, head_flag, hipcub::CountingInputIterator<int64_t>> heads(
        counting, hf);
    cub_bytes = ws.cub_bytes;
    KIMG_HIP(hipcub::DeviceScan::InclusiveSum(base + ws.cub, cub_bytes, heads, runs, (int) num_vis, s));
#define MERGE(PP) store_merge_kernel<PP><<<blocks, 256, 0, s>>>(index.Current(), runs, num_vis, uv2, \
        w_plane, weights, vis2, uv_o, out_w_plane, out_weights, vis_o, \
        reinterpret_cast<unsigned long long *>(out_count))
    switch (num_polarizations) {
    case 1: MERGE(1); break;
    case 2: MERGE(2); break;
    case 3: MERGE(3); break;
    default: MERGE(4); break;
    }
#undef MERGE
    return kimg_launch_status();
}

// (kimg_preload, api.hip)
KIMG_PRELOAD_THIS_UNIT(strip_key_kernel)
