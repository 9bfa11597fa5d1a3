// Visibility preprocessing on the device: polarization (Mueller) conversion, w-flip, weight
// pre-multiply, NaN squash, UVW quantisation, adjacent-merge compression and the stable bucket
// sort by w-slice.  Replaces visibility_collector<P>::add_impl2 / compress of the reference
// (preprocess.cpp:334-372, 390-513), which runs on host cores under OpenMP.
//
// All of it is streaming integer/byte work bound by HBM:
//   convert   reads 12 + 12 Q B and writes 12 + 12 P B per input visibility,
//   compress  is two prefix sums (hipCUB), a head-of-run merge and, for several w-slices,
//             one stable radix sort on the slice id.
// The merge keeps the reference's arithmetic: a run of adjacent equal keys is summed left to
// right in float32 by the thread that owns the head of the run, so sums are bit-identical to
// the sequential host loop.
#include "kimg_common.h"
#include <hipcub/hipcub.hpp>

namespace {

struct pp_matrices {
    float2 stokes[16];      // P x Q (no feed angles) or P x 4
    float2 circular[16];    // 4 x Q
};

struct pp_channel {
    float uv_scale, w_scale, half_planes;
    int w_planes, oversample, max_slice_plane;
};

__device__ inline float2 cmul(float2 a, float2 b)
{
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// MulZ<T>::operator* (mulz.h:37-40): a product with an exact zero is zero even if the other
// factor is not finite.
__device__ inline float2 mulz(float2 a, float2 b)
{
    bool nz = (a.x != 0.0f || a.y != 0.0f) && (b.x != 0.0f || b.y != 0.0f);
    return nz ? cmul(a, b) : make_float2(0.0f, 0.0f);
}

__device__ inline float mulz(float a, float b)
{
    return (a != 0.0f && b != 0.0f) ? a * b : 0.0f;
}

// subpixel_coord, preprocess.cpp:313-323
__device__ inline void subpixel_coord(float x, int oversample, short &pixel, short &sub)
{
    int xs = (int) floorf(x * (float) oversample);
    int p = xs / oversample, s = xs % oversample;
    if (s < 0) {
        p--;
        s += oversample;
    }
    pixel = (short) p;
    sub = (short) s;
}

template <int P, int Q>
__global__ __launch_bounds__(256) void pp_convert_kernel(
    int64_t n, const float *__restrict__ uvw, const float *__restrict__ weights,
    const float2 *__restrict__ vis, const float *__restrict__ fa1, const float *__restrict__ fa2,
    pp_matrices mat, pp_channel ch,
    short *__restrict__ key, float *__restrict__ out_w, float2 *__restrict__ out_vis)
{
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    float wi[Q];
    float2 vi[Q];
    bool flagged = false;
#pragma unroll
    for (int q = 0; q < Q; q++) {
        wi[q] = weights[i * Q + q];
        vi[q] = vis[i * Q + q];
        flagged |= (wi[q] == 0.0f);                       // :446
    }
    int *key32 = reinterpret_cast<int *>(key + 6 * i);
    if (flagged) {
        key32[0] = key32[1] = key32[2] = 0;
#pragma unroll
        for (int p = 0; p < P; p++) {
            out_w[i * P + p] = 0.0f;
            out_vis[i * P + p] = make_float2(0.0f, 0.0f);
        }
        return;
    }
    float2 M[P][Q];
    if (fa1 == nullptr) {
#pragma unroll
        for (int p = 0; p < P; p++)
#pragma unroll
            for (int q = 0; q < Q; q++)
                M[p][q] = mat.stokes[p * Q + q];
    } else {                                              // :244-258
        float s1, c1, s2, c2;
        sincosf(fa1[i], &s1, &c1);
        sincosf(fa2[i], &s2, &c2);
        float2 r1 = make_float2(c1, s1), r2 = make_float2(c2, s2);
        float2 rr = cmul(r1, make_float2(r2.x, -r2.y)), rl = cmul(r1, r2);
        float2 scale[4] = {rr, rl, make_float2(rl.x, -rl.y), make_float2(rr.x, -rr.y)};
        float2 mu[4][Q];
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int q = 0; q < Q; q++)
                mu[k][q] = cmul(mat.circular[k * Q + q], scale[k]);
#pragma unroll
        for (int p = 0; p < P; p++)
#pragma unroll
            for (int q = 0; q < Q; q++) {
                float2 acc = make_float2(0.0f, 0.0f);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    float2 t = cmul(mat.stokes[p * 4 + k], mu[k][q]);
                    acc.x += t.x;
                    acc.y += t.y;
                }
                M[p][q] = acc;
            }
    }
    float2 xvis[P];
    float xw[P];
#pragma unroll
    for (int p = 0; p < P; p++) {
        float2 acc = make_float2(0.0f, 0.0f);
        float var = 0.0f;
#pragma unroll
        for (int q = 0; q < Q; q++) {
            float2 t = mulz(M[p][q], vi[q]);              // :456
            acc.x += t.x;
            acc.y += t.y;
            float m2 = M[p][q].x * M[p][q].x + M[p][q].y * M[p][q].y;
            var += mulz(m2, 1.0f / fabsf(wi[q]));         // :468-471
        }
        xvis[p] = acc;
        xw[p] = 1.0f / var;
    }
    float u = uvw[3 * i], v = uvw[3 * i + 1], w = uvw[3 * i + 2];
    if (w < 0.0f) {                                       // :476-482
        u = -u;
        v = -v;
        w = -w;
#pragma unroll
        for (int p = 0; p < P; p++)
            xvis[p].y = -xvis[p].y;
    }
#pragma unroll
    for (int p = 0; p < P; p++) {                         // :483-496
        float weight = xw[p];
        float2 s = make_float2(xvis[p].x * weight, xvis[p].y * weight);
        if (!isfinite(s.x) || !isfinite(s.y)) {
            s = make_float2(0.0f, 0.0f);
            weight = 0.0f;
        }
        out_vis[i * P + p] = s;
        out_w[i * P + p] = weight;
    }
    u = u * ch.uv_scale;
    v = v * ch.uv_scale;
    w = truncf(w * ch.w_scale + ch.half_planes);          // :501
    int wsp = min((int) w, ch.max_slice_plane);
    short k0, k1, k2, k3;
    subpixel_coord(u, ch.oversample, k0, k2);
    subpixel_coord(v, ch.oversample, k1, k3);
    short k4 = (short) (wsp % ch.w_planes), k5 = (short) (wsp / ch.w_planes);
    key32[0] = (int) (unsigned short) k0 | ((int) k1 << 16);
    key32[1] = (int) (unsigned short) k2 | ((int) k3 << 16);
    key32[2] = (int) (unsigned short) k4 | ((int) k5 << 16);
}

// ---- compress ---------------------------------------------------------------------------

// Run length per w-slice from the sorted slice ids (a same-address atomic per record would
// serialise in L2): counts[s] = lower_bound(s + 1) - lower_bound(s).  Unused tail entries
// carry the sentinel w_slices and sort behind everything.
__global__ void pp_slice_counts_kernel(
    int64_t n, const unsigned short *__restrict__ sorted, int w_slices,
    unsigned long long *__restrict__ counts)
{
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= w_slices)
        return;
    auto lower_bound = [&](int value) {
        int64_t lo = 0, hi = n;
        while (lo < hi) {
            int64_t mid = (lo + hi) >> 1;
            if ((int) sorted[mid] < value)
                lo = mid + 1;
            else
                hi = mid;
        }
        return lo;
    };
    counts[s] = (unsigned long long) (lower_bound(s + 1) - lower_bound(s));
}

// Apply the stable slice order: output t takes merged record perm[t].
template <int P>
__global__ __launch_bounds__(256) void pp_gather_stream_kernel(
    int64_t n, const int *__restrict__ ipos, const int *__restrict__ perm,
    const short *__restrict__ m_uv, const short *__restrict__ m_wplane,
    const float *__restrict__ m_w, const float2 *__restrict__ m_vis,
    short *__restrict__ out_uv, short *__restrict__ out_wplane, float *__restrict__ out_w,
    float2 *__restrict__ out_vis)
{
    int64_t t = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n)
        return;
    const int m = ipos[n - 1];
    if (t >= m)
        return;
    int64_t s = perm[t];
    const int2 uv = *reinterpret_cast<const int2 *>(m_uv + 4 * s);
    *reinterpret_cast<int2 *>(out_uv + 4 * t) = uv;
    out_wplane[t] = m_wplane[s];
#pragma unroll
    for (int p = 0; p < P; p++) {
        out_w[t * P + p] = m_w[s * P + p];
        out_vis[t * P + p] = m_vis[s * P + p];
    }
}

__global__ __launch_bounds__(256) void pp_real_to_complex_kernel(
    int64_t n, const float *__restrict__ src, float2 *__restrict__ dst)
{
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        dst[i] = make_float2(src[i], 0.0f);
}

// ---- compress, fused (round 2): two scans with the index kernels folded into their input
// iterators, and a merge that walks the uncompacted stream -----------------------------------
//   scan 1 (inclusive max): pv[i] = index of the last valid record at or before i (-1: none),
//           input computed on the fly from the weights (the flag test of :339,:352);
//   scan 2 (inclusive sum): ipos[i] = number of run heads at or before i, a head being a valid
//           record whose key differs from the previous valid record's (:354), input computed on the
//           fly from pv and the keys;
//   head index: hidx[o] = input index of the head of run o;
//   merge:  one thread per output run sums it in arrival order (skipping flagged records).
// Seven launches per buffer instead of thirteen (each scan is a rocPRIM look-back scan with its
// small state-initialisation kernel), no compaction pass and no index indirection in the merge.
struct pp_valid_index {
    const float *w;
    int P;
    __host__ __device__ int operator()(int i) const
    {
#if defined(__HIP_DEVICE_COMPILE__)
        return w[(int64_t) i * P] != 0.0f ? i : -1;
#else
        return -1;
#endif
    }
};

struct pp_max_op {
    __host__ __device__ int operator()(int a, int b) const { return a > b ? a : b; }
};

struct pp_head_flag {
    const int *pv;
    const short *key;
    int window;         // merging never crosses a multiple of `window` records (0: no cuts)
    __host__ __device__ int operator()(int i) const
    {
#if defined(__HIP_DEVICE_COMPILE__)
        if (pv[i] != i)
            return 0;                           // flagged record
        const int prev = i > 0 ? pv[i - 1] : -1;
        // (the first unflagged record of a window starts a run whatever came before: the windows
        // are the reference's buffers, each compressed on its own, preprocess.cpp:431-509)
        if (prev < 0 || (window > 0 && prev < i - i % window))
            return 1;
        const int *a = reinterpret_cast<const int *>(key + 6 * (int64_t) i);
        const int *b = reinterpret_cast<const int *>(key + 6 * (int64_t) prev);
        return (a[0] != b[0]) | (a[1] != b[1]) | (a[2] != b[2]);
#else
        return 0;
#endif
    }
};

// hidx[o] = input index of the head of output run o (one thread per input record)
__global__ __launch_bounds__(256) void pp_headidx_stream_kernel(
    int64_t n, const int *__restrict__ pv, const int *__restrict__ ipos, int *__restrict__ hidx)
{
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const int here = ipos[i];
    if (pv[i] == i && here != (i > 0 ? ipos[i - 1] : 0))
        hidx[here - 1] = (int) i;
}

// One thread per OUTPUT record (dense waves): it walks the input stream from its head to the next
// run's head, adding the unflagged records in arrival order (:356-360, bit-identical to the host
// loop), and writes the merged record.  With one w-slice the record goes straight to its final place.
template <int P>
__global__ __launch_bounds__(256) void pp_merge_stream_kernel(
    int64_t n, const int *__restrict__ ipos, const int *__restrict__ hidx,
    const short *__restrict__ key, const float *__restrict__ w, const float2 *__restrict__ vis,
    int single_slice,
    short *__restrict__ m_uv, short *__restrict__ m_wplane, float *__restrict__ m_w,
    float2 *__restrict__ m_vis, unsigned short *__restrict__ skey,
    unsigned long long *__restrict__ counts)
{
    const int64_t o = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n)
        return;
    const int m = ipos[n - 1];
    if (single_slice && o == 0)
        counts[0] = (unsigned long long) m;
    if (o >= m)
        return;
    const int64_t i = hidx[o];
    const int64_t i_end = (o + 1 < m) ? hidx[o + 1] : n;
    const int *kp = reinterpret_cast<const int *>(key + 6 * i);
    const int k01 = kp[0], k23 = kp[1], k45 = kp[2];
    float aw[P];
    float2 av[P];
#pragma unroll
    for (int p = 0; p < P; p++) {
        aw[p] = w[i * P + p];
        av[p] = vis[i * P + p];
    }
    // Long runs (slowly moving short baselines) are latency chains: fetch UNROLL records ahead,
    // then add them in arrival order; flagged records (weight 0) inside the run are skipped.
    constexpr int UNROLL = 8;
    int64_t j = i + 1;
    for (; j + UNROLL <= i_end; j += UNROLL) {
        float2 tv[UNROLL][P];
        float tw[UNROLL][P];
#pragma unroll
        for (int u = 0; u < UNROLL; u++)
#pragma unroll
            for (int p = 0; p < P; p++) {
                tv[u][p] = vis[(j + u) * P + p];
                tw[u][p] = w[(j + u) * P + p];
            }
#pragma unroll
        for (int u = 0; u < UNROLL; u++)
            if (tw[u][0] != 0.0f) {
#pragma unroll
                for (int p = 0; p < P; p++) {
                    av[p].x += tv[u][p].x;
                    av[p].y += tv[u][p].y;
                    aw[p] += tw[u][p];
                }
            }
    }
    for (; j < i_end; j++) {
        if (w[j * P] == 0.0f)
            continue;
#pragma unroll
        for (int p = 0; p < P; p++) {
            const float2 t = vis[j * P + p];
            av[p].x += t.x;
            av[p].y += t.y;
            aw[p] += w[j * P + p];
        }
    }
    int *uv32 = reinterpret_cast<int *>(m_uv + 4 * o);
    uv32[0] = k01;
    uv32[1] = k23;
    m_wplane[o] = (short) (k45 & 0xffff);
#pragma unroll
    for (int p = 0; p < P; p++) {
        m_w[o * P + p] = aw[p];
        m_vis[o * P + p] = av[p];
    }
    if (!single_slice)
        skey[o] = (unsigned short) (k45 >> 16);
}

// (several w-slices only) the sort arrays: sentinel keys, identity values
__global__ __launch_bounds__(256) void pp_sort_init_kernel(
    int64_t n, unsigned short *__restrict__ skey, int *__restrict__ sval, int w_slices)
{
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    skey[i] = (unsigned short) w_slices;      // sorts behind every real slice
    sval[i] = (int) i;
}

inline size_t align_up(size_t x) { return (x + 255) & ~(size_t) 255; }

struct pp_workspace {
    int *valid, *pos, *hidx, *sval_in, *sval_out;   // valid = pv (last valid index), pos = ipos (heads so far)
    unsigned short *skey_in, *skey_out;
    short *m_uv, *m_wplane;
    float *m_w;
    float2 *m_vis;
    void *cub;
    size_t cub_bytes, total;
};

int slice_bits(int w_slices)
{
    int bits = 1;
    while ((1 << bits) <= w_slices)     // keys 0..w_slices (sentinel) must fit
        bits++;
    return bits;
}

hipError_t layout_workspace(int64_t n, int P, char *base, pp_workspace &ws)
{
    size_t scan_bytes = 0, sort_bytes = 0, scan2_bytes = 0;
    hipError_t e = hipcub::DeviceScan::InclusiveSum(nullptr, scan_bytes, (int *) nullptr,
                                                    (int *) nullptr, (int) n);
    if (e != hipSuccess)
        return e;
    e = hipcub::DeviceScan::InclusiveScan(nullptr, scan2_bytes, (int *) nullptr, (int *) nullptr,
                                          pp_max_op(), (int) n);
    if (e != hipSuccess)
        return e;
    if (scan2_bytes > scan_bytes)
        scan_bytes = scan2_bytes;
    e = hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, (unsigned short *) nullptr,
                                           (unsigned short *) nullptr, (int *) nullptr,
                                           (int *) nullptr, (int) n, 0, 16);
    if (e != hipSuccess)
        return e;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char *p = base ? base + off : nullptr;
        off += align_up(bytes);
        return p;
    };
    size_t ni = (size_t) n * sizeof(int);
    ws.valid = (int *) take(ni);
    ws.pos = (int *) take(ni);
    ws.hidx = (int *) take(ni);
    ws.sval_in = (int *) take(ni);
    ws.sval_out = (int *) take(ni);
    ws.skey_in = (unsigned short *) take((size_t) n * 2);
    ws.skey_out = (unsigned short *) take((size_t) n * 2);
    ws.m_uv = (short *) take((size_t) n * 8);
    ws.m_wplane = (short *) take((size_t) n * 2);
    ws.m_w = (float *) take((size_t) n * P * 4);
    ws.m_vis = (float2 *) take((size_t) n * P * 8);
    ws.cub_bytes = scan_bytes > sort_bytes ? scan_bytes : sort_bytes;
    ws.cub = take(ws.cub_bytes);
    ws.total = off;
    return hipSuccess;
}

template <int P, int Q>
int convert_launch(int64_t n, const float *uvw, const float *weights, const float2 *vis,
                   const float *fa1, const float *fa2, const pp_matrices &mat,
                   const pp_channel &ch, short *key, float *out_w, float2 *out_vis,
                   hipStream_t stream)
{
    pp_convert_kernel<P, Q><<<kimg_divup(n, 256), 256, 0, stream>>>(
        n, uvw, weights, vis, fa1, fa2, mat, ch, key, out_w, out_vis);
    return kimg_launch_status();
}

template <int P>
int convert_q(int Q, int64_t n, const float *uvw, const float *weights, const float2 *vis,
              const float *fa1, const float *fa2, const pp_matrices &mat, const pp_channel &ch,
              short *key, float *out_w, float2 *out_vis, hipStream_t stream)
{
    switch (Q) {
    case 1: return convert_launch<P, 1>(n, uvw, weights, vis, fa1, fa2, mat, ch, key, out_w, out_vis, stream);
    case 2: return convert_launch<P, 2>(n, uvw, weights, vis, fa1, fa2, mat, ch, key, out_w, out_vis, stream);
    case 3: return convert_launch<P, 3>(n, uvw, weights, vis, fa1, fa2, mat, ch, key, out_w, out_vis, stream);
    default: return convert_launch<P, 4>(n, uvw, weights, vis, fa1, fa2, mat, ch, key, out_w, out_vis, stream);
    }
}

template <int P>
int compress_impl(int64_t n, int w_slices, int window, const short *key, const float *w, const float2 *vis,
                  short *out_uv, short *out_wplane, float *out_w, float2 *out_vis,
                  unsigned long long *counts, pp_workspace &ws, hipStream_t stream)
{
    const int blocks = kimg_divup(n, 256);
    const int single = (w_slices == 1);
    int *pv = ws.valid, *ipos = ws.pos;         // (the arrays of the unfused pipeline, reused)
    size_t cb = ws.cub_bytes;
    hipcub::CountingInputIterator<int> index(0);
    hipcub::TransformInputIterator<int, pp_valid_index, hipcub::CountingInputIterator<int>>
        valid_index(index, pp_valid_index{w, P});
    KIMG_HIP(hipcub::DeviceScan::InclusiveScan(ws.cub, cb, valid_index, pv, pp_max_op(), (int) n,
                                               stream));
    hipcub::TransformInputIterator<int, pp_head_flag, hipcub::CountingInputIterator<int>>
        head_flag(index, pp_head_flag{pv, key, window});
    cb = ws.cub_bytes;
    KIMG_HIP(hipcub::DeviceScan::InclusiveSum(ws.cub, cb, head_flag, ipos, (int) n, stream));
    if (!single)
        pp_sort_init_kernel<<<blocks, 256, 0, stream>>>(n, ws.skey_in, ws.sval_in, w_slices);
    pp_headidx_stream_kernel<<<blocks, 256, 0, stream>>>(n, pv, ipos, ws.hidx);
    pp_merge_stream_kernel<P><<<blocks, 256, 0, stream>>>(
        n, ipos, ws.hidx, key, w, vis, single,
        single ? out_uv : ws.m_uv, single ? out_wplane : ws.m_wplane, single ? out_w : ws.m_w,
        single ? out_vis : ws.m_vis, ws.skey_in, counts);
    if (!single) {
        cb = ws.cub_bytes;
        KIMG_HIP(hipcub::DeviceRadixSort::SortPairs(ws.cub, cb, ws.skey_in, ws.skey_out, ws.sval_in,
                                                    ws.sval_out, (int) n, 0, slice_bits(w_slices),
                                                    stream));
        pp_slice_counts_kernel<<<kimg_divup(w_slices, 64), 64, 0, stream>>>(n, ws.skey_out, w_slices,
                                                                           counts);
        pp_gather_stream_kernel<P><<<blocks, 256, 0, stream>>>(
            n, ipos, ws.sval_out, ws.m_uv, ws.m_wplane, ws.m_w, ws.m_vis,
            out_uv, out_wplane, out_w, out_vis);
    }
    return kimg_launch_status();
}

}  // namespace

extern "C" int kimg_preprocess_convert(
    int num_pols, int num_in_pols, int64_t num_vis,
    const float *uvw, const float *weights, const void *vis,
    const float *feed_angle1, const float *feed_angle2,
    const float *mueller_stokes_host, const float *mueller_circular_host,
    float max_w, int w_slices, int w_planes, int oversample, float cell_size,
    int16_t *key, float *out_weights, void *out_vis, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    KIMG_CHECK_ARG(num_pols >= 1 && num_pols <= 4 && num_in_pols >= 1 && num_in_pols <= 4);
    KIMG_CHECK_ARG(num_vis >= 0 && num_vis < ((int64_t) 1 << 31));
    KIMG_CHECK_ARG(w_slices >= 1 && w_planes >= 1 && oversample >= 1 && cell_size > 0 && max_w > 0);
    KIMG_CHECK_ARG(w_slices <= 32767 && w_planes <= 32767);      // both are stored as int16
    KIMG_CHECK_ARG(mueller_stokes_host != nullptr);
    KIMG_CHECK_ARG((feed_angle1 == nullptr) == (feed_angle2 == nullptr));
    KIMG_CHECK_ARG((feed_angle1 == nullptr) == (mueller_circular_host == nullptr));   // :592-603
    if (num_vis == 0)
        return 0;
    KIMG_CHECK_ARG(uvw && weights && vis && key && out_weights && out_vis);
    pp_matrices mat = {};
    const int stokes_cols = feed_angle1 ? 4 : num_in_pols;
    for (int i = 0; i < num_pols * stokes_cols; i++)
        mat.stokes[i] = make_float2(mueller_stokes_host[2 * i], mueller_stokes_host[2 * i + 1]);
    if (feed_angle1)
        for (int i = 0; i < 4 * num_in_pols; i++)
            mat.circular[i] = make_float2(mueller_circular_host[2 * i], mueller_circular_host[2 * i + 1]);
    pp_channel ch;
    ch.uv_scale = 1.0f / cell_size;                                        // :428
    ch.w_scale = ((float) w_slices - 0.5f) * (float) w_planes / max_w;     // :429
    ch.half_planes = (float) w_planes * 0.5f;
    ch.w_planes = w_planes;
    ch.oversample = oversample;
    ch.max_slice_plane = w_slices * w_planes - 1;                          // :430
    const float2 *v = static_cast<const float2 *>(vis);
    float2 *ov = static_cast<float2 *>(out_vis);
    switch (num_pols) {
    case 1: return convert_q<1>(num_in_pols, num_vis, uvw, weights, v, feed_angle1, feed_angle2, mat, ch, key, out_weights, ov, stream);
    case 2: return convert_q<2>(num_in_pols, num_vis, uvw, weights, v, feed_angle1, feed_angle2, mat, ch, key, out_weights, ov, stream);
    case 3: return convert_q<3>(num_in_pols, num_vis, uvw, weights, v, feed_angle1, feed_angle2, mat, ch, key, out_weights, ov, stream);
    default: return convert_q<4>(num_in_pols, num_vis, uvw, weights, v, feed_angle1, feed_angle2, mat, ch, key, out_weights, ov, stream);
    }
}

extern "C" size_t kimg_preprocess_workspace_bytes(int64_t num_vis, int num_pols)
{
    if (num_vis <= 0 || num_vis >= ((int64_t) 1 << 31) || num_pols < 1 || num_pols > 4)
        return 0;
    pp_workspace ws;
    if (layout_workspace(num_vis, num_pols, nullptr, ws) != hipSuccess)
        return 0;
    return ws.total;
}

extern "C" int kimg_preprocess_compress(
    int num_pols, int64_t num_vis, int w_slices,
    const int16_t *key, const float *weights, const void *vis,
    int16_t *out_uv, int16_t *out_w_plane, float *out_weights, void *out_vis,
    uint64_t *counts, int64_t merge_window, void *workspace, size_t workspace_bytes, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    KIMG_CHECK_ARG(num_pols >= 1 && num_pols <= 4);
    KIMG_CHECK_ARG(num_vis >= 0 && num_vis < ((int64_t) 1 << 31));
    KIMG_CHECK_ARG(merge_window >= 0 && merge_window < ((int64_t) 1 << 31));
    const int window = (int) merge_window;
    KIMG_CHECK_ARG(w_slices >= 1 && w_slices <= 32767 && counts != nullptr);
    if (num_vis == 0) {
        KIMG_HIP(hipMemsetAsync(counts, 0, sizeof(uint64_t) * w_slices, stream));
        return 0;
    }
    KIMG_CHECK_ARG(key && weights && vis && out_uv && out_w_plane && out_weights && out_vis);
    pp_workspace ws;
    KIMG_HIP(layout_workspace(num_vis, num_pols, static_cast<char *>(workspace), ws));
    if (workspace == nullptr || workspace_bytes < ws.total)
        return KIMG_EWORKSPACE;
    const float2 *v = static_cast<const float2 *>(vis);
    float2 *ov = static_cast<float2 *>(out_vis);
    unsigned long long *c = reinterpret_cast<unsigned long long *>(counts);
    switch (num_pols) {
    case 1: return compress_impl<1>(num_vis, w_slices, window, key, weights, v, out_uv, out_w_plane, out_weights, ov, c, ws, stream);
    case 2: return compress_impl<2>(num_vis, w_slices, window, key, weights, v, out_uv, out_w_plane, out_weights, ov, c, ws, stream);
    case 3: return compress_impl<3>(num_vis, w_slices, window, key, weights, v, out_uv, out_w_plane, out_weights, ov, c, ws, stream);
    default: return compress_impl<4>(num_vis, w_slices, window, key, weights, v, out_uv, out_w_plane, out_weights, ov, c, ws, stream);
    }
}

extern "C" int kimg_real_to_complex(void *dst, const float *src, int64_t count, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    KIMG_CHECK_ARG(count >= 0);
    if (count == 0)
        return 0;
    KIMG_CHECK_ARG(dst && src);
    pp_real_to_complex_kernel<<<kimg_divup(count, 256), 256, 0, stream>>>(
        count, src, static_cast<float2 *>(dst));
    return kimg_launch_status();
}

// (kimg_preload, api.hip)
KIMG_PRELOAD_THIS_UNIT(pp_slice_counts_kernel)
