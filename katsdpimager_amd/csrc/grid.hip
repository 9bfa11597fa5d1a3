// Generic (any kernel width / plane count / 1-4 polarizations) convolutional gridder and
// degridder, and the direct (DFT) predictor.  Mirrors grid.py:786-867 (+grid.mako),
// grid.py:985-1029 (+degrid.mako) and predict.py:386-416 (+predict.mako) of the reference.
//
// The generic gridder is the fallback for parameter combinations the MFMA window kernel
// (grid_mfma.hip) does not cover: every tap is a global float atomic, so it runs at the
// chip's atomic rate (~1.3 TB/s of added bytes), not at the FMA rate.
#include "kimg_common.h"

int kimg_grid_mfma(void *grid, int64_t grid_row_stride, int64_t grid_pol_stride, int grid_size,
                   int P, const float *weights_grid, int64_t wg_row_stride, int64_t wg_pol_stride,
                   const int16_t *uv, const int16_t *w_plane, const void *vis, int64_t num_vis,
                   const void *convolve_kernel, int w_planes, int oversample, int kernel_width,
                   void *workspace, size_t workspace_bytes, int arith, hipStream_t stream);
int kimg_grid_binned(void *grid, int64_t grid_row_stride, int64_t grid_pol_stride, int grid_size,
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
int kimg_degrid_binned(const void *grid, int64_t grid_row_stride, int64_t grid_pol_stride,
                       int grid_size, int P, const int16_t *uv, const int16_t *w_plane,
                       const float *weights, void *vis, int64_t num_vis, const void *convolve_kernel,
                       int w_planes, int oversample, int kernel_width, void *workspace,
                       size_t workspace_bytes, int arith, hipStream_t stream);
size_t kimg_degrid_mfma_workspace_bytes(int P, int w_planes, int oversample, int kernel_width);
bool kimg_degrid_mfma_supported(int P, int w_planes, int oversample, int kernel_width);

namespace {

__device__ inline float2 cmul(float2 a, float2 b)
{
    return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}

// a * conj(b)
__device__ inline float2 cmul_conj(float2 a, float2 b)
{
    return make_float2(fmaf(a.x, b.x, a.y * b.y), fmaf(a.y, b.x, -a.x * b.y));
}

struct vis_coord {
    int u, v, sub_u, sub_v;
};

__device__ inline vis_coord load_uv(const int16_t *__restrict__ uv, int64_t i)
{
    const int2 packed = reinterpret_cast<const int2 *>(uv)[i];
    vis_coord c;
    c.u = (short) (packed.x & 0xffff);
    c.v = (short) (packed.x >> 16);
    c.sub_u = (short) (packed.y & 0xffff);
    c.sub_v = (short) (packed.y >> 16);
    return c;
}

// One wave per visibility; lanes sweep the K x K footprint (lane%32 along u).
template <int P>
__global__ __launch_bounds__(256) void grid_generic_kernel(
    float *__restrict__ grid, int64_t row_stride, int64_t pol_stride, int Gg,
    const float *__restrict__ weights_grid, int64_t wg_row_stride, int64_t wg_pol_stride,
    const int16_t *__restrict__ uv, const int16_t *__restrict__ w_plane,
    const float2 *__restrict__ vis, int64_t num_vis,
    const float2 *__restrict__ kern, int oversample, int K)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t) blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t) gridDim.x * (blockDim.x >> 6);
    const int uv_bias = (K - 1) / 2 - Gg / 2;                   // grid.py:1038
    for (int64_t i = wave; i < num_vis; i += nwaves) {
        const vis_coord c = load_uv(uv, i);
        const int wp = w_plane[i];
        const int u0 = c.u - uv_bias, v0 = c.v - uv_bias;
        const int64_t wa = (int64_t) (c.v + Gg / 2) * wg_row_stride + (c.u + Gg / 2);
        float2 sample[P];
#pragma unroll
        for (int p = 0; p < P; p++) {
            const float wgt = weights_grid[wa + p * wg_pol_stride];
            const float2 s = vis[i * P + p];
            sample[p] = make_float2(s.x * wgt, s.y * wgt);
        }
        const float2 *kv = kern + ((int64_t) wp * oversample + c.sub_v) * K;
        const float2 *ku = kern + ((int64_t) wp * oversample + c.sub_u) * K;
        for (int k = lane & 31; k < K; k += 32) {
            const float2 wu = ku[k];
            for (int j = lane >> 5; j < K; j += 2) {
                float2 wgt = cmul(kv[j], wu);
                const int64_t a = 2 * ((int64_t) (v0 + j) * row_stride + (u0 + k));
#pragma unroll
                for (int p = 0; p < P; p++) {
                    const float2 upd = cmul_conj(sample[p], wgt);
                    atomicAdd(&grid[a + 2 * p * pol_stride], upd.x);
                    atomicAdd(&grid[a + 2 * p * pol_stride + 1], upd.y);
                }
            }
        }
    }
}

// One wave per visibility; lanes sweep the footprint, partial sums reduced by shuffles.
template <int P>
__global__ __launch_bounds__(256) void degrid_generic_kernel(
    const float2 *__restrict__ grid, int64_t row_stride, int64_t pol_stride, int Gg,
    const int16_t *__restrict__ uv, const int16_t *__restrict__ w_plane,
    const float *__restrict__ weights, float2 *__restrict__ vis, int64_t num_vis,
    const float2 *__restrict__ kern, int oversample, int K)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t) blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t) gridDim.x * (blockDim.x >> 6);
    const int uv_bias = (K - 1) / 2 - Gg / 2;                   // grid.py:1141
    for (int64_t i = wave; i < num_vis; i += nwaves) {
        const vis_coord c = load_uv(uv, i);
        const int wp = w_plane[i];
        const int u0 = c.u - uv_bias, v0 = c.v - uv_bias;
        const float2 *kv = kern + ((int64_t) wp * oversample + c.sub_v) * K;
        const float2 *ku = kern + ((int64_t) wp * oversample + c.sub_u) * K;
        float2 acc[P];
#pragma unroll
        for (int p = 0; p < P; p++)
            acc[p] = make_float2(0.0f, 0.0f);
        for (int k = lane & 31; k < K; k += 32) {
            const float2 wu = ku[k];
            // column sum first: t[p] = sum_j kv[j] * grid[p][v0+j][u0+k]
            float2 t[P];
#pragma unroll
            for (int p = 0; p < P; p++)
                t[p] = make_float2(0.0f, 0.0f);
            for (int j = lane >> 5; j < K; j += 2) {
                const float2 wv = kv[j];
                const int64_t a = (int64_t) (v0 + j) * row_stride + (u0 + k);
#pragma unroll
                for (int p = 0; p < P; p++) {
                    const float2 g = grid[a + p * pol_stride];
                    t[p].x = fmaf(wv.x, g.x, fmaf(-wv.y, g.y, t[p].x));
                    t[p].y = fmaf(wv.x, g.y, fmaf(wv.y, g.x, t[p].y));
                }
            }
#pragma unroll
            for (int p = 0; p < P; p++) {
                acc[p].x = fmaf(wu.x, t[p].x, fmaf(-wu.y, t[p].y, acc[p].x));
                acc[p].y = fmaf(wu.x, t[p].y, fmaf(wu.y, t[p].x, acc[p].y));
            }
        }
#pragma unroll
        for (int p = 0; p < P; p++) {
            acc[p].x = wave_sum(acc[p].x);
            acc[p].y = wave_sum(acc[p].y);
        }
        if (lane < P) {
            float2 a = acc[0];
#pragma unroll
            for (int p = 1; p < P; p++)
                if (lane == p)
                    a = acc[p];
            const float wgt = weights[i * P + lane];
            float2 old = vis[i * P + lane];
            old.x -= wgt * a.x;                                 // grid.py:1154
            old.y -= wgt * a.y;
            vis[i * P + lane] = old;
        }
    }
}

// One thread per visibility; sources staged through LDS in chunks (predict.mako:38-74) as
// (l, m, n-1, flux[0]) records so that a source costs one broadcast 16-byte LDS read.
// Transcendental-bound: per (visibility, source) 3 FMA-class ops for the phase, v_fract (the
// range reduction: v_sin / v_cos take turns), v_sin + v_cos (quarter rate) and 2 FMA per
// polarization; 4 sources are in flight per thread to cover the transcendental latency.
template <int P>
__global__ __launch_bounds__(256) void predict_kernel(
    float2 *__restrict__ vis, const int16_t *__restrict__ uv, const int16_t *__restrict__ w_plane,
    const float *__restrict__ weights, const float *__restrict__ lmn,
    const float *__restrict__ flux, int64_t num_vis, int num_sources, int oversample,
    float uv_scale, float w_scale, float w_bias)
{
    constexpr int CHUNK = 256;
    __shared__ float4 src[CHUNK];                       // l, m, n-1, flux[0]
    __shared__ float sb[P > 1 ? P - 1 : 1][CHUNK];      // flux[1..P-1]
    const int64_t gid = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = gid < num_vis;
    float u = 0, v = 0, w = 0;
    if (live) {
        const vis_coord c = load_uv(uv, gid);
        u = ((float) (c.u * oversample + c.sub_u) + 0.5f) * uv_scale;     // predict.py:428-430
        v = ((float) (c.v * oversample + c.sub_v) + 0.5f) * uv_scale;
        w = (float) w_plane[gid] * w_scale + w_bias;
    }
    float2 acc[P];
#pragma unroll
    for (int p = 0; p < P; p++)
        acc[p] = make_float2(0.0f, 0.0f);
    for (int start = 0; start < num_sources; start += CHUNK) {
        const int batch = min(CHUNK, num_sources - start);
        __syncthreads();
        {
            // pad the chunk to a multiple of 4 with zero-flux sources
            const int s = start + threadIdx.x;
            const bool real = (int) threadIdx.x < batch;
            src[threadIdx.x] = real ? make_float4(lmn[3 * s], lmn[3 * s + 1], lmn[3 * s + 2],
                                                  flux[s * P])
                                    : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
            for (int p = 1; p < P; p++)
                sb[p - 1][threadIdx.x] = real ? flux[s * P + p] : 0.0f;
        }
        __syncthreads();
        const int padded = (batch + 3) & ~3;
        for (int s0 = 0; s0 < padded; s0 += 4) {
            float cs[4], sn[4];
            float4 rec[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                rec[i] = src[s0 + i];
                float phase = rec[i].x * u + rec[i].y * v + rec[i].z * w;     // turns
                phase = __builtin_amdgcn_fractf(phase);
                // e^{-2 pi i phase}; v_sin / v_cos take their argument in turns
                cs[i] = __builtin_amdgcn_cosf(phase);
                sn[i] = __builtin_amdgcn_sinf(phase);
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                acc[0].x = fmaf(cs[i], rec[i].w, acc[0].x);
                acc[0].y = fmaf(-sn[i], rec[i].w, acc[0].y);
#pragma unroll
                for (int p = 1; p < P; p++) {
                    const float f = sb[p - 1][s0 + i];
                    acc[p].x = fmaf(cs[i], f, acc[p].x);
                    acc[p].y = fmaf(-sn[i], f, acc[p].y);
                }
            }
        }
    }
    if (!live)
        return;
#pragma unroll
    for (int p = 0; p < P; p++) {
        const float wgt = weights[gid * P + p];
        float2 old = vis[gid * P + p];
        old.x -= acc[p].x * wgt;
        old.y -= acc[p].y * wgt;
        vis[gid * P + p] = old;
    }
}

int check_grid_args(int grid_size, int P, int64_t num_vis, int w_planes, int oversample, int K)
{
    if (grid_size <= 0 || grid_size % 2 || num_vis < 0 || w_planes <= 0 || oversample <= 0
        || K <= 0 || K > grid_size)
        return KIMG_EINVAL;
    if (P < 1 || P > 4)
        return KIMG_EUNSUPPORTED;
    return 0;
}

} // namespace

extern "C" size_t kimg_grid_workspace_bytes(int64_t max_vis, int num_polarizations, int w_planes,
                                           int oversample, int kernel_width)
{
    (void) max_vis;
    return kimg_grid_mfma_workspace_bytes(num_polarizations, w_planes, oversample, kernel_width);
}

extern "C" int kimg_grid(void *grid, int64_t grid_row_stride, int64_t grid_pol_stride,
                         int grid_size, int num_polarizations, const float *weights_grid,
                         int64_t wg_row_stride, int64_t wg_pol_stride, const int16_t *uv,
                         const int16_t *w_plane, const void *vis, int64_t num_vis,
                         const void *convolve_kernel, int w_planes, int oversample,
                         int kernel_width, void *workspace, size_t workspace_bytes, int variant,
                         int arith, void *stream)
{
    KIMG_CHECK_ARG(grid && weights_grid && uv && w_plane && vis && convolve_kernel);
    KIMG_CHECK_ARG(arith == KIMG_ARITH_FP32 || arith == KIMG_ARITH_SPLIT_FP16);
    KIMG_CHECK_ARG(variant >= 0 && (variant >> 8) <= 256);
    const kimg_window_cus_scope cus(variant >> 8);      // KIMG_WINDOW_CUS(n): this call's share of the CUs
    variant &= 0xff;
    KIMG_CHECK_ARG(variant == KIMG_VARIANT_AUTO || variant == KIMG_VARIANT_GENERIC
                   || variant == KIMG_VARIANT_MFMA || variant == KIMG_VARIANT_BINNED);
    int rc = check_grid_args(grid_size, num_polarizations, num_vis, w_planes, oversample,
                             kernel_width);
    if (rc)
        return rc;
    if (num_vis == 0)
        return 0;                                               // grid.py:810-811
    hipStream_t s = (hipStream_t) stream;
    // (the window kernel packs first-tap coordinates into 16 bits and addresses a polarization's
    // plane with 32-bit byte offsets)
    const bool mfma_ok = grid_size <= 32000
                         && (int64_t) grid_size * grid_row_stride * 8 < ((int64_t) 1 << 32)
                         && kimg_grid_mfma_supported(num_polarizations, w_planes, oversample,
                                                     kernel_width);
    if ((variant == KIMG_VARIANT_MFMA || variant == KIMG_VARIANT_BINNED) && !mfma_ok)
        return KIMG_EUNSUPPORTED;
    if (variant == KIMG_VARIANT_BINNED)
        return kimg_grid_binned(grid, grid_row_stride, grid_pol_stride, grid_size,
                                num_polarizations, weights_grid, wg_row_stride, wg_pol_stride, uv,
                                w_plane, vis, num_vis, convolve_kernel, w_planes, oversample,
                                kernel_width, workspace, workspace_bytes, arith, s);
    if (variant == KIMG_VARIANT_MFMA || (variant == KIMG_VARIANT_AUTO && mfma_ok))
        return kimg_grid_mfma(grid, grid_row_stride, grid_pol_stride, grid_size,
                              num_polarizations, weights_grid, wg_row_stride, wg_pol_stride, uv,
                              w_plane, vis, num_vis, convolve_kernel, w_planes, oversample,
                              kernel_width, workspace, workspace_bytes, arith, s);
    int blocks = kimg_divup(num_vis, 4);
    if (blocks > 8192)
        blocks = 8192;
#define LAUNCH(P) grid_generic_kernel<P><<<blocks, 256, 0, s>>>( \
        (float *) grid, grid_row_stride, grid_pol_stride, grid_size, weights_grid, wg_row_stride, \
        wg_pol_stride, uv, w_plane, (const float2 *) vis, num_vis, \
        (const float2 *) convolve_kernel, oversample, kernel_width)
    switch (num_polarizations) {
    case 1: LAUNCH(1); break;
    case 2: LAUNCH(2); break;
    case 3: LAUNCH(3); break;
    case 4: LAUNCH(4); break;
    }
#undef LAUNCH
    return kimg_launch_status();
}

extern "C" size_t kimg_degrid_workspace_bytes(int num_polarizations, int w_planes, int oversample,
                                             int kernel_width)
{
    return kimg_degrid_mfma_workspace_bytes(num_polarizations, w_planes, oversample, kernel_width);
}

extern "C" int kimg_degrid(const void *grid, int64_t grid_row_stride, int64_t grid_pol_stride,
                           int grid_size, int num_polarizations, const int16_t *uv,
                           const int16_t *w_plane, const float *weights, void *vis,
                           int64_t num_vis, const void *convolve_kernel, int w_planes,
                           int oversample, int kernel_width, void *workspace,
                           size_t workspace_bytes, int variant, int arith, void *stream)
{
    KIMG_CHECK_ARG(grid && uv && w_plane && weights && vis && convolve_kernel);
    KIMG_CHECK_ARG(arith == KIMG_ARITH_FP32 || arith == KIMG_ARITH_SPLIT_FP16);
    KIMG_CHECK_ARG(variant >= 0 && (variant >> 8) <= 256);
    const kimg_window_cus_scope cus(variant >> 8);
    variant &= 0xff;
    KIMG_CHECK_ARG(variant == KIMG_VARIANT_AUTO || variant == KIMG_VARIANT_GENERIC
                   || variant == KIMG_VARIANT_MFMA || variant == KIMG_VARIANT_BINNED);
    int rc = check_grid_args(grid_size, num_polarizations, num_vis, w_planes, oversample,
                             kernel_width);
    if (rc)
        return rc;
    if (num_vis == 0)
        return 0;                                               // grid.py:989-990
    hipStream_t s = (hipStream_t) stream;
    const bool mfma_ok = kimg_degrid_mfma_supported(num_polarizations, w_planes, oversample,
                                                    kernel_width);
    if ((variant == KIMG_VARIANT_MFMA || variant == KIMG_VARIANT_BINNED) && !mfma_ok)
        return KIMG_EUNSUPPORTED;
    if (variant == KIMG_VARIANT_BINNED)
        return kimg_degrid_binned(grid, grid_row_stride, grid_pol_stride, grid_size,
                                  num_polarizations, uv, w_plane, weights, vis, num_vis,
                                  convolve_kernel, w_planes, oversample, kernel_width, workspace,
                                  workspace_bytes, arith, s);
    if (variant != KIMG_VARIANT_GENERIC && mfma_ok)
        return kimg_degrid_mfma(grid, grid_row_stride, grid_pol_stride, grid_size,
                                num_polarizations, uv, w_plane, weights, vis, num_vis,
                                convolve_kernel, w_planes, oversample, kernel_width, workspace,
                                workspace_bytes, arith, s);
    int blocks = kimg_divup(num_vis, 4);
    if (blocks > 16384)
        blocks = 16384;
#define LAUNCH(P) degrid_generic_kernel<P><<<blocks, 256, 0, s>>>( \
        (const float2 *) grid, grid_row_stride, grid_pol_stride, grid_size, uv, w_plane, weights, \
        (float2 *) vis, num_vis, (const float2 *) convolve_kernel, oversample, kernel_width)
    switch (num_polarizations) {
    case 1: LAUNCH(1); break;
    case 2: LAUNCH(2); break;
    case 3: LAUNCH(3); break;
    case 4: LAUNCH(4); break;
    }
#undef LAUNCH
    return kimg_launch_status();
}

extern "C" int kimg_predict(void *vis, const int16_t *uv, const int16_t *w_plane,
                            const float *weights, const float *lmn, const float *flux,
                            int64_t num_vis, int num_sources, int num_polarizations,
                            int oversample, float uv_scale, float w_scale, float w_bias,
                            void *stream)
{
    KIMG_CHECK_ARG(vis && uv && w_plane && weights && num_vis >= 0 && num_sources >= 0);
    if (num_polarizations < 1 || num_polarizations > 4)
        return KIMG_EUNSUPPORTED;
    if (num_vis == 0 || num_sources == 0)
        return 0;                                               // predict.py:387-388
    KIMG_CHECK_ARG(lmn && flux);
    hipStream_t s = (hipStream_t) stream;
    const int blocks = kimg_divup(num_vis, 256);
#define LAUNCH(P) predict_kernel<P><<<blocks, 256, 0, s>>>( \
        (float2 *) vis, uv, w_plane, weights, lmn, flux, num_vis, num_sources, oversample, \
        uv_scale, w_scale, w_bias)
    switch (num_polarizations) {
    case 1: LAUNCH(1); break;
    case 2: LAUNCH(2); break;
    case 3: LAUNCH(3); break;
    case 4: LAUNCH(4); break;
    }
#undef LAUNCH
    return kimg_launch_status();
}

// (kimg_preload, api.hip)
KIMG_PRELOAD_THIS_UNIT(grid_generic_kernel<1>)
