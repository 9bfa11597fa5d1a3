// Imaging-weight kernels: un-convolved scatter of statistical weights, mean weight,
// density weights and fill.  Mirrors weight.py:155-176, 261-284, 357-376 of the reference
// (kernels grid_weights.mako, density_weights.mako, mean_weight.mako); HBM-bound streams.
#include "kimg_common.h"

namespace {

// One lane per visibility.  Consecutive visibilities of a track fall into the same cell many
// times over (a same-address float atomic per visibility serialises in L2), so each wave first
// sums its runs of equal (u, v) with a segmented scan and only the last lane of a run issues
// the atomics.
template <int P>
__global__ __launch_bounds__(256) void grid_weights_kernel(
    float *__restrict__ grid, int64_t row_stride, int64_t pol_stride, int half_u, int half_v,
    const int16_t *__restrict__ uv, const float *__restrict__ weights, int64_t num_vis)
{
    const int64_t gid = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool live = gid < num_vis;
    // first two of the four int16 are (u, v): one 4-byte load
    const int packed = live ? reinterpret_cast<const int *>(uv)[2 * gid] : 0;
    float w[P];
#pragma unroll
    for (int p = 0; p < P; p++)
        w[p] = live ? weights[gid * P + p] : 0.0f;
    const int prev = __shfl_up(packed, 1, WAVE);
    // a dead lane continues its predecessor's run with zero weight
    int head = lane == 0 || (live && packed != prev);
    const int next_head = __shfl_down(head, 1, WAVE);
    const bool tail = lane == 63 || next_head;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const int f_up = __shfl_up(head, d, WAVE);
        float w_up[P];
#pragma unroll
        for (int p = 0; p < P; p++)
            w_up[p] = __shfl_up(w[p], d, WAVE);
        if (lane >= d && !head) {
#pragma unroll
            for (int p = 0; p < P; p++)
                w[p] += w_up[p];
            head = f_up;
        }
    }
    // the run's key: dead lanes inherit it from the run they extend
    const unsigned long long heads = __ballot(lane == 0 || (live && packed != prev));
    const int run_head = 63 - __builtin_clzll(heads & (~0ull >> (63 - lane)));
    const int key = __shfl(packed, run_head, WAVE);
    const bool any_live = __shfl((int) live, run_head, WAVE);
    if (!tail || !any_live)
        return;
    const int u = (short) (key & 0xffff);
    const int v = (short) (key >> 16);
    const int64_t addr = (int64_t) (v + half_v) * row_stride + (u + half_u);
#pragma unroll
    for (int p = 0; p < P; p++)
        atomicAdd(&grid[addr + p * pol_stride], w[p]);
}

// Block-wide sum of NS doubles; lane 0 of wave 0 issues the global atomics.
template <int NS>
__device__ inline void block_accumulate(double (&v)[NS], double *__restrict__ sums)
{
    __shared__ double scratch[NS][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NS; i++) {
        v[i] = wave_sum(v[i]);
        if (lane == 0)
            scratch[i][wv] = v[i];
    }
    __syncthreads();
    if (threadIdx.x < NS) {
        double t = 0;
        for (int w = 0; w < (int) (blockDim.x >> 6); w++)
            t += scratch[threadIdx.x][w];
        atomicAdd(&sums[threadIdx.x], t);
    }
}

__global__ __launch_bounds__(256) void mean_weight_kernel(
    double *__restrict__ sums, const float *__restrict__ grid, int64_t row_stride,
    int width, int height)
{
    double acc[2] = {0, 0};
    // (four rows per round, their loads issued together: with few, long-lived workgroups -- see
    // image_grid -- a thread's loop would otherwise be one memory round trip per row)
    constexpr int ROWS = 4;
    for (int y0 = blockIdx.y; y0 < height; y0 += gridDim.y * ROWS)
        for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < width; x += gridDim.x * blockDim.x) {
            float w[ROWS];
#pragma unroll
            for (int r = 0; r < ROWS; r++) {
                const int y = y0 + r * gridDim.y;
                w[r] = y < height ? grid[(int64_t) y * row_stride + x] : 0.0f;
            }
#pragma unroll
            for (int r = 0; r < ROWS; r++) {
                acc[0] += w[r];
                acc[1] += (double) w[r] * w[r];
            }
        }
    block_accumulate<2>(acc, sums);
}

// mean_sums (optional): (sum W, sum W^2) as kimg_mean_weight leaves them; a is then robust / (their
// quotient), worked out as the host would (weight.py:529: doubles, rounded to the float32 it passes)
__global__ __launch_bounds__(256) void density_weights_kernel(
    double *__restrict__ sums, float *__restrict__ grid, int64_t row_stride, int64_t pol_stride,
    int width, int height, int num_pols, float a, float b, const double *__restrict__ mean_sums,
    double robust)
{
    if (mean_sums) {
        const double mean_weight = mean_sums[1] / mean_sums[0];
        a = (float) (robust / mean_weight);
    }
    double acc[3] = {0, 0, 0};
    constexpr int ROWS = 4;
    for (int y0 = blockIdx.y; y0 < height; y0 += gridDim.y * ROWS)
        for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < width; x += gridDim.x * blockDim.x)
            for (int p = 0; p < num_pols; p++) {
                float w[ROWS];
#pragma unroll
                for (int r = 0; r < ROWS; r++) {
                    const int y = y0 + r * gridDim.y;
                    w[r] = y < height ? grid[(int64_t) y * row_stride + x + p * pol_stride] : 0.0f;
                }
#pragma unroll
                for (int r = 0; r < ROWS; r++) {
                    const int y = y0 + r * gridDim.y;
                    if (y >= height)
                        continue;
                    // weight.py:596-597: 1 / (w*S2 + 1), zero where no visibilities fell
                    const float d = (w[r] != 0.0f) ? 1.0f / (a * w[r] + b) : 0.0f;
                    if (p == 0) {
                        const double dw = (double) d * w[r];
                        acc[0] += w[r];
                        acc[1] += dw;
                        acc[2] += d * dw;
                    }
                    grid[(int64_t) y * row_stride + x + p * pol_stride] = d;
                }
            }
    block_accumulate<3>(acc, sums);
}

__global__ __launch_bounds__(256) void fill_kernel(float *__restrict__ data, int64_t count, float value)
{
    const int64_t vec_count = count & ~(int64_t) 3;
    const int64_t stride = (int64_t) gridDim.x * blockDim.x * 4;
    for (int64_t i = ((int64_t) blockIdx.x * blockDim.x + threadIdx.x) * 4; i < vec_count; i += stride)
        *reinterpret_cast<float4 *>(data + i) = make_float4(value, value, value, value);
    if (blockIdx.x == 0 && threadIdx.x < (count & 3))
        data[vec_count + threadIdx.x] = value;
}

} // namespace

extern "C" int kimg_grid_weights(float *grid, int64_t row_stride, int64_t pol_stride, int width,
                                 int height, int num_polarizations, const int16_t *uv,
                                 const float *weights, int64_t num_vis, void *stream)
{
    KIMG_CHECK_ARG(grid && uv && weights && num_vis >= 0 && width > 0 && height > 0);
    KIMG_CHECK_ARG(width % 2 == 0 && height % 2 == 0);     // weight.py:131-132
    if (num_vis == 0)
        return 0;
    hipStream_t s = (hipStream_t) stream;
    dim3 grid_dim(kimg_divup(num_vis, 256)), block(256);
#define LAUNCH(P) grid_weights_kernel<P><<<grid_dim, block, 0, s>>>( \
        grid, row_stride, pol_stride, width / 2, height / 2, uv, weights, num_vis)
    switch (num_polarizations) {
    case 1: LAUNCH(1); break;
    case 2: LAUNCH(2); break;
    case 3: LAUNCH(3); break;
    case 4: LAUNCH(4); break;
    default: return KIMG_EUNSUPPORTED;
    }
#undef LAUNCH
    return kimg_launch_status();
}

static dim3 image_grid(int width, int height)
{
    // Four workgroups per CU, grid-stride the rest: every workgroup ends with two or three double
    // atomics on the SAME addresses, which the memory system serves one after the other (~30 ns each:
    // with 4096 workgroups that chain was longer than the pass over the grid)
    int bx = kimg_divup(width, 256);
    int by = height < 1024 / bx ? height : 1024 / bx;
    return dim3(bx, by > 0 ? by : 1);
}

extern "C" int kimg_mean_weight(double *sums, const float *grid, int64_t row_stride, int width,
                                int height, void *stream)
{
    KIMG_CHECK_ARG(sums && grid && width > 0 && height > 0);
    hipStream_t s = (hipStream_t) stream;
    KIMG_HIP(hipMemsetAsync(sums, 0, 2 * sizeof(double), s));
    mean_weight_kernel<<<image_grid(width, height), 256, 0, s>>>(sums, grid, row_stride, width, height);
    return kimg_launch_status();
}

extern "C" int kimg_density_weights(double *sums, float *grid, int64_t row_stride,
                                    int64_t pol_stride, int width, int height,
                                    int num_polarizations, float a, float b, void *stream)
{
    KIMG_CHECK_ARG(sums && grid && width > 0 && height > 0 && num_polarizations > 0);
    hipStream_t s = (hipStream_t) stream;
    KIMG_HIP(hipMemsetAsync(sums, 0, 3 * sizeof(double), s));
    density_weights_kernel<<<image_grid(width, height), 256, 0, s>>>(
        sums, grid, row_stride, pol_stride, width, height, num_polarizations, a, b, nullptr, 0.0);
    return kimg_launch_status();
}

extern "C" int kimg_density_weights_robust(double *sums, float *grid, int64_t row_stride,
                                           int64_t pol_stride, int width, int height,
                                           int num_polarizations, const double *mean_sums,
                                           double robust, float b, void *stream)
{
    KIMG_CHECK_ARG(sums && grid && mean_sums && width > 0 && height > 0 && num_polarizations > 0);
    hipStream_t s = (hipStream_t) stream;
    KIMG_HIP(hipMemsetAsync(sums, 0, 3 * sizeof(double), s));
    density_weights_kernel<<<image_grid(width, height), 256, 0, s>>>(
        sums, grid, row_stride, pol_stride, width, height, num_polarizations, 0.0f, b, mean_sums, robust);
    return kimg_launch_status();
}

extern "C" int kimg_fill(float *data, int64_t count, float value, void *stream)
{
    KIMG_CHECK_ARG(data && count >= 0);
    if (count == 0)
        return 0;
    KIMG_CHECK_ARG((reinterpret_cast<uintptr_t>(data) & 15) == 0);
    int blocks = kimg_divup(count, 1024);
    if (blocks > 2048)
        blocks = 2048;
    fill_kernel<<<blocks, 256, 0, (hipStream_t) stream>>>(data, count, value);
    return kimg_launch_status();
}

// (kimg_preload, api.hip)
KIMG_PRELOAD_THIS_UNIT(mean_weight_kernel)
