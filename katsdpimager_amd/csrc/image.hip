// Image-plane kernels: grid<->layer quadrant copies, layer<->image (fftshift + W-stack phase
// + n-term + taper), scale, add_image, apply_primary_beam.  All HBM-bound streams.
// Mirrors image.py:153-180, 351-367, 439-458, 539-558, 649-673, 716-740 of the reference.
// The arithmetic order follows the reference HOST classes (image.py:781-799, 836-848) so that
// float32 results agree to rounding; this file is built with -ffp-contract=off.
#include "kimg_common.h"

namespace {

struct c64 { float re, im; };

// layer[ly][lx] = grid cell with the same (centred) frequency, or 0 outside the grid.
// Fuses the reference's layer.zero() + 4 copy_region calls (image.py:660-671).
__global__ __launch_bounds__(256) void grid_to_layer_kernel(
    float2 *__restrict__ layer, int G, const float2 *__restrict__ grid, int64_t grid_row_stride,
    int Gg)
{
    const int lx = blockIdx.x * blockDim.x + threadIdx.x;
    const int ly = blockIdx.y;
    if (lx >= G)
        return;
    const int half = Gg / 2;
    // centred coordinates of this layer pixel: 0..G/2-1 positive, G/2.. negative
    const int cx = lx < G - half ? lx : lx - G;
    const int cy = ly < G - half ? ly : ly - G;
    float2 v = make_float2(0.0f, 0.0f);
    if (cx >= -half && cx < half && cy >= -half && cy < half)
        v = grid[(int64_t) (cy + half) * grid_row_stride + (cx + half)];
    layer[(int64_t) ly * G + lx] = v;
}

// The same for a transform that is only wanted for its REAL part (the W-stack phase is 1 at
// w = 0): Re F^-1[g] = F^-1[h] with h(k) = (g(k) + conj g(-k)) / 2, and h is Hermitian, so half
// of it -- columns 0 .. G/2 -- feeds a complex-to-real transform of half the size (the
// "opportunity" noted at image.py:561-566 of the reference).  half[ly][lx], row length G/2 + 1.
__global__ __launch_bounds__(256) void grid_to_half_layer_kernel(
    float2 *__restrict__ half_layer, int G, const float2 *__restrict__ grid,
    int64_t grid_row_stride, int Gg)
{
    const int lx = blockIdx.x * blockDim.x + threadIdx.x;
    const int ly = blockIdx.y;
    const int W = G / 2 + 1;
    if (lx >= W)
        return;
    const int half = Gg / 2;
    const int cx = lx < G - half ? lx : lx - G;
    const int cy = ly < G - half ? ly : ly - G;
    float2 a = make_float2(0.0f, 0.0f), b = make_float2(0.0f, 0.0f);
    if (cx >= -half && cx < half && cy >= -half && cy < half)
        a = grid[(int64_t) (cy + half) * grid_row_stride + (cx + half)];
    // -k modulo G: the Nyquist row and column (a grid as large as the layer has them) are their
    // own mirrors
    const int mlx = lx ? G - lx : 0, mly = ly ? G - ly : 0;
    const int mx = mlx < G - half ? mlx : mlx - G;
    const int my = mly < G - half ? mly : mly - G;
    if (mx >= -half && mx < half && my >= -half && my < half)
        b = grid[(int64_t) (my + half) * grid_row_stride + (mx + half)];
    half_layer[(int64_t) ly * W + lx] = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y - b.y));
}

__global__ __launch_bounds__(256) void layer_to_grid_kernel(
    float2 *__restrict__ grid, int64_t grid_row_stride, int Gg, const float2 *__restrict__ layer,
    int G)
{
    const int gx = blockIdx.x * blockDim.x + threadIdx.x;
    const int gy = blockIdx.y;
    if (gx >= Gg)
        return;
    const int half = Gg / 2;
    int lx = gx - half, ly = gy - half;
    if (lx < 0) lx += G;
    if (ly < 0) ly += G;
    grid[(int64_t) gy * grid_row_stride + gx] = layer[(int64_t) ly * G + lx];
}

// e^{2 pi i x} with the reference's range reduction (fast_math.py:14-15).
__device__ inline void expj2pi(float x, float &c, float &s)
{
    float r = x - rintf(x);
    sincospif(2.0f * r, &s, &c);
}

// n(l, m) following GridToImageHost.__call__ (image.py:785-790) operation by operation.
__device__ inline float lm_coord(int i, float lm_scale, float lm_bias)
{
    return (float) i * lm_scale + lm_bias;
}

// One thread: one image pixel pair... kept simple: one pixel per thread, x fastest.
// image[y][x] += Re(layer[(y+G/2)%G][(x+G/2)%G] * e^{2 pi i w (n-1)}) * n / (k[y] k[x])
__global__ __launch_bounds__(256) void layer_to_image_kernel(
    float *__restrict__ image, int64_t image_row_stride, const float2 *__restrict__ layer, int G,
    const float *__restrict__ kernel1d, float lm_scale, float lm_bias, float w)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= G)
        return;
    const int half = G / 2;
    const int sx = x < half ? x + half : x - half;
    const int sy = y < half ? y + half : y - half;
    const float2 v = layer[(int64_t) sy * G + sx];
    const float l = lm_coord(x, lm_scale, lm_bias);
    const float m = lm_coord(y, lm_scale, lm_bias);
    const float l2 = l * l, m2 = m * m;
    const float n = sqrtf(1.0f - (m2 + l2));
    float c, s;
    expj2pi(w * (n - 1.0f), c, s);
    const float rotated = v.x * c - v.y * s;
    const float taper = kernel1d[y] * kernel1d[x];
    image[(int64_t) y * image_row_stride + x] += (rotated * n) / taper;
}

// layer_to_image for w = 0 from the real output of the complex-to-real transform (rows of
// `layer_row_stride` floats): the phase factor is exactly (1, 0), so "rotated" is the real part.
__global__ __launch_bounds__(256) void real_layer_to_image_kernel(
    float *__restrict__ image, int64_t image_row_stride, const float *__restrict__ layer,
    int64_t layer_row_stride, int G, const float *__restrict__ kernel1d, float lm_scale, float lm_bias)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= G)
        return;
    const int half = G / 2;
    const int sx = x < half ? x + half : x - half;
    const int sy = y < half ? y + half : y - half;
    const float rotated = layer[(int64_t) sy * layer_row_stride + sx];
    const float l = lm_coord(x, lm_scale, lm_bias);
    const float m = lm_coord(y, lm_scale, lm_bias);
    const float l2 = l * l, m2 = m * m;
    const float n = sqrtf(1.0f - (m2 + l2));
    const float taper = kernel1d[y] * kernel1d[x];
    image[(int64_t) y * image_row_stride + x] += (rotated * n) / taper;
}

// layer[(y+G/2)%G][(x+G/2)%G] = image[y][x] / (k[y] k[x] n) * e^{-2 pi i w (n-1)}
__global__ __launch_bounds__(256) void image_to_layer_kernel(
    float2 *__restrict__ layer, const float *__restrict__ image, int64_t image_row_stride, int G,
    const float *__restrict__ kernel1d, float lm_scale, float lm_bias, float w)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= G)
        return;
    const int half = G / 2;
    const int sx = x < half ? x + half : x - half;
    const int sy = y < half ? y + half : y - half;
    const float l = lm_coord(x, lm_scale, lm_bias);
    const float m = lm_coord(y, lm_scale, lm_bias);
    const float l2 = l * l, m2 = m * m;
    const float n = sqrtf(1.0f - (m2 + l2));
    float c, s;
    expj2pi(-w * (n - 1.0f), c, s);
    const float taper = kernel1d[y] * kernel1d[x];
    const float v = image[(int64_t) y * image_row_stride + x] / (taper * n);
    layer[(int64_t) sy * G + sx] = make_float2(v * c, v * s);
}

// image_to_layer for w = 0: the layer is real (phase factor exactly (1, 0)); rows of
// `layer_row_stride` floats, ready for an in-place real-to-complex transform.
__global__ __launch_bounds__(256) void image_to_real_layer_kernel(
    float *__restrict__ layer, int64_t layer_row_stride, const float *__restrict__ image,
    int64_t image_row_stride, int G, const float *__restrict__ kernel1d, float lm_scale, float lm_bias)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= G)
        return;
    const int half = G / 2;
    const int sx = x < half ? x + half : x - half;
    const int sy = y < half ? y + half : y - half;
    const float l = lm_coord(x, lm_scale, lm_bias);
    const float m = lm_coord(y, lm_scale, lm_bias);
    const float l2 = l * l, m2 = m * m;
    const float n = sqrtf(1.0f - (m2 + l2));
    const float taper = kernel1d[y] * kernel1d[x];
    layer[(int64_t) sy * layer_row_stride + sx] = image[(int64_t) y * image_row_stride + x] / (taper * n);
}

// layer_to_grid from the half spectrum of a real layer: F(-k) = conj F(k).
__global__ __launch_bounds__(256) void half_layer_to_grid_kernel(
    float2 *__restrict__ grid, int64_t grid_row_stride, int Gg, const float2 *__restrict__ half_layer,
    int G)
{
    const int gx = blockIdx.x * blockDim.x + threadIdx.x;
    const int gy = blockIdx.y;
    if (gx >= Gg)
        return;
    const int half = Gg / 2, W = G / 2 + 1;
    int lx = gx - half, ly = gy - half;
    if (lx < 0) lx += G;
    if (ly < 0) ly += G;
    float2 v;
    if (lx < W) {
        v = half_layer[(int64_t) ly * W + lx];
    } else {
        v = half_layer[(int64_t) (ly ? G - ly : 0) * W + (G - lx)];
        v.y = -v.y;
    }
    grid[(int64_t) gy * grid_row_stride + gx] = v;
}

struct scale_t { float v[4]; };

__global__ __launch_bounds__(256) void scale_kernel(
    float *__restrict__ image, int64_t row_stride, int64_t pol_stride, int width, int num_pols,
    scale_t scale)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= width)
        return;
    int64_t addr = (int64_t) blockIdx.y * row_stride + x;
    for (int p = 0; p < num_pols; p++, addr += pol_stride)
        image[addr] *= scale.v[p];
}

__global__ __launch_bounds__(256) void add_image_kernel(
    float *__restrict__ dest, int64_t dest_row_stride, int64_t dest_pol_stride,
    const float *__restrict__ src, int64_t src_row_stride, int64_t src_pol_stride,
    int width, int num_pols)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= width)
        return;
    int64_t d = (int64_t) blockIdx.y * dest_row_stride + x;
    int64_t s = (int64_t) blockIdx.y * src_row_stride + x;
    for (int p = 0; p < num_pols; p++, d += dest_pol_stride, s += src_pol_stride)
        dest[d] += src[s];
}

__global__ __launch_bounds__(256) void apply_primary_beam_kernel(
    float *__restrict__ image, int64_t row_stride, int64_t pol_stride,
    const float *__restrict__ beam_power, int64_t beam_row_stride, int width, int num_pols,
    float threshold, float replacement)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= width)
        return;
    const float beam = beam_power[(int64_t) blockIdx.y * beam_row_stride + x];
    int64_t addr = (int64_t) blockIdx.y * row_stride + x;
    for (int p = 0; p < num_pols; p++, addr += pol_stride)
        image[addr] = beam < threshold ? replacement : image[addr] / beam;
}

} // namespace

extern "C" int kimg_grid_to_layer(void *layer, int layer_size, const void *grid,
                                  int64_t grid_row_stride, int grid_size, void *stream)
{
    KIMG_CHECK_ARG(layer && grid && layer_size > 0 && grid_size > 0 && grid_size <= layer_size);
    KIMG_CHECK_ARG(layer_size % 2 == 0 && grid_size % 2 == 0);      // image.py:655-656
    dim3 g(kimg_divup(layer_size, 256), layer_size);
    grid_to_layer_kernel<<<g, 256, 0, (hipStream_t) stream>>>(
        (float2 *) layer, layer_size, (const float2 *) grid, grid_row_stride, grid_size);
    return kimg_launch_status();
}

extern "C" int kimg_grid_to_half_layer(void *half_layer, int layer_size, const void *grid,
                                       int64_t grid_row_stride, int grid_size, void *stream)
{
    KIMG_CHECK_ARG(half_layer && grid && layer_size > 0 && layer_size % 2 == 0 && grid_size > 0
                   && grid_size % 2 == 0 && grid_size <= layer_size && grid_row_stride >= grid_size);
    const dim3 blocks(kimg_divup(layer_size / 2 + 1, 256), layer_size);
    grid_to_half_layer_kernel<<<blocks, 256, 0, (hipStream_t) stream>>>(
        static_cast<float2 *>(half_layer), layer_size, static_cast<const float2 *>(grid),
        grid_row_stride, grid_size);
    return kimg_launch_status();
}

extern "C" int kimg_real_layer_to_image(float *image, int64_t image_row_stride, const float *layer,
                                        int64_t layer_row_stride, int size, const float *kernel1d,
                                        float lm_scale, float lm_bias, void *stream)
{
    KIMG_CHECK_ARG(image && layer && kernel1d && size > 0 && size % 2 == 0
                   && image_row_stride >= size && layer_row_stride >= size);
    const dim3 blocks(kimg_divup(size, 256), size);
    real_layer_to_image_kernel<<<blocks, 256, 0, (hipStream_t) stream>>>(
        image, image_row_stride, layer, layer_row_stride, size, kernel1d, lm_scale, lm_bias);
    return kimg_launch_status();
}

extern "C" int kimg_image_to_real_layer(float *layer, int64_t layer_row_stride, const float *image,
                                        int64_t image_row_stride, int size, const float *kernel1d,
                                        float lm_scale, float lm_bias, void *stream)
{
    KIMG_CHECK_ARG(image && layer && kernel1d && size > 0 && size % 2 == 0
                   && image_row_stride >= size && layer_row_stride >= size);
    const dim3 blocks(kimg_divup(size, 256), size);
    image_to_real_layer_kernel<<<blocks, 256, 0, (hipStream_t) stream>>>(
        layer, layer_row_stride, image, image_row_stride, size, kernel1d, lm_scale, lm_bias);
    return kimg_launch_status();
}

extern "C" int kimg_half_layer_to_grid(void *grid, int64_t grid_row_stride, int grid_size,
                                       const void *half_layer, int layer_size, void *stream)
{
    KIMG_CHECK_ARG(half_layer && grid && layer_size > 0 && layer_size % 2 == 0 && grid_size > 0
                   && grid_size % 2 == 0 && grid_size <= layer_size && grid_row_stride >= grid_size);
    const dim3 blocks(kimg_divup(grid_size, 256), grid_size);
    half_layer_to_grid_kernel<<<blocks, 256, 0, (hipStream_t) stream>>>(
        static_cast<float2 *>(grid), grid_row_stride, grid_size,
        static_cast<const float2 *>(half_layer), layer_size);
    return kimg_launch_status();
}

extern "C" int kimg_layer_to_grid(void *grid, int64_t grid_row_stride, int grid_size,
                                  const void *layer, int layer_size, void *stream)
{
    KIMG_CHECK_ARG(layer && grid && layer_size > 0 && grid_size > 0 && grid_size <= layer_size);
    KIMG_CHECK_ARG(layer_size % 2 == 0 && grid_size % 2 == 0);
    dim3 g(kimg_divup(grid_size, 256), grid_size);
    layer_to_grid_kernel<<<g, 256, 0, (hipStream_t) stream>>>(
        (float2 *) grid, grid_row_stride, grid_size, (const float2 *) layer, layer_size);
    return kimg_launch_status();
}

extern "C" int kimg_layer_to_image(float *image, int64_t image_row_stride, const void *layer,
                                   int size, const float *kernel1d, float lm_scale,
                                   float lm_bias, float w, void *stream)
{
    KIMG_CHECK_ARG(image && layer && kernel1d && size > 0 && size % 2 == 0);   // image.py:127-128
    dim3 g(kimg_divup(size, 256), size);
    layer_to_image_kernel<<<g, 256, 0, (hipStream_t) stream>>>(
        image, image_row_stride, (const float2 *) layer, size, kernel1d, lm_scale, lm_bias, w);
    return kimg_launch_status();
}

extern "C" int kimg_image_to_layer(void *layer, const float *image, int64_t image_row_stride,
                                   int size, const float *kernel1d, float lm_scale,
                                   float lm_bias, float w, void *stream)
{
    KIMG_CHECK_ARG(image && layer && kernel1d && size > 0 && size % 2 == 0);
    dim3 g(kimg_divup(size, 256), size);
    image_to_layer_kernel<<<g, 256, 0, (hipStream_t) stream>>>(
        (float2 *) layer, image, image_row_stride, size, kernel1d, lm_scale, lm_bias, w);
    return kimg_launch_status();
}

extern "C" int kimg_scale(float *image, int64_t row_stride, int64_t pol_stride, int width,
                          int height, int num_polarizations, const float *scale_host, void *stream)
{
    KIMG_CHECK_ARG(image && scale_host && width > 0 && height > 0);
    if (num_polarizations < 1 || num_polarizations > 4)
        return KIMG_EUNSUPPORTED;
    scale_t sc = {};
    for (int p = 0; p < num_polarizations; p++)
        sc.v[p] = scale_host[p];
    dim3 g(kimg_divup(width, 256), height);
    scale_kernel<<<g, 256, 0, (hipStream_t) stream>>>(image, row_stride, pol_stride, width,
                                                      num_polarizations, sc);
    return kimg_launch_status();
}

extern "C" int kimg_add_image(float *dest, int64_t dest_row_stride, int64_t dest_pol_stride,
                              const float *src, int64_t src_row_stride, int64_t src_pol_stride,
                              int width, int height, int num_polarizations, void *stream)
{
    KIMG_CHECK_ARG(dest && src && width > 0 && height > 0 && num_polarizations > 0);
    dim3 g(kimg_divup(width, 256), height);
    add_image_kernel<<<g, 256, 0, (hipStream_t) stream>>>(
        dest, dest_row_stride, dest_pol_stride, src, src_row_stride, src_pol_stride, width,
        num_polarizations);
    return kimg_launch_status();
}

extern "C" int kimg_apply_primary_beam(float *image, int64_t row_stride, int64_t pol_stride,
                                       const float *beam_power, int64_t beam_row_stride,
                                       int width, int height, int num_polarizations,
                                       float threshold, float replacement, void *stream)
{
    KIMG_CHECK_ARG(image && beam_power && width > 0 && height > 0 && num_polarizations > 0);
    dim3 g(kimg_divup(width, 256), height);
    apply_primary_beam_kernel<<<g, 256, 0, (hipStream_t) stream>>>(
        image, row_stride, pol_stride, beam_power, beam_row_stride, width, num_polarizations,
        threshold, replacement);
    return kimg_launch_status();
}

// ---- restoring beam in the Fourier domain: beam.py:283-311 + fourier_beam.mako ----------
namespace {
__global__ __launch_bounds__(256) void fourier_beam_kernel(
    float2 *__restrict__ data, int64_t stride, float amplitude, float a, float b, float c,
    int width, int height)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y;
    if (x >= width)
        return;
    float u = (float) x;                 // only the non-negative half of this axis is stored
    float v = (float) ((y * 2 >= height) ? y - height : y);
    float power = (a * v + b * u) * v + c * u * u;
    float ft = amplitude * expf(power);
    int64_t addr = (int64_t) y * stride + x;
    float2 value = data[addr];
    value.x *= ft;
    value.y *= ft;
    data[addr] = value;
}
}  // namespace

extern "C" int kimg_fourier_beam(void *data, int64_t row_stride, int width, int height,
                                 float amplitude, float a, float b, float c, void *stream)
{
    KIMG_CHECK_ARG(width >= 0 && height >= 0 && row_stride >= width);
    if (width == 0 || height == 0)
        return 0;
    KIMG_CHECK_ARG(data != nullptr && height <= 65535);
    dim3 grid(kimg_divup(width, 256), height);
    fourier_beam_kernel<<<grid, 256, 0, (hipStream_t) stream>>>(
        static_cast<float2 *>(data), row_stride, amplitude, a, b, c, width, height);
    return kimg_launch_status();
}

// ---- output statistics of the restore step: frontend.py:171-209 -------------------------------
namespace {
// find_peak (frontend.py:171-194): max |image| over pixels with |image| * pbeam > 7.5 noise
// (comparisons with NaN are false, as on the host).  |x| >= 0, so the float maximum is the
// maximum of the bit patterns.
__global__ __launch_bounds__(256) void image_peak_kernel(
    const float *__restrict__ image, int64_t row_stride, int64_t pol_stride,
    const float *__restrict__ pbeam, int64_t beam_row_stride, int width, int height, int P,
    float limit, unsigned int *__restrict__ out)
{
    float peak = 0.0f;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x < width)
        for (int y = blockIdx.y; y < height; y += gridDim.y) {
            const float pb = pbeam ? pbeam[(int64_t) y * beam_row_stride + x] : 1.0f;
            for (int p = 0; p < P; p++) {
                const float v = fabsf(image[p * pol_stride + (int64_t) y * row_stride + x]);
                if (v > peak && v * pb > limit)
                    peak = v;
            }
        }
    peak = fmaxf(peak, __shfl_xor(peak, 32, WAVE));
#pragma unroll
    for (int off = 16; off > 0; off >>= 1)
        peak = fmaxf(peak, __shfl_xor(peak, off, WAVE));
    if ((threadIdx.x & 63) == 0 && peak > 0.0f)
        atomicMax(out, __float_as_uint(peak));
}

// get_totals (frontend.py:197-209): per-polarization sum ignoring NaNs, in float64.
__global__ __launch_bounds__(256) void image_nansum_kernel(
    const float *__restrict__ image, int64_t row_stride, int64_t pol_stride, int width,
    int height, double *__restrict__ sums)
{
    const int p = blockIdx.z;
    double acc = 0.0;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x < width)
        for (int y = blockIdx.y; y < height; y += gridDim.y) {
            const float v = image[p * pol_stride + (int64_t) y * row_stride + x];
            if (v == v)
                acc += (double) v;
        }
    acc = wave_sum(acc);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0)
        part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < (int) (blockDim.x >> 6); w++)
            t += part[w];
        atomicAdd(&sums[p], t);
    }
}
}  // namespace

extern "C" int kimg_image_peak(const float *image, int64_t row_stride, int64_t pol_stride,
                               const float *pbeam, int64_t beam_row_stride, int width, int height,
                               int num_polarizations, float noise, float *peak, void *stream)
{
    KIMG_CHECK_ARG(image && peak && width > 0 && height > 0 && num_polarizations >= 1);
    hipStream_t s = (hipStream_t) stream;
    KIMG_HIP(hipMemsetAsync(peak, 0, sizeof(float), s));
    int by = height < 256 ? height : 256;
    dim3 grid(kimg_divup(width, 256), by);
    image_peak_kernel<<<grid, 256, 0, s>>>(image, row_stride, pol_stride, pbeam, beam_row_stride,
                                           width, height, num_polarizations, 7.5f * noise,
                                           reinterpret_cast<unsigned int *>(peak));
    return kimg_launch_status();
}

extern "C" int kimg_image_nansum(const float *image, int64_t row_stride, int64_t pol_stride,
                                 int width, int height, int num_polarizations, double *sums,
                                 void *stream)
{
    KIMG_CHECK_ARG(image && sums && width > 0 && height > 0 && num_polarizations >= 1
                   && num_polarizations <= 65535);
    hipStream_t s = (hipStream_t) stream;
    KIMG_HIP(hipMemsetAsync(sums, 0, sizeof(double) * num_polarizations, s));
    int by = height < 128 ? height : 128;
    dim3 grid(kimg_divup(width, 256), by, num_polarizations);
    image_nansum_kernel<<<grid, 256, 0, s>>>(image, row_stride, pol_stride, width, height, sums);
    return kimg_launch_status();
}
