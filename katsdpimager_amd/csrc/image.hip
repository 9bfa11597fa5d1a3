// Image-plane kernels: grid<->layer quadrant copies, layer<->image (fftshift + W-stack phase
// + n-term + taper), scale, add_image, apply_primary_beam.  All HBM-bound streams.
// Mirrors image.py:153-180, 351-367, 439-458, 539-558, 649-673, 716-740 of the reference.
// The arithmetic order follows the reference HOST classes (image.py:781-799, 836-848) so that
// float32 results agree to rounding; this file is built with -ffp-contract=off.
#include "kimg_common.h"
#include <map>
#include <mutex>
#include <utility>

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
__device__ inline float2 half_layer_value(const float2 *__restrict__ grid, int64_t grid_row_stride,
                                          int Gg, int G, int lx, int ly)
{
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
    return make_float2(0.5f * (a.x + b.x), 0.5f * (a.y - b.y));
}

__global__ __launch_bounds__(256) void grid_to_half_layer_kernel(
    float2 *__restrict__ half_layer, int G, const float2 *__restrict__ grid,
    int64_t grid_row_stride, int Gg)
{
    const int lx = blockIdx.x * blockDim.x + threadIdx.x;
    const int ly = blockIdx.y;
    const int W = G / 2 + 1;
    if (lx >= W)
        return;
    half_layer[(int64_t) ly * W + lx] = half_layer_value(grid, grid_row_stride, Gg, G, lx, ly);
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

// The same with the factors where an earlier kernel left them (kimg_pixel_reciprocal): a driver that
// scales by 1 / (a pixel of the image) need not read that pixel back first.
__global__ __launch_bounds__(256) void scale_device_kernel(
    float *__restrict__ image, int64_t row_stride, int64_t pol_stride, int width, int num_pols,
    const float *__restrict__ scale)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= width)
        return;
    int64_t addr = (int64_t) blockIdx.y * row_stride + x;
    for (int p = 0; p < num_pols; p++, addr += pol_stride)
        image[addr] *= scale[p];
}

__global__ void pixel_reciprocal_kernel(const float *__restrict__ image, int64_t pol_stride, int64_t offset,
                                        int num_pols, float *__restrict__ out)
{
    const int p = threadIdx.x;
    if (p < num_pols)
        out[p] = 1.0f / image[p * pol_stride + offset];     // (np.reciprocal of a float32: one rounding)
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

// ---- grid <-> image at w = 0 with transforms of our own ------------------------------------
// The library route above is six launches (pad + fold, transpose, column transforms, transpose,
// row transforms, image correction) over a layer of G x (G/2 + 1) cells, of which only the
// (Gg/2 + 1) columns the grid reaches are not zero.  Here: one launch of column transforms for
// those columns only, reading the grid directly (fold and padding on the fly), and one launch of
// row transforms, two real rows per complex transform, with the image correction as its
// epilogue; what passes between them is (Gg/2 + 1) x G cells (20 MB at 4096 / 1244).
// Transforms: in LDS, decimation in time in stages of radix 4, 2, 3, 5, 7 on digit-reversed
// input (any size 2^a 3^b 5^c 7^d up to 8192 -- the sizes the reference picks, parameters.py:17-25),
// one workgroup per transform, twiddles from a table.
constexpr int FFT_THREADS = 512;
constexpr int FFT_BATCH = 4;        // global loads a thread has in flight in the kernels' prologues
constexpr int FFT_UNROLL = 2;       // butterflies a thread has in flight (4096 cells: all it has)

__device__ inline float2 cmul(float2 a, float2 b)
{
    return make_float2(fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x));
}

__device__ inline float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ inline float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

// LDS index of cell i: one cell of padding per 32, so that the strided accesses of the first
// stages (and the digit-reversed scatter before them) spread over the banks.
__device__ __host__ inline int fft_pad(int i) { return i + (i >> 5); }

// A transform of G = 2^a 3^b 5^c 7^d cells: decimation in time, in place, stages of radix 4 (two
// radix-2 steps in one pass over LDS), 2, 3, 5 and 7 in that order; the input goes in at the cells
// the generalised digit reversal names (`perm`, made once per size on the host), the result comes
// out in natural order.  Twiddles, one contiguous run per stage (a single table indexed
// j G / rq would be read with strides that put all 64 lanes on one bank): for the stage of radix r
// over blocks of q cells, q cells e^{2 pi i j / rq} (the stage's other twiddles are their powers).
struct fft_plan {
    const float2 *twiddle;
    const unsigned short *perm;     // LDS cell (before padding) of input index n
    int twiddles;                   // cells of `twiddle`
    int stages;
    int log2size;                   // log2 G when G is a power of two (perm is then the bit reversal)
    unsigned char radix[16];
};

// ODD: the size has factors 3, 5 or 7.  Powers of two get kernels without those stages (fewer
// registers: three workgroups per CU instead of two) that compute the bit reversal themselves.
template<bool ODD>
__device__ inline int fft_cell(const fft_plan &plan, int n)
{
    return ODD ? (int) plan.perm[n] : (int) (__brev((unsigned) n) >> (32 - plan.log2size));
}

template<bool INVERSE>
__device__ inline float2 fft_conj_if_forward(float2 w) { return INVERSE ? w : make_float2(w.x, -w.y); }

// Radix 3, 5, 7: out[c] = sum_k in[k] e^{+-2 pi i k c / R} with the R-th roots of unity as constants
template<int R> struct fft_roots;
template<> struct fft_roots<3> {
    static constexpr float c[3] = {1.0f, -0.5f, -0.5f};
    static constexpr float s[3] = {0.0f, 0.86602540378443865f, -0.86602540378443865f};
};
template<> struct fft_roots<5> {
    static constexpr float c[5] = {1.0f, 0.30901699437494742f, -0.80901699437494742f,
                                   -0.80901699437494742f, 0.30901699437494742f};
    static constexpr float s[5] = {0.0f, 0.95105651629515357f, 0.58778525229247313f,
                                   -0.58778525229247313f, -0.95105651629515357f};
};
template<> struct fft_roots<7> {
    static constexpr float c[7] = {1.0f, 0.62348980185873353f, -0.22252093395631440f,
                                   -0.90096886790241913f, -0.90096886790241913f,
                                   -0.22252093395631440f, 0.62348980185873353f};
    static constexpr float s[7] = {0.0f, 0.78183148246802981f, 0.97492791218182361f,
                                   0.43388373911755812f, -0.43388373911755812f,
                                   -0.97492791218182361f, -0.78183148246802981f};
};

template<bool INVERSE, int R>
__device__ inline void fft_stage_odd(float2 *x, const float2 *tw, int G, int q)
{
    for (int t = threadIdx.x; t < G / R; t += FFT_THREADS) {
        const int block = t / q, j = t - block * q;
        const int base = block * (R * q) + j;
        const float2 w = fft_conj_if_forward<INVERSE>(tw[j]);
        float2 in[R];
        int at[R];
        float2 wk = make_float2(1.0f, 0.0f);
#pragma unroll
        for (int k = 0; k < R; k++) {
            at[k] = fft_pad(base + k * q);
            in[k] = k ? cmul(wk, x[at[k]]) : x[at[k]];
            wk = k ? cmul(wk, w) : w;
        }
#pragma unroll
        for (int c = 0; c < R; c++) {
            float2 acc = in[0];
#pragma unroll
            for (int k = 1; k < R; k++) {
                const int m = (k * c) % R;
                const float2 root = make_float2(fft_roots<R>::c[m],
                                                INVERSE ? fft_roots<R>::s[m] : -fft_roots<R>::s[m]);
                acc = cadd(acc, cmul(root, in[k]));
            }
            x[at[c]] = acc;
        }
    }
}

// x: G cells in `perm` order on entry, the transform (sign + for INVERSE, unnormalised) in natural
// order on return.  Called by the whole workgroup; starts and ends with a barrier.
template<bool INVERSE, bool ODD>
__device__ inline void lds_fft(float2 *x, const float2 *tw, int G, const fft_plan &plan)
{
    int q = 1;
    // (a power of two: the stages follow from G -- radix 4 while it fits, then one radix 2 --
    // and nothing is read from the plan between the barriers)
    const int stages = ODD ? plan.stages : (plan.log2size + 1) / 2;
    // (the radices are fetched once, up front: a scalar load per stage would sit between its barriers)
    unsigned radix_words[4] = {0, 0, 0, 0};
    if (ODD) {
#pragma unroll
        for (int i = 0; i < 4; i++)
            radix_words[i] = (unsigned) plan.radix[4 * i] | ((unsigned) plan.radix[4 * i + 1] << 8)
                             | ((unsigned) plan.radix[4 * i + 2] << 16) | ((unsigned) plan.radix[4 * i + 3] << 24);
    }
    for (int stage = 0; stage < stages; stage++) {
        const unsigned word = stage < 4 ? radix_words[0] : stage < 8 ? radix_words[1]
                              : stage < 12 ? radix_words[2] : radix_words[3];
        const int r = ODD ? (int) ((word >> (8 * (stage & 3))) & 255u) : (4 * q <= G ? 4 : 2);
        __syncthreads();
        if (r == 4) {
            // (q is a power of two here: the radix-4 and radix-2 stages come first)
            const int shift = ODD ? 31 - __clz(q) : 2 * stage;
            for (int t0 = threadIdx.x; t0 < G / 4; t0 += FFT_UNROLL * FFT_THREADS) {
                float2 a[FFT_UNROLL][4], w2[FFT_UNROLL];
                int at[FFT_UNROLL][4];
#pragma unroll
                for (int u = 0; u < FFT_UNROLL; u++) {
                    const int t = t0 + u * FFT_THREADS;
                    if (t < G / 4) {
                        const int j = t & (q - 1);
                        const int base = ((t >> shift) << (shift + 2)) + j;
                        w2[u] = tw[j];
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            at[u][k] = fft_pad(base + k * q);
                            a[u][k] = x[at[u][k]];
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < FFT_UNROLL; u++) {
                    if (t0 + u * FFT_THREADS < G / 4) {
                        if (!INVERSE)
                            w2[u].y = -w2[u].y;
                        const float2 w1 = cmul(w2[u], w2[u]);
                        const float2 a0 = a[u][0], a1 = cmul(w1, a[u][1]);
                        const float2 a2 = a[u][2], a3 = cmul(w1, a[u][3]);
                        const float2 b0 = cadd(a0, a1), b1 = csub(a0, a1);
                        const float2 b2 = cmul(w2[u], cadd(a2, a3));
                        const float2 t3 = cmul(w2[u], csub(a2, a3));
                        // the twiddle of the second pair is w2 times e^{+-i pi / 2}
                        const float2 b3 = INVERSE ? make_float2(-t3.y, t3.x) : make_float2(t3.y, -t3.x);
                        x[at[u][0]] = cadd(b0, b2);
                        x[at[u][2]] = csub(b0, b2);
                        x[at[u][1]] = cadd(b1, b3);
                        x[at[u][3]] = csub(b1, b3);
                    }
                }
            }
        } else if (r == 2) {
            const int shift = 31 - __clz(q);
            for (int t = threadIdx.x; t < G / 2; t += FFT_THREADS) {
                const int j = t & (q - 1);
                const int base = ((t >> shift) << (shift + 1)) + j;
                const float2 w = fft_conj_if_forward<INVERSE>(tw[j]);
                const int i0 = fft_pad(base), i1 = fft_pad(base + q);
                const float2 a0 = x[i0], a1 = cmul(w, x[i1]);
                x[i0] = cadd(a0, a1);
                x[i1] = csub(a0, a1);
            }
        } else if (ODD && r == 3) {
            fft_stage_odd<INVERSE, 3>(x, tw, G, q);
        } else if (ODD && r == 5) {
            fft_stage_odd<INVERSE, 5>(x, tw, G, q);
        } else if (ODD) {
            fft_stage_odd<INVERSE, 7>(x, tw, G, q);
        }
        tw += q;
        q *= r;
    }
    __syncthreads();
}

// LDS of the transform kernels: the (padded) cells, then the twiddles
__host__ __device__ inline int fft_lds_cells(int G) { return fft_pad(G) + 1; }

__device__ inline void fft_lds_setup(float2 *x, float2 *tw, const fft_plan &plan, int G, bool clear)
{
    for (int i = threadIdx.x; i < plan.twiddles; i += FFT_THREADS)
        tw[i] = plan.twiddle[i];
    if (clear)
        for (int i = threadIdx.x; i < fft_lds_cells(G); i += FFT_THREADS)
            x[i] = make_float2(0.0f, 0.0f);
    __syncthreads();
}

__device__ inline int fft_shift(int i, int half) { return i < half ? i + half : i - half; }

// Workgroups are dealt to the 8 XCDs in turn; neighbouring rows share the cache lines of T (a
// row pair is 16 bytes of each) and neighbouring columns those of the grid, so give every XCD
// (and its L2) a contiguous run of them.
__device__ inline int xcd_contiguous(int block, int blocks)
{
    const int xcd = block % 8, each = blocks / 8, extra = blocks % 8;
    return xcd * each + min(xcd, extra) + block / 8;
}

// Column lx of the half layer (never stored): T[lx][sy] = sum_ly half_layer[ly][lx] e^{2 pi i ly sy / G}
template<bool ODD>
__global__ __launch_bounds__(FFT_THREADS) void g2i_columns_kernel(
    float2 *__restrict__ T, const float2 *__restrict__ grid, int64_t grid_row_stride, int Gg, int G,
    fft_plan plan)
{
    extern __shared__ float2 fft_lds[];
    float2 *x = fft_lds, *tw = fft_lds + fft_lds_cells(G);
    fft_lds_setup(x, tw, plan, G, true);
    const int lx = xcd_contiguous(blockIdx.x, gridDim.x), half = Gg / 2;
    // the rows the grid (or its mirror image) reaches: centred -half .. half
    const int rows = 2 * half == G ? G : Gg + 1;
    // (FFT_BATCH cells per thread and round: their loads are issued before the first of them is used)
    for (int r0 = threadIdx.x; r0 < rows; r0 += FFT_BATCH * FFT_THREADS) {
        float2 value[FFT_BATCH];
        int cell[FFT_BATCH];
#pragma unroll
        for (int k = 0; k < FFT_BATCH; k++) {
            const int r = r0 + k * FFT_THREADS;
            const int cy = r - half;
            const int ly = cy < 0 ? cy + G : cy;
            cell[k] = r < rows ? ly : -1;
            if (r < rows)
                value[k] = half_layer_value(grid, grid_row_stride, Gg, G, lx, ly);
        }
#pragma unroll
        for (int k = 0; k < FFT_BATCH; k++)
            if (cell[k] >= 0)
                x[fft_pad(fft_cell<ODD>(plan, cell[k]))] = value[k];
    }
    lds_fft<true, ODD>(x, tw, G, plan);
    for (int sy = threadIdx.x; sy < G; sy += FFT_THREADS)
        T[(int64_t) lx * G + sy] = x[fft_pad(sy)];
}

// Rows sy1 = 2 * blockIdx.x and sy1 + 1 of the real transform: both Hermitian sequences in one
// complex transform (z = row1 + i row2 comes out with row1 in its real part, row2 in its
// imaginary part), then real_layer_to_image_kernel's arithmetic.
// (at most 85 registers: three workgroups per CU, as many as the LDS allows)
template<bool ACCUMULATE, bool ODD>
__global__ __launch_bounds__(FFT_THREADS, 6) void g2i_rows_kernel(
    float *__restrict__ image, int64_t image_row_stride, const float2 *__restrict__ T, int Gg, int G,
    fft_plan plan, const float *__restrict__ kernel1d,
    float lm_scale, float lm_bias)
{
    extern __shared__ float2 fft_lds[];
    float2 *x = fft_lds, *tw = fft_lds + fft_lds_cells(G);
    fft_lds_setup(x, tw, plan, G, true);
    const int sy1 = 2 * xcd_contiguous(blockIdx.x, gridDim.x), half = Gg / 2;
    for (int n0 = threadIdx.x; n0 <= half; n0 += FFT_BATCH * FFT_THREADS) {
        // (t.x, t.y) = T[n][sy1], (t.z, t.w) = T[n][sy1 + 1]
        float4 t4[FFT_BATCH];
#pragma unroll
        for (int k = 0; k < FFT_BATCH; k++) {
            const int n = n0 + k * FFT_THREADS;
            if (n <= half)
                t4[k] = *reinterpret_cast<const float4 *>(T + (int64_t) n * G + sy1);
        }
#pragma unroll
        for (int k = 0; k < FFT_BATCH; k++) {
            const int n = n0 + k * FFT_THREADS;
            if (n > half)
                continue;
            const float4 t = t4[k];
            if (n == 0 || 2 * n == G) {
                x[fft_pad(fft_cell<ODD>(plan, n))] = make_float2(t.x, t.z);       // (real up to rounding)
            } else {
                x[fft_pad(fft_cell<ODD>(plan, n))] = make_float2(t.x - t.w, t.y + t.z);
                x[fft_pad(fft_cell<ODD>(plan, G - n))] = make_float2(t.x + t.w, t.z - t.y);
            }
        }
    }
    lds_fft<true, ODD>(x, tw, G, plan);
    const int hG = G / 2;
    const int ya = fft_shift(sy1, hG), yb = fft_shift(sy1 + 1, hG);
    const float ma = lm_coord(ya, lm_scale, lm_bias), mb = lm_coord(yb, lm_scale, lm_bias);
    const float ma2 = ma * ma, mb2 = mb * mb;
    const float ka = kernel1d[ya], kb = kernel1d[yb];
    float *rowa = image + (int64_t) ya * image_row_stride, *rowb = image + (int64_t) yb * image_row_stride;
    for (int sx0 = threadIdx.x; sx0 < G; sx0 += 2 * FFT_THREADS) {
        float olda[2] = {0.0f, 0.0f}, oldb[2] = {0.0f, 0.0f}, kx[2];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int sx = sx0 + k * FFT_THREADS;
            if (sx < G) {
                const int xx = fft_shift(sx, hG);
                kx[k] = kernel1d[xx];
                if (ACCUMULATE) {
                    olda[k] = rowa[xx];
                    oldb[k] = rowb[xx];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int sx = sx0 + k * FFT_THREADS;
            if (sx >= G)
                continue;
            const int xx = fft_shift(sx, hG);
            const float2 both = x[fft_pad(sx)];
            const float l = lm_coord(xx, lm_scale, lm_bias);
            const float l2 = l * l;
            const float na = sqrtf(1.0f - (ma2 + l2)), nb = sqrtf(1.0f - (mb2 + l2));
            const float va = (both.x * na) / (ka * kx[k]), vb = (both.y * nb) / (kb * kx[k]);
            rowa[xx] = ACCUMULATE ? olda[k] + va : va;
            rowb[xx] = ACCUMULATE ? oldb[k] + vb : vb;
        }
    }
}

// image -> grid, rows: two real layer rows (image_to_real_layer_kernel's arithmetic) as one
// complex sequence, forward transform, the two half spectra taken apart;
// T[lx][sy] for lx = 0 .. Gg / 2.
template<bool ODD>
__global__ __launch_bounds__(FFT_THREADS, 6) void i2g_rows_kernel(
    float2 *__restrict__ T, const float *__restrict__ image, int64_t image_row_stride, int Gg, int G,
    fft_plan plan, const float *__restrict__ kernel1d,
    float lm_scale, float lm_bias)
{
    extern __shared__ float2 fft_lds[];
    float2 *x = fft_lds, *tw = fft_lds + fft_lds_cells(G);
    fft_lds_setup(x, tw, plan, G, false);
    const int sy1 = 2 * xcd_contiguous(blockIdx.x, gridDim.x), half = Gg / 2, hG = G / 2;
    const int y1 = fft_shift(sy1, hG), y2 = fft_shift(sy1 + 1, hG);
    const float ma = lm_coord(y1, lm_scale, lm_bias), mb = lm_coord(y2, lm_scale, lm_bias);
    const float ma2 = ma * ma, mb2 = mb * mb;
    const float ka = kernel1d[y1], kb = kernel1d[y2];
    for (int sx0 = threadIdx.x; sx0 < G; sx0 += FFT_BATCH * FFT_THREADS) {
        float pa[FFT_BATCH], pb[FFT_BATCH], kx[FFT_BATCH];
#pragma unroll
        for (int k = 0; k < FFT_BATCH; k++) {
            const int sx = sx0 + k * FFT_THREADS;
            if (sx < G) {
                const int xx = fft_shift(sx, hG);
                kx[k] = kernel1d[xx];
                pa[k] = image[(int64_t) y1 * image_row_stride + xx];
                pb[k] = image[(int64_t) y2 * image_row_stride + xx];
            }
        }
#pragma unroll
        for (int k = 0; k < FFT_BATCH; k++) {
            const int sx = sx0 + k * FFT_THREADS;
            if (sx >= G)
                continue;
            const int xx = fft_shift(sx, hG);
            const float l = lm_coord(xx, lm_scale, lm_bias);
            const float l2 = l * l;
            const float na = sqrtf(1.0f - (ma2 + l2)), nb = sqrtf(1.0f - (mb2 + l2));
            const float va = pa[k] / ((ka * kx[k]) * na);
            const float vb = pb[k] / ((kb * kx[k]) * nb);
            x[fft_pad(fft_cell<ODD>(plan, sx))] = make_float2(va, vb);
        }
    }
    lds_fft<false, ODD>(x, tw, G, plan);
    for (int lx = threadIdx.x; lx <= half; lx += FFT_THREADS) {
        const float2 z = x[fft_pad(lx)], zm = x[fft_pad(lx ? G - lx : 0)];
        // row a: (z + conj zm) / 2, row b: (z - conj zm) / 2i
        const float4 out = make_float4(0.5f * (z.x + zm.x), 0.5f * (z.y - zm.y),
                                       0.5f * (z.y + zm.y), 0.5f * (zm.x - z.x));
        *reinterpret_cast<float4 *>(T + (int64_t) lx * G + sy1) = out;
    }
}

// image -> grid, columns: F[ly][lx] = sum_sy T[lx][sy] e^{-2 pi i ly sy / G}; grid column
// half + lx from it, column half - lx from its mirror image (half_layer_to_grid_kernel).
template<bool ODD>
__global__ __launch_bounds__(FFT_THREADS) void i2g_columns_kernel(
    float2 *__restrict__ grid, int64_t grid_row_stride, const float2 *__restrict__ T, int Gg, int G,
    fft_plan plan)
{
    extern __shared__ float2 fft_lds[];
    float2 *x = fft_lds, *tw = fft_lds + fft_lds_cells(G);
    fft_lds_setup(x, tw, plan, G, false);
    const int lx = xcd_contiguous(blockIdx.x, gridDim.x), half = Gg / 2;
    for (int sy0 = threadIdx.x; sy0 < G; sy0 += FFT_BATCH * FFT_THREADS) {
        float2 value[FFT_BATCH];
#pragma unroll
        for (int k = 0; k < FFT_BATCH; k++)
            if (sy0 + k * FFT_THREADS < G)
                value[k] = T[(int64_t) lx * G + sy0 + k * FFT_THREADS];
#pragma unroll
        for (int k = 0; k < FFT_BATCH; k++)
            if (sy0 + k * FFT_THREADS < G)
                x[fft_pad(fft_cell<ODD>(plan, sy0 + k * FFT_THREADS))] = value[k];
    }
    lds_fft<false, ODD>(x, tw, G, plan);
    for (int gy = threadIdx.x; gy < Gg; gy += FFT_THREADS) {
        const int cy = gy - half;
        const int ly = cy < 0 ? cy + G : cy;
        if (lx < half)
            grid[(int64_t) gy * grid_row_stride + half + lx] = x[fft_pad(ly)];
        if (lx == half && 2 * half == G) {
            grid[(int64_t) gy * grid_row_stride] = x[fft_pad(ly)];      // the Nyquist column is its own mirror
        } else if (lx > 0) {
            const float2 v = x[fft_pad(ly ? G - ly : 0)];
            grid[(int64_t) gy * grid_row_stride + half - lx] = make_float2(v.x, -v.y);
        }
    }
}

// ---- the same two-launch scheme for any w (complex layer, no Hermitian symmetry to use) -------
// Columns: the Gg columns of the layer the grid reaches (centred -Gg/2 .. Gg/2 - 1), index c.
// Rows: every row needs its own complex transform; a workgroup takes two neighbouring rows one
// after the other so that it reads (or writes) T in 16-byte pieces.
constexpr int FFT_ROW_CELLS = 8192 / FFT_THREADS;     // cells of one row of T a thread may hold

__device__ inline int grid_to_layer_index(int c, int half, int G)
{
    const int centred = c - half;
    return centred < 0 ? centred + G : centred;
}

template<bool ODD>
__global__ __launch_bounds__(FFT_THREADS) void g2iw_columns_kernel(
    float2 *__restrict__ T, const float2 *__restrict__ grid, int64_t grid_row_stride, int Gg, int G,
    fft_plan plan)
{
    extern __shared__ float2 fft_lds[];
    float2 *x = fft_lds, *tw = fft_lds + fft_lds_cells(G);
    fft_lds_setup(x, tw, plan, G, true);
    const int c = xcd_contiguous(blockIdx.x, gridDim.x), half = Gg / 2;
    for (int r0 = threadIdx.x; r0 < Gg; r0 += FFT_BATCH * FFT_THREADS) {
        float2 value[FFT_BATCH];
#pragma unroll
        for (int k = 0; k < FFT_BATCH; k++)
            if (r0 + k * FFT_THREADS < Gg)
                value[k] = grid[(int64_t) (r0 + k * FFT_THREADS) * grid_row_stride + c];
#pragma unroll
        for (int k = 0; k < FFT_BATCH; k++)
            if (r0 + k * FFT_THREADS < Gg)
                x[fft_pad(fft_cell<ODD>(plan, grid_to_layer_index(r0 + k * FFT_THREADS, half, G)))] = value[k];
    }
    lds_fft<true, ODD>(x, tw, G, plan);
    for (int sy = threadIdx.x; sy < G; sy += FFT_THREADS)
        T[(int64_t) c * G + sy] = x[fft_pad(sy)];
}

// layer_to_image_kernel's arithmetic on rows sy1 = 2 * blockIdx.x and sy1 + 1
// (at most 128 registers: two workgroups per CU)
template<bool ACCUMULATE, bool ODD>
__global__ __launch_bounds__(FFT_THREADS, 4) void g2iw_rows_kernel(
    float *__restrict__ image, int64_t image_row_stride, const float2 *__restrict__ T, int Gg, int G,
    fft_plan plan, const float *__restrict__ kernel1d,
    float lm_scale, float lm_bias, float w)
{
    extern __shared__ float2 fft_lds[];
    float2 *x = fft_lds, *tw = fft_lds + fft_lds_cells(G);
    fft_lds_setup(x, tw, plan, G, true);
    const int sy1 = 2 * xcd_contiguous(blockIdx.x, gridDim.x), half = Gg / 2, hG = G / 2;
    float2 second[FFT_ROW_CELLS];
#pragma unroll
    for (int k = 0; k < FFT_ROW_CELLS; k++) {
        const int c = threadIdx.x + k * FFT_THREADS;
        if (c < Gg) {
            const float4 t = *reinterpret_cast<const float4 *>(T + (int64_t) c * G + sy1);
            x[fft_pad(fft_cell<ODD>(plan, grid_to_layer_index(c, half, G)))] = make_float2(t.x, t.y);
            second[k] = make_float2(t.z, t.w);
        }
    }
#pragma unroll
    for (int r = 0; r < 2; r++) {
        if (r == 1) {
            __syncthreads();
            for (int i = threadIdx.x; i < fft_lds_cells(G); i += FFT_THREADS)
                x[i] = make_float2(0.0f, 0.0f);
            __syncthreads();
#pragma unroll
            for (int k = 0; k < FFT_ROW_CELLS; k++) {
                const int c = threadIdx.x + k * FFT_THREADS;
                if (c < Gg)
                    x[fft_pad(fft_cell<ODD>(plan, grid_to_layer_index(c, half, G)))] = second[k];
            }
        }
        lds_fft<true, ODD>(x, tw, G, plan);
        const int y = fft_shift(sy1 + r, hG);
        const float m = lm_coord(y, lm_scale, lm_bias);
        const float m2 = m * m;
        const float ky = kernel1d[y];
        float *row = image + (int64_t) y * image_row_stride;
        for (int sx = threadIdx.x; sx < G; sx += FFT_THREADS) {
            const int xx = fft_shift(sx, hG);
            const float2 v = x[fft_pad(sx)];
            const float l = lm_coord(xx, lm_scale, lm_bias);
            const float l2 = l * l;
            const float n = sqrtf(1.0f - (m2 + l2));
            float c, s;
            expj2pi(w * (n - 1.0f), c, s);
            const float rotated = v.x * c - v.y * s;
            const float taper = ky * kernel1d[xx];
            const float out = (rotated * n) / taper;
            row[xx] = ACCUMULATE ? row[xx] + out : out;
        }
    }
}

// image_to_layer_kernel's arithmetic on two rows, forward transforms, the Gg columns kept
template<bool ODD>
__global__ __launch_bounds__(FFT_THREADS, 4) void i2gw_rows_kernel(
    float2 *__restrict__ T, const float *__restrict__ image, int64_t image_row_stride, int Gg, int G,
    fft_plan plan, const float *__restrict__ kernel1d,
    float lm_scale, float lm_bias, float w)
{
    extern __shared__ float2 fft_lds[];
    float2 *x = fft_lds, *tw = fft_lds + fft_lds_cells(G);
    fft_lds_setup(x, tw, plan, G, false);
    const int sy1 = 2 * xcd_contiguous(blockIdx.x, gridDim.x), half = Gg / 2, hG = G / 2;
    float2 first[FFT_ROW_CELLS];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int y = fft_shift(sy1 + r, hG);
        const float m = lm_coord(y, lm_scale, lm_bias);
        const float m2 = m * m;
        const float ky = kernel1d[y];
        if (r == 1)
            __syncthreads();        // (the first row's results have been taken out of x)
        for (int sx0 = threadIdx.x; sx0 < G; sx0 += FFT_BATCH * FFT_THREADS) {
            float pixel[FFT_BATCH], kx[FFT_BATCH];
#pragma unroll
            for (int k = 0; k < FFT_BATCH; k++) {
                const int sx = sx0 + k * FFT_THREADS;
                if (sx < G) {
                    const int xx = fft_shift(sx, hG);
                    pixel[k] = image[(int64_t) y * image_row_stride + xx];
                    kx[k] = kernel1d[xx];
                }
            }
#pragma unroll
            for (int k = 0; k < FFT_BATCH; k++) {
                const int sx = sx0 + k * FFT_THREADS;
                if (sx >= G)
                    continue;
                const int xx = fft_shift(sx, hG);
                const float l = lm_coord(xx, lm_scale, lm_bias);
                const float l2 = l * l;
                const float n = sqrtf(1.0f - (m2 + l2));
                float c, s;
                expj2pi(-w * (n - 1.0f), c, s);
                const float taper = ky * kx[k];
                const float v = pixel[k] / (taper * n);
                x[fft_pad(fft_cell<ODD>(plan, sx))] = make_float2(v * c, v * s);
            }
        }
        lds_fft<false, ODD>(x, tw, G, plan);
#pragma unroll
        for (int k = 0; k < FFT_ROW_CELLS; k++) {
            const int c = threadIdx.x + k * FFT_THREADS;
            if (c < Gg) {
                const float2 v = x[fft_pad(grid_to_layer_index(c, half, G))];
                if (r == 0)
                    first[k] = v;
                else
                    *reinterpret_cast<float4 *>(T + (int64_t) c * G + sy1) =
                        make_float4(first[k].x, first[k].y, v.x, v.y);
            }
        }
    }
}

template<bool ODD>
__global__ __launch_bounds__(FFT_THREADS) void i2gw_columns_kernel(
    float2 *__restrict__ grid, int64_t grid_row_stride, const float2 *__restrict__ T, int Gg, int G,
    fft_plan plan)
{
    extern __shared__ float2 fft_lds[];
    float2 *x = fft_lds, *tw = fft_lds + fft_lds_cells(G);
    fft_lds_setup(x, tw, plan, G, false);
    const int c = xcd_contiguous(blockIdx.x, gridDim.x), half = Gg / 2;
    for (int sy0 = threadIdx.x; sy0 < G; sy0 += FFT_BATCH * FFT_THREADS) {
        float2 value[FFT_BATCH];
#pragma unroll
        for (int k = 0; k < FFT_BATCH; k++)
            if (sy0 + k * FFT_THREADS < G)
                value[k] = T[(int64_t) c * G + sy0 + k * FFT_THREADS];
#pragma unroll
        for (int k = 0; k < FFT_BATCH; k++)
            if (sy0 + k * FFT_THREADS < G)
                x[fft_pad(fft_cell<ODD>(plan, sy0 + k * FFT_THREADS))] = value[k];
    }
    lds_fft<false, ODD>(x, tw, G, plan);
    for (int gy = threadIdx.x; gy < Gg; gy += FFT_THREADS)
        grid[(int64_t) gy * grid_row_stride + c] = x[fft_pad(grid_to_layer_index(gy, half, G))];
}

// ---- restoring-beam convolution (beam.py:351-398) on the same transforms -----------------------
// image -> rows (two real rows per complex transform, no shift) -> T[u][y] for u = 0 .. G/2 ->
// per column: forward transform, times the beam's Fourier transform, inverse transform, all in LDS
// -> rows back.  Three launches; the half spectrum is written once and read once more than it must.
template<bool ODD>
__global__ __launch_bounds__(FFT_THREADS) void cb_rows_forward_kernel(
    float2 *__restrict__ T, const float *__restrict__ image, int64_t row_stride, int G, fft_plan plan)
{
    extern __shared__ float2 fft_lds[];
    float2 *x = fft_lds, *tw = fft_lds + fft_lds_cells(G);
    fft_lds_setup(x, tw, plan, G, false);
    const int y1 = 2 * xcd_contiguous(blockIdx.x, gridDim.x);
    for (int sx0 = threadIdx.x; sx0 < G; sx0 += FFT_BATCH * FFT_THREADS) {
        float2 value[FFT_BATCH];
#pragma unroll
        for (int k = 0; k < FFT_BATCH; k++)
            if (sx0 + k * FFT_THREADS < G)
                value[k] = make_float2(image[(int64_t) y1 * row_stride + sx0 + k * FFT_THREADS],
                                       image[(int64_t) (y1 + 1) * row_stride + sx0 + k * FFT_THREADS]);
#pragma unroll
        for (int k = 0; k < FFT_BATCH; k++)
            if (sx0 + k * FFT_THREADS < G)
                x[fft_pad(fft_cell<ODD>(plan, sx0 + k * FFT_THREADS))] = value[k];
    }
    lds_fft<false, ODD>(x, tw, G, plan);
    for (int lx = threadIdx.x; lx <= G / 2; lx += FFT_THREADS) {
        const float2 z = x[fft_pad(lx)], zm = x[fft_pad(lx ? G - lx : 0)];
        *reinterpret_cast<float4 *>(T + (int64_t) lx * G + y1) =
            make_float4(0.5f * (z.x + zm.x), 0.5f * (z.y - zm.y), 0.5f * (z.y + zm.y), 0.5f * (zm.x - z.x));
    }
}

template<bool ODD>
__global__ __launch_bounds__(FFT_THREADS) void cb_columns_kernel(
    float2 *__restrict__ T, int G, fft_plan plan, float amplitude, float a, float b, float c)
{
    extern __shared__ float2 fft_lds[];
    float2 *x = fft_lds, *tw = fft_lds + fft_lds_cells(G);
    fft_lds_setup(x, tw, plan, G, false);
    const int lx = xcd_contiguous(blockIdx.x, gridDim.x);
    float2 *column = T + (int64_t) lx * G;
    for (int y0 = threadIdx.x; y0 < G; y0 += FFT_BATCH * FFT_THREADS) {
        float2 value[FFT_BATCH];
#pragma unroll
        for (int k = 0; k < FFT_BATCH; k++)
            if (y0 + k * FFT_THREADS < G)
                value[k] = column[y0 + k * FFT_THREADS];
#pragma unroll
        for (int k = 0; k < FFT_BATCH; k++)
            if (y0 + k * FFT_THREADS < G)
                x[fft_pad(fft_cell<ODD>(plan, y0 + k * FFT_THREADS))] = value[k];
    }
    lds_fft<false, ODD>(x, tw, G, plan);
    // fourier_beam_kernel's factor, and the result into the cells the inverse transform starts from
    const float u = (float) lx;
    float2 held[FFT_ROW_CELLS];
#pragma unroll
    for (int k = 0; k < FFT_ROW_CELLS; k++) {
        const int ly = threadIdx.x + k * FFT_THREADS;
        if (ly < G) {
            const float v = (float) (ly * 2 >= G ? ly - G : ly);
            const float power = (a * v + b * u) * v + c * u * u;
            const float ft = amplitude * expf(power);
            const float2 value = x[fft_pad(ly)];
            held[k] = make_float2(value.x * ft, value.y * ft);
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < FFT_ROW_CELLS; k++) {
        const int ly = threadIdx.x + k * FFT_THREADS;
        if (ly < G)
            x[fft_pad(fft_cell<ODD>(plan, ly))] = held[k];
    }
    lds_fft<true, ODD>(x, tw, G, plan);
    for (int y = threadIdx.x; y < G; y += FFT_THREADS)
        column[y] = x[fft_pad(y)];
}

template<bool ODD>
__global__ __launch_bounds__(FFT_THREADS) void cb_rows_inverse_kernel(
    float *__restrict__ image, int64_t row_stride, const float2 *__restrict__ T, int G, fft_plan plan)
{
    extern __shared__ float2 fft_lds[];
    float2 *x = fft_lds, *tw = fft_lds + fft_lds_cells(G);
    fft_lds_setup(x, tw, plan, G, false);
    const int y1 = 2 * xcd_contiguous(blockIdx.x, gridDim.x);
    for (int n = threadIdx.x; n <= G / 2; n += FFT_THREADS) {
        const float4 t = *reinterpret_cast<const float4 *>(T + (int64_t) n * G + y1);
        if (n == 0 || 2 * n == G) {
            x[fft_pad(fft_cell<ODD>(plan, n))] = make_float2(t.x, t.z);     // (imaginary parts ignored,
        } else {                                                          // as a complex-to-real plan does)
            x[fft_pad(fft_cell<ODD>(plan, n))] = make_float2(t.x - t.w, t.y + t.z);
            x[fft_pad(fft_cell<ODD>(plan, G - n))] = make_float2(t.x + t.w, t.z - t.y);
        }
    }
    lds_fft<true, ODD>(x, tw, G, plan);
    for (int sx = threadIdx.x; sx < G; sx += FFT_THREADS) {
        const float2 v = x[fft_pad(sx)];
        image[(int64_t) y1 * row_stride + sx] = v.x;
        image[(int64_t) (y1 + 1) * row_stride + sx] = v.y;
    }
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

extern "C" int kimg_pixel_reciprocal(const float *image, int64_t row_stride, int64_t pol_stride,
                                     int width, int height, int num_polarizations, int x, int y,
                                     float *out, void *stream)
{
    KIMG_CHECK_ARG(image && out && x >= 0 && x < width && y >= 0 && y < height);
    if (num_polarizations < 1 || num_polarizations > 4)
        return KIMG_EUNSUPPORTED;
    pixel_reciprocal_kernel<<<1, 64, 0, (hipStream_t) stream>>>(image, pol_stride, (int64_t) y * row_stride + x,
                                                                num_polarizations, out);
    return kimg_launch_status();
}

extern "C" int kimg_scale_device(float *image, int64_t row_stride, int64_t pol_stride, int width,
                                 int height, int num_polarizations, const float *scale, void *stream)
{
    KIMG_CHECK_ARG(image && scale && width > 0 && height > 0);
    if (num_polarizations < 1 || num_polarizations > 4)
        return KIMG_EUNSUPPORTED;
    dim3 g(kimg_divup(width, 256), height);
    scale_device_kernel<<<g, 256, 0, (hipStream_t) stream>>>(image, row_stride, pol_stride, width,
                                                             num_polarizations, scale);
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
    constexpr int ROWS = 4;     // rows in flight per thread
    if (x < width)
        for (int y0 = blockIdx.y; y0 < height; y0 += gridDim.y * ROWS)
            for (int p = 0; p < P; p++) {
                float v[ROWS], pb[ROWS];
#pragma unroll
                for (int r = 0; r < ROWS; r++) {
                    const int y = y0 + r * gridDim.y;
                    const bool ok = y < height;
                    v[r] = ok ? fabsf(image[p * pol_stride + (int64_t) y * row_stride + x]) : 0.0f;
                    pb[r] = (ok && pbeam) ? pbeam[(int64_t) y * beam_row_stride + x] : 1.0f;
                }
#pragma unroll
                for (int r = 0; r < ROWS; r++)
                    if (v[r] > peak && v[r] * pb[r] > limit)
                        peak = v[r];
            }
    peak = fmaxf(peak, __shfl_xor(peak, 32, WAVE));
#pragma unroll
    for (int off = 16; off > 0; off >>= 1)
        peak = fmaxf(peak, __shfl_xor(peak, off, WAVE));
    // one atomic per workgroup (atomics on one address are served one after the other)
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0)
        part[threadIdx.x >> 6] = peak;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int) (blockDim.x >> 6); w++)
            peak = fmaxf(peak, part[w]);
        if (peak > 0.0f)
            atomicMax(out, __float_as_uint(peak));
    }
}

// get_totals (frontend.py:197-209): per-polarization sum ignoring NaNs, in float64.
__global__ __launch_bounds__(256) void image_nansum_kernel(
    const float *__restrict__ image, int64_t row_stride, int64_t pol_stride, int width,
    int height, double *__restrict__ sums)
{
    const int p = blockIdx.z;
    double acc = 0.0;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    constexpr int ROWS = 4;
    if (x < width)
        for (int y0 = blockIdx.y; y0 < height; y0 += gridDim.y * ROWS) {
            float v[ROWS];
#pragma unroll
            for (int r = 0; r < ROWS; r++) {
                const int y = y0 + r * gridDim.y;
                v[r] = y < height ? image[p * pol_stride + (int64_t) y * row_stride + x] : 0.0f;
            }
#pragma unroll
            for (int r = 0; r < ROWS; r++)
                if (v[r] == v[r])
                    acc += (double) v[r];
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
    int by = height < 64 ? height : 64;
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
    int by = height < 32 ? height : 32;
    dim3 grid(kimg_divup(width, 256), by, num_polarizations);
    image_nansum_kernel<<<grid, 256, 0, s>>>(image, row_stride, pol_stride, width, height, sums);
    return kimg_launch_status();
}

// ---- grid <-> image at w = 0, own transforms ------------------------------------------------
#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

namespace {

struct plan_entry { int device, size; fft_plan plan; };
std::mutex plan_mutex;
std::vector<plan_entry> plans;

// Radices of the stages for G cells (4s, then at most one 2, then 3s, 5s, 7s), or false when G has
// another prime factor
bool fft_stages(int G, fft_plan *plan)
{
    int twos = 0, rest = G, n = 0;
    while (rest % 2 == 0) {
        rest /= 2;
        twos++;
    }
    for (int i = 0; i < twos / 2; i++)
        plan->radix[n++] = 4;
    if (twos % 2)
        plan->radix[n++] = 2;
    for (int r : {3, 5, 7})
        while (rest % r == 0) {
            if (n >= 16)
                return false;
            rest /= r;
            plan->radix[n++] = (unsigned char) r;
        }
    plan->stages = n;
    return rest == 1 && n > 0;
}

// The plan for transforms of G cells on the current device: twiddles and digit reversal made once
// on the host (in double) and copied over -- the stream is waited for, so that every other stream
// may use them --, never freed.
int fft_plan_for(int G, hipStream_t s, fft_plan *out)
{
    int device = 0;
    KIMG_HIP(hipGetDevice(&device));
    std::lock_guard<std::mutex> lock(plan_mutex);
    for (const plan_entry &e : plans)
        if (e.device == device && e.size == G) {
            *out = e.plan;
            return 0;
        }
    fft_plan plan;
    std::memset(&plan, 0, sizeof(plan));
    if (!fft_stages(G, &plan))
        return KIMG_EUNSUPPORTED;
    std::vector<float2> tw;
    std::vector<int> radix2;        // the radix-4 stages as the two radix-2 steps they are
    int q = 1;
    for (int i = 0; i < plan.stages; i++) {
        const int r = plan.radix[i];
        for (int j = 0; j < q; j++) {
            const double angle = 2.0 * M_PI * (double) j / (double) (r * q);
            tw.push_back(make_float2((float) cos(angle), (float) sin(angle)));
        }
        q *= r;
        if (r == 4) {
            radix2.push_back(2);
            radix2.push_back(2);
        } else {
            radix2.push_back(r);
        }
    }
    std::vector<unsigned short> perm(G);
    for (int n = 0; n < G; n++) {
        int rem = n, block = G, pos = 0;
        for (int i = (int) radix2.size() - 1; i >= 0; i--) {
            block /= radix2[i];
            pos += (rem % radix2[i]) * block;
            rem /= radix2[i];
        }
        perm[n] = (unsigned short) pos;
    }
    float2 *d_tw = nullptr;
    unsigned short *d_perm = nullptr;
    KIMG_HIP(hipMalloc((void **) &d_tw, sizeof(float2) * tw.size()));
    hipError_t e = hipMalloc((void **) &d_perm, sizeof(unsigned short) * perm.size());
    if (e == hipSuccess)
        e = hipMemcpyAsync(d_tw, tw.data(), sizeof(float2) * tw.size(), hipMemcpyHostToDevice, s);
    if (e == hipSuccess)
        e = hipMemcpyAsync(d_perm, perm.data(), sizeof(unsigned short) * perm.size(),
                           hipMemcpyHostToDevice, s);
    if (e == hipSuccess)
        e = hipStreamSynchronize(s);        // (also: the host vectors go out of scope)
    if (e != hipSuccess) {
        (void) hipFree(d_tw);
        (void) hipFree(d_perm);
        return -(int) e;
    }
    plan.twiddle = d_tw;
    plan.perm = d_perm;
    plan.twiddles = (int) tw.size();
    if ((G & (G - 1)) == 0)
        while ((1 << plan.log2size) < G)
            plan.log2size++;
    plans.push_back(plan_entry{device, G, plan});
    *out = plan;
    return 0;
}

// (once per kernel, device and size: the attribute is set when a launch needs more dynamic LDS than
// any before it, not with every transform)
template<typename Kernel>
int fft_lds_attribute(Kernel kernel, size_t lds)
{
    if (lds <= 64 * 1024)
        return 0;
    return kimg_dynamic_lds(reinterpret_cast<const void *>(kernel), lds);
}

size_t fft_lds_bytes(int G, const fft_plan &plan)
{
    return sizeof(float2) * (size_t) (fft_lds_cells(G) + plan.twiddles);
}

} // namespace

extern "C" int kimg_grid_image_real_supported(int layer_size, int grid_size)
{
    fft_plan plan;
    // (even sizes: rows go in pairs; 160 KB of LDS hold the cells and the twiddles up to 8192)
    return layer_size >= 16 && layer_size <= 8192 && layer_size % 2 == 0 && fft_stages(layer_size, &plan)
        && grid_size >= 2 && grid_size % 2 == 0 && grid_size <= layer_size;
}

extern "C" size_t kimg_grid_image_real_workspace_bytes(int layer_size, int grid_size)
{
    if (!kimg_grid_image_real_supported(layer_size, grid_size))
        return 0;
    return sizeof(float2) * (size_t) (grid_size / 2 + 1) * (size_t) layer_size;
}

extern "C" int kimg_grid_to_image_real(float *image, int64_t image_row_stride, int layer_size,
                                       const void *grid, int64_t grid_row_stride, int grid_size,
                                       const float *kernel1d, float lm_scale, float lm_bias,
                                       int accumulate, void *workspace, size_t workspace_bytes,
                                       void *stream)
{
    KIMG_CHECK_ARG(image && grid && kernel1d && workspace);
    KIMG_CHECK_ARG(kimg_grid_image_real_supported(layer_size, grid_size));
    KIMG_CHECK_ARG(image_row_stride >= layer_size && grid_row_stride >= grid_size);
    KIMG_CHECK_ARG(workspace_bytes >= kimg_grid_image_real_workspace_bytes(layer_size, grid_size));
    KIMG_CHECK_ARG(((uintptr_t) workspace & 15) == 0);
    hipStream_t s = (hipStream_t) stream;
    const int G = layer_size;
    fft_plan plan;
    int rc = fft_plan_for(G, s, &plan);
    if (rc)
        return rc;
    const size_t lds = fft_lds_bytes(G, plan);
    float2 *T = static_cast<float2 *>(workspace);
    const float2 *g = static_cast<const float2 *>(grid);
    const int columns = grid_size / 2 + 1;
#define G2I(ODD) do { \
        if ((rc = fft_lds_attribute(&g2i_columns_kernel<ODD>, lds)) \
            || (rc = fft_lds_attribute(&g2i_rows_kernel<true, ODD>, lds)) \
            || (rc = fft_lds_attribute(&g2i_rows_kernel<false, ODD>, lds))) \
            return rc; \
        g2i_columns_kernel<ODD><<<columns, FFT_THREADS, lds, s>>>(T, g, grid_row_stride, grid_size, G, plan); \
        if (accumulate) \
            g2i_rows_kernel<true, ODD><<<G / 2, FFT_THREADS, lds, s>>>( \
                image, image_row_stride, T, grid_size, G, plan, kernel1d, lm_scale, lm_bias); \
        else \
            g2i_rows_kernel<false, ODD><<<G / 2, FFT_THREADS, lds, s>>>( \
                image, image_row_stride, T, grid_size, G, plan, kernel1d, lm_scale, lm_bias); \
    } while (0)
    if (plan.log2size)
        G2I(false);
    else
        G2I(true);
#undef G2I
    return kimg_launch_status();
}

extern "C" int kimg_image_to_grid_real(void *grid, int64_t grid_row_stride, int grid_size,
                                       const float *image, int64_t image_row_stride, int layer_size,
                                       const float *kernel1d, float lm_scale, float lm_bias,
                                       void *workspace, size_t workspace_bytes, void *stream)
{
    KIMG_CHECK_ARG(image && grid && kernel1d && workspace);
    KIMG_CHECK_ARG(kimg_grid_image_real_supported(layer_size, grid_size));
    KIMG_CHECK_ARG(image_row_stride >= layer_size && grid_row_stride >= grid_size);
    KIMG_CHECK_ARG(workspace_bytes >= kimg_grid_image_real_workspace_bytes(layer_size, grid_size));
    KIMG_CHECK_ARG(((uintptr_t) workspace & 15) == 0);
    hipStream_t s = (hipStream_t) stream;
    const int G = layer_size;
    fft_plan plan;
    int rc = fft_plan_for(G, s, &plan);
    if (rc)
        return rc;
    const size_t lds = fft_lds_bytes(G, plan);
    float2 *T = static_cast<float2 *>(workspace);
    const int columns = grid_size / 2 + 1;
#define I2G(ODD) do { \
        if ((rc = fft_lds_attribute(&i2g_rows_kernel<ODD>, lds)) \
            || (rc = fft_lds_attribute(&i2g_columns_kernel<ODD>, lds))) \
            return rc; \
        i2g_rows_kernel<ODD><<<G / 2, FFT_THREADS, lds, s>>>( \
            T, image, image_row_stride, grid_size, G, plan, kernel1d, lm_scale, lm_bias); \
        i2g_columns_kernel<ODD><<<columns, FFT_THREADS, lds, s>>>( \
            static_cast<float2 *>(grid), grid_row_stride, T, grid_size, G, plan); \
    } while (0)
    if (plan.log2size)
        I2G(false);
    else
        I2G(true);
#undef I2G
    return kimg_launch_status();
}

// ---- the same for any w: complex layer ------------------------------------------------------
extern "C" size_t kimg_grid_image_w_workspace_bytes(int layer_size, int grid_size)
{
    if (!kimg_grid_image_real_supported(layer_size, grid_size))
        return 0;
    return sizeof(float2) * (size_t) grid_size * (size_t) layer_size;
}

extern "C" int kimg_grid_to_image_w(float *image, int64_t image_row_stride, int layer_size,
                                    const void *grid, int64_t grid_row_stride, int grid_size,
                                    const float *kernel1d, float lm_scale, float lm_bias, float w,
                                    int accumulate, void *workspace, size_t workspace_bytes,
                                    void *stream)
{
    KIMG_CHECK_ARG(image && grid && kernel1d && workspace);
    KIMG_CHECK_ARG(kimg_grid_image_real_supported(layer_size, grid_size));
    KIMG_CHECK_ARG(image_row_stride >= layer_size && grid_row_stride >= grid_size);
    KIMG_CHECK_ARG(workspace_bytes >= kimg_grid_image_w_workspace_bytes(layer_size, grid_size));
    KIMG_CHECK_ARG(((uintptr_t) workspace & 15) == 0);
    hipStream_t s = (hipStream_t) stream;
    const int G = layer_size;
    fft_plan plan;
    int rc = fft_plan_for(G, s, &plan);
    if (rc)
        return rc;
    const size_t lds = fft_lds_bytes(G, plan);
    float2 *T = static_cast<float2 *>(workspace);
    const float2 *g = static_cast<const float2 *>(grid);
#define G2IW(ODD) do { \
        if ((rc = fft_lds_attribute(&g2iw_columns_kernel<ODD>, lds)) \
            || (rc = fft_lds_attribute(&g2iw_rows_kernel<true, ODD>, lds)) \
            || (rc = fft_lds_attribute(&g2iw_rows_kernel<false, ODD>, lds))) \
            return rc; \
        g2iw_columns_kernel<ODD><<<grid_size, FFT_THREADS, lds, s>>>(T, g, grid_row_stride, grid_size, G, plan); \
        if (accumulate) \
            g2iw_rows_kernel<true, ODD><<<G / 2, FFT_THREADS, lds, s>>>( \
                image, image_row_stride, T, grid_size, G, plan, kernel1d, lm_scale, lm_bias, w); \
        else \
            g2iw_rows_kernel<false, ODD><<<G / 2, FFT_THREADS, lds, s>>>( \
                image, image_row_stride, T, grid_size, G, plan, kernel1d, lm_scale, lm_bias, w); \
    } while (0)
    if (plan.log2size)
        G2IW(false);
    else
        G2IW(true);
#undef G2IW
    return kimg_launch_status();
}

extern "C" int kimg_image_to_grid_w(void *grid, int64_t grid_row_stride, int grid_size,
                                    const float *image, int64_t image_row_stride, int layer_size,
                                    const float *kernel1d, float lm_scale, float lm_bias, float w,
                                    void *workspace, size_t workspace_bytes, void *stream)
{
    KIMG_CHECK_ARG(image && grid && kernel1d && workspace);
    KIMG_CHECK_ARG(kimg_grid_image_real_supported(layer_size, grid_size));
    KIMG_CHECK_ARG(image_row_stride >= layer_size && grid_row_stride >= grid_size);
    KIMG_CHECK_ARG(workspace_bytes >= kimg_grid_image_w_workspace_bytes(layer_size, grid_size));
    KIMG_CHECK_ARG(((uintptr_t) workspace & 15) == 0);
    hipStream_t s = (hipStream_t) stream;
    const int G = layer_size;
    fft_plan plan;
    int rc = fft_plan_for(G, s, &plan);
    if (rc)
        return rc;
    const size_t lds = fft_lds_bytes(G, plan);
    float2 *T = static_cast<float2 *>(workspace);
#define I2GW(ODD) do { \
        if ((rc = fft_lds_attribute(&i2gw_rows_kernel<ODD>, lds)) \
            || (rc = fft_lds_attribute(&i2gw_columns_kernel<ODD>, lds))) \
            return rc; \
        i2gw_rows_kernel<ODD><<<G / 2, FFT_THREADS, lds, s>>>( \
            T, image, image_row_stride, grid_size, G, plan, kernel1d, lm_scale, lm_bias, w); \
        i2gw_columns_kernel<ODD><<<grid_size, FFT_THREADS, lds, s>>>( \
            static_cast<float2 *>(grid), grid_row_stride, T, grid_size, G, plan); \
    } while (0)
    if (plan.log2size)
        I2GW(false);
    else
        I2GW(true);
#undef I2GW
    return kimg_launch_status();
}

// ---- restoring-beam convolution on the library's own transforms -------------------------------
extern "C" int kimg_convolve_beam(float *image, int64_t row_stride, int size, float amplitude,
                                  float a, float b, float c, void *workspace,
                                  size_t workspace_bytes, void *stream)
{
    KIMG_CHECK_ARG(image && workspace && kimg_grid_image_real_supported(size, size));
    KIMG_CHECK_ARG(row_stride >= size && ((uintptr_t) workspace & 15) == 0);
    KIMG_CHECK_ARG(workspace_bytes >= sizeof(float2) * (size_t) (size / 2 + 1) * (size_t) size);
    hipStream_t s = (hipStream_t) stream;
    const int G = size;
    fft_plan plan;
    int rc = fft_plan_for(G, s, &plan);
    if (rc)
        return rc;
    const size_t lds = fft_lds_bytes(G, plan);
    float2 *T = static_cast<float2 *>(workspace);
#define CONVOLVE(ODD) do { \
        if ((rc = fft_lds_attribute(&cb_rows_forward_kernel<ODD>, lds)) \
            || (rc = fft_lds_attribute(&cb_columns_kernel<ODD>, lds)) \
            || (rc = fft_lds_attribute(&cb_rows_inverse_kernel<ODD>, lds))) \
            return rc; \
        cb_rows_forward_kernel<ODD><<<G / 2, FFT_THREADS, lds, s>>>(T, image, row_stride, G, plan); \
        cb_columns_kernel<ODD><<<G / 2 + 1, FFT_THREADS, lds, s>>>(T, G, plan, amplitude, a, b, c); \
        cb_rows_inverse_kernel<ODD><<<G / 2, FFT_THREADS, lds, s>>>(image, row_stride, T, G, plan); \
    } while (0)
    if (plan.log2size)
        CONVOLVE(false);
    else
        CONVOLVE(true);
#undef CONVOLVE
    return kimg_launch_status();
}

// (kimg_preload, api.hip)
KIMG_PRELOAD_THIS_UNIT(grid_to_half_layer_kernel)
