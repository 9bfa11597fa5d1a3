// Separable anti-aliasing x W-correction kernel table, generated on the device.
//
// Replaces the host construction of grid.py:235-334 (antialias_w_kernel, called per W plane
// by ConvolutionKernel.__init__ :358-389).  On the host the table costs 3 ms at W=32, K=28
// and 73 ms at W=400, K=60 (the reference's default W step gives hundreds of planes) — more than
// imaging the channel takes on the device — and it is needed once per channel.
//
// One workgroup per (W plane, tile of 256 outputs):
//   1. the image-space samples of the plane, img[n] = aa(l_n) exp(2 pi i frac(phase_w(l_n))), l_n =
//      (n - n_img/2) step, and the n_img-th roots of unity go to LDS (float64);
//   2. each thread evaluates one output of the centred DFT directly,
//      U[f] = sum_n img[n] exp(-2 pi i f (n - n_img/2) / n_img), f = j - n_out/2,
//      which is what fft(ifftshift(img)) cropped to the central n_out samples computes;
//   3. the sub-pixel phases are de-interleaved (reversed) on the store:
//      table[w][s][t] = step * U[t OV + (OV - 1 - s) - n_out/2], rounded to complex64.
// Everything is float64 like the reference; the direct sum's rounding error (~1e-14 of the
// peak) is far below the float32 rounding of the result.
#include "kimg_common.h"

namespace {

constexpr int KT_THREADS = 256;

// np.sinc(sqrt(q)) continued analytically to q < 0 (grid.py:181-184)
__device__ double sinc_sqrt(double q)
{
    const double pi = 3.14159265358979323846;
    if (q == 0.0)
        return 1.0;
    if (q > 0.0) {
        double r = sqrt(q);
        return sinpi(r) / (pi * r);
    }
    double r = pi * sqrt(-q);
    return sinh(r) / r;
}

__global__ __launch_bounds__(KT_THREADS)
void kernel_table_kernel(float2 *__restrict__ table, const double *__restrict__ ws,
                         int width, int oversample, int n_img, double cell_wavelengths,
                         double step, double antialias_width, double beta_over_pi_sq,
                         double kbf_scale, double half_subcell)
{
    extern __shared__ double2 kt_lds[];
    double2 *img = kt_lds;            // [n_img]
    double2 *tw = kt_lds + n_img;     // [n_img]: exp(-2 pi i k / n_img)
    const int n_out = width * oversample;
    const double w = ws[blockIdx.x];
    for (int n = threadIdx.x; n < n_img; n += KT_THREADS) {
        double l = (double) (n - n_img / 2) * step;
        double f = l * cell_wavelengths;
        double q = (antialias_width * f) * (antialias_width * f) - beta_over_pi_sq;
        double aa = cell_wavelengths * (kbf_scale * sinc_sqrt(q));
        double l2 = l * l;
        double phase = (-w) * (-0.5 * l2 - (5.0 / 24.0) * l2 * l2) + half_subcell * l;
        phase -= rint(phase);
        double s, c;
        sincospi(2.0 * phase, &s, &c);
        img[n] = make_double2(aa * c, aa * s);
        sincospi(2.0 * (double) n / (double) n_img, &s, &c);
        tw[n] = make_double2(c, -s);
    }
    __syncthreads();
    const int j = blockIdx.y * KT_THREADS + threadIdx.x;
    if (j >= n_out)
        return;
    const int f = j - n_out / 2;
    // index of exp(-2 pi i f (n - n_img/2) / n_img), advanced by f per sample
    const int fm = ((f % n_img) + n_img) % n_img;
    int idx = (int) ((((long long) f * (-(n_img / 2))) % n_img + n_img) % n_img);
    double re = 0.0, im = 0.0;
    for (int n = 0; n < n_img; n++) {
        double2 x = img[n], t = tw[idx];
        re = fma(x.x, t.x, re);
        re = fma(-x.y, t.y, re);
        im = fma(x.x, t.y, im);
        im = fma(x.y, t.x, im);
        idx += fm;
        idx -= idx >= n_img ? n_img : 0;
    }
    const int t_ = j / oversample, s_ = oversample - 1 - j % oversample;
    table[((size_t) blockIdx.x * oversample + s_) * width + t_] =
        make_float2((float) (re * step), (float) (im * step));
}

// I0(x) by its power series (all terms positive: no cancellation)
double bessel_i0(double x)
{
    double h = 0.25 * x * x, term = 1.0, sum = 1.0;
    for (int k = 1; k < 500; k++) {
        term *= h / ((double) k * (double) k);
        sum += term;
        if (term < 1e-17 * sum)
            break;
    }
    return sum;
}

} // namespace

extern "C" int kimg_kernel_table(void *table, const double *ws, int w_planes, int kernel_width,
                                 int oversample, int image_oversample, double cell_wavelengths,
                                 double antialias_width, double beta, void *stream)
{
    KIMG_CHECK_ARG(table && ws && w_planes > 0 && kernel_width > 0 && oversample > 0
                   && image_oversample > 0 && cell_wavelengths > 0 && antialias_width > 0);
    const long long n_out = (long long) kernel_width * oversample;
    KIMG_CHECK_ARG(n_out % 2 == 0);                      // grid.py:268
    const long long n_img = n_out * image_oversample;
    const size_t lds = 2 * sizeof(double2) * (size_t) n_img;
    if (lds > 160 * 1024)
        return KIMG_EUNSUPPORTED;
    const double pi = 3.14159265358979323846;
    double step = 1.0 / (kernel_width * cell_wavelengths * image_oversample);
    if (lds > 64 * 1024) {
        const int rc = kimg_dynamic_lds((const void *) kernel_table_kernel, lds);
        if (rc)
            return rc;
    }
    dim3 grid(w_planes, (unsigned) ((n_out + KT_THREADS - 1) / KT_THREADS));
    hipLaunchKernelGGL(kernel_table_kernel, grid, dim3(KT_THREADS), lds, (hipStream_t) stream,
                       (float2 *) table, ws, kernel_width, oversample, (int) n_img,
                       cell_wavelengths, step, antialias_width, (beta / pi) * (beta / pi),
                       antialias_width / bessel_i0(beta), -0.5 * cell_wavelengths / oversample);
    return kimg_launch_status();
}

// (kimg_preload, api.hip)
KIMG_PRELOAD_THIS_UNIT(kernel_table_kernel)
