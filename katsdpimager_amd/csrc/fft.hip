// 2-D FFT plans on rocFFT (through the hipFFT front end): complex-to-complex for the
// grid <-> image transforms, real-to-complex / complex-to-real for the restoring-beam
// convolution (beam.py:323-349).
// Replaces katsdpsigproc.fft.FftTemplate as used by GridImageTemplate.make_fft_plan
// (image.py:585-600) and the transforms at image.py:629 (inverse) and :698 (forward):
// in place, unnormalised, complex64, row-major size_y x size_x.
#include "kimg_common.h"
#include <hipfft/hipfft.h>

namespace {
struct fft_plan {
    hipfftHandle handle;
    hipStream_t stream;
};

struct rfft_plan {
    hipfftHandle forward, inverse;
    hipStream_t stream;
};

int fft_status(hipfftResult r)
{
    // hipFFT result codes are small positive ints; keep them apart from hipError_t space
    return r == HIPFFT_SUCCESS ? 0 : -(20000 + (int) r);
}
} // namespace

extern "C" int kimg_fft_plan_create(void **plan, int size_y, int size_x)
{
    KIMG_CHECK_ARG(plan && size_y > 0 && size_x > 0);
    fft_plan *p = new fft_plan;
    p->stream = nullptr;
    hipfftResult r = hipfftPlan2d(&p->handle, size_y, size_x, HIPFFT_C2C);
    if (r != HIPFFT_SUCCESS) {
        delete p;
        return fft_status(r);
    }
    *plan = p;
    return 0;
}

extern "C" int kimg_fft_exec(void *plan, void *layer, int direction, void *stream)
{
    KIMG_CHECK_ARG(plan && layer && (direction == 1 || direction == -1));
    fft_plan *p = static_cast<fft_plan *>(plan);
    hipStream_t s = (hipStream_t) stream;
    if (s != p->stream) {
        hipfftResult r = hipfftSetStream(p->handle, s);
        if (r != HIPFFT_SUCCESS)
            return fft_status(r);
        p->stream = s;
    }
    return fft_status(hipfftExecC2C(p->handle, (hipfftComplex *) layer, (hipfftComplex *) layer,
                                    direction == 1 ? HIPFFT_BACKWARD : HIPFFT_FORWARD));
}

extern "C" int kimg_fft_plan_destroy(void *plan)
{
    if (!plan)
        return 0;
    fft_plan *p = static_cast<fft_plan *>(plan);
    hipfftResult r = hipfftDestroy(p->handle);
    delete p;
    return fft_status(r);
}

// ---- real <-> half-complex, out of place: image float32 [H][W] <-> fourier complex64 [H][W/2+1]
extern "C" int kimg_rfft_plan_create(void **plan, int height, int width)
{
    KIMG_CHECK_ARG(plan && height > 0 && width > 0);
    rfft_plan *p = new rfft_plan;
    p->stream = nullptr;
    hipfftResult r = hipfftPlan2d(&p->forward, height, width, HIPFFT_R2C);
    if (r != HIPFFT_SUCCESS) {
        delete p;
        return fft_status(r);
    }
    r = hipfftPlan2d(&p->inverse, height, width, HIPFFT_C2R);
    if (r != HIPFFT_SUCCESS) {
        hipfftDestroy(p->forward);
        delete p;
        return fft_status(r);
    }
    *plan = p;
    return 0;
}

extern "C" int kimg_rfft_exec(void *plan, float *image, void *fourier, int direction, void *stream)
{
    KIMG_CHECK_ARG(plan && image && fourier && (direction == 1 || direction == -1));
    rfft_plan *p = static_cast<rfft_plan *>(plan);
    hipStream_t s = (hipStream_t) stream;
    if (s != p->stream) {
        hipfftResult r = hipfftSetStream(p->forward, s);
        if (r == HIPFFT_SUCCESS)
            r = hipfftSetStream(p->inverse, s);
        if (r != HIPFFT_SUCCESS)
            return fft_status(r);
        p->stream = s;
    }
    if (direction == -1)
        return fft_status(hipfftExecR2C(p->forward, image, (hipfftComplex *) fourier));
    return fft_status(hipfftExecC2R(p->inverse, (hipfftComplex *) fourier, image));
}

extern "C" int kimg_rfft_plan_destroy(void *plan)
{
    if (!plan)
        return 0;
    rfft_plan *p = static_cast<rfft_plan *>(plan);
    hipfftResult r = hipfftDestroy(p->forward);
    hipfftResult r2 = hipfftDestroy(p->inverse);
    delete p;
    return fft_status(r != HIPFFT_SUCCESS ? r : r2);
}
