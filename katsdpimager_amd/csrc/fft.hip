// 2-D complex-to-complex FFT plans on rocFFT (through the hipFFT front end).
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
