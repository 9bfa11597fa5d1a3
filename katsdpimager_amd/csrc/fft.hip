// 2-D FFT plans on rocFFT, called directly (rounds 1-3 went through the hipFFT front end, a cuFFT-shaped
// layer over the same library): complex-to-complex for the grid <-> image transforms of the sizes the
// library's own transforms do not take (layers above 8192 or with a prime factor above 7, image.hip), and
// real-to-complex / complex-to-real for the restoring-beam convolution (beam.py:323-349) and the w = 0
// routes on the library's plan.
// Replaces katsdpsigproc.fft.FftTemplate as used by GridImageTemplate.make_fft_plan
// (image.py:585-600) and the transforms at image.py:629 (inverse) and :698 (forward):
// unnormalised, complex64 / float32, row-major size_y x size_x.
#include "kimg_common.h"
#include <rocfft/rocfft.h>
#include <mutex>

namespace {

// rocFFT status codes are small positive ints; keep them apart from the hipError_t space
int fft_status(rocfft_status r)
{
    return r == rocfft_status_success ? 0 : -(20000 + (int) r);
}

std::once_flag setup_once;

void library_setup()
{
    std::call_once(setup_once, []() { (void) rocfft_setup(); });
}

// One rocFFT plan with what it needs to run: its execution info and work buffer (allocated with the
// plan: the only device memory this library allocates).
struct roc_plan {
    rocfft_plan plan = nullptr;
    rocfft_execution_info info = nullptr;
    void *work = nullptr;

    // lengths: {fastest, slowest}.  Strides (in elements of the respective type) describe rows that
    // are longer than the data, as the in-place real transforms need; 0 = dense.
    int create(rocfft_result_placement placement, rocfft_transform_type type, size_t size_x, size_t size_y,
               size_t in_row_stride, size_t out_row_stride)
    {
        library_setup();
        const size_t lengths[2] = {size_x, size_y};
        rocfft_plan_description desc = nullptr;
        rocfft_status r = rocfft_status_success;
        if (in_row_stride || out_row_stride) {
            const bool forward_real = type == rocfft_transform_type_real_forward;
            const size_t in_strides[2] = {1, in_row_stride}, out_strides[2] = {1, out_row_stride};
            r = rocfft_plan_description_create(&desc);
            if (r == rocfft_status_success)
                r = rocfft_plan_description_set_data_layout(
                    desc, forward_real ? rocfft_array_type_real : rocfft_array_type_hermitian_interleaved,
                    forward_real ? rocfft_array_type_hermitian_interleaved : rocfft_array_type_real, nullptr,
                    nullptr, 2, in_strides, in_row_stride * size_y, 2, out_strides, out_row_stride * size_y);
        }
        if (r == rocfft_status_success)
            r = rocfft_plan_create(&plan, placement, type, rocfft_precision_single, 2, lengths, 1, desc);
        if (desc)
            (void) rocfft_plan_description_destroy(desc);
        if (r != rocfft_status_success)
            return fft_status(r);
        r = rocfft_execution_info_create(&info);
        if (r != rocfft_status_success)
            return fft_status(r);
        size_t bytes = 0;
        r = rocfft_plan_get_work_buffer_size(plan, &bytes);
        if (r != rocfft_status_success)
            return fft_status(r);
        if (bytes) {
            hipError_t e = hipMalloc(&work, bytes);
            if (e != hipSuccess)
                return -(int) e;
            r = rocfft_execution_info_set_work_buffer(info, work, bytes);
        }
        return fft_status(r);
    }

    int execute(void *in, void *out, hipStream_t s)
    {
        rocfft_status r = rocfft_execution_info_set_stream(info, s);
        if (r != rocfft_status_success)
            return fft_status(r);
        void *ins[1] = {in}, *outs[1] = {out};
        return fft_status(rocfft_execute(plan, ins, out ? outs : nullptr, info));
    }

    void destroy()
    {
        if (info)
            (void) rocfft_execution_info_destroy(info);
        if (plan)
            (void) rocfft_plan_destroy(plan);
        if (work)
            (void) hipFree(work);
        plan = nullptr;
        info = nullptr;
        work = nullptr;
    }
};

struct fft_plan {
    roc_plan forward, inverse;          // in place
};

// (made on first use: most callers use one placement only)
struct rfft_plan {
    int height, width;
    roc_plan forward, inverse;          // out of place, dense rows
    roc_plan forward_in, inverse_in;    // in place: real rows of 2 (width / 2 + 1) floats
    std::mutex mutex;
};

} // namespace

extern "C" int kimg_fft_plan_create(void **plan, int size_y, int size_x)
{
    KIMG_CHECK_ARG(plan && size_y > 0 && size_x > 0);
    fft_plan *p = new fft_plan;
    int rc = p->forward.create(rocfft_placement_inplace, rocfft_transform_type_complex_forward, size_x,
                               size_y, 0, 0);
    if (rc == 0)
        rc = p->inverse.create(rocfft_placement_inplace, rocfft_transform_type_complex_inverse, size_x,
                               size_y, 0, 0);
    if (rc) {
        p->forward.destroy();
        p->inverse.destroy();
        delete p;
        return rc;
    }
    *plan = p;
    return 0;
}

extern "C" int kimg_fft_exec(void *plan, void *layer, int direction, void *stream)
{
    KIMG_CHECK_ARG(plan && layer && (direction == 1 || direction == -1));
    fft_plan *p = static_cast<fft_plan *>(plan);
    return (direction == 1 ? p->inverse : p->forward).execute(layer, nullptr, (hipStream_t) stream);
}

extern "C" int kimg_fft_plan_destroy(void *plan)
{
    if (!plan)
        return 0;
    fft_plan *p = static_cast<fft_plan *>(plan);
    p->forward.destroy();
    p->inverse.destroy();
    delete p;
    return 0;
}

// ---- real <-> half-complex: image float32 [H][W] <-> fourier complex64 [H][W/2+1], out of place with
// dense rows, or in place (image == fourier) with real rows of W + 2 floats
extern "C" int kimg_rfft_plan_create(void **plan, int height, int width)
{
    KIMG_CHECK_ARG(plan && height > 0 && width > 0);
    rfft_plan *p = new rfft_plan;
    p->height = height;
    p->width = width;
    // (the out-of-place pair now, so that a size rocFFT does not take is refused here)
    int rc = p->forward.create(rocfft_placement_notinplace, rocfft_transform_type_real_forward, width,
                               height, 0, 0);
    if (rc == 0)
        rc = p->inverse.create(rocfft_placement_notinplace, rocfft_transform_type_real_inverse, width,
                               height, 0, 0);
    if (rc) {
        p->forward.destroy();
        p->inverse.destroy();
        delete p;
        return rc;
    }
    *plan = p;
    return 0;
}

extern "C" int kimg_rfft_exec(void *plan, float *image, void *fourier, int direction, void *stream)
{
    KIMG_CHECK_ARG(plan && image && fourier && (direction == 1 || direction == -1));
    rfft_plan *p = static_cast<rfft_plan *>(plan);
    hipStream_t s = (hipStream_t) stream;
    if (static_cast<void *>(image) != fourier) {
        if (direction == -1)
            return p->forward.execute(image, fourier, s);
        return p->inverse.execute(fourier, image, s);
    }
    // in place: rows of width / 2 + 1 complex cells = width + 2 floats
    const size_t half = (size_t) p->width / 2 + 1;
    roc_plan &rp = direction == -1 ? p->forward_in : p->inverse_in;
    {
        std::lock_guard<std::mutex> lock(p->mutex);
        if (!rp.plan) {
            const int rc = direction == -1
                ? rp.create(rocfft_placement_inplace, rocfft_transform_type_real_forward, p->width,
                            p->height, 2 * half, half)
                : rp.create(rocfft_placement_inplace, rocfft_transform_type_real_inverse, p->width,
                            p->height, half, 2 * half);
            if (rc) {
                rp.destroy();
                return rc;
            }
        }
    }
    return rp.execute(fourier, nullptr, s);
}

extern "C" int kimg_rfft_plan_destroy(void *plan)
{
    if (!plan)
        return 0;
    rfft_plan *p = static_cast<rfft_plan *>(plan);
    p->forward.destroy();
    p->inverse.destroy();
    p->forward_in.destroy();
    p->inverse_in.destroy();
    delete p;
    return 0;
}
