// Shared helpers for the libkimg HIP sources (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/kimg.h"

#define KIMG_CHECK_ARG(cond) do { if (!(cond)) return KIMG_EINVAL; } while (0)

// Return the negated hipError_t of the last launch on failure.
static inline int kimg_launch_status()
{
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int) e;
}

#define KIMG_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return -(int) e_; } while (0)

static inline int kimg_divup(int64_t a, int64_t b) { return (int) ((a + b - 1) / b); }

// CUs the window kernels (gridder, degridder) may fill with their resident workgroups
// (kimg_set_window_cus; api.hip)
int kimg_window_cus_now();
// (set for the duration of one kimg_grid / kimg_degrid call, on the calling thread)
struct kimg_window_cus_scope {
    explicit kimg_window_cus_scope(int cus);
    ~kimg_window_cus_scope();
    int before;
};

// hipFuncAttributeMaxDynamicSharedMemorySize of kernel `fn` on the current device raised to `bytes`
// (never lowered), once per kernel and device, under one lock (api.hip): channels imaged on several
// host threads reach their first launches together, and two threads setting the attribute of one
// kernel -- or one setting it while the other launches it -- is what their first pass must not do.
int kimg_dynamic_lds(const void *fn, size_t bytes);

// The stream graphs are captured on: one per host thread, the library's own.  A caller's stream is
// never put into capture mode -- streams can be shared between threads without either knowing
// (torch hands out the streams of a pool of 32 per device round robin), and what another thread
// launches on a capturing stream is recorded into the graph instead of run: its kernels never happen,
// and happen later, with its pointers, whenever the graph is launched.  (api.hip)
hipStream_t kimg_capture_stream();

// One kernel of every translation unit (= code object) of the library, for kimg_preload (api.hip):
// returns 0 so that a namespace-scope initialiser can call it.
int kimg_register_kernel(const void *fn);
// ... and a function that launches an empty kernel of the translation unit on the null stream: the
// one thing that is certain to make the runtime load the code object
int kimg_register_touch(void (*launch)());
#define KIMG_PRELOAD_THIS_UNIT(kernel) \
    namespace { __global__ void kimg_touch_kernel() {} \
                void kimg_touch_launch() { kimg_touch_kernel<<<1, 1, 0, nullptr>>>(); } } \
    static const int kimg_preload_registered = \
        kimg_register_kernel(reinterpret_cast<const void *>(&kernel)) + kimg_register_touch(&kimg_touch_launch);

// The multi-component form of the CLEAN loop (clean_multi.hip), reached through kimg_clean_cycles
int kimg_clean_multi_components(int patch_width, int patch_height, int tiles_x, int tiles_y);
size_t kimg_clean_multi_state_bytes(int tiles_x, int tiles_y);
int kimg_clean_multi_run(float *dirty, float *model, int64_t row_stride, int64_t pol_stride,
                         int width, int height, int num_polarizations, const float *psf,
                         int64_t psf_row_stride, int64_t psf_pol_stride, int psf_width,
                         int psf_height, int patch_width, int patch_height, int border, int mode,
                         float loop_gain, float threshold, float *tile_max, int32_t *tile_pos,
                         int tiles_x, int tiles_y, int max_cycles, int components, int repeats,
                         bool relative, double noise_threshold, double left_for_next,
                         void *state, float *log, hipStream_t s, int *cycles_done = nullptr,
                         float *first_peak = nullptr);

constexpr int WAVE = 64;    // gfx950 wavefront

// Wave-wide sum by DPP-backed shuffles; result valid in every lane.
__device__ inline float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_xor(v, off, WAVE);
    return v;
}

__device__ inline double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_xor(v, off, WAVE);
    return v;
}
