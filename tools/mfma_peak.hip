// Micro-benchmark: sustained rate of v_mfma_f32_32x32x2_f32 on this chip (all CUs, W waves/SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(1024) void k(float *out, int iters, float a0, float b0)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; i++) for (int j = 0; j < 16; j++) acc[i][j] = 0.f;
    float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; i++) for (int j = 0; j < 16; j++) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main()
{
    float *out; hipMalloc(&out, 256 * 1024 * sizeof(float) * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves = 4; waves <= 16; waves *= 2) {
        const int iters = 2048;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            k<2><<<256, waves * 64>>>(out, iters, 1.0f, 0.5f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double mfma_per_simd = (double) iters * 2 * (waves / 4.0);
            double cyc = ms * 1e-3 / mfma_per_simd;
            if (rep == 2)
                printf("waves/CU %2d: %.3f ms, %.1f ns per MFMA per SIMD -> %.1f TFLOP/s; 64 cycles => %.2f GHz\n",
                       waves, ms, cyc * 1e9, 256.0 * 4 * 4096 / cyc / 1e12, 64 / cyc / 1e9);
        }
    }
    return 0;
}
