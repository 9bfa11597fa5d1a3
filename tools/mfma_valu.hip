// Micro-benchmark: how many independent VALU / LDS instructions fit beside each
// v_mfma_f32_32x32x2_f32 before the MFMA rate drops (W waves per SIMD, all CUs).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NV, int NL>
__global__ __launch_bounds__(1024) void k(float *out, int iters, float a0, float b0)
{
    __shared__ float2 lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = make_float2(i, -i);
    __syncthreads();
    f32x16 acc0, acc1;
    for (int j = 0; j < 16; j++) { acc0[j] = 0.f; acc1[j] = 0.f; }
    float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    float v[16];
    for (int i = 0; i < 16; i++) v[i] = a * i;
    int addr = (threadIdx.x * 8) & 4095;
    for (int it = 0; it < iters; it++) {
        float2 l[NL > 0 ? NL : 1];
#pragma unroll
        for (int i = 0; i < NL; i++) l[i] = lds[(addr + i * 64 + it) & 4095];
#pragma unroll
        for (int i = 0; i < NV; i++) v[i & 15] = fmaf(v[i & 15], 1.0001f, b);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NV; i++) v[(i + 5) & 15] = fmaf(v[(i + 5) & 15], 0.9999f, a);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NL; i++) v[i] += l[i].x;
    }
    float s = 0;
    for (int j = 0; j < 16; j++) s += acc0[j] + acc1[j] + v[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NV, int NL>
void run(float *out)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves = 4; waves <= 16; waves *= 2) {
        const int iters = 1024;
        float ms = 0;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            k<NV, NL><<<256, waves * 64>>>(out, iters, 1.0f, 0.5f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        double per = ms * 1e6 / ((double) iters * 2 * (waves / 4.0));
        printf("VALU/MFMA %2d LDS/2MFMA %d waves/SIMD %d: %.1f ns per MFMA per SIMD\n", NV, NL, waves / 4, per);
    }
}
int main()
{
    float *out; hipMalloc(&out, 256 * 1024 * 4);
    run<0, 0>(out); run<4, 0>(out); run<8, 0>(out); run<12, 0>(out); run<16, 0>(out); run<24, 0>(out);
    run<8, 4>(out); run<8, 8>(out); run<8, 12>(out);
    return 0;
}
