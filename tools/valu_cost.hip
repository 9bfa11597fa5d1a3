// Micro-benchmark: issue cost of individual VALU instructions on gfx950, alone and next to
// v_mfma_f32_32x32x2_f32 (W waves per SIMD, all CUs).  Each variant runs a loop of UNROLL
// independent instructions of one kind; reports cycles per instruction per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x
template <int KIND, bool MFMA>
__global__ __launch_bounds__(1024) void k(float *out, int iters, float a0)
{
    f32x16 acc;
    for (int j = 0; j < 16; j++) acc[j] = 0.f;
    float a = a0 + threadIdx.x;
    float r0 = a, r1 = a + 1, r2 = a + 2, r3 = a + 3, r4 = a + 4, r5 = a + 5, r6 = a + 6, r7 = a + 7;
    v2f p0 = {a, a}, p1 = {a, a + 1}, p2 = {a, a + 2}, p3 = {a, a + 3};
    unsigned long long q0 = threadIdx.x, q1 = q0 + 1, q2 = q0 + 2, q3 = q0 + 3;
    unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3;
    for (int it = 0; it < iters; it++) {
        if (MFMA)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a0, acc, 0, 0, 0);
        if (KIND == 0) {        // v_fma_f32 x8
            asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                         "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a0));
        } else if (KIND == 1) { // v_pk_mul_f32 x8
            asm volatile(REP8("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n") "" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p3));
        } else if (KIND == 2) { // v_lshl_add_u64 x8
            asm volatile("v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4\n"
                         "v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4\n"
                         : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(q3));
        } else if (KIND == 3) { // v_add_u32 x8
            asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n"
                         "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n"
                         : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u3));
        } else if (KIND == 4) { // v_mul_f32 x8
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a0));
        } else if (KIND == 5) { // v_pk_fma_f32 x8
            asm volatile("v_pk_fma_f32 %0, %0, %4, %0\n v_pk_fma_f32 %1, %1, %4, %1\n v_pk_fma_f32 %2, %2, %4, %2\n v_pk_fma_f32 %3, %3, %4, %3\n"
                         "v_pk_fma_f32 %0, %0, %4, %0\n v_pk_fma_f32 %1, %1, %4, %1\n v_pk_fma_f32 %2, %2, %4, %2\n v_pk_fma_f32 %3, %3, %4, %3\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p3));
        } else if (KIND == 6) { // v_readlane_b32 x8 (to SGPR)
            int s;
            asm volatile("v_readlane_b32 %0, %1, 3\n v_readlane_b32 %0, %1, 5\n v_readlane_b32 %0, %1, 7\n v_readlane_b32 %0, %1, 9\n"
                         "v_readlane_b32 %0, %1, 11\n v_readlane_b32 %0, %1, 13\n v_readlane_b32 %0, %1, 15\n v_readlane_b32 %0, %1, 17\n"
                         : "=s"(s) : "v"(u0));
            u1 += s;
        } else if (KIND == 7) { // v_cvt_pk_f16_f32 (split-form packing) x8
            asm volatile("v_cvt_pk_f16_f32 %0, %0, %4\n v_cvt_pk_f16_f32 %1, %1, %4\n v_cvt_pk_f16_f32 %2, %2, %4\n v_cvt_pk_f16_f32 %3, %3, %4\n"
                         "v_cvt_pk_f16_f32 %0, %0, %4\n v_cvt_pk_f16_f32 %1, %1, %4\n v_cvt_pk_f16_f32 %2, %2, %4\n v_cvt_pk_f16_f32 %3, %3, %4\n"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a0));
        } else if (KIND == 8) { // v_perm_b32 x8
            asm volatile("v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5\n"
                         "v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5\n"
                         : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u3), "v"(u2));
        }
    }
    float s = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + p0.x + p1.y + p2.x + p3.y + (float) (q0 + q1 + q2 + q3)
              + (float) (u0 + u1 + u2 + u3);
    for (int j = 0; j < 16; j++) s += acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, bool MFMA>
void run(float *out, const char *name)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves = 4; waves <= 16; waves *= 2) {
        const int iters = 4096;
        float ms = 0;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            k<KIND, MFMA><<<256, waves * 64>>>(out, iters, 1.0f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        const double ns_per_iter_per_simd = ms * 1e6 / ((double) iters * (waves / 4.0));
        printf("%-18s mfma %d waves/SIMD %d: %7.2f ns per (8 instr%s) per SIMD\n", name, (int) MFMA, waves / 4,
               ns_per_iter_per_simd, MFMA ? " + 1 MFMA" : "");
    }
}
int main()
{
    float *out; hipMalloc(&out, 256 * 1024 * 4);
    run<0, false>(out, "v_fma_f32"); run<0, true>(out, "v_fma_f32");
    run<4, false>(out, "v_mul_f32"); run<4, true>(out, "v_mul_f32");
    run<3, false>(out, "v_add_u32"); run<3, true>(out, "v_add_u32");
    run<1, false>(out, "v_pk_mul_f32"); run<1, true>(out, "v_pk_mul_f32");
    run<5, false>(out, "v_pk_fma_f32"); run<5, true>(out, "v_pk_fma_f32");
    run<2, false>(out, "v_lshl_add_u64"); run<2, true>(out, "v_lshl_add_u64");
    run<6, false>(out, "v_readlane_b32"); run<6, true>(out, "v_readlane_b32");
    run<7, false>(out, "v_cvt_pk_f16_f32"); run<7, true>(out, "v_cvt_pk_f16_f32");
    run<8, false>(out, "v_perm_b32"); run<8, true>(out, "v_perm_b32");
    return 0;
}
