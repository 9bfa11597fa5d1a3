// Feasibility micro-benchmark: gridding update with fp32 operands split into three bf16 pieces
// and contracted by v_mfma_f32_32x32x16_bf16 (6 cross terms x (re, im) = 12 of 16 k-slots),
// against the exact-fp32 v_mfma_f32_32x32x2_f32 formulation.  Per "visibility": operand LDS
// reads, a = c.kv, split/pack of a and of two b values, 2 MFMAs.
// MODE 2: fp16 hi/lo split (hi*hi + hi*lo + lo*hi for re and im = 6 k-slots per visibility) with
// TWO visibilities per v_mfma_f32_32x32x16_f16: lanes 0-31 (k 0..7) carry one visibility, lanes
// 32-63 (k 8..15) the other; the column operand comes pre-split from the table (re_hi, re_lo,
// im_hi, im_lo per tap, the same 8 bytes) and is arranged with three v_perm and two sign flips.
// Throughput only: operand values are not meaningful.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ inline unsigned cvt_pk_f16(float lo, float hi)
{
    unsigned r;
    asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

__device__ inline float f16_lo_to_f32(unsigned packed)
{
    float r;
    asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(r) : "v"(packed));
    return r;
}

__device__ inline float f16_hi_to_f32(unsigned packed)
{
    float r;
    asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(r) : "v"(packed));
    return r;
}

__device__ inline unsigned cvt_pk_bf16(float lo, float hi)
{
    unsigned r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

// x -> [x1, x1, x1, x2 | x2, x3, 0, 0] (pattern A) or [x1, x2, x3, x1 | x2, x1, 0, 0] (pattern B)
template <bool PAT_A>
__device__ inline bf16x8 split3(float x)
{
    unsigned p11 = cvt_pk_bf16(x, x);                       // (x1, x1)
    float r1 = x - __uint_as_float(p11 << 16);
    unsigned p21 = cvt_pk_bf16(r1, x);                      // (x2, x1)
    float r2 = r1 - __uint_as_float(p21 << 16);
    unsigned p32 = cvt_pk_bf16(r2, r1);                     // (x3, x2)
    u32x4 v;
    if (PAT_A) {
        v[0] = p11;                                         // x1 x1
        v[1] = __builtin_amdgcn_perm(p21, p11, 0x01000504); // (x1, x2): lo = x1, hi = x2
        v[2] = p32 >> 16 | (p32 << 16);                     // (x2, x3)
        v[3] = 0;
    } else {
        v[0] = __builtin_amdgcn_perm(p21, p11, 0x01000504); // (x1, x2)
        v[1] = __builtin_amdgcn_perm(p11, p32, 0x01000504); // (x3, x1)
        v[2] = p21;                                         // (x2, x1)
        v[3] = 0;
    }
    return __builtin_bit_cast(bf16x8, v);
}

template <int MODE>     // 0: fp32 MFMA (current kernel's inner work), 1: bf16 split
__global__ __launch_bounds__(768) void k(float *out, int iters)
{
    __shared__ float2 table[8192];
    __shared__ float4 samples[64];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) table[i] = make_float2(1.0f + i * 1e-4f, 0.5f - i * 1e-4f);
    if (threadIdx.x < 64) samples[threadIdx.x] = make_float4(threadIdx.x, 1.0f, 1.0f, -(float) threadIdx.x);
    __syncthreads();
    f32x16 acc0, acc1;
    for (int j = 0; j < 16; j++) { acc0[j] = 0.f; acc1[j] = 0.f; }
    const int lane = threadIdx.x & 63;
    const unsigned char *tb = reinterpret_cast<const unsigned char *>(table);
    const int lane_v = (lane & 31) * 8, lane_u = ((lane & 31) >> 1) * 8 + ((lane & 1) ? 4 : 0);
    const float sign = (lane & 33) == 1 ? -1.0f : 1.0f;
    // MODE 2: per-parity byte selectors and sign masks of the column operand
    const int lane_u2 = ((lane & 31) >> 1) * 8;
    const bool odd = lane & 1;
    const unsigned sel0 = odd ? 0x07060504u : 0x03020100u;
    const unsigned sel1 = odd ? 0x01000504u : 0x05040100u;
    const unsigned sel2 = odd ? 0x01000302u : 0x05040706u;
    const unsigned sgn1 = odd ? 0u : 0x80000000u, sgn2 = odd ? 0u : 0x80008000u;
    int rv = (threadIdx.x >> 6) * 512, ru = rv + 256;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int t = 0; t < 4; t++) {
            rv = (rv + 520) & 0xfe00; ru = (ru + 1544) & 0xfe00;
            const float2 kv = *reinterpret_cast<const float2 *>(tb + rv + lane_v);
            const float b0 = *reinterpret_cast<const float *>(tb + ru + lane_u) * sign;
            const float b1 = *reinterpret_cast<const float *>(tb + ru + lane_u + 128) * sign;
            const float2 c = *reinterpret_cast<const float2 *>(
                reinterpret_cast<const unsigned char *>(samples + ((it * 4 + t) & 63)) + (lane >= 32 ? 8 : 0));
            const float a = fmaf(c.x, kv.x, c.y * kv.y);
            if (MODE == 2) {
                // this lane's visibility: both components of a, split and packed into 6 k-slots
                const float a_im = fmaf(c.y, kv.x, -c.x * kv.y);
                u32x4 A;
                A[0] = cvt_pk_f16(a, a);                                    // (re_hi, re_hi)
                const float re_lo = a - f16_lo_to_f32(A[0]);
                A[1] = cvt_pk_f16(re_lo, a_im);                             // (re_lo, im_hi)
                const float im_lo = a_im - f16_hi_to_f32(A[1]);
                A[2] = cvt_pk_f16(a_im, im_lo);                             // (im_hi, im_lo)
                A[3] = 0;
                // column operands of the two 16-column tiles from pre-split table entries
                const uint2 t0 = *reinterpret_cast<const uint2 *>(tb + ru + lane_u2);
                const uint2 t1 = *reinterpret_cast<const uint2 *>(tb + ru + lane_u2 + 128);
                u32x4 B0, B1;
                B0[0] = __builtin_amdgcn_perm(t0.y, t0.x, sel0);
                B0[1] = __builtin_amdgcn_perm(t0.y, t0.x, sel1) ^ sgn1;
                B0[2] = __builtin_amdgcn_perm(t0.y, t0.x, sel2) ^ sgn2;
                B0[3] = 0;
                B1[0] = __builtin_amdgcn_perm(t1.y, t1.x, sel0);
                B1[1] = __builtin_amdgcn_perm(t1.y, t1.x, sel1) ^ sgn1;
                B1[2] = __builtin_amdgcn_perm(t1.y, t1.x, sel2) ^ sgn2;
                B1[3] = 0;
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A),
                                                              __builtin_bit_cast(f16x8, B0), acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A),
                                                              __builtin_bit_cast(f16x8, B1), acc1, 0, 0, 0);
            } else if (MODE == 0) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
            } else {
                const bf16x8 A = split3<true>(a);
                const bf16x8 B0 = split3<false>(b0);
                const bf16x8 B1 = split3<false>(b1);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B1, acc1, 0, 0, 0);
            }
        }
    }
    float s = 0;
    for (int j = 0; j < 16; j++) s += acc0[j] + acc1[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(float *out, const char *name)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves = 4; waves <= 12; waves += 4) {
        const int iters = 2048;
        float ms = 0;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            k<MODE><<<256, waves * 64>>>(out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        // "visibilities" per SIMD: iters * 4 * (waves / 4), twice that when an MFMA pair covers two
        double ns = ms * 1e6 / ((double) iters * 4 * (MODE == 2 ? 2 : 1) * (waves / 4.0));
        printf("%s waves/SIMD %d: %.1f ns per visibility per SIMD -> %.2f Gvis/s chip\n", name, waves / 4, ns,
               1024.0 / ns);
    }
}

int main()
{
    float *out; hipMalloc(&out, 256 * 768 * 4);
    run<0>(out, "fp32 mfma 32x32x2 ");
    run<1>(out, "bf16x3 split 32x32x16");
    run<2>(out, "fp16x2 split, 2 vis/MFMA");
    return 0;
}
