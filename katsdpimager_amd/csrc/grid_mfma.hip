// MFMA window gridder for gfx950 (CDNA4) -- the hot kernel of the path.
//
// Replaces Gridder.static_run / grid.mako:63-197 of the reference for kernel widths <= 32.
// Same result as GridderHost/_grid (grid.py:1032-1052):
//     grid[p][v0+j][u0+k] += vis[p]*wgt[p] * conj(kern[w][sv][j]) * conj(kern[w][su][k]).
//
// Design (CDNA4-first, not the reference's thread-per-cell Romein kernel):
//  * A wave owns a moving 32x32 window of the grid held in MFMA accumulators
//    (v_mfma_f32_32x32x2_f32: exact fp32, 16 VGPRs for Re + 16 for Im per polarization).
//    Cell (r, c) of the window is grid point (Wv + ((r-Wv)&31), Wu + ((c-Wu)&31)), so the
//    window slides without moving data; cells are flushed with float atomics only when
//    their mapping changes (the window outran them) and once at the end of the wave's range.
//  * A visibility's K x K update is a rank-1 outer product a (x) b with
//    a[j] = vis*wgt*conj(kv[j]), b[k] = conj(ku[k]).  Two visibilities form the K=2
//    dimension of one MFMA: D[32x32] += A[32x2] * B[2x32]; complex = 4 real MFMAs.
//    All per-tap arithmetic therefore runs on the matrix pipe; the VALU only prepares
//    one A and one B element per lane per visibility pair.
//  * The separable kernel table (W x OV rows, padded to 32 taps) lives in LDS; visibility
//    records are staged per wave through LDS in batches of 64.
//  * Window slack 32-K lets consecutive visibilities whose footprints differ by a few
//    cells share accumulators without any flush.
//
// Algorithmic work per visibility: 8*K*K*P flop (complex MAC per tap) + 6*K*P (a-vector);
// executed: 4 MFMA x 2048 MAC per visibility pair per polarization (32x32 window).
#include "kimg_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WIN = 32;

template <int P>
struct window_acc {
    f32x16 re[P];
    f32x16 im[P];
};

// Flush (atomically add and clear) every accumulator cell whose grid mapping differs
// between window origin (Wu, Wv) and (nWu, nWv); `full` flushes everything.
template <int P>
__device__ inline void flush_window(window_acc<P> &acc, float *__restrict__ grid,
                                    int64_t row_stride, int64_t pol_stride, int Gg,
                                    int Wu, int Wv, int nWu, int nWv, bool full, int lane)
{
    const int c = lane & 31;
    const int h4 = (lane >> 5) * 4;
    const int xold = Wu + ((c - Wu) & 31);
    const int xnew = nWu + ((c - nWu) & 31);
    const bool col_changed = full || (xold != xnew);
    const bool x_ok = (unsigned) xold < (unsigned) Gg;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int r = (k & 3) + 8 * (k >> 2) + h4;      // MFMA 32x32 C/D row of register k
        const int yold = Wv + ((r - Wv) & 31);
        const int ynew = nWv + ((r - nWv) & 31);
        if (col_changed || yold != ynew) {
            const bool ok = x_ok && (unsigned) yold < (unsigned) Gg;
            float *cell = grid + 2 * ((int64_t) yold * row_stride + xold);
#pragma unroll
            for (int p = 0; p < P; p++) {
                const float vr = acc.re[p][k], vi = acc.im[p][k];
                if (ok && (vr != 0.0f || vi != 0.0f)) {
                    atomicAdd(cell + 2 * p * pol_stride, vr);
                    atomicAdd(cell + 2 * p * pol_stride + 1, vi);
                }
                acc.re[p][k] = 0.0f;
                acc.im[p][k] = 0.0f;
            }
        }
    }
}

template <int P, int NW>
__global__ __launch_bounds__(NW * 64) void grid_mfma_kernel(
    float *__restrict__ grid, int64_t row_stride, int64_t pol_stride, int Gg,
    const float *__restrict__ weights_grid, int64_t wg_row_stride, int64_t wg_pol_stride,
    const int16_t *__restrict__ uv, const int16_t *__restrict__ w_plane,
    const float2 *__restrict__ vis, int64_t num_vis,
    const float2 *__restrict__ kern, int W, int OV, int K, int64_t vis_per_wave)
{
    extern __shared__ __align__(16) unsigned char smem[];
    float2 *table = reinterpret_cast<float2 *>(smem);                       // [W*OV][32]
    const int table_rows = W * OV;
    unsigned char *rec_base = smem + (size_t) table_rows * WIN * sizeof(float2);
    const int wave_in_block = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    int4 *heads = reinterpret_cast<int4 *>(rec_base) + wave_in_block * 64;   // (mu, mv, tu, tv)
    float2 *samples = reinterpret_cast<float2 *>(rec_base + (size_t) NW * 64 * sizeof(int4))
                      + wave_in_block * 64 * P;                             // [P][64]

    // Stage the kernel table, zero-padded from K to 32 taps per row.
    for (int idx = threadIdx.x; idx < table_rows * WIN; idx += NW * 64) {
        const int row = idx >> 5, t = idx & 31;
        table[idx] = t < K ? kern[(int64_t) row * K + t] : make_float2(0.0f, 0.0f);
    }
    __syncthreads();

    const int64_t wave = (int64_t) blockIdx.x * NW + wave_in_block;
    const int64_t start = wave * vis_per_wave;
    const int64_t end = start + vis_per_wave < num_vis ? start + vis_per_wave : num_vis;
    if (start >= end)
        return;

    const int uv_bias = (K - 1) / 2 - Gg / 2;           // grid.py:1038
    const int half = Gg / 2;
    const int S = WIN - K;                              // window slack
    const int c = lane & 31, h = lane >> 5;

    window_acc<P> acc;
#pragma unroll
    for (int p = 0; p < P; p++)
        for (int k = 0; k < 16; k++) {
            acc.re[p][k] = 0.0f;
            acc.im[p][k] = 0.0f;
        }
    bool have = false;
    int Wu = 0, Wv = 0;

    for (int64_t b = start; b < end; b += 64) {
        // ---- load a batch of up to 64 visibilities, lane i <-> visibility b+i ----------
        {
            const int64_t i = b + lane;
            const bool valid = i < end;
            const int64_t ii = valid ? i : end - 1;
            const int2 packed = reinterpret_cast<const int2 *>(uv)[ii];
            const int u = (short) (packed.x & 0xffff), v = (short) (packed.x >> 16);
            const int su = (short) (packed.y & 0xffff), sv = (short) (packed.y >> 16);
            const int wp = w_plane[ii];
            const int wu = u + half, wv = v + half;
            const bool ok = valid && (unsigned) wu < (unsigned) Gg && (unsigned) wv < (unsigned) Gg
                            && (unsigned) su < (unsigned) OV && (unsigned) sv < (unsigned) OV
                            && (unsigned) wp < (unsigned) W;
            int4 head;
            head.x = u - uv_bias;
            head.y = v - uv_bias;
            head.z = ok ? wp * OV + su : 0;
            head.w = ok ? wp * OV + sv : 0;
            heads[lane] = head;
            const int64_t wa = ok ? (int64_t) wv * wg_row_stride + wu : 0;
#pragma unroll
            for (int p = 0; p < P; p++) {
                float2 s = make_float2(0.0f, 0.0f);
                if (ok) {
                    const float wgt = weights_grid[wa + p * wg_pol_stride];
                    const float2 raw = vis[ii * P + p];
                    s = make_float2(raw.x * wgt, raw.y * wgt);      // grid.py:1046
                }
                samples[p * 64 + lane] = s;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        const int count = end - b < 64 ? (int) (end - b) : 64;
        const int npairs = (count + 1) >> 1;
        for (int q = 0; q < npairs; q++) {
            // lanes 0-31 take visibility 2q, lanes 32-63 visibility 2q+1
            const int ri = 2 * q + h;
            const int4 head = heads[ri];
            float2 s[P];
            bool nonzero = false;
#pragma unroll
            for (int p = 0; p < P; p++) {
                s[p] = samples[p * 64 + ri];
                nonzero |= (s[p].x != 0.0f) | (s[p].y != 0.0f);
            }
            const float2 kv = table[head.w * WIN + ((c - head.y) & 31)];
            const float2 ku = table[head.z * WIN + ((c - head.x) & 31)];
            // a = s * conj(kv)
            float ar[P], ai[P];
#pragma unroll
            for (int p = 0; p < P; p++) {
                ar[p] = fmaf(s[p].x, kv.x, s[p].y * kv.y);
                ai[p] = fmaf(s[p].y, kv.x, -s[p].x * kv.y);
            }
            const int muA = __builtin_amdgcn_readfirstlane(head.x);
            const int mvA = __builtin_amdgcn_readfirstlane(head.y);
            const int muB = __builtin_amdgcn_readlane(head.x, 32);
            const int mvB = __builtin_amdgcn_readlane(head.y, 32);
            const unsigned long long nz = __ballot(nonzero);
            int pend = ((nz & 0xffffffffull) ? 1 : 0) | ((nz >> 32) ? 2 : 0);
            while (pend) {
                const bool first_b = !(pend & 1);
                const int fmu = first_b ? muB : muA, fmv = first_b ? mvB : mvA;
                if (!have) {
                    Wu = fmu - S / 2;
                    Wv = fmv - S / 2;
                    have = true;
                } else if ((unsigned) (fmu - Wu) > (unsigned) S || (unsigned) (fmv - Wv) > (unsigned) S) {
                    const int nWu = (unsigned) (fmu - Wu) > (unsigned) S ? fmu - S / 2 : Wu;
                    const int nWv = (unsigned) (fmv - Wv) > (unsigned) S ? fmv - S / 2 : Wv;
                    const bool full = nWu - Wu >= WIN || Wu - nWu >= WIN
                                      || nWv - Wv >= WIN || Wv - nWv >= WIN;
                    flush_window<P>(acc, grid, row_stride, pol_stride, Gg, Wu, Wv, nWu, nWv,
                                    full, lane);
                    Wu = nWu;
                    Wv = nWv;
                }
                int act = first_b ? 2 : 1;
                if (pend == 3 && (unsigned) (muB - Wu) <= (unsigned) S
                    && (unsigned) (mvB - Wv) <= (unsigned) S)
                    act = 3;
                const bool on = (act >> h) & 1;
#pragma unroll
                for (int p = 0; p < P; p++) {
                    const float xr = on ? ar[p] : 0.0f;
                    const float xi = on ? ai[p] : 0.0f;
                    // (xr + i xi) * (ku.x - i ku.y)
                    acc.re[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(xr, ku.x, acc.re[p], 0, 0, 0);
                    acc.re[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(xi, ku.y, acc.re[p], 0, 0, 0);
                    acc.im[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(xi, ku.x, acc.im[p], 0, 0, 0);
                    acc.im[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(-xr, ku.y, acc.im[p], 0, 0, 0);
                }
                pend &= ~act;
            }
        }
        // the next batch overwrites this wave's staging area
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (have)
        flush_window<P>(acc, grid, row_stride, pol_stride, Gg, Wu, Wv, Wu, Wv, true, lane);
}

template <int P>
constexpr int waves_per_block()
{
    return P == 1 ? 16 : 8;
}

size_t lds_bytes(int P, int NW, int W, int OV)
{
    return (size_t) W * OV * WIN * sizeof(float2) + (size_t) NW * 64 * (sizeof(int4) + P * sizeof(float2));
}

template <int P>
int launch(float *grid, int64_t row_stride, int64_t pol_stride, int Gg, const float *wg,
           int64_t wg_row_stride, int64_t wg_pol_stride, const int16_t *uv,
           const int16_t *w_plane, const float2 *vis, int64_t num_vis, const float2 *kern,
           int W, int OV, int K, hipStream_t stream)
{
    constexpr int NW = waves_per_block<P>();
    const size_t lds = lds_bytes(P, NW, W, OV);
    static bool attr_set = false;
    if (!attr_set) {
        KIMG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&grid_mfma_kernel<P, NW>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    // One resident block per CU; every wave streams a contiguous range, a multiple of 64.
    const int blocks_max = 256;
    int64_t vis_per_wave = (num_vis + (int64_t) blocks_max * NW - 1) / ((int64_t) blocks_max * NW);
    vis_per_wave = (vis_per_wave + 63) / 64 * 64;
    if (vis_per_wave < 64)
        vis_per_wave = 64;
    const int64_t nwaves = (num_vis + vis_per_wave - 1) / vis_per_wave;
    const int blocks = (int) ((nwaves + NW - 1) / NW);
    grid_mfma_kernel<P, NW><<<blocks, NW * 64, lds, stream>>>(
        grid, row_stride, pol_stride, Gg, wg, wg_row_stride, wg_pol_stride, uv, w_plane, vis,
        num_vis, kern, W, OV, K, vis_per_wave);
    return kimg_launch_status();
}

} // namespace

bool kimg_grid_mfma_supported(int P, int w_planes, int oversample, int kernel_width)
{
    if (P < 1 || P > 4 || kernel_width > WIN || kernel_width < 1)
        return false;
    const int NW = P == 1 ? 16 : 8;
    return lds_bytes(P, NW, w_planes, oversample) <= 160 * 1024;
}

size_t kimg_grid_mfma_workspace_bytes(int64_t max_vis, int P)
{
    (void) max_vis;
    (void) P;
    return 0;       // records are staged through LDS; no HBM scratch needed
}

int kimg_grid_mfma(void *grid, int64_t grid_row_stride, int64_t grid_pol_stride, int grid_size,
                   int P, const float *weights_grid, int64_t wg_row_stride, int64_t wg_pol_stride,
                   const int16_t *uv, const int16_t *w_plane, const void *vis, int64_t num_vis,
                   const void *convolve_kernel, int w_planes, int oversample, int kernel_width,
                   void *workspace, size_t workspace_bytes, hipStream_t stream)
{
    (void) workspace;
    (void) workspace_bytes;
#define LAUNCH(PP) return launch<PP>((float *) grid, grid_row_stride, grid_pol_stride, grid_size, \
        weights_grid, wg_row_stride, wg_pol_stride, uv, w_plane, (const float2 *) vis, num_vis, \
        (const float2 *) convolve_kernel, w_planes, oversample, kernel_width, stream)
    switch (P) {
    case 1: LAUNCH(1);
    case 2: LAUNCH(2);
    case 3: LAUNCH(3);
    case 4: LAUNCH(4);
    }
#undef LAUNCH
    return KIMG_EUNSUPPORTED;
}
