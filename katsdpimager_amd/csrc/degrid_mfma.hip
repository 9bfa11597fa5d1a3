// MFMA window degridder for gfx950 (CDNA4).
//
// Replaces Degridder.static_run / degrid.mako:77-199 of the reference for kernel widths <= 64
// (widths above 32 as 2 x 2 tap blocks).
// Same result as DegridderHost/_degrid (grid.py:1138-1154):
//     vis[r][p] -= weights[r][p] * sum_{j,k} kern[w][sv][j] * kern[w][su][k] * grid[p][v0+j][u0+k].
//
// Mirror image of the MFMA gridder (grid_mfma.hip): a wave keeps a moving 32x32 window of the
// grid in registers -- here as the MFMA *A operand* -- and contracts it with the separable
// kernel of 16 visibilities at a time:
//   step 1 (matrix pipe): T[x][b] = sum_y G[y][x] * kv_b[y]          32 MFMA (one per window row)
//        A[i = x][k] = (Re, Im) of G[y][x]   (lane = window column: row loads are coalesced)
//        B[k][n = 2b + part] = [[kv.re, kv.im], [-kv.im, kv.re]]      (re/im interleaved outputs)
//   step 2 (VALU): vis_b = sum_x ku_b[x] * T[x][b], 16 complex MACs per lane + one cross-half add.
// Window cell (y, x) <-> grid point (Wv + ((y-Wv)&31), Wu + ((x-Wu)&31)); when the window moves
// only the cells whose mapping changed are re-read.  Every visibility owns its output columns,
// so a visibility that does not fit the current window is simply masked and handled in a later
// pass of the same 16-group (no cross-talk).
// The kernel table lives in LDS with rows of 32 zero-padded taps stored twice (so "tap
// (index - first_tap) mod 32" is an immediate offset) and a row stride of 65 taps (so that the
// per-visibility gathers of different table rows fall on different banks).
//
// Algorithmic work per visibility: 8*K*K*P flop; executed 2 MFMA x 4096 flop per polarization.
#include "kimg_common.h"
#include <limits.h>
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WIN = 32;
// LDS table rows: TAPS = 64 (32 zero-padded taps stored twice: the tap index needs no wrap) when
// the table fits, else TAPS = 32 (wrap = 2 more VALU per read); one pad tap per row either way.
template <int TAPS>
constexpr int row_bytes() { return (TAPS + 1) * 8; }
constexpr int BATCH = 16;               // visibilities per MFMA pass (N = 32 = 16 x (re, im))

__device__ inline uint32_t changed_mask(int W, int nW)
{
    const int d = nW - W;
    if (d == 0)
        return 0u;
    if (d >= WIN || d <= -WIN)
        return 0xffffffffu;
    const uint32_t offsets = d > 0 ? (1u << d) - 1u : ~((1u << (WIN + d)) - 1u);
    const int rot = W & 31;
    return rot ? (offsets << rot) | (offsets >> (32 - rot)) : offsets;
}

// min / max over each aligned group of 16 lanes (one DPP row)
__device__ inline int row16_min(int v)
{
    v = min(v, __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true));     // quad_perm [1,0,3,2]
    v = min(v, __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, true));     // quad_perm [2,3,0,1]
    v = min(v, __builtin_amdgcn_mov_dpp(v, 0x141, 0xf, 0xf, true));    // row_half_mirror
    v = min(v, __builtin_amdgcn_mov_dpp(v, 0x140, 0xf, 0xf, true));    // row_mirror
    return v;
}

__device__ inline int row16_max(int v)
{
    v = max(v, __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true));
    v = max(v, __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, true));
    v = max(v, __builtin_amdgcn_mov_dpp(v, 0x141, 0xf, 0xf, true));
    v = max(v, __builtin_amdgcn_mov_dpp(v, 0x140, 0xf, 0xf, true));
    return v;
}

template <int P>
struct window_regs {
    float g[P][WIN];        // g[p][y]: (lane < 32 ? Re : Im) of G[p][row y][column lane & 31]
};

// Kernel widths above 32 are degridded as 2 x 2 blocks of taps, one launch per block, each
// subtracting its partial sum (see grid_mfma.hip): row taps [tv0, tv0 + Kv) and column taps
// [tu0, tu0 + Ku) of the K-tap kernel; off-diagonal blocks need TWO tables in LDS.
struct tap_split {
    int K;
    int tv0, Kv;
    int tu0, Ku;
};

// TG: table(s) too large for LDS (hundreds of W planes) stay in HBM as [rows][32] zero-padded
// taps (pad_rows_kernel, rebuilt per call in the caller's workspace).  Each wave copies the 2 x 16
// rows its current 16 visibilities need into a private LDS cache (coalesced 16-byte loads: a row
// is 256 bytes) and then works exactly like the single-row LDS form (TAPS must be 32).
// ---- fp16 hi/lo form (F16 = true; one polarization, table in LDS) ---------------------------
// As in the gridder (grid_mfma.hip): fp32 operands travel as fp16 hi/lo pairs, a complex product
// needs 6 of the 16 k-slots of v_mfma_f32_32x32x16_f16, so one 8-pass instruction contracts TWO
// window rows (lanes 0-31: row 2i, lanes 32-63: row 2i + 1) where the exact 16-pass instruction
// takes one.  The window is split when it is (re)loaded, scaled by a power of two S_g chosen from
// its largest value; the row taps are split on the fly (the table stays fp32: step 2 needs it so),
// scaled by S from the largest tap; the sums are multiplied by 1 / (S_g S) at the end.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// (round to nearest: v_cvt_pk_f16_f32; with truncation the errors of the 10^7 products of an
// inner product all point the same way and the adjointness check sees them)
__device__ inline unsigned dg_cvt_pk_f16(float lo, float hi)
{
    typedef float v2f __attribute__((ext_vector_type(2)));
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const v2f x = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(x, h2));
}

__device__ inline float dg_f16_to_f32(unsigned bits16)
{
    return (float) __builtin_bit_cast(_Float16, (unsigned short) bits16);
}

// 2^(13 - e) for a largest magnitude with bit pattern m in [2^e, 2^(e+1)); 1 for m == 0
__device__ inline float dg_scale_for(unsigned m)
{
    int e = (int) (m >> 23) - 127;
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    return (m && m < 0x7f800000u) ? __uint_as_float((unsigned) (13 - e + 127) << 23) : 1.0f;
}

template <int P, int NW, int TAPS, bool TWO, bool TG = false, bool F16 = false>
__global__ __launch_bounds__(NW * 64) void degrid_mfma_kernel(
    const float *__restrict__ grid, int64_t row_stride, int64_t pol_stride, int Gg,
    const int16_t *__restrict__ uv, const int16_t *__restrict__ w_plane,
    const float *__restrict__ weights, float *__restrict__ vis, int64_t num_vis,
    const float2 *__restrict__ kern, int W, int OV, tap_split ts, int64_t vis_per_block,
    int p_total, const unsigned char *__restrict__ padded, const unsigned *__restrict__ tab_max,
    int64_t chunk, unsigned long long *queue)
{
    static_assert(!TWO || TAPS == 32 || TG, "two tables only fit LDS with single rows");
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int ROW_BYTES = row_bytes<TAPS>();
    const unsigned char *tbytes = TG ? padded : smem;
    const int table_rows = W * OV;
    const int table_bytes = table_rows * ROW_BYTES;
    const int u_table = TWO ? table_bytes : 0;
    static_assert(!TG || TAPS == 32, "the row cache uses single rows");
    unsigned char *rec_base = smem + (TG ? 0 : (size_t) table_bytes * (TWO ? 2 : 1));
    constexpr int CACHE_ROW = 272;                      // bytes per cached row (34 taps: 16-byte aligned)
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    // per-wave staging of 64 visibilities: (table offset for kv, for ku, mu, mv); mu = INT_MIN
    // marks a visibility to skip
    int4 *recs = reinterpret_cast<int4 *>(rec_base) + wib * 64;
    // TG: per-wave cache of the rows of the current 16 visibilities (row taps, then column taps)
    unsigned char *cache_v = rec_base + (size_t) NW * 64 * sizeof(int4) + (size_t) wib * 2 * BATCH * CACHE_ROW;
    unsigned char *cache_u = cache_v + BATCH * CACHE_ROW;

    // Stage the kernel table: rows zero-padded to 32 taps, stored twice, stride 65 taps.
    // A thread issues all loads of a round before its first LDS write (one L2 round trip per
    // round instead of one per element).
    auto stage_table = [&](unsigned char *dst_base, int tap0, int Kp) __attribute__((always_inline)) {
        constexpr int STG = 8;
        const int total = table_rows * 32;
        for (int base = threadIdx.x; base < total; base += NW * 64 * STG) {
            float2 v[STG];
#pragma unroll
            for (int i = 0; i < STG; i++) {
                const int idx = base + i * NW * 64;
                const int row = idx >> 5, t = idx & 31;
                v[i] = (idx < total && t < Kp) ? kern[(int64_t) row * ts.K + tap0 + t]
                                               : make_float2(0.0f, 0.0f);
            }
#pragma unroll
            for (int i = 0; i < STG; i++) {
                const int idx = base + i * NW * 64;
                const int row = idx >> 5, t = idx & 31;
                if (idx < total) {
                    float2 *dst = reinterpret_cast<float2 *>(dst_base + (size_t) row * ROW_BYTES + t * 8);
                    dst[0] = v[i];
                    if (TAPS == 64)
                        dst[32] = v[i];
                }
            }
        }
    };
    if (!TG) {
        stage_table(smem, ts.tv0, ts.Kv);
        if (TWO)
            stage_table(smem + table_bytes, ts.tu0, ts.Ku);
    }
    __syncthreads();
    float S_kv = 1.0f;
    if (F16 && TG) {
        S_kv = dg_scale_for(*tab_max);          // table in HBM: maximum found by dg_table_max_kernel
    } else if (F16) {
        // largest |component| of the row-tap table (block reduction; the staging area is still free)
        unsigned *s_max = reinterpret_cast<unsigned *>(rec_base);
        // (taps 0..31 of every row: the padding tap of the 65-tap rows is never written)
        unsigned m = 0;
        for (int i = threadIdx.x; i < table_rows * 32; i += NW * 64) {
            const uint2 t = *reinterpret_cast<const uint2 *>(smem + (size_t) (i >> 5) * ROW_BYTES + (i & 31) * 8);
            m = max(m, max(t.x & 0x7fffffffu, t.y & 0x7fffffffu));
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            m = max(m, (unsigned) __shfl_xor((int) m, off, WAVE));
        if (lane == 0)
            s_max[wib] = m;
        __syncthreads();
        for (int w = 0; w < NW; w++)
            m = max(m, s_max[w]);
        S_kv = dg_scale_for(m);
        __syncthreads();
    }

    const int64_t block_start = (int64_t) blockIdx.x * vis_per_block;
    int64_t block_end = block_start + vis_per_block;
    if (block_end > num_vis)
        block_end = num_vis;
    const int64_t span = block_end > block_start ? block_end - block_start : 0;
    const int64_t start = block_start + (span * wib / NW) / 64 * 64;
    const int64_t end_static = wib == NW - 1 ? block_end : block_start + (span * (wib + 1) / NW) / 64 * 64;
    // What a wave degrids (as in grid_mfma_kernel): chunk == 0: the contiguous range [start,
    // end_static); chunk > 0 (long launches): chunks of `chunk` visibilities, the first one by the
    // wave's number, the following ones drawn from the counter `queue` (or, without one, in steps of
    // the number of waves), so that every wave is busy until the launch ends.
    struct batch_pos {
        int64_t b, e;       // first visibility of the batch; end of its chunk (b >= e: no batch)
    };
    const int64_t wave_id = (int64_t) blockIdx.x * NW + wib, waves = (int64_t) gridDim.x * NW;
    const int64_t chunks_total = chunk > 0 ? (num_vis + chunk - 1) / chunk : 0;
    // (chunks are handed out in stream order: the degridder only reads the grid, neighbouring
    // chunks share cache lines instead of contending for them)
    auto chunk_pos = [&](int64_t c) __attribute__((always_inline)) {
        batch_pos p;
        p.b = c < chunks_total ? c * chunk : 0;
        p.e = c < chunks_total ? (p.b + chunk < num_vis ? p.b + chunk : num_vis) : 0;
        return p;
    };
    unsigned long long ticket = 0;          // lane 0: value returned by the last draw
    int64_t static_next = wave_id + waves;
    auto draw = [&]() __attribute__((always_inline)) {
        if (queue != nullptr) {
            if (lane == 0)
                ticket = atomicAdd(queue, 1ull);
        } else {
            ticket = (unsigned long long) static_next;
            static_next += waves;
        }
    };
    auto drawn = [&]() __attribute__((always_inline)) {
        const unsigned lo = (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) ticket);
        const unsigned hi = (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) (ticket >> 32));
        const int64_t t = (int64_t) (((unsigned long long) hi << 32) | lo);
        return queue != nullptr ? t + waves : t;
    };
    auto next_pos = [&](const batch_pos &p) __attribute__((always_inline)) {
        batch_pos n = p;
        n.b = p.b + 64;
        if (n.b >= p.e && chunk > 0 && p.b < p.e) {
            n = chunk_pos(drawn());
            if (n.b < n.e)
                draw();
        }
        return n;
    };
    batch_pos pos;
    if (chunk > 0) {
        pos = chunk_pos(wave_id);
        draw();
    } else {
        pos.b = start;
        pos.e = end_static;
    }
    if (pos.b >= pos.e)
        return;
    // the next batch's coordinates are fetched while the current one is worked on
    auto load_coords = [&](const batch_pos &p, int2 &packed, int &wp) __attribute__((always_inline)) {
        const int64_t i = p.b + lane;
        const int64_t ii = i < p.e ? i : p.e - 1;
        packed = reinterpret_cast<const int2 *>(uv)[ii];
        wp = w_plane[ii];
    };
    int2 next_packed;
    int next_wp;
    load_coords(pos, next_packed, next_wp);

    const int uv_bias = (ts.K - 1) / 2 - Gg / 2;        // grid.py:1141
    const int half = Gg / 2;
    const int Su = WIN - ts.Ku, Sv = WIN - ts.Kv;
    const bool h = lane >= 32;
    const int x_lane = lane & 31;                       // window column held by this lane
    const int part = lane & 1;                          // output column parity: 0 re, 1 im
    const int b_lane = (lane & 31) >> 1;                // visibility of the 16-group served
    // B operand (k = h): part 0 -> (kv.re, -kv.im), part 1 -> (kv.im, kv.re)
    const bool b_take_im = (h != (part != 0));
    const float b_sign = (h && !part) ? -1.0f : 1.0f;
    const int b_comp = b_take_im ? 4 : 0;
    // step 2: own column holds T_part, the neighbour lane T_(1-part):
    // part 0: sum += ku.re*Tr - ku.im*Ti ; part 1: sum += ku.re*Ti + ku.im*Tr
    const float o_sign = part ? 1.0f : -1.0f;
    const int h32 = h ? 32 : 0;                         // rows of C held by this half: r_k + 4h

    window_regs<P> win;
#pragma unroll
    for (int p = 0; p < P; p++)
#pragma unroll
        for (int y = 0; y < WIN; y++)
            win.g[p][y] = 0.0f;
    // fp16 form: this lane's 16 rows (2 i + (lane >> 5)) of its column as complex values, the
    // same split into the row-pair operands of the matrix instruction, and the scales
    unsigned a_hi[P][F16 ? WIN / 2 : 1], a_lo[P][F16 ? WIN / 2 : 1];   // (re_hi, im_hi), (re_lo, im_lo)
    float S_g = 1.0f, inv_scale = 1.0f;
    const int hrow = lane >> 5;
    const unsigned sel = part ? 0x01000302u : 0x03020100u;      // odd outputs swap (re, im)
    const unsigned flip = part ? 0u : 0x80000000u;              // even outputs negate im
    bool have = false;
    int Wu = 0, Wv = 0;

    // (Re-)load the window cells whose mapping changes when the origin moves to (nWu, nWv).
    auto load_window = [&](int nWu, int nWv) __attribute__((always_inline)) {
        const uint32_t row_mask = have ? changed_mask(Wv, nWv) : 0xffffffffu;
        const uint32_t col_mask = have ? changed_mask(Wu, nWu) : 0xffffffffu;
        const int gx = nWu + ((x_lane - nWu) & 31);
        const bool col_changed = (col_mask >> x_lane) & 1u;
        const bool x_ok = (unsigned) gx < (unsigned) Gg;
        if constexpr (F16) {
            // The whole window is (re)read at every move: the split needs one scale for all of it,
            // and keeping the fp32 values as well would not fit the register budget.  Moves are rare.
            float2 g2[P][WIN / 2];
#pragma unroll
            for (int i = 0; i < WIN / 2; i++) {
                const int y = 2 * i + hrow;
                const int gy = nWv + ((y - nWv) & 31);
                const bool y_ok = (unsigned) gy < (unsigned) Gg;
                const float2 *cell = reinterpret_cast<const float2 *>(grid)
                                     + ((int64_t) gy * row_stride + gx);
#pragma unroll
                for (int p = 0; p < P; p++)
                    g2[p][i] = (x_ok && y_ok) ? cell[p * pol_stride] : make_float2(0.0f, 0.0f);
            }
            // scale from the largest magnitude in the window, then split every row pair
            unsigned m = 0;
#pragma unroll
            for (int p = 0; p < P; p++)
#pragma unroll
                for (int i = 0; i < WIN / 2; i++)
                    m = max(m, max(__float_as_uint(g2[p][i].x) & 0x7fffffffu,
                                   __float_as_uint(g2[p][i].y) & 0x7fffffffu));
            m = (unsigned) row16_max((int) (m >= 0x7f800000u ? 0u : m));
            m = max(max((unsigned) __builtin_amdgcn_readlane((int) m, 0),
                        (unsigned) __builtin_amdgcn_readlane((int) m, 16)),
                    max((unsigned) __builtin_amdgcn_readlane((int) m, 32),
                        (unsigned) __builtin_amdgcn_readlane((int) m, 48)));
            S_g = dg_scale_for(m);
            inv_scale = 1.0f / (S_g * S_kv);
#pragma unroll
            for (int p = 0; p < P; p++)
#pragma unroll
                for (int i = 0; i < WIN / 2; i++) {
                    const float re = g2[p][i].x * S_g, im = g2[p][i].y * S_g;
                    a_hi[p][i] = dg_cvt_pk_f16(re, im);
                    a_lo[p][i] = dg_cvt_pk_f16(re - dg_f16_to_f32(a_hi[p][i] & 0xffffu),
                                               im - dg_f16_to_f32(a_hi[p][i] >> 16));
                }
            Wu = nWu;
            Wv = nWv;
            have = true;
            return;
        }
#pragma unroll
        for (int y = 0; y < WIN; y++) {
            const bool row_changed = (row_mask >> y) & 1u;          // uniform
            if (!row_changed && col_mask == 0)
                continue;
            const int gy = nWv + ((y - nWv) & 31);                  // uniform
            const bool y_ok = (unsigned) gy < (unsigned) Gg;
            if (row_changed || col_changed) {
                const float *cell = grid + 2 * ((int64_t) gy * row_stride + gx) + (h ? 1 : 0);
#pragma unroll
                for (int p = 0; p < P; p++)
                    win.g[p][y] = (x_ok && y_ok) ? cell[2 * p * pol_stride] : 0.0f;
            }
        }
        Wu = nWu;
        Wv = nWv;
        have = true;
    };

    for (; pos.b < pos.e;) {
        const int64_t b0 = pos.b, end = pos.e;
        const batch_pos following = next_pos(pos);
        // ---- stage 64 visibilities (lane i <-> visibility b0 + i) ---------------------------
        int gmin_u, gmax_u, gmin_v, gmax_v;
        {
            const int64_t i = b0 + lane;
            const int2 packed = next_packed;
            const int wp = next_wp;
            if (following.b < following.e)
                load_coords(following, next_packed, next_wp);
            const int u = (short) (packed.x & 0xffff), v = (short) (packed.x >> 16);
            const int su = (short) (packed.y & 0xffff), sv = (short) (packed.y >> 16);
            const bool ok = i < end && (unsigned) (u + half) < (unsigned) Gg
                            && (unsigned) (v + half) < (unsigned) Gg && (unsigned) su < (unsigned) OV
                            && (unsigned) sv < (unsigned) OV && (unsigned) wp < (unsigned) W;
            const int mu = u - uv_bias + ts.tu0, mv = v - uv_bias + ts.tv0;
            int4 r;
            // TAPS == 64: byte address of tap ((-m) & 31) of the row; TAPS == 32: row number in
            // the high half and the tap offset in the low byte (wrapped at read time)
            if (TAPS == 64) {
                r.x = ok ? (wp * OV + sv) * ROW_BYTES + ((-mv) & 31) * 8 : 0;
                r.y = ok ? (wp * OV + su) * ROW_BYTES + ((-mu) & 31) * 8 : 0;
            } else {
                r.x = ok ? ((wp * OV + sv) << 16) | (((-mv) & 31) * 8) : 0;
                r.y = ok ? ((wp * OV + su) << 16) | (((-mu) & 31) * 8) : 0;
            }
            r.z = ok ? mu : INT_MIN;
            r.w = mv;
            recs[lane] = r;
            gmin_u = row16_min(ok ? mu : INT_MAX);
            gmax_u = row16_max(ok ? mu : INT_MIN);
            gmin_v = row16_min(ok ? mv : INT_MAX);
            gmax_v = row16_max(ok ? mv : INT_MIN);
        }
        __builtin_amdgcn_wave_barrier();

        const int count = end - b0 < 64 ? (int) (end - b0) : 64;
        for (int g = 0; g * BATCH < count; g++) {
            const int first = g * BATCH;
            const int lo_u = __builtin_amdgcn_readlane(gmin_u, first);
            const int hi_u = __builtin_amdgcn_readlane(gmax_u, first);
            const int lo_v = __builtin_amdgcn_readlane(gmin_v, first);
            const int hi_v = __builtin_amdgcn_readlane(gmax_v, first);
            if (lo_u > hi_u)
                continue;                               // nothing to do in this group
            const int4 rec = recs[first + b_lane];      // this lane's visibility of the group
            if (TG) {
                // lane l copies quarter l & 3 (64 bytes) of the rows of visibility l >> 2
                const int4 rr = recs[first + (lane >> 2)];
                const unsigned row_v = (unsigned) rr.x >> 16, row_u = (unsigned) rr.y >> 16;
                const float4 *src_v = reinterpret_cast<const float4 *>(
                    padded + (size_t) row_v * 256) + (lane & 3) * 4;
                const float4 *src_u = reinterpret_cast<const float4 *>(
                    padded + (TWO ? (size_t) table_rows * 256 : 0) + (size_t) row_u * 256) + (lane & 3) * 4;
                float4 tv[4], tu[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    tv[i] = src_v[i];
                    tu[i] = src_u[i];
                }
                float4 *dst_v = reinterpret_cast<float4 *>(cache_v + (lane >> 2) * CACHE_ROW) + (lane & 3) * 4;
                float4 *dst_u = reinterpret_cast<float4 *>(cache_u + (lane >> 2) * CACHE_ROW) + (lane & 3) * 4;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    dst_v[i] = tv[i];
                    dst_u[i] = tu[i];
                }
                __builtin_amdgcn_wave_barrier();
            }
            const int mu = rec.z, mv = rec.w;
            const bool mine = mu != INT_MIN;
            // the visibility and weight this lane will update: fetched now, so that their latency
            // hides behind the MFMA chain instead of ending the batch
            float old_vis[P], old_wgt[P];
            {
                const int64_t r = b0 + first + b_lane;
                const bool fetch = mine && !h && r < end;
#pragma unroll
                for (int p = 0; p < P; p++) {
                    old_vis[p] = fetch ? vis[(r * p_total + p) * 2 + part] : 0.0f;
                    old_wgt[p] = fetch ? weights[r * p_total + p] : 0.0f;
                }
            }
            unsigned long long pending = __ballot(mine);
            const bool whole = hi_u - lo_u <= Su && hi_v - lo_v <= Sv;
            while (pending) {
                // ---- choose / move the window ------------------------------------------------
                // the whole group when it fits one window position, else the first pending
                // visibility; all slack is left ahead in the direction of travel
                int lu = lo_u, hu = hi_u, lv = lo_v, hv = hi_v;
                if (!whole) {
                    const int src = __builtin_ctzll(pending);
                    lu = hu = __builtin_amdgcn_readlane(mu, src);
                    lv = hv = __builtin_amdgcn_readlane(mv, src);
                }
                auto place = [&](int lo, int hi, int cur, int S) __attribute__((always_inline)) {
                    if (!have)
                        return lo - (S - (hi - lo)) / 2;
                    if (lo >= cur && hi <= cur + S)
                        return cur;
                    return hi > cur + S ? lo : hi - S;
                };
                const int nWu = place(lu, hu, Wu, Su);
                const int nWv = place(lv, hv, Wv, Sv);
                if (!have || nWu != Wu || nWv != Wv)
                    load_window(nWu, nWv);
                const bool fit = mine && (unsigned) (mu - Wu) <= (unsigned) Su
                                 && (unsigned) (mv - Wv) <= (unsigned) Sv;
                const unsigned long long done = __ballot(fit) & pending;
                pending &= ~done;
                const bool active = (done >> lane) & 1ull;

                // ---- step 1: T[x][n] = sum_y G[y][x] * B_y[n] ---------------------------------
                f32x16 acc[P];
#pragma unroll
                for (int p = 0; p < P; p++)
#pragma unroll
                    for (int k = 0; k < 16; k++)
                        acc[p][k] = 0.0f;
                const unsigned char *pv = TG ? cache_v + b_lane * CACHE_ROW + b_comp
                                             : tbytes + (TAPS == 64 ? rec.x : (rec.x >> 16) * ROW_BYTES)
                                                   + b_comp;
                const int off_v = rec.x & 0xff;
                if constexpr (F16) {
                    // two window rows per instruction; this lane's row is 2 i + hrow, its tap is
                    // split here: (kre_hi, -kim_hi, ...) for real outputs, (kim_hi, kre_hi, ...) else
                    const unsigned char *pv16 = pv - b_comp;
#pragma unroll
                    for (int i = 0; i < WIN / 2; i++) {
                        const int y = 2 * i + hrow;
                        const float2 kv = *reinterpret_cast<const float2 *>(
                            TAPS == 64 ? pv16 + 8 * y : pv16 + ((off_v + 8 * y) & 0xf8));
                        const float kre = kv.x * S_kv, kim = kv.y * S_kv;
                        const unsigned hi = dg_cvt_pk_f16(kre, kim);
                        const unsigned lo = dg_cvt_pk_f16(kre - dg_f16_to_f32(hi & 0xffffu),
                                                          kim - dg_f16_to_f32(hi >> 16));
                        u32x4 B;
                        B[0] = __builtin_amdgcn_perm(hi, hi, sel) ^ flip;
                        B[1] = __builtin_amdgcn_perm(lo, lo, sel) ^ flip;
                        B[2] = B[0];
                        B[3] = 0;
#pragma unroll
                        for (int p = 0; p < P; p++) {
                            u32x4 A;
                            A[0] = a_hi[p][i];
                            A[1] = a_hi[p][i];
                            A[2] = a_lo[p][i];
                            A[3] = 0;
                            acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                                __builtin_bit_cast(f16x8, A), __builtin_bit_cast(f16x8, B), acc[p], 0, 0, 0);
                        }
                        if (i % 4 == 3)
                            __builtin_amdgcn_sched_barrier(0);      // bounds the operands in flight
                    }
                } else {
                typedef float dg_v2f __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int y = 0; y < WIN; y += 2) {
                    // two rows' taps with one sign multiplication (v_pk_mul_f32)
                    dg_v2f bv;
                    bv.x = *reinterpret_cast<const float *>(
                        TAPS == 64 ? pv + 8 * y : pv + ((off_v + 8 * y) & 0xf8));
                    bv.y = *reinterpret_cast<const float *>(
                        TAPS == 64 ? pv + 8 * (y + 1) : pv + ((off_v + 8 * (y + 1)) & 0xf8));
                    bv = bv * dg_v2f{b_sign, b_sign};
#pragma unroll
                    for (int p = 0; p < P; p++)
                        acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(win.g[p][y], bv.x, acc[p], 0, 0, 0);
#pragma unroll
                    for (int p = 0; p < P; p++)
                        acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(win.g[p][y + 1], bv.y, acc[p], 0, 0, 0);
                }
                }
                // ---- step 2: vis = sum_x ku[x] * T[x] ------------------------------------------
                const unsigned char *pu = TG ? cache_u + b_lane * CACHE_ROW
                                             : tbytes + u_table
                                                   + (TAPS == 64 ? rec.y + h32 : (rec.y >> 16) * ROW_BYTES);
                const int off_u = (rec.y & 0xff) + h32;
                // (the two products of the complex multiplication are summed apart, so that the sign
                // of the cross term is applied once instead of to every tap)
                float sum[P], cross[P];
#pragma unroll
                for (int p = 0; p < P; p++)
                    sum[p] = cross[p] = 0.0f;
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int rk = (k & 3) + 8 * (k >> 2);
                    const float2 ku = *reinterpret_cast<const float2 *>(
                        TAPS == 64 ? pu + 8 * rk : pu + ((off_u + 8 * rk) & 0xf8));
#pragma unroll
                    for (int p = 0; p < P; p++) {
                        const float own = acc[p][k];
                        const float other = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(
                            __builtin_bit_cast(int, own), 0xB1, 0xf, 0xf, true));
                        sum[p] = fmaf(ku.x, own, sum[p]);
                        cross[p] = fmaf(ku.y, other, cross[p]);
                    }
                }
#pragma unroll
                for (int p = 0; p < P; p++) {
                    sum[p] = fmaf(o_sign, cross[p], sum[p]);
                    sum[p] += __shfl_xor(sum[p], 32, WAVE);
                }
                // ---- write back: lanes 0-31 hold (vis b, part) -----------------------------------
                if (active && !h) {
                    const int64_t r = b0 + first + b_lane;
#pragma unroll
                    for (int p = 0; p < P; p++)
                        vis[(r * p_total + p) * 2 + part] =
                            old_vis[p] - old_wgt[p] * (F16 ? sum[p] * inv_scale : sum[p]);         // grid.py:1154
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        pos = following;
    }
}

// HBM copy of the table in the LDS row layout: (TAPS + 1) taps per row, the 32 zero-padded taps
// of [tap0, tap0 + Kp) once (TAPS = 32) or twice (TAPS = 64).
template <int TAPS>
__global__ __launch_bounds__(256) void pad_table_kernel(
    const float2 *__restrict__ kern, int rows, int K, int tap0, int Kp, float2 *__restrict__ out)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * (TAPS + 1))
        return;
    const int row = idx / (TAPS + 1), c = idx % (TAPS + 1), t = c & 31;
    out[idx] = (c < TAPS && t < Kp) ? kern[(int64_t) row * K + tap0 + t] : make_float2(0.0f, 0.0f);
}

// Largest |component| of the raw table (bit pattern), for the fp16 form with the table in HBM.
__global__ __launch_bounds__(256) void dg_table_max_kernel(const float *__restrict__ kern, int64_t n,
                                                            unsigned *__restrict__ out)
{
    unsigned m = 0;
    for (int64_t i = blockIdx.x * (int64_t) blockDim.x + threadIdx.x; i < n;
         i += (int64_t) gridDim.x * blockDim.x)
        m = max(m, __float_as_uint(kern[i]) & 0x7fffffffu);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        m = max(m, (unsigned) __shfl_xor((int) m, off, WAVE));
    if ((threadIdx.x & 63) == 0 && m)
        atomicMax(out, m);
}

// TG: [rows][32] zero-padded taps (256-byte rows).
__global__ __launch_bounds__(256) void pad_rows_kernel(
    const float2 *__restrict__ kern, int rows, int K, int tap0, int Kp, float2 *__restrict__ out)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * 32)
        return;
    const int row = idx >> 5, t = idx & 31;
    out[idx] = t < Kp ? kern[(int64_t) row * K + tap0 + t] : make_float2(0.0f, 0.0f);
}

size_t lds_bytes(int NW, int W, int OV, int taps, int tables = 1)
{
    return (size_t) tables * W * OV * (taps + 1) * 8 + (size_t) NW * 64 * sizeof(int4);
}

constexpr size_t LDS_LIMIT = 160 * 1024;

// The chunk counter of the call in progress on this host thread (in the workspace's tail, see
// kimg_degrid_mfma): handed to launch() this way because every launch site is a macro.
thread_local unsigned long long *tls_queue = nullptr;

template <int P, int NW, int TAPS, bool TWO, bool TG = false, bool F16 = false>
int launch(const float *grid, int64_t row_stride, int64_t pol_stride, int Gg, const int16_t *uv,
           const int16_t *w_plane, const float *weights, float *vis, int64_t num_vis,
           const float2 *kern, int W, int OV, const tap_split &ts, int p_total, hipStream_t stream,
           unsigned char *padded = nullptr, size_t tab_max_offset = 0)
{
    const size_t lds = TG ? lds_bytes(NW, 0, 0, TAPS) + (size_t) NW * 2 * BATCH * 272
                          : lds_bytes(NW, W, OV, TAPS, TWO ? 2 : 1);
    unsigned *tab_max = nullptr;
    if (TG) {
        const int rows = W * OV;
        float2 *out = reinterpret_cast<float2 *>(padded);
        if (F16) {
            // (the last 256 bytes of the workspace hold the table maximum)
            tab_max = reinterpret_cast<unsigned *>(padded + tab_max_offset);
            KIMG_HIP(hipMemsetAsync(tab_max, 0, sizeof(unsigned), stream));
            const int64_t n = (int64_t) rows * ts.K * 2;
            dg_table_max_kernel<<<kimg_divup(n, 256 * 8), 256, 0, stream>>>(
                reinterpret_cast<const float *>(kern), n, tab_max);
        }
        pad_rows_kernel<<<kimg_divup(rows * 32, 256), 256, 0, stream>>>(kern, rows, ts.K, ts.tv0,
                                                                       ts.Kv, out);
        if (TWO)
            pad_rows_kernel<<<kimg_divup(rows * 32, 256), 256, 0, stream>>>(
                kern, rows, ts.K, ts.tu0, ts.Ku, out + (size_t) rows * 32);
    }
    {
        const int rc = kimg_dynamic_lds(
            reinterpret_cast<const void *>(&degrid_mfma_kernel<P, NW, TAPS, TWO, TG, F16>), LDS_LIMIT);
        if (rc)
            return rc;
    }
    const int blocks_max = kimg_window_cus_now();
    int64_t vis_per_block = (num_vis + blocks_max - 1) / blocks_max;
    vis_per_block = (vis_per_block + 63) / 64 * 64;
    if (vis_per_block < 64 * NW)
        vis_per_block = 64 * NW;
    const int blocks = (int) ((num_vis + vis_per_block - 1) / vis_per_block);
    // long launches: work by the chunk (see batch_pos in the kernel)
    const int64_t waves = (int64_t) blocks * NW;
// (512 until round 3: on a W-slice of the resident store -- 6-7 M records, three chunks of ~700 per
// wave -- the waves' last chunks ended up to 170 us apart; 256 measures 4-6 % faster there, 384 / 192 /
// 128 do not; a chunk's end costs the degridder no flush, only a window reload)
#ifndef KIMG_DEGRID_MIN_CHUNK
#define KIMG_DEGRID_MIN_CHUNK 256
#endif
#ifndef KIMG_DEGRID_MAX_PARTS
#define KIMG_DEGRID_MAX_PARTS 32
#endif
    int64_t parts = num_vis / (waves * KIMG_DEGRID_MIN_CHUNK);
    parts = parts > KIMG_DEGRID_MAX_PARTS ? KIMG_DEGRID_MAX_PARTS : parts;
    int64_t chunk = 0;
    if (parts >= 2)
        chunk = ((num_vis + waves * parts - 1) / (waves * parts) + 63) / 64 * 64;
    unsigned long long *queue = chunk > 0 ? tls_queue : nullptr;
    if (queue != nullptr)
        KIMG_HIP(hipMemsetAsync(queue, 0, sizeof(unsigned long long), stream));
    degrid_mfma_kernel<P, NW, TAPS, TWO, TG, F16><<<blocks, NW * 64, lds, stream>>>(
        grid, row_stride, pol_stride, Gg, uv, w_plane, weights, vis, num_vis, kern, W, OV, ts,
        vis_per_block, p_total, padded, tab_max, chunk, queue);
    return kimg_launch_status();
}

} // namespace

static bool tables_fit_lds(int w_planes, int oversample, int kernel_width)
{
    return lds_bytes(12, w_planes, oversample, 32, kernel_width > WIN ? 2 : 1) <= LDS_LIMIT;
}

bool kimg_degrid_mfma_supported(int P, int w_planes, int oversample, int kernel_width)
{
    return P >= 1 && P <= 4 && kernel_width >= 1 && kernel_width <= 2 * WIN
           && (int64_t) w_planes * oversample < 65536;        // row index packed in 16 bits
}

// Scratch: the padded HBM copy of the table (none when the tables fit LDS) and a tail of 256
// bytes (table maximum of the fp16 form at its start, the chunk counter of long launches 128 bytes in).
size_t kimg_degrid_mfma_workspace_bytes(int P, int w_planes, int oversample, int kernel_width)
{
    if (!kimg_degrid_mfma_supported(P, w_planes, oversample, kernel_width))
        return 0;
    if (tables_fit_lds(w_planes, oversample, kernel_width))
        return 256;
    return (size_t) w_planes * oversample * 65 * sizeof(float2) * (kernel_width > WIN ? 2 : 1) + 256;
}

int kimg_degrid_mfma(const void *grid, int64_t grid_row_stride, int64_t grid_pol_stride,
                     int grid_size, int P, const int16_t *uv, const int16_t *w_plane,
                     const float *weights, void *vis, int64_t num_vis, const void *convolve_kernel,
                     int w_planes, int oversample, int kernel_width, void *workspace,
                     size_t workspace_bytes, int arith, hipStream_t stream)
{
    const bool f16 = arith == KIMG_ARITH_SPLIT_FP16;
    const bool in_lds = tables_fit_lds(w_planes, oversample, kernel_width);
    if (!in_lds && (workspace == nullptr
                    || workspace_bytes < kimg_degrid_mfma_workspace_bytes(P, w_planes, oversample,
                                                                         kernel_width)))
        return KIMG_EWORKSPACE;
    unsigned char *padded = static_cast<unsigned char *>(workspace);
    tls_queue = (workspace != nullptr && workspace_bytes >= 256)
        ? reinterpret_cast<unsigned long long *>(padded + workspace_bytes - 128) : nullptr;
    const int K = kernel_width;
    const bool wide = K > WIN;
    const int Kh = wide ? (K + 1) / 2 : K;
    const int nblk = wide ? 2 : 1;
    // instantiated for 1 and 2 polarizations; 3 or 4 run as 2 + 1 / 2 + 2 (register budget)
    for (int p0 = 0; p0 < P; p0 += 2) {
        const int pn = P - p0 >= 2 ? 2 : 1;
        const float *g = (const float *) grid + 2 * p0 * grid_pol_stride;
        for (int jb = 0; jb < nblk; jb++)
            for (int kb = 0; kb < nblk; kb++) {
                tap_split ts;
                ts.K = K;
                ts.tv0 = jb * Kh;
                ts.Kv = jb ? K - Kh : Kh;
                ts.tu0 = kb * Kh;
                ts.Ku = kb ? K - Kh : Kh;
                int rc;
#define LAUNCH(PP, NWV, TAPSV, TWOV) rc = launch<PP, NWV, TAPSV, TWOV>(g, grid_row_stride, \
        grid_pol_stride, grid_size, uv, w_plane, weights + p0, (float *) vis + 2 * p0, num_vis, \
        (const float2 *) convolve_kernel, w_planes, oversample, ts, P, stream)
                const bool doubled = lds_bytes(12, w_planes, oversample, 64) <= LDS_LIMIT;
#define LAUNCH_TG(PP, NWV, TAPSV, TWOV) do { if (f16_tg) rc = launch<PP, NWV, TAPSV, TWOV, true, true>(g, \
        grid_row_stride, grid_pol_stride, grid_size, uv, w_plane, weights + p0, (float *) vis + 2 * p0, \
        num_vis, (const float2 *) convolve_kernel, w_planes, oversample, ts, P, stream, padded, \
        tab_max_offset); else rc = launch<PP, NWV, TAPSV, TWOV, true, false>(g, grid_row_stride, \
        grid_pol_stride, grid_size, uv, w_plane, weights + p0, (float *) vis + 2 * p0, num_vis, \
        (const float2 *) convolve_kernel, w_planes, oversample, ts, P, stream, padded, 0); } while (0)
                const bool f16_tg = f16;
                const size_t tab_max_offset = workspace_bytes >= 256 ? workspace_bytes - 256 : 0;
                // Diagonal blocks of a wide kernel use the same taps for rows and columns: one
                // table, handled exactly like a narrow kernel's.
                const bool two = wide && jb != kb;
                const bool single_in_lds = lds_bytes(12, w_planes, oversample, 32) <= LDS_LIMIT;
                if (two) {
                    if (!in_lds) {
                        if (pn == 1) LAUNCH_TG(1, 12, 32, true); else LAUNCH_TG(2, 8, 32, true);
                    } else {
                        if (pn == 1) LAUNCH(1, 12, 32, true); else LAUNCH(2, 8, 32, true);
                    }
                } else if (!single_in_lds) {
                    if (pn == 1) LAUNCH_TG(1, 12, 32, false); else LAUNCH_TG(2, 8, 32, false);
                } else if (pn == 1) {
                    // KIMG_ARITH_SPLIT_FP16: fp16 hi/lo form (two window rows per matrix instruction)
                    if (f16) {
                        if (doubled)
                            rc = launch<1, 12, 64, false, false, true>(g, grid_row_stride, grid_pol_stride,
                                grid_size, uv, w_plane, weights + p0, (float *) vis + 2 * p0, num_vis,
                                (const float2 *) convolve_kernel, w_planes, oversample, ts, P, stream);
                        else
                            rc = launch<1, 12, 32, false, false, true>(g, grid_row_stride, grid_pol_stride,
                                grid_size, uv, w_plane, weights + p0, (float *) vis + 2 * p0, num_vis,
                                (const float2 *) convolve_kernel, w_planes, oversample, ts, P, stream);
                    } else if (doubled) LAUNCH(1, 12, 64, false); else LAUNCH(1, 12, 32, false);
                } else {
                    if (f16) {
                        if (doubled)
                            rc = launch<2, 8, 64, false, false, true>(g, grid_row_stride, grid_pol_stride,
                                grid_size, uv, w_plane, weights + p0, (float *) vis + 2 * p0, num_vis,
                                (const float2 *) convolve_kernel, w_planes, oversample, ts, P, stream);
                        else
                            rc = launch<2, 8, 32, false, false, true>(g, grid_row_stride, grid_pol_stride,
                                grid_size, uv, w_plane, weights + p0, (float *) vis + 2 * p0, num_vis,
                                (const float2 *) convolve_kernel, w_planes, oversample, ts, P, stream);
                    } else if (doubled) LAUNCH(2, 8, 64, false); else LAUNCH(2, 8, 32, false);
                }
#undef LAUNCH
#undef LAUNCH_TG
                if (rc)
                    return rc;
            }
    }
    return 0;
}

// (kimg_preload, api.hip)
KIMG_PRELOAD_THIS_UNIT(dg_table_max_kernel)
