// MFMA window gridder for gfx950 (CDNA4) -- the hot kernel of the path.
//
// Replaces Gridder.static_run / grid.mako:63-197 of the reference for kernel widths <= 64
// (widths above 32 as 2 x 2 tap blocks, see tap_split).
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
//    a[j] = vis*wgt*conj(kv[j]), b[k] = conj(ku[k]).  The complex product is folded into ONE
//    real MFMA per 32x16-cell tile: the MFMA K=2 dimension carries (Re a, Im a) and the N
//    dimension carries interleaved (re, im) output columns, B = [[ku.re, -ku.im], [ku.im,
//    ku.re]].  Two tiles cover the 32x32 window: 2 MFMAs per visibility and polarization.
//    All per-tap arithmetic therefore runs on the matrix pipe; the VALU only prepares one A
//    and two B elements per lane per visibility.
//  * The separable kernel table (W x OV rows, padded to 32 taps) lives in LDS when it fits
//    (up to 512 rows); otherwise -- hundreds of W planes, or the two tables of a wide kernel --
//    the operands come from a zero-padded copy in HBM (template flag TG, served by L1/L2).
//    Visibilities are loaded 64 at a time (lane i <-> visibility i, coalesced, two batches
//    prefetched), staged per wave in LDS and read back as broadcasts.  The scalar unit is a
//    scarce resource (one issue slot shared by a SIMD's waves), so the window test runs once
//    per group of 8 visibilities on bounds reduced with DPP; the hot loop is a hand-pipelined
//    sequence of 4-visibility stages (operand reads one stage ahead of 8 back-to-back MFMAs)
//    with no flush code inside.
//  * Accumulator columns are (re, im)-interleaved, so a flush writes 128 contiguous bytes
//    per half-wave -- the shape global float atomics run at full rate for.
//  * Window slack 32-K lets consecutive visibilities whose footprints differ by a few
//    cells share accumulators without any flush.
//
// Algorithmic work per visibility: 8*K*K*P flop (complex MAC per tap) + 6*K*P (a-vector);
// executed: 2 MFMA x 2048 MAC per visibility per polarization (32x32 window).
#include "kimg_common.h"
#include <limits.h>
#include <string.h>
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WIN = 32;

// Accumulators of one 32x32 window of complex cells, as two MFMA tiles of 32 rows x 16
// complex columns with (re, im) interleaved along the MFMA N dimension:
// tile t, register k, lane l  <->  row (k&3) + 8*(k>>2) + 4*(l>>5),
//                                  complex column ((l&31)>>1) + 16*t, part l&1 (0 re, 1 im).
template <int P>
struct window_acc {
    f32x16 t0[P];
    f32x16 t1[P];
};

// 32-bit mask of the window indices i whose mapping W + ((i - W) & 31) changes when the
// origin moves from W to nW (all of them when |nW - W| >= 32).
__device__ inline uint32_t changed_mask(int W, int nW)
{
    const int d = nW - W;
    if (d == 0)
        return 0u;
    if (d >= WIN || d <= -WIN)
        return 0xffffffffu;
    // forward move: the d lowest offsets wrap forward; backward: the |d| highest wrap back
    const uint32_t offsets = d > 0 ? (1u << d) - 1u : ~((1u << (WIN + d)) - 1u);
    const int rot = W & 31;                         // offset o <-> index (W + o) & 31
    return rot ? (offsets << rot) | (offsets >> (32 - rot)) : offsets;
}

// Flush (atomically add and clear) every accumulator cell whose grid mapping differs
// between window origin (Wu, Wv) and (nWu, nWv); `full` flushes everything.  One wave
// instruction covers two grid rows x 128 contiguous bytes (16 complex cells), the shape
// global float atomics run at full rate for.  Registers whose rows and columns are all
// unaffected are skipped with a scalar test.
//
// Window moves are frequent on fast tracks (every few visibilities for the 30-tap blocks of a wide
// kernel, whose slack is 2 cells), and a wave issues at most one instruction every few cycles, so
// the flush is built for a short INSTRUCTION STREAM: 32-bit byte offsets on a scalar base
// (global_atomic ... s[base]), the grid row of register k as two vector instructions plus one
// multiply-add per tile, and no per-cell test that the kind of move makes unnecessary --
// ROWS says whether row predicates are needed at all (false for the common pure column move),
// and windows that lie wholly inside the grid (nearly all) skip the bounds tests.
// SCALED: multiply by scale[p] on the way out (fp16 form); the float32 form's scale is 1.
template <int P, bool SCALED, bool ROWS, bool COLS, bool BOUNDS>
__device__ __attribute__((always_inline)) inline void flush_cells(
    window_acc<P> &acc, float *__restrict__ grid, unsigned row_bytes, int64_t pol_stride, int Gg,
    int Wu, int Wv, uint32_t row_mask, uint32_t col_mask, int lane, const float (&scale)[P])
{
    const int part = lane & 1;
    const int cx0 = (lane & 31) >> 1;
    const int c = (lane >> 5) * 4 - Wv;             // window row of register k = (rk + c) & 31
    const uint32_t lane_rows = row_mask >> ((lane >> 5) * 4);   // bit rk <-> this lane's row of register k
    unsigned base[2];
    bool col_sel[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int cx = cx0 + 16 * t;
        const int xold = Wu + ((cx - Wu) & 31);
        base[t] = (unsigned) Wv * row_bytes + (unsigned) (2 * xold + part) * 4u;   // (wraps like the sum)
        col_sel[t] = COLS ? ((col_mask >> cx) & 1u) != 0 : false;
        if (BOUNDS && (unsigned) xold >= (unsigned) Gg)
            base[t] = 0xffffffffu;                  // marks a column outside the grid
    }
#pragma unroll
    for (int k = 0; k < 16; k++) {
        constexpr uint32_t one = 1u;
        const int rk = (k & 3) + 8 * (k >> 2);          // MFMA 32x32 C/D row of register k (lower half)
        if (ROWS && !COLS && (row_mask & ((one << rk) | (one << (rk + 4)))) == 0)
            continue;                                   // uniform: nothing of register k moves
        const unsigned rr = (unsigned) (rk + c) & 31u;
        const bool row_sel = ROWS ? ((lane_rows >> rk) & 1u) != 0 : false;
        const bool y_ok = BOUNDS ? (unsigned) (Wv + (int) rr) < (unsigned) Gg : true;
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const bool sel = (ROWS || COLS) ? (row_sel || col_sel[t]) : true;
            const unsigned off = rr * row_bytes + base[t];      // v_mad_u32_u24
            const bool inside = BOUNDS ? (y_ok && base[t] != 0xffffffffu) : true;
#pragma unroll
            for (int p = 0; p < P; p++) {
                const float v = t ? acc.t1[p][k] : acc.t0[p][k];
                if (sel && inside && v != 0.0f) {
                    float *cell = reinterpret_cast<float *>(
                        reinterpret_cast<char *>(grid + 2 * p * pol_stride) + off);
                    atomicAdd(cell, SCALED ? v * scale[p] : v);
                }
                if (t)
                    acc.t1[p][k] = sel ? 0.0f : v;
                else
                    acc.t0[p][k] = sel ? 0.0f : v;
            }
        }
    }
}

template <int P, bool SCALED>
__device__ __attribute__((always_inline)) inline void flush_window(
    window_acc<P> &acc, float *__restrict__ grid, int64_t row_stride, int64_t pol_stride,
    int Gg, int Wu, int Wv, int nWu, int nWv, bool full, int lane, const float (&scale)[P])
{
    const uint32_t row_mask = full ? 0xffffffffu : changed_mask(Wv, nWv);
    const uint32_t col_mask = full ? 0xffffffffu : changed_mask(Wu, nWu);
    if ((row_mask | col_mask) == 0)
        return;
    const unsigned row_bytes = (unsigned) row_stride * 8u;  // (grid_size * row_bytes < 2^32: checked on the host)
    const bool inside = Wu >= 0 && Wv >= 0 && Wu + WIN <= Gg && Wv + WIN <= Gg;
#define KIMG_FLUSH(ROWS, COLS, BOUNDS) flush_cells<P, SCALED, ROWS, COLS, BOUNDS>( \
        acc, grid, row_bytes, pol_stride, Gg, Wu, Wv, row_mask, col_mask, lane, scale)
    // the two common moves of a window inside the grid get their own short code; everything else
    // (both axes at once, jumps and other full flushes, windows at the grid's rim) the general form
    if (inside && !full && row_mask == 0)
        KIMG_FLUSH(false, true, false);         // along u only
    else if (inside && !full && col_mask == 0)
        KIMG_FLUSH(true, false, false);         // along v only
    else
        KIMG_FLUSH(true, true, true);
#undef KIMG_FLUSH
}

template <int P>
struct vis_raw {
    int2 uv;
    int wp;
    float2 v[P];
    float w[P];
};

typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int GROUP = 8;        // visibilities sharing one window check
constexpr int GROUP_COUNT = 64 / GROUP;
// LDS kernel-table rows hold 32 zero-padded taps.  When the LDS budget allows, each row is
// stored twice (ROW = 64) so that tap (lane - first_tap) mod 32 is a plain "lane + offset"
// address; otherwise (ROW = 32) the wrap costs two more VALU operations per address.

// Operands of SUB staged visibilities (LDS reads in flight or landed):
// c = (Re s, Im s) for lanes 0-31, (Im s, -Re s) for lanes 32-63, so that a = c . kv gives
// Re(s conj kv) resp. Im(s conj kv); kv = this lane's row tap; b0/b1 = this lane's column
// tap component (re or im, chosen by the lane's address) for the two tiles.
template <int P, int SUB>
struct sub_ops {
    float2 c[P][SUB];
    float2 kv[SUB];
    v2f b[SUB];             // (tile 0, tile 1) column operand: adjacent registers for v_pk_mul_f32
};

// min / max over each aligned group of 8 lanes with DPP (VALU rate, no LDS round trips)
__device__ inline int group8_min(int v)
{
    v = min(v, __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true));     // quad_perm [1,0,3,2]
    v = min(v, __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, true));     // quad_perm [2,3,0,1]
    v = min(v, __builtin_amdgcn_mov_dpp(v, 0x141, 0xf, 0xf, true));    // row_half_mirror
    return v;
}

__device__ inline int group8_max(int v)
{
    v = max(v, __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true));
    v = max(v, __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, true));
    v = max(v, __builtin_amdgcn_mov_dpp(v, 0x141, 0xf, 0xf, true));
    return v;
}

// Kernel widths above 32 are gridded as 2 x 2 blocks of taps, one launch per block: the
// launch handles row taps [tv0, tv0 + Kv) and column taps [tu0, tu0 + Ku) of the K-tap kernel
// (each at most 32 wide), which is itself a gridding with a narrower kernel and a shifted
// origin.  Off-diagonal blocks need different row and column taps, hence TWO tables in LDS.
struct tap_split {
    int K;              // full kernel width (row stride of the table in HBM, uv_bias)
    int tv0, Kv;        // row (v) taps of this launch
    int tu0, Ku;        // column (u) taps of this launch
};

// TG: the padded table(s) do not fit LDS (many W planes: the reference's default w-step gives
// hundreds per slice) and are read from a zero-padded copy in HBM instead ([rows][32] taps,
// built per call by pad_table_kernel; served by L1/L2 -- rows recur along a track).  Everything
// else is unchanged; only the operand reads are global loads.
// ---- fp16 hi/lo formulation (F16 = true) ------------------------------------------------------
// Each fp32 operand x is carried as two halfs, hi = fp16(x) and lo = fp16(x - hi) (22 bits), and
// a product as hi*hi + hi*lo + lo*hi (the dropped lo*lo is 2^-22 of it).  A complex multiply-add
// therefore needs 6 of the 16 k-slots of v_mfma_f32_32x32x16_f16, so ONE instruction (8 passes)
// carries TWO visibilities -- lanes 0-31 (k 0..7) hold one, lanes 32-63 (k 8..15) the other --
// against one visibility per 16-pass v_mfma_f32_32x32x2_f32: a quarter of the matrix time.
// fp16 has little exponent range, so the operands are scaled by powers of two (exact): the table
// by S (largest tap -> [2^13, 2^14)), the samples of a wave by T (largest sample -> [1, 2)); the
// accumulators then hold grid * S*S*T and are multiplied by 1/(S*S*T) when they are flushed.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// (lo | hi << 16), rounded to nearest (v_cvt_pk_f16_f32): truncation would bias every product
// the same way.  (The name dates from a first version that truncated.)
__device__ inline unsigned cvt_pk_f16_rtz(float lo, float hi)
{
    typedef float pk2f __attribute__((ext_vector_type(2)));
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const pk2f x = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(x, h2));
}

__device__ inline float f16_bits_to_f32(unsigned bits16)
{
    return (float) __builtin_bit_cast(_Float16, (unsigned short) bits16);
}

// x -> (hi | lo << 16)
__device__ inline unsigned split_f16(float x)
{
    const unsigned hi = cvt_pk_f16_rtz(x, 0.0f) & 0xffffu;
    const float rest = x - f16_bits_to_f32(hi);
    const unsigned lo = cvt_pk_f16_rtz(rest, 0.0f) & 0xffffu;
    return hi | (lo << 16);
}

// (hi | lo << 16) -> fp32
__device__ inline float join_f16(unsigned packed)
{
    return f16_bits_to_f32(packed & 0xffffu) + f16_bits_to_f32(packed >> 16);
}

// A complex tap as it is kept in the tables of the fp16 form: (re_hi | im_hi << 16, re_lo | im_lo << 16)
__device__ inline uint2 split_tap(float re, float im)
{
    const unsigned hi = cvt_pk_f16_rtz(re, im);
    const unsigned lo = cvt_pk_f16_rtz(re - f16_bits_to_f32(hi & 0xffffu), im - f16_bits_to_f32(hi >> 16));
    return make_uint2(hi, lo);
}

// Operands of SUBP staged PAIRS of visibilities (this lane's member of each pair)
template <int P, int SUBP>
struct pair_ops {
    float4 c[P][SUBP];      // (Re s, Im s, Im s, -Re s) * T
    uint2 kv[SUBP];         // row tap: (re_hi | im_hi << 16, re_lo | im_lo << 16), times S
    uint2 t0[SUBP], t1[SUBP];   // column taps of the two tiles, same format
};

#ifdef KIMG_GRID_TIMING
// (test builds only) per-wave {first, last} wall_clock64 of the last launch
__device__ long long *g_timing = nullptr;
#endif

template <int P, int NW, int SUB, int ROW, bool TWO, bool TG = false, bool F16 = false>
__global__ __launch_bounds__(NW * 64) void grid_mfma_kernel(
    float *__restrict__ grid, int64_t row_stride, int64_t pol_stride, int Gg,
    const float *__restrict__ weights_grid, int64_t wg_row_stride, int64_t wg_pol_stride,
    const int16_t *__restrict__ uv, const int16_t *__restrict__ w_plane,
    const float2 *__restrict__ vis, int64_t num_vis,
    const float2 *__restrict__ kern, int W, int OV, tap_split ts, int64_t vis_per_block,
    int p_total, int dbg, const unsigned char *__restrict__ padded,
    const unsigned *__restrict__ tab_max, int64_t chunk, int64_t scramble, unsigned long long *queue)
{
#ifdef KIMG_GRID_TIMING
    const long long t_begin = wall_clock64();
#endif
    static_assert(!TWO || ROW == 32 || TG, "two tables only fit LDS with single rows");
    static_assert(!F16 || SUB % 2 == 0, "fp16 form: visibilities go in pairs");
    extern __shared__ __align__(16) unsigned char smem[];
    const int table_rows = W * OV;
    const int table_bytes = table_rows * ROW * (int) sizeof(float2);
    const int u_table = TWO ? table_bytes : 0;           // byte offset of the column-tap table
    unsigned char *rec_base = smem + (TG ? 0 : (size_t) table_bytes * (TWO ? 2 : 1));
    // wave index: uniform by construction, but the compiler must be told (readfirstlane),
    // or every loop below is lowered to divergent (exec-masked) control flow
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    // per-wave staging of one batch of 64 visibilities:
    //   recs[64]       = byte offsets into the table of (row tu, tap (-mu) mod 32) and
    //                    (row tv, tap (-mv) mod 32)
    //   samples[P][64] = s = vis * density weight as (Re, Im, Im, -Re)
    //   origins[64]    = first-tap grid coordinates (mu, mv), used only when a group jumps
    int2 *recs = reinterpret_cast<int2 *>(rec_base) + wib * 64;
    float4 *samples = reinterpret_cast<float4 *>(rec_base + (size_t) NW * 64 * sizeof(int2))
                      + wib * 64 * P;
    int2 *origins = reinterpret_cast<int2 *>(rec_base + (size_t) NW * 64 * (sizeof(int2) + P * sizeof(float4)))
                    + wib * 64;
    const unsigned char *tbytes = TG ? padded : smem;

    // This wave's contiguous range.
    const int64_t block_start = (int64_t) blockIdx.x * vis_per_block;
    int64_t block_end = block_start + vis_per_block;
    if (block_end > num_vis)
        block_end = num_vis;
    const int64_t span = block_end > block_start ? block_end - block_start : 0;
    // The waves of a SIMD (wave index mod 4) get spans of unequal length, (100 - q) : 100 : (100 + q)
    // percent for the three waves of a 12-wave block (dbg bits 8-15; 0 = equal spans), capped at
    // about one batch of 64 visibilities for long spans.  The SIMD's total work is unchanged, but
    // its waves reach their end flush -- a burst of float atomics limited by the CU's atomic rate --
    // at different times, so that part of it overlaps the remaining waves' arithmetic.  (Measured
    // at 341 visibilities per wave: 89.5 -> 81 us per launch with q = 12, i.e. spans of 4-6 batches.)
    int stagger = (dbg >> 8) & 0xff;
    if (NW != 12)
        stagger = 0;
    else if (span > 0 && (int64_t) stagger * span > 100 * 64 * NW)
        stagger = (int) (100 * 64 * NW / span);         // about one batch per wave at most
    auto wave_edge = [&](int w) {
        int64_t cum = 0;                        // cumulative weight of waves [0, w)
        for (int i = 0; i < w; i++)
            cum += 100 + stagger * ((i >> 2) - 1);
        return span * cum / (100 * NW) / 64 * 64;
    };
    const int64_t start = block_start + wave_edge(wib);
    const int64_t end = wib == NW - 1 ? block_end : block_start + wave_edge(wib + 1);

    // What a wave grids.  chunk == 0: the contiguous range [start, end) above.  chunk > 0 (long
    // launches): the stream is cut into chunks of `chunk` visibilities (a multiple of 64); a wave
    // starts with the chunk of its own number and then takes the next free one from a counter in
    // device memory (`queue`), until none is left.  With one contiguous range per wave the launch
    // ends in a long tail: the three waves of a SIMD do not advance at the same pace (the oldest
    // wave wins the arbitration for the matrix pipe), so the first is done after 72 % of the
    // launch and the last runs alone for the final 13 % -- at a fraction of the SIMD's rate.  Taking
    // work by the chunk keeps all waves busy to the end; a chunk's end costs what a baseline
    // boundary costs, one window flush.  Without a counter (`queue` null: no workspace given) wave g
    // takes chunks g, g + waves, g + 2 waves, ... instead.
    struct batch_pos {
        int64_t b, e;       // first visibility of the batch; end of its chunk (b >= e: no batch)
    };
    const int64_t wave_id = (int64_t) blockIdx.x * NW + wib, waves = (int64_t) gridDim.x * NW;
    // Ticket t is chunk (t * scramble) mod chunks_total (scramble is coprime to the number of chunks):
    // the chunks in flight at any moment are spread over the whole stream.  Neighbouring chunks are
    // neighbouring pieces of one baseline's track, and waves that flush their windows into the same
    // region of the grid at the same time contend for the same cache lines in L2 (measured: handing
    // out the first 85 % of a launch in 8 x longer chunks, in stream order, costs 4 %; gridding a
    // stream that was physically interleaved gains 10 %).
    const int64_t chunks_total = chunk > 0 ? (num_vis + chunk - 1) / chunk : 0;
    auto chunk_pos = [&](int64_t t) __attribute__((always_inline)) {
        batch_pos p;
        const int64_t c = t < chunks_total ? (int64_t) ((unsigned long long) t * (unsigned long long) scramble
                                                        % (unsigned long long) chunks_total) : 0;
        p.b = t < chunks_total ? c * chunk : 0;
        p.e = t < chunks_total ? (p.b + chunk < num_vis ? p.b + chunk : num_vis) : 0;
        return p;
    };
    // the ticket for the chunk after the current one is drawn when the current one is entered, and
    // only looked at when it is left (the atomic's round trip is off the critical path)
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
    batch_pos p0;
    if (chunk > 0) {
        p0 = chunk_pos(wave_id);
        draw();
    } else {
        p0.b = start;
        p0.e = end;
    }
    batch_pos p1 = next_pos(p0), p2 = next_pos(p1);

    auto load_raw = [&](const batch_pos &pos, vis_raw<P> &raw) __attribute__((always_inline)) {
        const int64_t b = pos.b, end = pos.e;
        int64_t ii = b + lane;
        ii = ii < end ? ii : end - 1;
        raw.uv = reinterpret_cast<const int2 *>(uv)[ii];
        raw.wp = w_plane[ii];
#pragma unroll
        for (int p = 0; p < P; p++)
            raw.v[p] = vis[ii * p_total + p];
    };
    const int half = Gg / 2;
    auto coords_ok = [&](const batch_pos &pos, const vis_raw<P> &raw) __attribute__((always_inline)) {
        const int u = (short) (raw.uv.x & 0xffff), v = (short) (raw.uv.x >> 16);
        const int su = (short) (raw.uv.y & 0xffff), sv = (short) (raw.uv.y >> 16);
        return pos.b + lane < pos.e && (unsigned) (u + half) < (unsigned) Gg
               && (unsigned) (v + half) < (unsigned) Gg && (unsigned) su < (unsigned) OV
               && (unsigned) sv < (unsigned) OV && (unsigned) raw.wp < (unsigned) W;
    };
    auto gather = [&](const batch_pos &b, vis_raw<P> &raw) __attribute__((always_inline)) {
        const int u = (short) (raw.uv.x & 0xffff), v = (short) (raw.uv.x >> 16);
        const int64_t wa = coords_ok(b, raw) ? (int64_t) (v + half) * wg_row_stride + (u + half) : 0;
#pragma unroll
        for (int p = 0; p < P; p++)
            raw.w[p] = weights_grid[wa + p * wg_pol_stride];
    };

    // Two-deep prefetch of the visibility stream: issue the first loads before staging the
    // kernel table so that their latency hides behind it.
    vis_raw<P> r0, r1;
    const bool active = p0.b < p0.e;
    if (active) {
        load_raw(p0, r0);
        if (p1.b < p1.e)
            load_raw(p1, r1);
        else
            r1 = r0;
    }

    // Stage the kernel table(s): taps [tap0, tap0 + Kp) of every row, zero-padded to 32 taps
    // (and stored twice when ROW == 64).  Two taps (16 bytes) per thread when alignment allows.
    auto stage_table = [&](unsigned char *dst, int tap0, int Kp) __attribute__((always_inline)) {
        if (((ts.K | tap0 | Kp) & 1) == 0) {
            const float4 *kern2 = reinterpret_cast<const float4 *>(kern);
            float4 *table2 = reinterpret_cast<float4 *>(dst);
            // all of a thread's loads are issued before its first LDS write: one L2 round trip
            // instead of one per iteration
            constexpr int STG = 6;
            const int total = table_rows * 16;
            for (int base = threadIdx.x; base < total; base += NW * 64 * STG) {
                float4 v[STG];
#pragma unroll
                for (int i = 0; i < STG; i++) {
                    const int idx = base + i * NW * 64;
                    const int row = idx >> 4, t2 = idx & 15;
                    v[i] = (idx < total && 2 * t2 < Kp)
                               ? kern2[(((int64_t) row * ts.K + tap0) >> 1) + t2]
                               : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                }
#pragma unroll
                for (int i = 0; i < STG; i++) {
                    const int idx = base + i * NW * 64;
                    const int row = idx >> 4, t2 = idx & 15;
                    if (idx < total) {
                        table2[row * (ROW / 2) + t2] = v[i];
                        if (ROW == 64)
                            table2[row * (ROW / 2) + 16 + t2] = v[i];
                    }
                }
            }
        } else {
            float2 *table = reinterpret_cast<float2 *>(dst);
            for (int idx = threadIdx.x; idx < table_rows * ROW; idx += NW * 64) {
                const int row = idx / ROW, t = idx & 31;
                table[idx] = t < Kp ? kern[(int64_t) row * ts.K + tap0 + t] : make_float2(0.0f, 0.0f);
            }
        }
    };
    if (!TG) {
        stage_table(smem, ts.tv0, ts.Kv);
        if (TWO)
            stage_table(smem + table_bytes, ts.tu0, ts.Ku);
    }
    __syncthreads();
    float S_scale = 1.0f;
    if (F16 && TG) {
        // table in HBM: already split by pad_table_kernel, with S from the same maximum
        const unsigned m = *tab_max;
        const int e = (int) (m >> 23) - 127;
        S_scale = m ? __uint_as_float((unsigned) (13 - e + 127) << 23) : 1.0f;
    } else if (F16) {
        // largest |component| of the table -> S, then every tap is split in place
        unsigned *s_tabmax = reinterpret_cast<unsigned *>(rec_base);   // (staging area, not yet in use)
        const unsigned *words = reinterpret_cast<const unsigned *>(smem);
        const int nwords = table_bytes * (TWO ? 2 : 1) / 4;
        unsigned m = 0;
        for (int i = threadIdx.x; i < nwords; i += NW * 64)
            m = max(m, words[i] & 0x7fffffffu);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            m = max(m, (unsigned) __shfl_xor((int) m, off, WAVE));
        if (lane == 0)
            s_tabmax[wib] = m;
        __syncthreads();
        for (int w = 0; w < NW; w++)
            m = max(m, s_tabmax[w]);
        // m in [2^e, 2^(e+1)): S = 2^(13 - e)
        const int e = (int) (m >> 23) - 127;
        S_scale = m ? __uint_as_float((unsigned) (13 - e + 127) << 23) : 1.0f;
        uint2 *taps = reinterpret_cast<uint2 *>(smem);
        for (int i = threadIdx.x; i < nwords / 2; i += NW * 64) {
            const uint2 t = taps[i];
            taps[i] = split_tap(__uint_as_float(t.x) * S_scale, __uint_as_float(t.y) * S_scale);
        }
        __syncthreads();
    }
    if (!active)
        return;
    gather(p0, r0);

    const int uv_bias = (ts.K - 1) / 2 - Gg / 2;        // grid.py:1038
    const int Su = WIN - ts.Ku, Sv = WIN - ts.Kv;       // window slack along u and v
    const bool h = lane >= 32;                          // MFMA k index of this lane
    const int part = lane & 1;                          // 0: real part column, 1: imaginary
    // B operand: k=0 pairs with Re(a): (ku.re, -ku.im); k=1 with Im(a): (ku.im, ku.re).
    // Which component a lane needs is folded into its table address, the sign is one VALU op.
    const bool b_take_im = (h != (part != 0));
    const float b_sign = (part && !h) ? -1.0f : 1.0f;
    const int lane_v = (lane & 31) * 8;                                 // row tap, bytes
    const int lane_u = ((lane & 31) >> 1) * 8 + (b_take_im ? 4 : 0);    // column tap component
    const int lane_s = h ? 8 : 0;                                       // (Re,Im) or (Im,-Re)
    const unsigned lds_base = (unsigned) (uintptr_t) (__attribute__((address_space(3))) unsigned char *) smem;
    const uint64_t lane_uv = ((uint64_t) (lds_base + (unsigned) lane_v) << 32) | (lds_base + (unsigned) lane_u);
    // fp16 form: which member of a pair this lane serves, the (whole) column tap it reads, and how
    // its 6 k-slots come from the tap (re_hi, im_hi | re_lo, im_lo): even columns (real part of the
    // result) take (re_hi, im_hi, re_lo, im_lo, re_hi, im_hi) as stored, odd ones (-im_hi, re_hi,
    // -im_lo, re_lo, -im_hi, re_hi); the row operand is (re_hi, im_hi, re_hi, im_hi, re_lo, im_lo)
    const int member = lane >> 5;
    const int lane_u16 = ((lane & 31) >> 1) * 8;
    const unsigned sel = part ? 0x01000302u : 0x03020100u;      // odd columns swap (re, im)
    const unsigned flip = part ? 0x00008000u : 0u;              // ... and negate im
    // fp16 form: per-polarization sample scale T = 2^-E, kept as the integer E so that every decision
    // about it is scalar integer work (gfx950 has no scalar float compare).  E_cur = scale of what
    // the accumulators hold, E_stage = the scale the staging code last gave to a group,
    // out_scale = 1 / (S S T) = 2^E / S^2.
    constexpr int E_NONE = 0x7fff;      // no sample seen yet
    constexpr int E_WIDE = 0x7ffe;      // (per group) staged unscaled: one visibility at a time
    int E_cur[P], E_stage[P];
    float out_scale[P];
#pragma unroll
    for (int p = 0; p < P; p++) {
        E_cur[p] = E_stage[p] = E_NONE;
        out_scale[p] = 1.0f;
    }
    // A group (8 visibilities) keeps the scale of its predecessor while its largest sample, times
    // that scale, stays within [2^-6, 2) (density weights that differ by an order of magnitude
    // between neighbouring cells must not cost a flush per cell); otherwise the window is flushed and a new scale chosen
    // (both ways: a run of small samples after a large one gets its own scale).  A group whose own
    // samples spread over more than 2^11 is handled one visibility at a time; such a visibility
    // keeps the current scale while it stays within [2^-14, 2) of it (the same lower end as the
    // smallest member of an ordinary group: its row operand still has 17 good bits on the weakest
    // tap that matters), so that an occasional tiny sample costs no flush.
#ifndef KIMG_T_SPREAD
#define KIMG_T_SPREAD 11
#endif
    constexpr int T_SPREAD = KIMG_T_SPREAD;
    auto sample_exponent = [](unsigned bits) {      // |x| in [2^e, 2^(e+1)) -> e, clamped
        const int e = (int) (bits >> 23) - 127;
        return e < -100 ? -100 : (e > 100 ? 100 : e);
    };
    auto keeps_scale = [&](int e, int E, int low = -6) {    // 2^e * 2^-E within [2^low, 2)
        return E != E_NONE && e - E <= 0 && e - E >= low;
    };

    window_acc<P> acc;
#pragma unroll
    for (int p = 0; p < P; p++)
        for (int k = 0; k < 16; k++) {
            acc.t0[p][k] = 0.0f;
            acc.t1[p][k] = 0.0f;
        }
    bool have = false;
    int Wu = 0, Wv = 0;

    // Move the window so that [lo_u, hi_u] x [lo_v, hi_v] (first-tap coordinates) fits.
    auto fit_window = [&](int lo_u, int hi_u, int lo_v, int hi_v) __attribute__((always_inline)) {
        if (!have) {
            Wu = lo_u - (Su - (hi_u - lo_u)) / 2;
            Wv = lo_v - (Sv - (hi_v - lo_v)) / 2;
            have = true;
            return;
        }
        const bool bad_u = lo_u < Wu || hi_u > Wu + Su;
        const bool bad_v = lo_v < Wv || hi_v > Wv + Sv;
        if (!(bad_u || bad_v))
            return;
        // leave all the slack ahead in the direction of travel (tracks are smooth curves)
        const int nWu = !bad_u ? Wu : (hi_u > Wu + Su ? lo_u : hi_u - Su);
        const int nWv = !bad_v ? Wv : (hi_v > Wv + Sv ? lo_v : hi_v - Sv);
        const bool full = nWu - Wu >= WIN || Wu - nWu >= WIN || nWv - Wv >= WIN || Wv - nWv >= WIN;
        if (!(dbg & 2))
            flush_window<P, F16>(acc, grid, row_stride, pol_stride, Gg, Wu, Wv, nWu, nWv, full, lane,
                            out_scale);
        Wu = nWu;
        Wv = nWv;
    };

    // LDS byte addresses of this lane's row tap / column tap for a staged record
    // (unsigned: with the table in HBM the 32-bit offset then rides on the scalar base pointer
    // -- global_load v, v_off, s[base] -- instead of costing a sign extension and a 64-bit add)
    auto addr_v = [&](int ry) __attribute__((always_inline)) {
        return (unsigned) (ROW == 64 ? ry + lane_v : (ry & ~0xff) | ((ry + lane_v) & 0xf8));
    };
    auto addr_u = [&](int rx) __attribute__((always_inline)) {
        return (unsigned) (ROW == 64 ? rx + lane_u : (rx & ~0xff) | ((rx + lane_u) & 0xfc));
    };
    auto addr_u16 = [&](int rx) __attribute__((always_inline)) {         // whole column tap
        return (unsigned) (ROW == 64 ? rx + lane_u16 : (rx & ~0xff) | ((rx + lane_u16) & 0xf8));
    };

    // ---- software-pipeline stages over sub-blocks of SUB staged visibilities ---------------
    int2 rec[SUB];
    auto stage_a = [&](int first) __attribute__((always_inline)) {         // record reads
        if constexpr (F16) {
#pragma unroll
            for (int t = 0; t < SUB / 2; t++)
                rec[t] = recs[first + 2 * t + member];      // this lane's member of pair t
        } else {
#pragma unroll
            for (int t = 0; t < SUB; t++)
                rec[t] = recs[first + t];
        }
    };
    auto stage_b = [&](auto &o, int first) __attribute__((always_inline)) {     // sample + kernel-table reads
        if constexpr (F16) {
#pragma unroll
            for (int t = 0; t < SUB / 2; t++) {
                const unsigned au = addr_u16(rec[t].x);
                o.kv[t] = *reinterpret_cast<const uint2 *>(tbytes + addr_v(rec[t].y));
                o.t0[t] = *reinterpret_cast<const uint2 *>(tbytes + au);
                o.t1[t] = *reinterpret_cast<const uint2 *>(ROW == 64 ? tbytes + au + 128
                                                                     : tbytes + (au ^ 128u));
#pragma unroll
                for (int p = 0; p < P; p++)
                    o.c[p][t] = samples[p * 64 + first + 2 * t + member];
            }
        } else {
#pragma unroll
        for (int t = 0; t < SUB; t++) {
            if constexpr (ROW == 64 && !TG) {
                // both LDS addresses with ONE 64-bit add (v_lshl_add_u64): (rx, ry) + (lane_u,
                // lane_v) with the table's LDS base folded into the lane constants; no carry can
                // cross (LDS addresses are far below 2^32)
                typedef const __attribute__((address_space(3))) v2f *lds_f2;
                typedef const __attribute__((address_space(3))) float *lds_f;
                const uint64_t sum = (((uint64_t) (unsigned) rec[t].y << 32) | (unsigned) rec[t].x)
                                     + lane_uv;
                const unsigned au = (unsigned) sum, av = (unsigned) (sum >> 32);
                const v2f kvt = *reinterpret_cast<lds_f2>((uintptr_t) av);
                o.kv[t] = make_float2(kvt.x, kvt.y);
                o.b[t].x = *reinterpret_cast<lds_f>((uintptr_t) au);
                o.b[t].y = *reinterpret_cast<lds_f>((uintptr_t) (au + 128));        // column + 16
            } else {
            const unsigned au = addr_u(rec[t].x);
            o.kv[t] = *reinterpret_cast<const float2 *>(tbytes + addr_v(rec[t].y));
            o.b[t].x = *reinterpret_cast<const float *>(tbytes + au);
            o.b[t].y = *reinterpret_cast<const float *>(                                 // column + 16
                ROW == 64 ? tbytes + au + 128 : tbytes + (au ^ 128u));
            }
#pragma unroll
            for (int p = 0; p < P; p++)
                o.c[p][t] = *reinterpret_cast<const float2 *>(
                    reinterpret_cast<const unsigned char *>(samples + p * 64 + first + t) + lane_s);
        }
        }
    };
    // fp16 form: the 6 k-slots of this lane's visibility (row operand) and of its column taps
    auto join_tap = [&](const uint2 &tp) __attribute__((always_inline)) {        // -> (re, im) in fp32
        // (scalar on purpose: with v_pk_fma_f32 / v_pk_add_f32 here the results were not repeatable)
        v2f r;
        r.x = f16_bits_to_f32(tp.x & 0xffffu) + f16_bits_to_f32(tp.y & 0xffffu);
        r.y = f16_bits_to_f32(tp.x >> 16) + f16_bits_to_f32(tp.y >> 16);
        return r;
    };
    auto row_operand = [&](const float4 &c, v2f kv, bool zero) __attribute__((always_inline)) {
        // (Re, Im)(s conj kv) * S * T = (sx, sy) kv.re + (sy, -sx) kv.im, two lanes of packed math
        float are = fmaf(c.x, kv.x, c.y * kv.y);
        float aim = fmaf(c.z, kv.x, c.w * kv.y);
        if (zero)
            are = aim = 0.0f;
        u32x4 A;
        A[0] = cvt_pk_f16_rtz(are, aim);                            // (re_hi, im_hi)
        const float re_lo = are - f16_bits_to_f32(A[0] & 0xffffu);
        const float im_lo = aim - f16_bits_to_f32(A[0] >> 16);
        A[1] = A[0];
        A[2] = cvt_pk_f16_rtz(re_lo, im_lo);                        // (re_lo, im_lo)
        A[3] = 0;
        return A;
    };
    auto col_operand = [&](const uint2 &tp) __attribute__((always_inline)) {
        u32x4 B;
        B[0] = __builtin_amdgcn_perm(tp.x, tp.x, sel) ^ flip;
        B[1] = __builtin_amdgcn_perm(tp.y, tp.y, sel) ^ flip;
        B[2] = B[0];
        B[3] = 0;
        return B;
    };
    auto stage_c = [&](const auto &o) __attribute__((always_inline)) {
        if constexpr (F16) {
            u32x4 A[P][SUB / 2], B0[SUB / 2], B1[SUB / 2];
#pragma unroll
            for (int t = 0; t < SUB / 2; t++) {
                const v2f kv = join_tap(o.kv[t]);
#pragma unroll
                for (int p = 0; p < P; p++)
                    A[p][t] = row_operand(o.c[p][t], kv, false);
                B0[t] = col_operand(o.t0[t]);
                B1[t] = col_operand(o.t1[t]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < SUB / 2; t++)
#pragma unroll
                for (int p = 0; p < P; p++) {
                    acc.t0[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                        __builtin_bit_cast(f16x8, A[p][t]), __builtin_bit_cast(f16x8, B0[t]), acc.t0[p], 0, 0, 0);
                    acc.t1[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                        __builtin_bit_cast(f16x8, A[p][t]), __builtin_bit_cast(f16x8, B1[t]), acc.t1[p], 0, 0, 0);
                }
            return;
        } else {
        // all operands first, then the MFMAs back to back (no VALU -> MFMA wait states)
        float a[P][SUB];
        v2f b[SUB];
#pragma unroll
        for (int t = 0; t < SUB; t++) {
            b[t] = o.b[t] * v2f{b_sign, b_sign};        // one v_pk_mul_f32
#pragma unroll
            for (int p = 0; p < P; p++)
                a[p][t] = fmaf(o.c[p][t].x, o.kv[t].x, o.c[p][t].y * o.kv[t].y);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < SUB; t++)
#pragma unroll
            for (int p = 0; p < P; p++) {
                acc.t0[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[p][t], b[t].x, acc.t0[p], 0, 0, 0);
                acc.t1[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[p][t], b[t].y, acc.t1[p], 0, 0, 0);
            }
        }
    };
    // A group whose visibilities do not share one window position (a jump inside the group):
    // rolled loop, window positioned per visibility, operands re-read from LDS.  Rare.
    auto switch_scale = [&](const int (&E_new)[P]) __attribute__((always_inline)) {
        // the accumulators hold sums in units of the old scale: write them out, then go on
        // in the new one
        bool held = false;
#pragma unroll
        for (int p = 0; p < P; p++)
            held |= E_cur[p] != E_NONE;
        if (have && held)
            flush_window<P, F16>(acc, grid, row_stride, pol_stride, Gg, Wu, Wv, Wu, Wv, true, lane,
                            out_scale);
#pragma unroll
        for (int p = 0; p < P; p++)
            if (E_new[p] != E_NONE && E_new[p] != E_WIDE) {
                E_cur[p] = E_new[p];
                out_scale[p] = ldexpf(1.0f / (S_scale * S_scale), E_new[p]);
            }
    };
    // `unscaled` (fp16 form): the group's samples were staged as they are because they spread over
    // too many binades for one scale; every visibility then checks the accumulators' scale itself.
    auto slow_group = [&](int first, bool unscaled) __attribute__((always_inline)) {
        for (int t = 0; t < GROUP; t++) {
            const int idx = first + t;
            const int2 org = origins[idx];
            float2 c[P];
            bool nz = false;
#pragma unroll
            for (int p = 0; p < P; p++) {
                c[p] = *reinterpret_cast<const float2 *>(
                    reinterpret_cast<const unsigned char *>(samples + p * 64 + idx) + lane_s);
                nz |= (c[p].x != 0.0f) | (c[p].y != 0.0f);
            }
            if (!__builtin_amdgcn_readfirstlane((int) nz))
                continue;
            fit_window(__builtin_amdgcn_readfirstlane(org.x), __builtin_amdgcn_readfirstlane(org.x),
                       __builtin_amdgcn_readfirstlane(org.y), __builtin_amdgcn_readfirstlane(org.y));
            const int2 r = recs[idx];
            if constexpr (F16) {
                int vis_shift[P];
#pragma unroll
                for (int p = 0; p < P; p++)
                    vis_shift[p] = 0;
                if (unscaled) {
                    int E_new[P];
                    bool change = false;
#pragma unroll
                    for (int p = 0; p < P; p++) {
                        unsigned m = max(__float_as_uint(c[p].x) & 0x7fffffffu,
                                         __float_as_uint(c[p].y) & 0x7fffffffu);
                        m = (unsigned) __builtin_amdgcn_readfirstlane((int) m);
                        if (m >= 0x7f800000u)
                            m = 0;
                        E_new[p] = E_NONE;
                        if (m != 0 && !keeps_scale(sample_exponent(m), E_cur[p], -14)) {
                            E_new[p] = sample_exponent(m);
                            change = true;
                        }
                    }
                    if (change)
                        switch_scale(E_new);     // (fit_window above has already placed the window)
#pragma unroll
                    for (int p = 0; p < P; p++)
                        vis_shift[p] = E_cur[p] != E_NONE ? -E_cur[p] : 0;
                }
                // one visibility, carried by the first member of a pair; the second contributes 0
                const unsigned au16 = addr_u16(r.x);
                const uint2 kvp = *reinterpret_cast<const uint2 *>(tbytes + addr_v(r.y));
                const uint2 tp0 = *reinterpret_cast<const uint2 *>(tbytes + au16);
                const uint2 tp1 = *reinterpret_cast<const uint2 *>(ROW == 64 ? tbytes + au16 + 128
                                                                           : tbytes + (au16 ^ 128u));
                const v2f kvj = join_tap(kvp);
                const u32x4 B0 = col_operand(tp0), B1 = col_operand(tp1);
#pragma unroll
                for (int p = 0; p < P; p++) {
                    float4 cs = samples[p * 64 + idx];
                    cs.x = ldexpf(cs.x, vis_shift[p]);
                    cs.y = ldexpf(cs.y, vis_shift[p]);
                    cs.z = ldexpf(cs.z, vis_shift[p]);
                    cs.w = ldexpf(cs.w, vis_shift[p]);
                    const u32x4 A = row_operand(cs, kvj, member != 0);
                    acc.t0[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                        __builtin_bit_cast(f16x8, A), __builtin_bit_cast(f16x8, B0), acc.t0[p], 0, 0, 0);
                    acc.t1[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                        __builtin_bit_cast(f16x8, A), __builtin_bit_cast(f16x8, B1), acc.t1[p], 0, 0, 0);
                }
                continue;
            }
            const unsigned au = addr_u(r.x);
            const float2 kv = *reinterpret_cast<const float2 *>(tbytes + addr_v(r.y));
            const float b0 = *reinterpret_cast<const float *>(tbytes + au) * b_sign;
            const float b1 = *reinterpret_cast<const float *>(
                ROW == 64 ? tbytes + au + 128 : tbytes + (au ^ 128u)) * b_sign;
#pragma unroll
            for (int p = 0; p < P; p++) {
                const float a = fmaf(c[p].x, kv.x, c[p].y * kv.y);
                acc.t0[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc.t0[p], 0, 0, 0);
                acc.t1[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc.t1[p], 0, 0, 0);
            }
        }
    };

    for (; p0.b < p0.e; p0 = p1, p1 = p2, p2 = next_pos(p2)) {
        const batch_pos &b = p0;
        // ---- stage batch b (lane i <-> visibility b + i) and its per-group bounds ----------
        int gb_u, gb_v;         // per group of 8 lanes: min | max << 16 of the first-tap coordinates
        int gmax_bits[P];       // fp16 form: bits of the group's largest sample | (spread too wide)
        int Evec[P];            // fp16 form: the scale (exponent) this lane's group was staged with,
                                // E_WIDE (staged unscaled) or E_NONE (no sample yet)
        {
            const bool ok = coords_ok(b, r0);
            const int u = (short) (r0.uv.x & 0xffff), v = (short) (r0.uv.x >> 16);
            const int su = (short) (r0.uv.y & 0xffff), sv = (short) (r0.uv.y >> 16);
            const int mu = u - uv_bias + ts.tu0, mv = v - uv_bias + ts.tv0;
            // row byte offset (a multiple of ROW * 8) + tap byte offset (< 256)
            int2 r;
            r.x = (ok ? (r0.wp * OV + su) * (ROW * 8) : 0) + ((-mu) & 31) * 8 + u_table;
            r.y = (ok ? (r0.wp * OV + sv) * (ROW * 8) : 0) + ((-mv) & 31) * 8;
            recs[lane] = r;
            origins[lane] = make_int2(mu, mv);
            bool live = false;
            float2 sp[P];
#pragma unroll
            for (int p = 0; p < P; p++)
                sp[p] = ok ? make_float2(r0.v[p].x * r0.w[p], r0.v[p].y * r0.w[p])
                           : make_float2(0.0f, 0.0f);                       // grid.py:1046
            if constexpr (F16) {
                // keep the samples in fp16 range: a scale T = 2^-E per group of 8 visibilities and
                // polarization (see keeps_scale / T_SPREAD above), chosen group after group so that
                // it changes only when it must
#pragma unroll
                for (int p = 0; p < P; p++) {
                    unsigned m = max(__float_as_uint(sp[p].x) & 0x7fffffffu,
                                     __float_as_uint(sp[p].y) & 0x7fffffffu);
                    if (m >= 0x7f800000u)
                        m = 0;      // NaN / Inf samples poison their own footprint, not the scale
                    const int gmax = group8_max((int) m);
                    const int gmin = group8_min(m ? (int) m : INT_MAX);
                    // low bit of the group maximum := "spread too wide" (the mantissa's last bit
                    // does not matter to the choice of scale)
                    const bool wide = gmax != 0 && (gmax >> 23) - (gmin >> 23) > T_SPREAD;
                    gmax_bits[p] = (gmax & ~1) | (wide ? 1 : 0);
                }
                int Eg[GROUP_COUNT][P];
#pragma unroll
                for (int g = 0; g < GROUP_COUNT; g++) {
#pragma unroll
                    for (int p = 0; p < P; p++) {
                        const unsigned gm = (unsigned) __builtin_amdgcn_readlane(gmax_bits[p], g * GROUP);
                        if ((gm & ~1u) != 0 && !(gm & 1u)
                            && !keeps_scale(sample_exponent(gm), E_stage[p]))
                            E_stage[p] = sample_exponent(gm);
                        Eg[g][p] = E_stage[p];
                    }
                }
                // a group with a wide spread in ANY polarization is staged unscaled in all of them
                bool wide_lane = false;
#pragma unroll
                for (int p = 0; p < P; p++)
                    wide_lane |= (gmax_bits[p] & 1) != 0;
#pragma unroll
                for (int p = 0; p < P; p++) {
                    int e = E_NONE;     // (no sample of this polarization so far: nothing to scale)
#pragma unroll
                    for (int g = 0; g < GROUP_COUNT; g++)
                        e = (lane >> 3) == g ? Eg[g][p] : e;
                    Evec[p] = wide_lane ? E_WIDE : e;
                    const int shift = (wide_lane || e == E_NONE) ? 0 : -e;
                    sp[p].x = ldexpf(sp[p].x, shift);
                    sp[p].y = ldexpf(sp[p].y, shift);
                }
            }
#pragma unroll
            for (int p = 0; p < P; p++) {
                samples[p * 64 + lane] = make_float4(sp[p].x, sp[p].y, sp[p].y, -sp[p].x);
                live |= (sp[p].x != 0.0f) | (sp[p].y != 0.0f);
            }
            // bounds over each aligned group of 8 lanes; dead visibilities do not constrain
            // (first-tap coordinates fit 16 bits: |u| < 32768 and the kernel is at most 64 wide)
            const int gmin_u = group8_min(live ? mu : SHRT_MAX);
            const int gmax_u = group8_max(live ? mu : SHRT_MIN);
            const int gmin_v = group8_min(live ? mv : SHRT_MAX);
            const int gmax_v = group8_max(live ? mv : SHRT_MIN);
            gb_u = (gmax_u << 16) | (gmin_u & 0xffff);
            gb_v = (gmax_v << 16) | (gmin_v & 0xffff);
        }
        // advance the global prefetch pipeline: b+64 gets its (dependent) weight gather,
        // b+128 its raw loads
        r0 = r1;
        if (p1.b < p1.e)
            gather(p1, r0);
        if (p2.b < p2.e)
            load_raw(p2, r1);
        __builtin_amdgcn_wave_barrier();        // LDS is in-order per wave; just pin the order

        const int count = b.e - b.b < 64 ? (int) (b.e - b.b) : 64;
        const int npairs = (count + 2 * SUB - 1) / (2 * SUB);
        typename std::conditional<F16, pair_ops<P, SUB / 2>, sub_ops<P, SUB>>::type X, Y;
        stage_a(0);
        stage_b(X, 0);
        stage_a(SUB);
        __builtin_amdgcn_sched_barrier(0);
        // Outer loop: decide how the group starting at pair q is handled (this is where the
        // window moves and cells are flushed).  Inner loop: the hot path, free of any flush
        // code so that the accumulators stay put in their registers; it runs on through the
        // following groups for as long as they fit the current window.
        // (the decision is a plain scalar branch: each arm has its own copy of the pipeline step, so
        // that no loop-carried flag has to live in a vector register)
        auto group_bounds = [&](int first, int &lo_u, int &hi_u, int &lo_v, int &hi_v)
            __attribute__((always_inline)) {
            // (min | max << 16) per axis: two v_readlane instead of four
            const int bu = __builtin_amdgcn_readlane(gb_u, first);
            const int bv = __builtin_amdgcn_readlane(gb_v, first);
            lo_u = (short) (bu & 0xffff);
            hi_u = bu >> 16;
            lo_v = (short) (bv & 0xffff);
            hi_v = bv >> 16;
        };
        auto step = [&](int q, auto live_tag) __attribute__((always_inline)) {
            constexpr bool LIVE = decltype(live_tag)::value;
            const int first = q * 2 * SUB;
            // clamp look-ahead indices at the end of the batch (those operands are unused)
            const int next = q + 1 < npairs ? first + 2 * SUB : first;
            stage_b(Y, first + SUB);
            stage_a(next);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (LIVE)
                stage_c(X);
            __builtin_amdgcn_sched_barrier(0);
            stage_b(X, next);
            stage_a(next + SUB);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (LIVE)
                stage_c(Y);
            __builtin_amdgcn_sched_barrier(0);
        };
        // fp16 form: the scale group `first` was staged with, against the accumulators' scale:
        // 0 = the same (or nothing staged), 1 = another one (flush, then switch), 2 = staged
        // unscaled (one visibility at a time)
        int group_E[P];
        auto group_scale = [&](int first) __attribute__((always_inline)) {
            int verdict = 0;
            if constexpr (F16) {
#pragma unroll
                for (int p = 0; p < P; p++) {
                    group_E[p] = __builtin_amdgcn_readlane(Evec[p], first);
                    if (group_E[p] == E_WIDE)
                        verdict = 2;
                    else if (group_E[p] != E_NONE && group_E[p] != E_cur[p] && verdict == 0)
                        verdict = 1;
                }
            }
            return verdict;
        };
        int q = 0;
        while (q < npairs) {
            int lo_u, hi_u, lo_v, hi_v;
            group_bounds(q * 2 * SUB, lo_u, hi_u, lo_v, hi_v);
            const bool any = lo_u <= hi_u;
            const bool jump = hi_u - lo_u > Su || hi_v - lo_v > Sv;
            const int scale_verdict = any ? group_scale(q * 2 * SUB) : 0;
            if (scale_verdict == 1)
                switch_scale(group_E);
            if (any && !jump && scale_verdict != 2) {
                fit_window(lo_u, hi_u, lo_v, hi_v);     // whole group shares one window
                for (;;) {
                    step(q, std::true_type());
                    q++;
                    if (q >= npairs)
                        break;
                    if ((q * 2 * SUB) % GROUP == 0) {
                        // peek at the next group: stay in the hot loop only if it needs no flush
                        group_bounds(q * 2 * SUB, lo_u, hi_u, lo_v, hi_v);
                        const bool fits = lo_u <= hi_u && lo_u >= Wu && hi_u <= Wu + Su
                                          && lo_v >= Wv && hi_v <= Wv + Sv
                                          && group_scale(q * 2 * SUB) == 0;
                        if (!fits)
                            break;
                    }
                }
            } else {
                if (any)
                    slow_group(q * 2 * SUB, scale_verdict == 2);
                // the group's operands still pass through the pipeline (unused)
                do {
                    step(q, std::false_type());
                    q++;
                } while (q < npairs && (q * 2 * SUB) % GROUP != 0);
            }
        }
        // the next batch overwrites this wave's staging area
        __builtin_amdgcn_wave_barrier();
    }
    if (have && !(dbg & 1))
        flush_window<P, F16>(acc, grid, row_stride, pol_stride, Gg, Wu, Wv, Wu, Wv, true, lane, out_scale);
#ifdef KIMG_GRID_TIMING
    if (g_timing && lane == 0) {
        g_timing[2 * wave_id] = t_begin;
        g_timing[2 * wave_id + 1] = wall_clock64();
    }
#endif
}

template <int P>
constexpr int waves_per_block()
{
    return 8;
}

// Zero-padded copy of taps [tap0, tap0 + Kp) of every table row: [rows][ROW] float2 (ROW = 64:
// the 32 taps twice).
// F16: taps scaled by S (from the largest |component| of the whole table, *tab_max) and split
// into fp16 hi/lo pairs, as the kernel does in LDS for tables that live there.
template <int ROW, bool F16 = false>
__global__ __launch_bounds__(256) void pad_table_kernel(
    const float2 *__restrict__ kern, int rows, int K, int tap0, int Kp, float2 *__restrict__ out,
    const unsigned *__restrict__ tab_max = nullptr)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * ROW)
        return;
    const int row = idx / ROW, t = idx & 31;
    float2 v = t < Kp ? kern[(int64_t) row * K + tap0 + t] : make_float2(0.0f, 0.0f);
    if (F16) {
        const unsigned m = *tab_max;
        const int e = (int) (m >> 23) - 127;
        const float S = m ? __uint_as_float((unsigned) (13 - e + 127) << 23) : 1.0f;
        const uint2 t = split_tap(v.x * S, v.y * S);
        v = make_float2(__uint_as_float(t.x), __uint_as_float(t.y));
    }
    out[idx] = v;
}

__global__ __launch_bounds__(256) void table_max_kernel(const float *__restrict__ kern, int64_t n,
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

size_t lds_bytes(int P, int NW, int W, int OV, int row, int tables = 1)
{
    return (size_t) tables * W * OV * row * sizeof(float2)
           + (size_t) NW * 64 * (sizeof(int2) + P * sizeof(float4) + sizeof(int2));
}

constexpr size_t LDS_LIMIT = 160 * 1024;
// (512 until round 3; 384 measures 3-4 % faster on launches of ~7 M records -- a W-slice of the
// re-ordered resident store, where a wave only gets four or five chunks and the tail of the launch
// is one chunk long -- and the same on 50 M-record launches, whose chunks the cap below sets)
#ifndef KIMG_INTERLEAVE_MIN_CHUNK
#define KIMG_INTERLEAVE_MIN_CHUNK 384
#endif
#ifndef KIMG_INTERLEAVE_MAX_PARTS
#define KIMG_INTERLEAVE_MAX_PARTS 16
#endif
constexpr int64_t INTERLEAVE_MIN_CHUNK = KIMG_INTERLEAVE_MIN_CHUNK;     // visibilities: bounds the extra window flushes
constexpr int64_t INTERLEAVE_MAX_PARTS = KIMG_INTERLEAVE_MAX_PARTS;

template <int P, int ROW, int NW, bool TWO, bool TG = false, bool F16 = false>
int launch(float *grid, int64_t row_stride, int64_t pol_stride, int Gg, const float *wg,
           int64_t wg_row_stride, int64_t wg_pol_stride, const int16_t *uv,
           const int16_t *w_plane, const float2 *vis, int64_t num_vis, const float2 *kern,
           int W, int OV, const tap_split &ts, int p_total, hipStream_t stream,
           unsigned char *padded = nullptr, size_t tab_max_offset = 0,
           unsigned long long *queue = nullptr)
{
    constexpr int SUB = (P == 1 && NW <= 12) ? 4 : 2;       // pipeline depth bounded by the VGPR budget
    const size_t lds = TG ? lds_bytes(P, NW, 0, 0, ROW) : lds_bytes(P, NW, W, OV, ROW, TWO ? 2 : 1);
    unsigned *tab_max = nullptr;
    if (TG) {
        // the padded table(s) of this launch: row taps first, column taps behind them
        const int rows = W * OV;
        float2 *out = reinterpret_cast<float2 *>(padded);
        if (F16) {
            // (the last 256 bytes of the workspace hold the table maximum)
            tab_max = reinterpret_cast<unsigned *>(padded + tab_max_offset);
            KIMG_HIP(hipMemsetAsync(tab_max, 0, sizeof(unsigned), stream));
            const int64_t n = (int64_t) rows * ts.K * 2;
            table_max_kernel<<<kimg_divup(n, 256 * 8), 256, 0, stream>>>(
                reinterpret_cast<const float *>(kern), n, tab_max);
        }
        pad_table_kernel<ROW, F16><<<kimg_divup(rows * ROW, 256), 256, 0, stream>>>(
            kern, rows, ts.K, ts.tv0, ts.Kv, out, tab_max);
        if (TWO)
            pad_table_kernel<ROW, F16><<<kimg_divup(rows * ROW, 256), 256, 0, stream>>>(
                kern, rows, ts.K, ts.tu0, ts.Ku, out + (size_t) rows * ROW, tab_max);
    }
    {
        const int rc = kimg_dynamic_lds(
            reinterpret_cast<const void *>(&grid_mfma_kernel<P, NW, SUB, ROW, TWO, TG, F16>), LDS_LIMIT);
        if (rc)
            return rc;
    }
    // bits 8-15: span stagger of a SIMD's waves, percent.  Bits 0-1 (test builds with
    // -DKIMG_NO_ATOMICS, the counterpart of the reference's NO_ATOMICS switch,
    // imager_kernels/atomic.mako:29-40): no end flush / no window flushes -- wrong results, for
    // measuring what the float atomics cost.
#ifdef KIMG_NO_ATOMICS
    const int dbg = (12 << 8) | 3;
#else
    const int dbg = 12 << 8;
#endif
    // as many blocks resident per CU as the LDS (kernel table + staging) allows;
    // every block streams a contiguous span (a multiple of 64).
    const int per_cu = (!TG && lds <= LDS_LIMIT / 2) ? 2 : 1;
    const int blocks_max = kimg_window_cus_now() * per_cu;
    int64_t vis_per_block = (num_vis + blocks_max - 1) / blocks_max;
    vis_per_block = (vis_per_block + 63) / 64 * 64;
    if (vis_per_block < 64 * NW)
        vis_per_block = 64 * NW;
    const int blocks = (int) ((num_vis + vis_per_block - 1) / vis_per_block);
    // Long launches: every wave takes its work from up to 16 places of the stream (chunks of at
    // least ~1000 visibilities, see batch_pos in the kernel) instead of one contiguous range
    const int64_t waves = (int64_t) blocks * NW;
    int64_t parts = num_vis / (waves * INTERLEAVE_MIN_CHUNK);
    parts = parts > INTERLEAVE_MAX_PARTS ? INTERLEAVE_MAX_PARTS : parts;
    int64_t chunk = 0;
    if (parts >= 2)
        chunk = ((num_vis + waves * parts - 1) / (waves * parts) + 63) / 64 * 64;
    // chunk numbers are scrambled by a multiplier coprime to their count
    int64_t scramble = 1;
    if (chunk > 0) {
        const int64_t total = (num_vis + chunk - 1) / chunk;
        static const int64_t primes[] = {7919, 7907, 7901, 7883, 7879, 7877, 7873};
        for (int64_t m : primes)
            if (total % m != 0) {
                scramble = m;
                break;
            }
    }
    if (chunk > 0 && queue != nullptr)
        KIMG_HIP(hipMemsetAsync(queue, 0, sizeof(unsigned long long), stream));
    grid_mfma_kernel<P, NW, SUB, ROW, TWO, TG, F16><<<blocks, NW * 64, lds, stream>>>(
        grid, row_stride, pol_stride, Gg, wg, wg_row_stride, wg_pol_stride, uv, w_plane, vis,
        num_vis, kern, W, OV, ts, vis_per_block, p_total, dbg, padded, tab_max, chunk, scramble,
        chunk > 0 ? queue : nullptr);
    return kimg_launch_status();
}

} // namespace

// The kernels are instantiated for 1 and 2 polarizations (32 accumulator registers each keep
// two waves per SIMD without spills); 3 or 4 polarizations run as 2 + 1 / 2 + 2.  Kernel widths
// 33..64 run as 2 x 2 tap blocks with two single-row tables in LDS.
static bool tables_fit_lds(int P, int w_planes, int oversample, int kernel_width)
{
    const int tables = kernel_width > WIN ? 2 : 1;
    return lds_bytes(P > 1 ? 2 : 1, 8, w_planes, oversample, 32, tables) <= LDS_LIMIT;
}

bool kimg_grid_mfma_supported(int P, int w_planes, int oversample, int kernel_width)
{
    // any number of W planes: tables that do not fit LDS are read from a padded copy in HBM
    return P >= 1 && P <= 4 && kernel_width >= 1 && kernel_width <= 2 * WIN
           && (int64_t) w_planes * oversample * 256 * 2 < ((int64_t) 1 << 31);
}

// Where the kernel reads its table from: LDS when it fits -- except for the off-diagonal tap
// blocks of wide kernels, whose two single-row tables in LDS are no faster than doubled rows in
// HBM (the diagonal blocks need one table and always take the LDS form when it fits).
static bool table_in_lds(int P, int w_planes, int oversample, int kernel_width)
{
    return tables_fit_lds(P, w_planes, oversample, kernel_width) && kernel_width <= WIN;
}

// Scratch: the padded table copy (none when the kernel reads its table from LDS) and a tail of 256
// bytes -- the table's maximum (fp16 form, tables in HBM) at its start, the chunk counter of long
// launches 128 bytes in.  (A caller that gives a kernel with its table in LDS no scratch still
// works: its waves then take their chunks in a fixed order.)
size_t kimg_grid_mfma_workspace_bytes(int P, int w_planes, int oversample, int kernel_width)
{
    if (!kimg_grid_mfma_supported(P, w_planes, oversample, kernel_width))
        return 0;
    const bool fits = tables_fit_lds(P, w_planes, oversample, kernel_width);
    if (fits && kernel_width <= WIN)
        return 256;
    return (size_t) w_planes * oversample * 64 * sizeof(float2) * (kernel_width > WIN ? 2 : 1) + 256;
}

int kimg_grid_mfma(void *grid, int64_t grid_row_stride, int64_t grid_pol_stride, int grid_size,
                   int P, const float *weights_grid, int64_t wg_row_stride, int64_t wg_pol_stride,
                   const int16_t *uv, const int16_t *w_plane, const void *vis, int64_t num_vis,
                   const void *convolve_kernel, int w_planes, int oversample, int kernel_width,
                   void *workspace, size_t workspace_bytes, int arith, hipStream_t stream)
{
    const bool f16 = arith == KIMG_ARITH_SPLIT_FP16;
    const bool in_lds = table_in_lds(P, w_planes, oversample, kernel_width);
    if (!in_lds && (workspace == nullptr
                    || workspace_bytes < kimg_grid_mfma_workspace_bytes(P, w_planes, oversample,
                                                                       kernel_width)))
        return KIMG_EWORKSPACE;
    unsigned char *padded = static_cast<unsigned char *>(workspace);
    unsigned long long *queue = (workspace != nullptr && workspace_bytes >= 256)
        ? reinterpret_cast<unsigned long long *>(padded + workspace_bytes - 128) : nullptr;
    const int K = kernel_width;
    const bool wide = K > WIN;
    const int Kh = wide ? (K + 1) / 2 : K;              // taps per block along one axis
    const int nblk = wide ? 2 : 1;
    for (int p0 = 0; p0 < P; p0 += 2) {
        const int pn = P - p0 >= 2 ? 2 : 1;
        float *g = (float *) grid + 2 * p0 * grid_pol_stride;
        const float *wg = weights_grid + p0 * wg_pol_stride;
        const float2 *v = (const float2 *) vis + p0;
        const float2 *kern = (const float2 *) convolve_kernel;
        for (int jb = 0; jb < nblk; jb++)
            for (int kb = 0; kb < nblk; kb++) {
                tap_split ts;
                ts.K = K;
                ts.tv0 = jb * Kh;
                ts.Kv = jb ? K - Kh : Kh;
                ts.tu0 = kb * Kh;
                ts.Ku = kb ? K - Kh : Kh;
                int rc;
                // GO(P, ROW, NW, TWO, TG): exact-fp32 instruction, or with KIMG_ARITH_SPLIT_FP16 the
                // fp16 hi/lo form (two visibilities per matrix instruction)
#define GO(PP, ROWV, NWV, TWOV, TGV) do { \
        if (f16) rc = launch<PP, ROWV, NWV, TWOV, TGV, true>(g, grid_row_stride, grid_pol_stride, \
            grid_size, wg, wg_row_stride, wg_pol_stride, uv, w_plane, v, num_vis, kern, w_planes, \
            oversample, ts, P, stream, padded, tab_max_offset, queue); \
        else rc = launch<PP, ROWV, NWV, TWOV, TGV, false>(g, grid_row_stride, grid_pol_stride, \
            grid_size, wg, wg_row_stride, wg_pol_stride, uv, w_plane, v, num_vis, kern, w_planes, \
            oversample, ts, P, stream, padded, tab_max_offset, queue); } while (0)
                const size_t tab_max_offset = workspace_bytes >= 256 ? workspace_bytes - 256 : 0;
                // Diagonal blocks of a wide kernel take row and column taps from the same half of
                // the table: one table, which fits LDS whenever a narrow kernel's would.
                const bool two = wide && jb != kb;
                const bool single_in_lds = two ? false
                    : wide ? lds_bytes(pn, 8, w_planes, oversample, 32) <= LDS_LIMIT
                           : in_lds;
                if (two) {
                    // two tables: doubled rows in HBM, 12-wave blocks (P = 1) as for the LDS form
                    if (!in_lds) {
                        if (pn == 1)
                            GO(1, 64, 12, true, true);
                        else
                            GO(2, 64, 8, true, true);
                    } else if (pn == 1) {
                        if (lds_bytes(1, 12, w_planes, oversample, 32, 2) <= LDS_LIMIT)
                            GO(1, 32, 12, true, false);
                        else
                            GO(1, 32, 8, true, false);
                    } else {
                        GO(2, 32, 8, true, false);
                    }
                } else if (!single_in_lds) {
                    if (pn == 1)
                        GO(1, 64, 12, false, true);     // doubled rows: 5 % faster than single rows
                    else
                        GO(2, 64, 8, false, true);
                } else if (pn == 1) {
                    // 12-wave blocks, one per CU (LDS-bound), when the doubled table leaves room
#ifdef KIMG_GRID_NW16
                    if (lds_bytes(1, 16, w_planes, oversample, 64) <= LDS_LIMIT)
                        GO(1, 64, 16, false, false);
                    else
#endif
                    if (lds_bytes(1, 12, w_planes, oversample, 64) <= LDS_LIMIT)
                        GO(1, 64, 12, false, false);
                    else if (lds_bytes(1, 8, w_planes, oversample, 64) <= LDS_LIMIT)
                        GO(1, 64, 8, false, false);
                    else
                        GO(1, 32, 8, false, false);
                } else {
                    if (lds_bytes(2, 8, w_planes, oversample, 64) <= LDS_LIMIT)
                        GO(2, 64, 8, false, false);
                    else
                        GO(2, 32, 8, false, false);
                }
#undef GO
                if (rc)
                    return rc;
            }
    }
    return 0;
}

#ifdef KIMG_GRID_TIMING
extern "C" int kimg_debug_grid_timing(void *buffer)
{
    long long *p = static_cast<long long *>(buffer);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_timing), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#endif

// (kimg_preload, api.hip)
KIMG_PRELOAD_THIS_UNIT(table_max_kernel)
