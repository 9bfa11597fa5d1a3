// Hogbom CLEAN kernels: per-tile peak, global peak, PSF subtract, PSF patch bound, the
// radix-select passes of the noise estimate, and a device-resident minor-cycle loop.
// Mirrors clean.py:123-163, 295-353, 451-480, 566-587, 683-726, 848-891 of the reference.
//
// Peak selection is BIT-EXACT with the reference host path (CleanHost, clean.py:946-1075):
//  - within a tile: first strict maximum in row-major order (clean.py:953-958), and a tile
//    with no positive metric keeps value 0 and the (x0, y0) initial position (clean.py:950);
//  - across tiles: first maximum in row-major tile order (np.argmax, clean.py:1062);
//  - subtraction is dirty -= (loop_gain*pixel) * psf with separately rounded multiply and
//    subtract (clean.py:1044-1046): this file is built with -ffp-contract=off.
#include "kimg_common.h"
#include <limits.h>
#include <string.h>
#include <mutex>

namespace {

constexpr int TILE = 32;            // clean.py:996

struct best_t {
    float value;
    int idx;                        // row-major index; INT_MAX = none yet
};

__device__ inline bool better(const best_t &a, const best_t &b)
{
    return a.value > b.value || (a.value == b.value && a.idx < b.idx);
}

__device__ inline best_t wave_best(best_t b)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        best_t o;
        o.value = __shfl_xor(b.value, off, WAVE);
        o.idx = __shfl_xor(b.idx, off, WAVE);
        if (better(o, b))
            b = o;
    }
    return b;
}

// Block-wide argmax with the tie-break of better(); result valid in thread 0.
__device__ inline best_t block_best(best_t b)
{
    __shared__ best_t scratch[16];
    b = wave_best(b);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0)
        scratch[wv] = b;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = blockDim.x >> 6;
        for (int w = 1; w < nw; w++)
            if (better(scratch[w], b))
                b = scratch[w];
    }
    return b;
}

// (value, index) as one unsigned key: larger value first, then smaller index.  Values are
// non-negative floats (never NaN: a NaN metric never replaces a tile's best), whose bit patterns
// order like the numbers.  Key 0 = nothing.
typedef unsigned long long key_t;

__device__ inline key_t make_key(float value, int idx)
{
    return ((key_t) __float_as_uint(value) << 32) | (unsigned) ~idx;
}

__device__ inline key_t key_max(key_t a, key_t b) { return a > b ? a : b; }

template <int CTRL>
__device__ inline key_t key_dpp(key_t k)
{
    const unsigned lo = __builtin_amdgcn_mov_dpp((unsigned) k, CTRL, 0xf, 0xf, true);
    const unsigned hi = __builtin_amdgcn_mov_dpp((unsigned) (k >> 32), CTRL, 0xf, 0xf, true);
    return ((key_t) hi << 32) | lo;
}

// Maximum over each 16-lane row, in every lane of the row (DPP butterflies: ALU latency only)
__device__ inline key_t row_max_key(key_t k)
{
    k = key_max(k, key_dpp<0xB1>(k));       // quad_perm [1,0,3,2]
    k = key_max(k, key_dpp<0x4E>(k));       // quad_perm [2,3,0,1]
    k = key_max(k, key_dpp<0x141>(k));      // row_half_mirror
    k = key_max(k, key_dpp<0x140>(k));      // row_mirror
    return k;
}

__device__ inline key_t read_lane_key(key_t k, int lane)
{
    return ((key_t) (unsigned) __builtin_amdgcn_readlane((int) (k >> 32), lane) << 32)
           | (unsigned) __builtin_amdgcn_readlane((int) k, lane);
}

// Maximum over a block of up to 1024 threads (a multiple of 64), the same (uniform) value in
// every thread.  `s_keys` [16] is shared scratch; two uses must be separated by a barrier.
__device__ inline key_t block_max_key(key_t k, key_t *s_keys)
{
    k = row_max_key(k);
    const key_t w = key_max(key_max(read_lane_key(k, 0), read_lane_key(k, 16)),
                            key_max(read_lane_key(k, 32), read_lane_key(k, 48)));
    if ((threadIdx.x & 63) == 0)
        s_keys[threadIdx.x >> 6] = w;
    __syncthreads();
    const int nw = blockDim.x >> 6, e = threadIdx.x & 15;
    k = row_max_key(e < nw ? s_keys[e] : 0);
    return read_lane_key(k, 0);
}

// A workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every global
// store of the wave (its release semantics), which puts the latency of stores nobody is waiting
// for on a latency-critical chain.
__device__ inline void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// block_max_key with that barrier
__device__ inline key_t block_max_key_lds(key_t k, key_t *s_keys)
{
    k = row_max_key(k);
    const key_t w = key_max(key_max(read_lane_key(k, 0), read_lane_key(k, 16)),
                            key_max(read_lane_key(k, 32), read_lane_key(k, 48)));
    if ((threadIdx.x & 63) == 0)
        s_keys[threadIdx.x >> 6] = w;
    lds_barrier();
    const int nw = blockDim.x >> 6, e = threadIdx.x & 15;
    k = row_max_key(e < nw ? s_keys[e] : 0);
    return read_lane_key(k, 0);
}

template <int MODE>
__device__ inline float clean_metric(const float *__restrict__ dirty, int64_t addr,
                                     int64_t pol_stride, int P)
{
    if (MODE == KIMG_CLEAN_I)
        return fabsf(dirty[addr]);
    float value = 0.0f;                                // clean.py:962-964
    for (int p = 0; p < P; p++) {
        float pix = dirty[addr + p * pol_stride];
        value += pix * pix;
    }
    return value;
}

// Scan tile (tx, ty): pixels [x0,x1) x [y0,y1); 256 threads, 4 pixels each in row-major order.
template <int MODE>
__device__ inline void tile_peak(const float *__restrict__ dirty, int64_t row_stride,
                                 int64_t pol_stride, int width, int height, int P, int border,
                                 int tx, int ty, float *__restrict__ tile_max,
                                 int32_t *__restrict__ tile_pos, int tiles_x)
{
    const int x0 = tx * TILE + border, y0 = ty * TILE + border;
    best_t b = {0.0f, INT_MAX};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int idx = threadIdx.x + k * 256;
        const int x = x0 + (idx & 31), y = y0 + (idx >> 5);
        if (x < width - border && y < height - border) {
            float v = clean_metric<MODE>(dirty, (int64_t) y * row_stride + x, pol_stride, P);
            if (v > b.value) {
                b.value = v;
                b.idx = idx;
            }
        }
    }
    b = block_best(b);
    if (threadIdx.x == 0) {
        const int t = ty * tiles_x + tx;
        tile_max[t] = b.value;
        if (b.idx == INT_MAX) {             // clean.py:950 best_pos = (x0, y0)
            tile_pos[2 * t] = x0;
            tile_pos[2 * t + 1] = y0;
        } else {
            tile_pos[2 * t] = y0 + (b.idx >> 5);
            tile_pos[2 * t + 1] = x0 + (b.idx & 31);
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void update_tiles_kernel(
    const float *__restrict__ dirty, int64_t row_stride, int64_t pol_stride, int width,
    int height, int P, int border, float *__restrict__ tile_max, int32_t *__restrict__ tile_pos,
    int tiles_x, int tile_x0, int tile_y0)
{
    tile_peak<MODE>(dirty, row_stride, pol_stride, width, height, P, border,
                    tile_x0 + blockIdx.x, tile_y0 + blockIdx.y, tile_max, tile_pos, tiles_x);
}

// Global argmax over tiles, by a 1024-thread block; every thread returns the winning tile index
// (or -1 when there are no tiles).  All of a thread's loads are issued together and clamped
// instead of predicated (a duplicate of the last tile under a larger index never wins).
__device__ inline int peak_tile(const float *__restrict__ tile_max, int num_tiles, float &value)
{
    __shared__ key_t s_peak[16];
    key_t best = 0;
    constexpr int ROUND = 16;
    for (int base = threadIdx.x; base < num_tiles; base += ROUND * blockDim.x) {
        float v[ROUND];
#pragma unroll
        for (int k = 0; k < ROUND; k++)
            v[k] = tile_max[min(base + k * (int) blockDim.x, num_tiles - 1)];
        key_t c[ROUND];
#pragma unroll
        for (int k = 0; k < ROUND; k++)
            c[k] = make_key(v[k], base + k * (int) blockDim.x);
#pragma unroll
        for (int w = ROUND / 2; w > 0; w >>= 1)
#pragma unroll
            for (int k = 0; k < w; k++)
                c[k] = key_max(c[k], c[k + w]);
        best = key_max(best, c[0]);
    }
    best = block_max_key(best, s_peak);
    value = best ? __uint_as_float((unsigned) (best >> 32)) : -1.0f;
    return best ? ~(int) (unsigned) best : -1;
}

__global__ __launch_bounds__(1024) void find_peak_kernel(
    const float *__restrict__ dirty, int64_t row_stride, int64_t pol_stride, int P,
    const float *__restrict__ tile_max, const int32_t *__restrict__ tile_pos, int num_tiles,
    float *__restrict__ peak_value, int32_t *__restrict__ peak_pos, float *__restrict__ peak_pixel)
{
    float value;
    int t = peak_tile(tile_max, num_tiles, value);
    if (threadIdx.x == 0 && t >= 0) {
        const int y = tile_pos[2 * t], x = tile_pos[2 * t + 1];
        *peak_value = value;
        peak_pos[0] = y;
        peak_pos[1] = x;
        for (int p = 0; p < P; p++)
            peak_pixel[p] = dirty[p * pol_stride + (int64_t) y * row_stride + x];
    }
}

struct pixel_t { float v[4]; };

__global__ __launch_bounds__(256) void subtract_psf_kernel(
    float *__restrict__ dirty, float *__restrict__ model, int64_t row_stride, int64_t pol_stride,
    int width, int height, int P, const float *__restrict__ psf, int64_t psf_row_stride,
    int64_t psf_pol_stride, int psf_x0, int psf_y0, int patch_w, int patch_h,
    const float *__restrict__ peak_pixel, int pos_x, int pos_y, int start_x, int start_y,
    float loop_gain)
{
    const int gx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int gy = blockIdx.y * 4 + (threadIdx.x >> 6);
    float scale[4];
    for (int p = 0; p < P; p++)
        scale[p] = loop_gain * peak_pixel[p];
    if (gx == 0 && gy == 0)
        for (int p = 0; p < P; p++)
            model[p * pol_stride + (int64_t) pos_y * row_stride + pos_x] += scale[p];
    if (gx >= patch_w || gy >= patch_h)
        return;
    const int x = start_x + gx, y = start_y + gy;
    if (x < 0 || x >= width || y < 0 || y >= height)
        return;
    const int64_t pa = (int64_t) (psf_y0 + gy) * psf_row_stride + (psf_x0 + gx);
    const int64_t ia = (int64_t) y * row_stride + x;
    for (int p = 0; p < P; p++) {
        const float t = scale[p] * psf[p * psf_pol_stride + pa];
        dirty[p * pol_stride + ia] -= t;
    }
}

// ---- device-resident minor cycles ------------------------------------------------------
struct clean_state {
    int count;          // cycles completed
    int done;           // threshold reached (or `limit` cycles done)
    int limit;          // maximum number of cycles for this call
    float threshold;    // stop when the peak metric falls below it (kept here, not in the kernel
                        // arguments, so that one captured graph serves every threshold)
    int pos_y, pos_x;
    int pad[2];
    float scale[4];     // loop_gain * pixel at the current peak
};

template <int MODE>
__global__ __launch_bounds__(1024) void cycle_find_peak_kernel(
    const float *__restrict__ dirty, float *__restrict__ model, int64_t row_stride,
    int64_t pol_stride, int P, const float *__restrict__ tile_max,
    const int32_t *__restrict__ tile_pos, int num_tiles, float loop_gain,
    clean_state *__restrict__ state, float *__restrict__ log)
{
    // The kernel is a chain of dependent memory round trips; keep it short: the state words are
    // fetched together with the tile maxima (not before them), and the pixel and model values of
    // all polarizations are fetched together before anything is stored.
    const int4 st = *reinterpret_cast<const int4 *>(state);    // count, done, limit, threshold
    const int count = st.x, done = st.y, limit = st.z;
    const float threshold = __int_as_float(st.w);
    float value;
    const int t = peak_tile(tile_max, num_tiles, value);
    const int p = threadIdx.x;          // one thread per polarization from here on
    if (p >= P || done)
        return;
    if (t < 0 || value < threshold || count >= limit) {   // clean.py:1065-1066
        if (p == 0)
            state->done = 1;
        return;
    }
    const int2 pos = *reinterpret_cast<const int2 *>(tile_pos + 2 * t);
    const int y = pos.x, x = pos.y;
    const int64_t a = p * pol_stride + (int64_t) y * row_stride + x;
    const float pix = dirty[a], mod = model[a];
    float *entry = log + (int64_t) count * (3 + P);
    const float s = loop_gain * pix;            // clean.py:1044
    state->scale[p] = s;
    entry[3 + p] = s;
    model[a] = mod + s;                         // clean.py:1047
    if (p == 0) {
        entry[0] = value;
        entry[1] = __int_as_float(y);
        entry[2] = __int_as_float(x);
        state->pos_y = y;
        state->pos_x = x;
        state->count = count + 1;
    }
}

// One workgroup per 32x32 block of the tile lattice that the PSF patch can touch: subtract
// the scaled PSF from the block's pixels that lie in the patch, then (if the block is a real
// tile) rescan the tile.  Fuses _subtract_psf + _update_tile of clean.py:1067-1074.
template <int MODE>
__global__ __launch_bounds__(256) void cycle_subtract_update_kernel(
    float *__restrict__ dirty, int64_t row_stride, int64_t pol_stride, int width, int height,
    int P, const float *__restrict__ psf, int64_t psf_row_stride, int64_t psf_pol_stride,
    int psf_w, int psf_h, int patch_w, int patch_h, int border,
    float *__restrict__ tile_max, int32_t *__restrict__ tile_pos, int tiles_x, int tiles_y,
    const clean_state *__restrict__ state)
{
    // one round trip for all the state words (they share a cache line)
    const int4 st = *reinterpret_cast<const int4 *>(state);    // count, done, limit, threshold
    const int2 pos = *reinterpret_cast<const int2 *>(&state->pos_y);
    const int done = st.y, py = pos.x, px = pos.y;
    const float4 sc = *reinterpret_cast<const float4 *>(state->scale);
    const float scale[4] = {sc.x, sc.y, sc.z, sc.w};
    if (done)
        return;
    const int x0 = px - patch_w / 2, y0 = py - patch_h / 2;      // clean.py:1024-1027
    // floor division: the lattice extends into the border with negative indices
    const int bx0 = (x0 - border) >= 0 ? (x0 - border) / TILE : -((border - x0 + TILE - 1) / TILE);
    const int by0 = (y0 - border) >= 0 ? (y0 - border) / TILE : -((border - y0 + TILE - 1) / TILE);
    const int tx = bx0 + (int) blockIdx.x, ty = by0 + (int) blockIdx.y;
    const int ox = tx * TILE + border, oy = ty * TILE + border;
    const int psf_dx = psf_w / 2 - px, psf_dy = psf_h / 2 - py;  // psf index = image index + d
    const bool is_tile = tx >= 0 && tx < tiles_x && ty >= 0 && ty < tiles_y;

    best_t b = {0.0f, INT_MAX};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int idx = threadIdx.x + k * 256;
        const int x = ox + (idx & 31), y = oy + (idx >> 5);
        if (x < 0 || x >= width || y < 0 || y >= height)
            continue;
        const int64_t ia = (int64_t) y * row_stride + x;
        const bool in_patch = x >= x0 && x < x0 + patch_w && y >= y0 && y < y0 + patch_h;
        const bool in_tile = is_tile && x < width - border && y < height - border;
        float metric = 0.0f;
        if (MODE == KIMG_CLEAN_I) {
            float d = dirty[ia];
            if (in_patch) {
                const float t = scale[0] * psf[(int64_t) (y + psf_dy) * psf_row_stride + (x + psf_dx)];
                d -= t;
                dirty[ia] = d;
                for (int p = 1; p < P; p++) {
                    const float tp = scale[p] * psf[p * psf_pol_stride
                                                    + (int64_t) (y + psf_dy) * psf_row_stride + (x + psf_dx)];
                    dirty[p * pol_stride + ia] -= tp;
                }
            }
            metric = fabsf(d);
        } else {
            for (int p = 0; p < P; p++) {
                float d = dirty[p * pol_stride + ia];
                if (in_patch) {
                    const float t = scale[p] * psf[p * psf_pol_stride
                                                   + (int64_t) (y + psf_dy) * psf_row_stride + (x + psf_dx)];
                    d -= t;
                    dirty[p * pol_stride + ia] = d;
                }
                metric += d * d;
            }
        }
        if (in_tile && metric > b.value) {
            b.value = metric;
            b.idx = idx;
        }
    }
    if (!is_tile)
        return;
    __shared__ key_t s_keys[16];
    const key_t kb = block_max_key(b.idx == INT_MAX ? 0 : make_key(b.value, b.idx), s_keys);
    if (threadIdx.x == 0) {
        const int t = ty * tiles_x + tx;
        if (kb == 0) {                      // clean.py:950 best_pos = (x0, y0), value 0
            tile_max[t] = 0.0f;
            tile_pos[2 * t] = ox;
            tile_pos[2 * t + 1] = oy;
        } else {
            const int idx = ~(int) (unsigned) kb;
            tile_max[t] = __uint_as_float((unsigned) (kb >> 32));
            tile_pos[2 * t] = oy + (idx >> 5);
            tile_pos[2 * t + 1] = ox + (idx & 31);
        }
    }
}

// ---- one launch per minor cycle ----------------------------------------------------------
// The two-launch cycle above is a chain of dependent memory round trips with a kernel boundary
// in the middle.  Here every workgroup of the subtract/update launch finds the global peak
// ITSELF, so the cycle is one launch with no communication between its workgroups.
//
// What a launch needs to know the peak.  Every tile is either one the LAST cycle rewrote -- its new
// record is a "delta": at most (patch / 32 + 2)^2 of them (30 for the 133 x 111 patch of a measured
// PSF), written by the lattice workgroups of the last launch into the slot of their lattice
// position -- or it is not, and the best of THOSE was worked out, off the critical path, during the
// last launch (`rest`).  So a workgroup's first round trip is 48 bytes of state and one delta per
// thread of its first few waves, every candidate arrives with its record (position, pixel values: no
// dependent load), and the peak is one block reduction away: 1.25 us after the first instruction.
// (Rounds 1-2: every workgroup read a 48-byte delta slot and a 16-byte best-two record per THREAD,
// 64 KB, and then fetched its candidate's record: 1.9 us.)
//
// Workgroups of a launch, per channel:
//   * lattice workgroups, one per 32 x 32 block of the tile lattice that the PSF patch can touch:
//     peak, then subtract the PSF from the block, rescan it, and write the new tile record to the
//     OTHER delta table (double-buffered by launch parity), never to the base arrays, so that
//     slower workgroups of the same launch still see the inputs unchanged;
//   * the "keeper": peak, then the log entry, the model pixel, the next state -- and `rest` for the
//     next launch: the best tile outside THIS cycle's lattice, from a table of every owner's best
//     three tiles (thread (b, a) of 1024 owns the tiles with (ty % 32, tx % 32) = (b, a); a patch
//     spans fewer than 32 tiles either way, so a cycle rewrites at most one tile per owner) that is
//     exact up to the cycle before last, plus the last cycle's delta of the owner, if any: three
//     known tiles always decide the best two after one of them changed;
//   * the "folder": folds the last cycle's deltas into the base arrays (plain stores nobody in this
//     launch reads back) and into the owners' table -- a rescan of the 16 (64 at 8192^2) tile maxima
//     of every owner that has one -- written to the OTHER copy of the table, for the next launch's
//     keeper.  Nobody waits for it within the launch.
// The keeper's chain (state + table, peak, candidate, record || reduction, store) and the folder's
// (deltas, rescans, stores) are each about as long as a lattice workgroup's; in round 3's first
// version one workgroup did both and was twice as long as the lattice workgroups (time stamps:
// 6.4 against 3.2 us), which made the cycle slower than before, not faster.
// Tile records carry the pixel values at the tile's peak (tile_pix), which saves the dependent
// load of the peak pixel.  Selection and arithmetic are those of the two-launch form, bit for bit.
constexpr int FUSED_MAX_BLOCKS = 256;     // workgroups of a launch incl. the two bookkeepers: one per CU
constexpr int FUSED_ROUND = 16;           // tile maxima per thread and round of a rescan
constexpr int FUSED_MAX_SLOTS = 4 * FUSED_ROUND;     // 32x32-tile groups: up to 8192^2 pixels

// A tile record rewritten by one cycle and consumed by the next, stored at the slot 32 * (lattice
// row) + (lattice column) of the workgroup that wrote it.  `tag` = 2 + the cycle that wrote it: a
// record is live for exactly the cycle after (0 = never written; the table is cleared per call).
struct __attribute__((aligned(16))) delta_t {
    int tag;
    int tile;
    float value;
    int y, x;
    float pix[4];
    int owner;              // the thread (32 * (ty % 32) + tx % 32) that owns the tile
    int pad[2];
};

struct fused_state {
    int count, done, limit;
    float threshold;
    int pad[12];
};

// The best three tiles among those a thread owns, ordered by (larger value, lower tile index);
// value -1 = none.  Three, because when ONE owned tile gets a new value the owner's best TWO are
// then known without looking at its other tiles (they are no better than the third), and two are
// what the keeper needs: the best tile outside the current cycle's lattice, which holds at most
// one tile of the owner.
struct __attribute__((aligned(16))) owner3_t {
    float v[3];
    int t[3];
    int pad[2];
};
static_assert(sizeof(owner3_t) == 32, "two 16-byte accesses");

// The best tile among those the LAST cycle did not rewrite (before the first cycle: among all), with
// its record.
struct __attribute__((aligned(16))) rest_t {
    float value;            // -1: no such tile
    int tile;
    int y, x;
    float pix[4];
};
static_assert(sizeof(rest_t) == 32, "two 16-byte accesses");

struct fused_scratch {
    fused_state st[2];
    delta_t deltas[2][1024];
    rest_t rest[2];
    owner3_t owner3[2][1024];
    // float tile_pix[tiles][4] follows
};

__device__ inline bool better_tile(float va, int ta, float vb, int tb)
{
    return va > vb || (va == vb && ta < tb);
}

// The tiles owned by thread (b, a) = (tid >> 5, tid & 31) of a 1024-thread block: (ty, tx) with
// ty % 32 == b and tx % 32 == a, visited as "slots" = 32x32-tile groups in row-major order (i.e.
// in increasing tile index).  Uniform bookkeeping, no divisions.
struct owned_tiles {
    int own, last, own_x, own_y, sup_x, tiles_x, sx, off, lim_x, lim_y, slots;

    __device__ owned_tiles(int tid, int tiles_x_, int tiles_y)
    {
        tiles_x = tiles_x_;
        sup_x = (tiles_x + 31) >> 5;
        slots = sup_x * ((tiles_y + 31) >> 5);
        last = tiles_x * tiles_y - 1;
        own_x = tid & 31;
        own_y = tid >> 5;
        own = own_y * tiles_x + own_x;
        sx = 0;
        off = 0;
        lim_x = tiles_x;
        lim_y = tiles_y;
    }

    // Tile index of the next slot (clamped into the array) and whether the slot is on the lattice
    __device__ int next(bool &valid)
    {
        const int i = min(own + off, last);
        valid = own_x < lim_x && own_y < lim_y;
        sx++;
        off += 32;
        lim_x -= 32;
        if (sx == sup_x) {
            sx = 0;
            off += 32 * tiles_x - 32 * sup_x;
            lim_x = tiles_x;
            lim_y -= 32;
        }
        return i;
    }

    // The best three owned tiles, with tile `ptile` (if >= 0) taking the value `pvalue` instead of
    // the stored one.  All of a round's loads are issued together; slots come in increasing tile
    // order, so strict comparisons keep the lowest index among equals.
    __device__ owner3_t best(const float *tile_max, int ptile, float pvalue)
    {
        owner3_t b;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            b.v[k] = -1.0f;
            b.t[k] = 0;
        }
        b.pad[0] = b.pad[1] = 0;
        for (int r = 0; r * FUSED_ROUND < slots; r++) {
            int ti[FUSED_ROUND];
            float v[FUSED_ROUND];
            bool ok[FUSED_ROUND];
#pragma unroll
            for (int k = 0; k < FUSED_ROUND; k++) {
                ti[k] = next(ok[k]);
                v[k] = tile_max[ti[k]];
            }
#pragma unroll
            for (int k = 0; k < FUSED_ROUND; k++) {
                const float val = ti[k] == ptile ? pvalue : v[k];
                if (r * FUSED_ROUND + k < slots && ok[k]) {
                    if (val > b.v[0]) {
                        b.v[2] = b.v[1];
                        b.t[2] = b.t[1];
                        b.v[1] = b.v[0];
                        b.t[1] = b.t[0];
                        b.v[0] = val;
                        b.t[0] = ti[k];
                    } else if (val > b.v[1]) {
                        b.v[2] = b.v[1];
                        b.t[2] = b.t[1];
                        b.v[1] = val;
                        b.t[1] = ti[k];
                    } else if (val > b.v[2]) {
                        b.v[2] = val;
                        b.t[2] = ti[k];
                    }
                }
            }
        }
        return b;
    }
};

// (the whole-loop kernel further down keeps an owner's best two in registers)
struct owner_best_t {
    float v1;
    int t1;
    float v2;
    int t2;
};

__device__ inline owner_best_t best_two(const owner3_t &b)
{
    owner_best_t o = {b.v[0], b.t[0], b.v[1], b.t[1]};
    return o;
}

__device__ inline void apply_delta(const delta_t &d, float *tile_max, int32_t *tile_pos,
                                   float *tile_pix)
{
    tile_max[d.tile] = d.value;
    tile_pos[2 * d.tile] = d.y;
    tile_pos[2 * d.tile + 1] = d.x;
#pragma unroll
    for (int p = 0; p < 4; p++)
        tile_pix[4 * d.tile + p] = d.pix[p];
}

#ifdef KIMG_CLEAN_STAMPS
#define STAMP(i) do { if (bid == 0 && tid == 0) stamps[i] = (int) wall_clock64(); } while (0)
#define KSTAMP(i) do { if (role == ROLE_KEEPER && tid == 0) stamps[i] = (int) wall_clock64(); } while (0)
#else
#define STAMP(i) do { } while (0)
#define KSTAMP(i) do { } while (0)
#endif

constexpr int ROLE_LATTICE = 0, ROLE_KEEPER = 1, ROLE_FOLDER = 2;

// The kernel is one chain of dependent steps executed once, so what counts is the latency of
// every step on the chain (global round trips ~0.8 us, LDS round trips and barriers ~0.1 us, and --
// with sixteen waves on a CU -- 0.75 us per hundred instructions of per-thread work), not
// throughput: reductions use DPP and one LDS exchange, state words travel as one 16-byte load.
// The cycle of one channel, executed by the workgroup that serves lattice block (blk_x, blk_y) of
// its PSF patch, or by one of the channel's two bookkeeping workgroups.  Two kernels call it: one
// channel per launch (cycle_fused_kernel) and several channels per launch (cycle_fused_batch_kernel).
template <int MODE>
__device__ __attribute__((always_inline)) inline void fused_cycle(
    float *dirty, float *model, int64_t row_stride, int64_t pol_stride, int width, int height,
    int P, const float *__restrict__ psf, int64_t psf_row_stride, int64_t psf_pol_stride,
    int psf_w, int psf_h, int patch_w, int patch_h, int border, float *tile_max,
    int32_t *tile_pos, int tiles_x, int tiles_y, float loop_gain,
    fused_scratch *scratch, int parity, float *log, int blk_x, int blk_y, int role)
{
    __shared__ key_t s_keys[16];
    __shared__ int s_pos[2];
    __shared__ float s_pix[4];
    const fused_state *cur = &scratch->st[parity];
    fused_state *next = &scratch->st[parity ^ 1];
    const delta_t *din = scratch->deltas[parity];
    delta_t *dout = scratch->deltas[parity ^ 1];
    float *tile_pix = reinterpret_cast<float *>(scratch + 1);
    const int tid = threadIdx.x;
    const bool keeper = role == ROLE_KEEPER;
#ifdef KIMG_CLEAN_STAMPS
    const int bid = (role == ROLE_LATTICE && blk_x == 0 && blk_y == 0) ? 0 : 1;   // stamps: one lattice block
    int stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    STAMP(0);
    KSTAMP(0);

    // ---- round trip 1: state, best of the rest, the records the last cycle rewrote ----------------
    const int4 st = *reinterpret_cast<const int4 *>(cur);      // count, done, limit, threshold
    const int count = st.x, done = st.y, limit = st.z;
    const float threshold = __int_as_float(st.w);
    const int lat_x = (patch_w + TILE - 1) / TILE + 1, lat_y = (patch_h + TILE - 1) / TILE + 1;
    const bool slot = (tid & 31) < lat_x && (tid >> 5) < lat_y;
    delta_t d;
    d.tag = 0;
    if (slot)
        d = din[tid];
    const rest_t rest = scratch->rest[parity];
    if (done) {
        if (keeper && tid == 0)
            *reinterpret_cast<int4 *>(next) = make_int4(count, 1, limit, st.w);
        return;
    }
    const bool live = slot && d.tag == count + 1;
    STAMP(1);
    // The bookkeeping workgroups get the last cycle's records by OWNER through LDS (a second copy of
    // the table in memory, indexed by owner, was measured: 48 KB more to fetch cold cost more than
    // the two barriers), and their owners' best three -- 32 KB, requested only now, so that the
    // state and the deltas do not queue behind them; they arrive during the reduction below.
    __shared__ delta_t s_delta[1024];
    delta_t mine;           // the last cycle's record of the one tile of THIS owner it rewrote, if any
    mine.tag = 0;
    int4 w0 = make_int4(0, 0, 0, 0), w1 = w0;
    if (role != ROLE_LATTICE) {
        s_delta[tid].tag = 0;
        lds_barrier();
        if (live)
            s_delta[d.owner] = d;
        lds_barrier();
        mine = s_delta[tid];
        const int4 *p3 = reinterpret_cast<const int4 *>(&scratch->owner3[parity][tid]);
        w0 = p3[0];
        w1 = p3[1];
    }
    const bool pending = mine.tag == count + 1;
    if (role == ROLE_FOLDER) {
        owner3_t own3;
        own3.v[0] = __int_as_float(w0.x);
        own3.v[1] = __int_as_float(w0.y);
        own3.v[2] = __int_as_float(w0.z);
        own3.t[0] = w0.w;
        own3.t[1] = w1.x;
        own3.t[2] = w1.y;
        // Nobody waits for this workgroup within the launch.  Base arrays: the records of the last
        // cycle; owners' table: the exact best three of every owner up to the last cycle, into the
        // copy the next launch's keeper reads.
        if (live)
            apply_delta(d, tile_max, tile_pos, tile_pix);
        if (pending) {
            owned_tiles walk(tid, tiles_x, tiles_y);
            own3 = walk.best(tile_max, mine.tile, mine.value);
        }
        int4 *q3 = reinterpret_cast<int4 *>(&scratch->owner3[parity ^ 1][tid]);
        q3[0] = make_int4(__float_as_int(own3.v[0]), __float_as_int(own3.v[1]),
                          __float_as_int(own3.v[2]), own3.t[0]);
        q3[1] = make_int4(own3.t[1], own3.t[2], 0, 0);
        return;
    }
    KSTAMP(1);
    const key_t mykey = live ? make_key(d.value, d.tile) : 0;
    // (a barrier that orders LDS only: the keeper's table loads are still in flight)
    key_t best = block_max_key_lds(mykey, s_keys);
    // (the rest never contains a tile that has a delta: keys of different tiles differ)
    const key_t rest_key = rest.value >= 0.0f ? make_key(rest.value, rest.tile) : 0;
    const bool from_rest = rest_key > best;
    if (from_rest)
        best = rest_key;
    STAMP(2);
    KSTAMP(2);
    const float value = __uint_as_float((unsigned) (best >> 32));
    if (best == 0 || value < threshold || count >= limit) {     // clean.py:1065-1066
        if (keeper && tid == 0)
            *reinterpret_cast<int4 *>(next) = make_int4(count, 1, limit, st.w);
        return;
    }
    if (from_rest ? tid == 0 : mykey == best) {     // exactly one thread
        s_pos[0] = from_rest ? rest.y : d.y;
        s_pos[1] = from_rest ? rest.x : d.x;
#pragma unroll
        for (int p = 0; p < 4; p++)
            s_pix[p] = from_rest ? rest.pix[p] : d.pix[p];
    }
    lds_barrier();
    const int py = s_pos[0], px = s_pos[1];
    if (value == 0.0f) {
        // a tile without any positive metric won: its record holds the (x0, y0) start position
        // of clean.py:950, whose pixel is read now, as the two-launch form does
        __syncthreads();
        if (tid < 4) {
            const bool ok = py >= 0 && py < height && px >= 0 && px < width && tid < P;
            s_pix[tid] = ok ? dirty[tid * pol_stride + (int64_t) py * row_stride + px] : 0.0f;
        }
        __syncthreads();
    }
    float scale[4];
#pragma unroll
    for (int p = 0; p < 4; p++)
        scale[p] = loop_gain * s_pix[p];                            // clean.py:1044
    STAMP(3);
    const int x0 = px - patch_w / 2, y0 = py - patch_h / 2;      // clean.py:1024-1027
    // floor division: the lattice extends into the border with negative indices
    const int bx0 = (x0 - border) >= 0 ? (x0 - border) / TILE : -((border - x0 + TILE - 1) / TILE);
    const int by0 = (y0 - border) >= 0 ? (y0 - border) / TILE : -((border - y0 + TILE - 1) / TILE);
    if (keeper) {
        owner3_t own3;
        own3.v[0] = __int_as_float(w0.x);
        own3.v[1] = __int_as_float(w0.y);
        own3.v[2] = __int_as_float(w0.z);
        own3.t[0] = w0.w;
        own3.t[1] = w1.x;
        own3.t[2] = w1.y;
        // (the model pixel is fetched now and used at the very end: its round trip must not hold
        // the first wave, and with it the reduction below, back)
        float *mp = model + (tid < P ? tid : 0) * pol_stride + (int64_t) py * row_stride + px;
        float mod = 0.0f;
        if (tid < P)
            mod = *mp;
        // The best of the rest for the NEXT launch: the best tile outside THIS cycle's lattice.  An
        // owner's candidates are its best three up to the cycle before last, with the tile the last
        // cycle rewrote (if any) at its new value, without the one tile of its own that this
        // cycle's lattice contains: whatever two of them are set aside, a tile of the three
        // remains, and no other tile of the owner is better than it.
        const int txc = bx0 + (((tid & 31) - bx0) & 31), tyc = by0 + (((tid >> 5) - by0) & 31);
        const bool in_lattice = txc < bx0 + lat_x && tyc < by0 + lat_y && txc >= 0 && txc < tiles_x
                                && tyc >= 0 && tyc < tiles_y;
        const int excluded = in_lattice ? tyc * tiles_x + txc : -1;
        const int ptile = pending ? mine.tile : -1;
        float cv = -1.0f;
        int ct = 0;
#pragma unroll
        for (int k = 0; k < 3; k++)
            if (own3.v[k] >= 0.0f && own3.t[k] != ptile && own3.t[k] != excluded
                && (cv < 0.0f || better_tile(own3.v[k], own3.t[k], cv, ct))) {
                cv = own3.v[k];
                ct = own3.t[k];
            }
        if (pending && ptile != excluded && (cv < 0.0f || better_tile(mine.value, ptile, cv, ct))) {
            cv = mine.value;
            ct = ptile;
        }
        // the candidate's record is fetched before it is known whether the candidate wins: the
        // load overlaps the block reduction instead of following it (the record of the tile the
        // last cycle rewrote is in `mine`: the folder may not have stored it yet)
        int2 cpos = make_int2(mine.y, mine.x);
        float4 cpix = make_float4(mine.pix[0], mine.pix[1], mine.pix[2], mine.pix[3]);
        if (cv >= 0.0f && ct != ptile) {
            cpos = *reinterpret_cast<const int2 *>(tile_pos + 2 * ct);
            cpix = *reinterpret_cast<const float4 *>(tile_pix + 4 * ct);
        }
        const key_t ckey = cv >= 0.0f ? make_key(cv, ct) : 0;
        KSTAMP(3);
        // (barriers that order LDS only: __syncthreads() would wait for the record loads)
        lds_barrier();                          // (s_keys is free again)
        const key_t rbest = block_max_key_lds(ckey, s_keys);
        KSTAMP(4);
        rest_t *rout = &scratch->rest[parity ^ 1];
        if (rbest == 0) {
            if (tid == 0)
                rout->value = -1.0f;
        } else if (ckey == rbest) {
            rest_t r;
            r.value = cv;
            r.tile = ct;
            r.y = cpos.x;
            r.x = cpos.y;
            r.pix[0] = cpix.x;
            r.pix[1] = cpix.y;
            r.pix[2] = cpix.z;
            r.pix[3] = cpix.w;
            *rout = r;
        }
        if (tid < P) {
            float *entry = log + (int64_t) count * (3 + P);
            if (tid == 0) {
                entry[0] = value;
                entry[1] = __int_as_float(py);
                entry[2] = __int_as_float(px);
                *reinterpret_cast<int4 *>(next) = make_int4(count + 1, 0, limit, st.w);
            }
            entry[3 + tid] = scale[tid];
            *mp = mod + scale[tid];                                 // clean.py:1047
        }
#ifdef KIMG_CLEAN_STAMPS
        if (cpix.x == 12345.678f)
            stamps[7] = 1;                      // (forces the record load to complete before the stamp)
        KSTAMP(5);
        if (tid == 0)
            for (int i = 0; i < 6; i++)
                next->pad[6 + i] = stamps[i];
#endif
        return;
    }

    // ---- round trip 2: this block's pixels -----------------------------------------------
    const int tx = bx0 + blk_x, ty = by0 + blk_y;
    const int ox = tx * TILE + border, oy = ty * TILE + border;
    const int psf_dx = psf_w / 2 - px, psf_dy = psf_h / 2 - py;
    const bool is_tile = tx >= 0 && tx < tiles_x && ty >= 0 && ty < tiles_y;
    const int x = ox + (tid & 31), y = oy + (tid >> 5);
    const bool inside = x >= 0 && x < width && y >= 0 && y < height;
    const int64_t ia = (int64_t) y * row_stride + x;
    const bool in_patch = inside && x >= x0 && x < x0 + patch_w && y >= y0 && y < y0 + patch_h;
    const bool in_tile = inside && is_tile && x < width - border && y < height - border;
    float dv[4] = {0.0f, 0.0f, 0.0f, 0.0f}, pv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (inside)
        for (int p = 0; p < P; p++)
            dv[p] = dirty[p * pol_stride + ia];
    if (in_patch)
        for (int p = 0; p < P; p++)
            pv[p] = psf[p * psf_pol_stride + (int64_t) (y + psf_dy) * psf_row_stride + (x + psf_dx)];
    float metric = 0.0f;
#ifdef KIMG_CLEAN_STAMPS
    if (dv[0] + pv[0] == 12345.678f)
        stamps[7] = 1;
    STAMP(4);
#endif
    for (int p = 0; p < P; p++) {
        if (in_patch) {
            const float tp = scale[p] * pv[p];
            dv[p] -= tp;
            dirty[p * pol_stride + ia] = dv[p];
        }
        if (MODE == KIMG_CLEAN_I) {
            if (p == 0)
                metric = fabsf(dv[0]);
        } else {
            metric += dv[p] * dv[p];
        }
    }
    if (!is_tile)
        return;
    // first strict maximum in row-major order; only positive metrics count (clean.py:953-958)
    const key_t tb = block_max_key((in_tile && metric > 0.0f) ? make_key(metric, tid) : 0, s_keys);
    const int widx = ~(int) (unsigned) tb;
#ifdef KIMG_CLEAN_STAMPS
    STAMP(5);
    if (bid == 0 && tid == 0)
        for (int i = 0; i < 6; i++)
            next->pad[i] = stamps[i];
#endif
    if (tb == 0 ? tid == 0 : tid == widx) {
        delta_t o;
        o.tag = count + 2;
        o.tile = ty * tiles_x + tx;
        if (tb == 0) {
            // no positive metric: value 0 and the (x0, y0) initial position of clean.py:950; the
            // pixel there is read when (if ever) this tile wins, see above
            o.value = 0.0f;
            o.y = ox;
            o.x = oy;
#pragma unroll
            for (int p = 0; p < 4; p++)
                o.pix[p] = 0.0f;
        } else {
            o.value = metric;
            o.y = y;
            o.x = x;
#pragma unroll
            for (int p = 0; p < 4; p++)
                o.pix[p] = dv[p];
        }
        o.owner = (ty & 31) * 32 + (tx & 31);
        o.pad[0] = o.pad[1] = 0;
        dout[blk_y * 32 + blk_x] = o;
    }
}

template <int MODE>
__global__ __launch_bounds__(1024) void cycle_fused_kernel(
    float *dirty, float *model, int64_t row_stride, int64_t pol_stride, int width, int height,
    int P, const float *__restrict__ psf, int64_t psf_row_stride, int64_t psf_pol_stride,
    int psf_w, int psf_h, int patch_w, int patch_h, int border, float *tile_max,
    int32_t *tile_pos, int tiles_x, int tiles_y, float loop_gain,
    fused_scratch *scratch, int parity, float *log)
{
    // The FIRST row of the grid (dispatched first) holds the two bookkeeping workgroups instead of
    // lattice blocks (its other members have nothing to do; a lattice is at least two blocks wide).
    int role = ROLE_LATTICE;
    if (blockIdx.y == 0) {
        if (blockIdx.x > 1)
            return;
        role = blockIdx.x == 0 ? ROLE_KEEPER : ROLE_FOLDER;
    }
    fused_cycle<MODE>(dirty, model, row_stride, pol_stride, width, height, P, psf, psf_row_stride,
                      psf_pol_stride, psf_w, psf_h, patch_w, patch_h, border, tile_max, tile_pos,
                      tiles_x, tiles_y, loop_gain, scratch, parity, log, (int) blockIdx.x,
                      (int) blockIdx.y - 1, role);
}

// ---- several channels per launch -------------------------------------------------------------
// A minor cycle is a latency chain (kernel boundary 2.1 us + ~3 us of dependent work) that
// occupies 32 of the 256 CUs.  Channels of a band are independent and have
// images of the same shape, so cycle i of up to KIMG_CLEAN_BATCH_MAX channels runs as ONE launch:
// blockIdx.z is the channel, whose pointers, patch size and (through its own state words)
// threshold, cycle limit and stop flag are its own; a finished channel's workgroups return after
// their first load.  The boundary and the cold load are paid once for all of them.  The table of
// channels travels in the kernel arguments (scalar loads with a uniform index: no extra round trip).
struct batch_channel {
    float *dirty, *model;
    const float *psf;
    float *tile_max;
    int32_t *tile_pos;
    fused_scratch *scratch;
    float *log;
    int patch_w, patch_h;
};

struct batch_table {
    batch_channel ch[KIMG_CLEAN_BATCH_MAX];
};

template <int MODE>
__global__ __launch_bounds__(1024) void cycle_fused_batch_kernel(
    batch_table tab, int64_t row_stride, int64_t pol_stride, int width, int height, int P,
    int64_t psf_row_stride, int64_t psf_pol_stride, int psf_w, int psf_h, int border, int tiles_x,
    int tiles_y, float loop_gain, int parity)
{
    // grid = (largest number of lattice blocks of any channel + 2, 1, channels): blocks 0 and 1 of a
    // channel keep its books, blocks 2 .. bx * by + 1 serve its lattice row by row; no workgroup is
    // launched only to find that it has nothing to do unless the channels' patches differ in size
    // (8 channels with the 6 x 5 lattice blocks of a 133 x 111 patch are 256 workgroups: one per CU)
    const batch_channel &ch = tab.ch[blockIdx.z];
    const int bx = (ch.patch_w + TILE - 1) / TILE + 1, by = (ch.patch_h + TILE - 1) / TILE + 1;
    const int role = blockIdx.x == 0 ? ROLE_KEEPER : blockIdx.x == 1 ? ROLE_FOLDER : ROLE_LATTICE;
    int blk_x = (int) blockIdx.x - 2, blk_y = 0;
    if (blk_x >= bx * by)
        return;
    while (blk_x >= bx) {           // (uniform; at most 31 rounds, typically < 6)
        blk_x -= bx;
        blk_y++;
    }
    fused_cycle<MODE>(ch.dirty, ch.model, row_stride, pol_stride, width, height, P, ch.psf,
                      psf_row_stride, psf_pol_stride, psf_w, psf_h, ch.patch_w, ch.patch_h, border,
                      ch.tile_max, ch.tile_pos, tiles_x, tiles_y, loop_gain, ch.scratch, parity,
                      ch.log, blk_x, blk_y, role);
}

// ---- the whole minor-cycle loop in ONE launch ----------------------------------------------
// The one-launch-per-cycle form above still pays a kernel boundary per cycle (2.1 us between
// dependent graph nodes + a cold first load).  Here the lattice workgroups of a small PSF patch
// (at most PERSIST_MAX_WGS, all resident at once, one per CU) stay alive for the whole call and
// hand their results to each other through memory:
//   * after its subtraction a workgroup drains its pixel stores (every wave s_waitcnt vmcnt(0),
//     barrier) and publishes ONE record -- the rewritten tile, or "no tile" -- as eight 8-byte
//     {word, tag} granules, each written by a single write-through (sc1) store: a reader that sees
//     all eight tags of cycle c has a consistent record and, because the pixel stores were
//     write-through and drained first, may read those pixels (MI355X_MICROARCH.md, hand-offs with
//     sc1 stores / loads and data-tagged granules);
//   * at the top of a cycle the first waves poll the records of all workgroups (sc1 loads), route
//     them through LDS to the threads that own the tiles, and every workgroup then finds the peak
//     itself exactly as in the one-launch form -- so all agree on it without further exchange;
//   * pixels are read and written with sc1 (L1-bypassing, write-through) accesses, since another
//     workgroup's CU may have written them a cycle earlier;
//   * everything a workgroup needs besides is PRIVATE to it: the tile maxima live in its LDS, the
//     owners' best-two in registers, tile positions / peak pixels in a replica of its own in the
//     scratch buffer (workgroup 0 uses the real arrays), all updated from the same records;
//   * records are double-buffered by cycle parity: a workgroup can only be one cycle ahead of the
//     slowest one.
// Selection and arithmetic are those of the other forms, bit for bit.  Every wait is bounded: a
// workgroup that does not see its peers within PERSIST_SPIN_LIMIT polls raises `error` and
// everybody leaves (kimg_clean_cycles then reports KIMG_ETIMEOUT).
//
// MEASURED (MI355X, 4096^2, 111 x 133 patch = 30 workgroups; wall_clock64 stamps of workgroup 0,
// build flag -DKIMG_CLEAN_STAMPS): 10.3-11.0 us per cycle against 6.2 for the one-launch-per-cycle
// form, so KIMG_CLEAN_FORM_AUTO does NOT take this form; it stays selectable and tested.  Per cycle:
// 2.6-3.0 us until the records of all peers are visible (an agent-scope hand-off is a fabric write
// plus a fabric read, and the poll itself is a 1-us round trip), 0.4 routing / applying, 2.4-2.9
// until the peak is known (the waves whose owners got a record rescan their tiles and fetch a
// record before they reach the reduction's barrier -- work the one-launch form leaves to its
// keeper workgroup, off the critical path), 1.05 for the block's pixels (sc1 loads are served from
// the memory side), 1.0 subtract + rescan + stores, 0.1 drain.  Moving the rescan behind the
// publication would bring it to about 5.7 us: the hand-off alone costs what the kernel boundary
// cost, which is why none of the in-launch exchanges tried so far beat one launch per cycle.
constexpr int PERSIST_MAX_WGS = 64;
constexpr int PERSIST_SPIN_LIMIT = 1 << 21;
constexpr size_t PERSIST_LDS_LIMIT = 160 * 1024 - 2048;

struct persist_header {
    unsigned long long records[2][PERSIST_MAX_WGS][8];      // {word | tag << 32}
    int error;
    int pad[15];
    // replicas follow: [wgs - 1][tiles] x { int y, x; float pix[4] }
};

struct replica_t {
    int y, x;
    float pix[4];
};

__device__ inline float load_sc1(const float *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ inline void store_sc1(float *p, float v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int MODE>
__global__ __launch_bounds__(1024) void cycle_persistent_kernel(
    float *dirty, float *model, int64_t row_stride, int64_t pol_stride, int width, int height,
    int P, const float *__restrict__ psf, int64_t psf_row_stride, int64_t psf_pol_stride,
    int psf_w, int psf_h, int patch_w, int patch_h, int border, float *tile_max,
    int32_t *tile_pos, int tiles_x, int tiles_y, float loop_gain, float threshold, int limit,
    fused_scratch *scratch, persist_header *hdr, replica_t *replicas, float *log)
{
    extern __shared__ __align__(16) unsigned char persist_smem[];
    __shared__ key_t s_keys[16];
    __shared__ int s_pos[2];
    __shared__ float s_pix[4];
    __shared__ int s_abort;
    const int tiles = tiles_x * tiles_y;
    float *s_tile_max = reinterpret_cast<float *>(persist_smem);
    delta_t *s_delta = reinterpret_cast<delta_t *>(persist_smem + (((size_t) tiles * 4 + 15) & ~(size_t) 15));
    float *tile_pix = reinterpret_cast<float *>(scratch + 1);
    const int tid = threadIdx.x;
    const int nwgs = gridDim.x * gridDim.y;
    const int wg = blockIdx.y * gridDim.x + blockIdx.x;
    // this workgroup's own copy of (tile position, peak pixels): workgroup 0 keeps the real arrays
    replica_t *mine = wg ? replicas + (size_t) (wg - 1) * tiles : nullptr;

    for (int i = tid; i < tiles; i += 1024) {
        s_tile_max[i] = tile_max[i];
        if (wg) {
            replica_t r;
            r.y = tile_pos[2 * i];
            r.x = tile_pos[2 * i + 1];
#pragma unroll
            for (int p = 0; p < 4; p++)
                r.pix[p] = tile_pix[4 * i + p];
            mine[i] = r;
        }
    }
    s_delta[tid].tag = 0;
    if (tid == 0)
        s_abort = 0;
    __syncthreads();
    owner_best_t ob;
    {
        owned_tiles walk(tid, tiles_x, tiles_y);
        ob = best_two(walk.best(s_tile_max, -1, 0.0f));
    }
    // (position, peak pixels) of this owner's best tile, kept in registers: the peak search then
    // needs no memory access at all
    auto load_record = [&](int t) {
        replica_t r;
        if (wg) {
            r = mine[t];
        } else {
            r.y = tile_pos[2 * t];
            r.x = tile_pos[2 * t + 1];
#pragma unroll
            for (int p = 0; p < 4; p++)
                r.pix[p] = tile_pix[4 * t + p];
        }
        return r;
    };
    replica_t best_rec = {0, 0, {0.0f, 0.0f, 0.0f, 0.0f}};
    if (ob.v1 >= 0.0f)
        best_rec = load_record(ob.t1);

#ifdef KIMG_CLEAN_STAMPS
    long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt = wall_clock64();
#define PSTAMP(i) do { if (wg == 0 && tid == 0) { const long long n_ = wall_clock64(); pacc[i] += n_ - pt; pt = n_; } } while (0)
#else
#define PSTAMP(i) do { } while (0)
#endif
    int count = 0;
    for (;;) {
        PSTAMP(0);
        // ---- the records of the previous cycle ------------------------------------------------
        delta_t d;
        d.tag = 0;
        if (count > 0) {
            if (tid < nwgs) {
                const unsigned long long *rec = hdr->records[(count - 1) & 1][tid];
                const unsigned tag = (unsigned) count + 1;
                unsigned long long g[8];
                int spins = 0;
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        g[k] = __hip_atomic_load(rec + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok &= (unsigned) (g[k] >> 32) == tag;
                    }
                    if (ok)
                        break;
                    if (++spins > PERSIST_SPIN_LIMIT
                        || __hip_atomic_load(&hdr->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                        __hip_atomic_store(&hdr->error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        s_abort = 1;
                        break;
                    }
                }
                const int tile = (int) (unsigned) g[0];
                if (!s_abort && tile >= 0) {
                    // route the record to the thread that owns the tile
                    delta_t o;
                    o.tag = (int) tag;
                    o.tile = tile;
                    o.value = __uint_as_float((unsigned) g[1]);
                    o.y = (int) (unsigned) g[2];
                    o.x = (int) (unsigned) g[3];
#pragma unroll
                    for (int p = 0; p < 4; p++)
                        o.pix[p] = __uint_as_float((unsigned) g[4 + p]);
                    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
                    s_delta[(ty & 31) * 32 + (tx & 31)] = o;
                }
            }
            PSTAMP(1);
            __syncthreads();
            if (s_abort)
                return;
            d = s_delta[tid];
            if (d.tag == count + 1) {
                s_tile_max[d.tile] = d.value;           // (only ever read by this thread: it owns the tile)
                if (wg) {
                    replica_t r;
                    r.y = d.y;
                    r.x = d.x;
#pragma unroll
                    for (int p = 0; p < 4; p++)
                        r.pix[p] = d.pix[p];
                    mine[d.tile] = r;
                } else {
                    tile_pos[2 * d.tile] = d.y;
                    tile_pos[2 * d.tile + 1] = d.x;
#pragma unroll
                    for (int p = 0; p < 4; p++)
                        tile_pix[4 * d.tile + p] = d.pix[p];
                }
                owned_tiles walk(tid, tiles_x, tiles_y);
                ob = best_two(walk.best(s_tile_max, -1, 0.0f));
                if (ob.v1 >= 0.0f) {
                    if (ob.t1 == d.tile) {
                        best_rec.y = d.y;
                        best_rec.x = d.x;
#pragma unroll
                        for (int p = 0; p < 4; p++)
                            best_rec.pix[p] = d.pix[p];
                    } else {
                        best_rec = load_record(ob.t1);
                    }
                }
            }
        }
        if (count >= limit)
            break;
        PSTAMP(2);

        // ---- the peak: every workgroup for itself ----------------------------------------------
        const float bv = ob.v1;
        const int bi = ob.t1;
        const int2 cpos = make_int2(best_rec.y, best_rec.x);
        const float4 cpix = make_float4(best_rec.pix[0], best_rec.pix[1], best_rec.pix[2], best_rec.pix[3]);
        const key_t mykey = bv < 0.0f ? 0 : make_key(bv, bi);
        const key_t best = block_max_key_lds(mykey, s_keys);
        const float value = __uint_as_float((unsigned) (best >> 32));
        PSTAMP(3);
        if (best == 0 || value < threshold)                         // clean.py:1065-1066
            break;
        if (mykey == best) {        // exactly one thread: every tile has one owner
            s_pos[0] = cpos.x;
            s_pos[1] = cpos.y;
            s_pix[0] = cpix.x;
            s_pix[1] = cpix.y;
            s_pix[2] = cpix.z;
            s_pix[3] = cpix.w;
        }
        lds_barrier();
        const int py = s_pos[0], px = s_pos[1];
        if (value == 0.0f) {
            // a tile without any positive metric won: its record holds the (x0, y0) start position
            // of clean.py:950, whose pixel is read now, as the other forms do
            __syncthreads();
            if (tid < 4) {
                const bool ok = py >= 0 && py < height && px >= 0 && px < width && tid < P;
                s_pix[tid] = ok ? load_sc1(dirty + tid * pol_stride + (int64_t) py * row_stride + px) : 0.0f;
            }
            __syncthreads();
        }
        float scale[4];
#pragma unroll
        for (int p = 0; p < 4; p++)
            scale[p] = loop_gain * s_pix[p];                            // clean.py:1044
        if (wg == 0 && tid < P) {
            float *entry = log + (int64_t) count * (3 + P);
            float *mp = model + tid * pol_stride + (int64_t) py * row_stride + px;
            if (tid == 0) {
                entry[0] = value;
                entry[1] = __int_as_float(py);
                entry[2] = __int_as_float(px);
            }
            entry[3 + tid] = scale[tid];
            *mp = *mp + scale[tid];                                     // clean.py:1047
        }

        // ---- this workgroup's lattice block ----------------------------------------------------
        const int x0 = px - patch_w / 2, y0 = py - patch_h / 2;      // clean.py:1024-1027
        const int bx0 = (x0 - border) >= 0 ? (x0 - border) / TILE : -((border - x0 + TILE - 1) / TILE);
        const int by0 = (y0 - border) >= 0 ? (y0 - border) / TILE : -((border - y0 + TILE - 1) / TILE);
        const int tx = bx0 + (int) blockIdx.x, ty = by0 + (int) blockIdx.y;
        const int ox = tx * TILE + border, oy = ty * TILE + border;
        const int psf_dx = psf_w / 2 - px, psf_dy = psf_h / 2 - py;
        const bool is_tile = tx >= 0 && tx < tiles_x && ty >= 0 && ty < tiles_y;
        const int x = ox + (tid & 31), y = oy + (tid >> 5);
        const bool inside = x >= 0 && x < width && y >= 0 && y < height;
        const int64_t ia = (int64_t) y * row_stride + x;
        const bool in_patch = inside && x >= x0 && x < x0 + patch_w && y >= y0 && y < y0 + patch_h;
        const bool in_tile = inside && is_tile && x < width - border && y < height - border;
        float dv[4] = {0.0f, 0.0f, 0.0f, 0.0f}, pv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (inside)
            for (int p = 0; p < P; p++)
                dv[p] = load_sc1(dirty + p * pol_stride + ia);
        if (in_patch)
            for (int p = 0; p < P; p++)
                pv[p] = psf[p * psf_pol_stride + (int64_t) (y + psf_dy) * psf_row_stride + (x + psf_dx)];
        float metric = 0.0f;
#ifdef KIMG_CLEAN_STAMPS
        if (dv[0] + pv[0] == 12345.678f)
            pacc[7] = 1;
        PSTAMP(4);
#endif
        for (int p = 0; p < P; p++) {
            if (in_patch) {
                const float tp = scale[p] * pv[p];
                dv[p] -= tp;
                store_sc1(dirty + p * pol_stride + ia, dv[p]);
            }
            if (MODE == KIMG_CLEAN_I) {
                if (p == 0)
                    metric = fabsf(dv[0]);
            } else {
                metric += dv[p] * dv[p];
            }
        }
        // first strict maximum in row-major order; only positive metrics count (clean.py:953-958)
        // (the barrier inside also orders s_pos / s_pix against the next cycle's writes)
        const key_t tb = block_max_key((in_tile && metric > 0.0f) ? make_key(metric, tid) : 0, s_keys);
        const int widx = ~(int) (unsigned) tb;
        PSTAMP(5);
        // every wave's pixel stores have left the CU before the record says so
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        PSTAMP(6);
        if (tb == 0 ? tid == 0 : tid == widx) {
            unsigned w[8];
            w[0] = (unsigned) (is_tile ? ty * tiles_x + tx : -1);
            if (tb == 0) {
                // no positive metric: value 0 and the (x0, y0) initial position of clean.py:950
                w[1] = __float_as_uint(0.0f);
                w[2] = (unsigned) ox;
                w[3] = (unsigned) oy;
                w[4] = w[5] = w[6] = w[7] = 0u;
            } else {
                w[1] = __float_as_uint(metric);
                w[2] = (unsigned) y;
                w[3] = (unsigned) x;
#pragma unroll
                for (int p = 0; p < 4; p++)
                    w[4 + p] = __float_as_uint(dv[p]);
            }
            unsigned long long *rec = hdr->records[count & 1][wg];
            const unsigned long long tag = (unsigned long long) (unsigned) (count + 2) << 32;
#pragma unroll
            for (int k = 0; k < 8; k++)
                __hip_atomic_store(rec + k, tag | w[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        count++;
    }
    __syncthreads();        // (the owners' last updates of the LDS tile maxima, read by other threads below)
    if (wg == 0) {
        // the tile maxima go back to the caller's array; positions / peak pixels are already there
        for (int i = tid; i < tiles; i += 1024)
            tile_max[i] = s_tile_max[i];
        if (tid == 0) {
            scratch->st[0].count = count;
            scratch->st[0].done = 1;
#ifdef KIMG_CLEAN_STAMPS
            for (int i = 0; i < 8; i++)
                scratch->st[0].pad[i] = (int) (pacc[i] / (count > 0 ? count : 1));
#endif
        }
    }
}

// state[1] := 2 when the persistent loop gave up (see kimg.h)
__global__ void persist_status_kernel(fused_scratch *scratch, const persist_header *hdr)
{
    if (hdr->error)
        scratch->st[0].done = 2;
}

// ---- the whole minor-cycle loop in ONE WORKGROUP ---------------------------------------------
// For a small PSF patch a cycle moves little data (a 111 x 133 patch rewrites 59 KB and its 30
// lattice tiles hold 123 KB) and is otherwise a chain of latencies.  The forms above pay a kernel
// boundary (one launch per cycle) or a hand-off between workgroups (persistent form) per cycle;
// one workgroup on one CU pays neither: everything it needs between cycles stays in its LDS, and
// the only synchronisation is its own barrier.  What it pays instead is instruction issue: a wave
// issues about one instruction per 5 clocks, so the ~31 K pixels of a cycle must cost few
// instructions per wave.
//   * LDS holds, per tile, the metric maximum (float) and a 16-bit record: the peak's index inside
//     the tile, a flag for "no positive metric" (clean.py:950) and -- CLEAN_I -- the sign of the
//     pixel there, so that the peak pixel is known without a load; and the PSF patch, its rows
//     padded with zeros so that a pixel next to the patch subtracts exactly nothing;
//   * thread (b, a) owns the tiles (ty % 32, tx % 32) = (b, a) and keeps their best two keys in
//     registers; the global peak is one block reduction of registers; the owner of a rewritten
//     tile gets its new best from the new record and those two, and looks at its (at most 16) LDS
//     values again only later, while the next cycle's loads are in flight;
//   * a wave takes whole rows of the patch's tile lattice, four pixels per lane (lane = tile
//     column x 8 + group of four): one load, one store and one LDS read per row and lane, and
//     everything that depends on the row (inside the image / the patch / a tile) is wave-uniform
//     control flow.  Groups that straddle the image's edge go pixel by pixel;
//   * all loads of a wave's rows are issued first, then one wait, then the arithmetic with its
//     stores: loads and stores return out of order with respect to each other, so a wave that
//     waits for a load while it has stores in flight waits for the stores, too.  The stores are
//     only waited for (vmcnt(0) + barrier) after the NEXT peak search.
// Selection and arithmetic are those of the other forms, bit for bit.
//
// MEASURED (MI355X, 4096^2, 111 x 133 patch; wall_clock64 stamps, build flag -DKIMG_CLEAN_STAMPS,
// -DKIMG_SOLO_STAMP_TID=<thread>): 7.0 us per cycle against 6.3 for the one-launch-per-cycle form,
// so KIMG_CLEAN_FORM_AUTO does NOT take this form; it stays selectable and tested.  Per cycle: peak
// search 0.7 (two DPP/LDS reduction stages, the second behind the owners' update), store drain
// 0.35, then per wave: 1.6 (first wave) to 3.1 us (ninth) until its twelve 640-byte row loads have
// landed -- ONE CU pulls the ~150 KB of a cycle's lattice tiles out of L2 / MALL at only ~50 GB/s --
// 1.25 for the patch rows (LDS PSF reads + 12 stores), 1.2 for the tile maxima (3 VALU
// instructions per pixel: four waves per SIMD share one VALU, ~31 K pixels are ~2300 VALU clocks
// per SIMD), and up to 3 us of waiting for the slowest wave (the rows of a wave are not equally
// expensive).  Earlier layouts of the same idea: one pixel per lane with per-row predicates 16.1
// us; units of 8 rows x 64 pixels double-buffered 9.3 (every wait for a load behind a store was a
// wait for the store); the same with all loads first 7.8.  What would still help: loading only
// the patch's pixels of tiles whose old maximum lies outside the patch (halves the bytes), rows
// dealt out by cost.  Neither brings one CU below ~4.5 us: the form trades two microseconds of
// kernel boundary for one CU's memory pipe and VALU, which is not a good trade at this patch size.
constexpr size_t SOLO_LDS_LIMIT = 160 * 1024 - 1024;
constexpr int SOLO_MAX_BX = 8;              // lattice columns of a patch: 8 lanes each
constexpr int SOLO_MAX_BLOCKS = 64;         // lattice blocks of a patch
constexpr int SOLO_AUTO_BLOCKS = 0;         // KIMG_CLEAN_FORM_AUTO takes this form up to here: never (see MEASURED)
constexpr int SOLO_ROWS = 12;               // rows of a wave whose loads are issued together
constexpr int SOLO_PAD = 4;                 // zeros either side of a PSF row in LDS
constexpr unsigned SOLO_NONE = 0x4000, SOLO_SIGN = 0x8000;

// tile-local key: metric, then lower index inside the tile; bit 0 carries the pixel's sign
__device__ inline key_t solo_pixel_key(float metric, int idx, bool negative)
{
    return ((key_t) __float_as_uint(metric) << 32) | (unsigned) (((1023 - idx) << 1) | (negative ? 1 : 0));
}

// global key: metric, then lower (ty, tx) in row-major order (no division to get them back)
__device__ inline key_t solo_tile_key(float value, int ty, int tx)
{
    return ((key_t) __float_as_uint(value) << 32) | (unsigned) ~((ty << 16) | tx);
}

struct solo_f4 { float v[4]; };             // four pixels, 4-byte aligned

template <int MODE>
__global__ __launch_bounds__(1024) void cycle_solo_kernel(
    float *dirty, float *model, int64_t row_stride, int width, int height,
    const float *__restrict__ psf, int64_t psf_row_stride, int psf_w, int psf_h, int patch_w,
    int patch_h, int border, float *tile_max, int32_t *tile_pos, int tiles_x, int tiles_y,
    float loop_gain, float threshold, int limit, fused_scratch *scratch, float *log)
{
    constexpr bool FAST = MODE == KIMG_CLEAN_I;     // the peak pixel is +-metric
    extern __shared__ __align__(16) unsigned char solo_smem[];
    __shared__ key_t s_keys[16];
    __shared__ key_t s_tkey[SOLO_MAX_BLOCKS];       // per lattice block: best pixel key of the cycle
    const int tiles = tiles_x * tiles_y;
    float *s_val = reinterpret_cast<float *>(solo_smem);
    unsigned short *s_rec = reinterpret_cast<unsigned short *>(solo_smem + (size_t) tiles * 4);
    float *s_psf = reinterpret_cast<float *>(solo_smem + (((size_t) tiles * 6 + 15) & ~(size_t) 15));
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int own_x = tid & 31, own_y = tid >> 5;
    const int ppx = psf_w / 2 - patch_w / 2, ppy = psf_h / 2 - patch_h / 2;   // patch origin in the PSF
    const int pstride = patch_w + 2 * SOLO_PAD;

    for (int t = tid; t < tiles; t += 1024) {
        const float v = tile_max[t];
        const int2 pos = *reinterpret_cast<const int2 *>(tile_pos + 2 * t);
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        unsigned rec = SOLO_NONE;
        if (v != 0.0f) {
            rec = (unsigned) ((pos.x - (ty * TILE + border)) * TILE + (pos.y - (tx * TILE + border)));
            if (FAST && dirty[(int64_t) pos.x * row_stride + pos.y] < 0.0f)
                rec |= SOLO_SIGN;
        }
        s_val[t] = v;
        s_rec[t] = (unsigned short) rec;
    }
    for (int i = tid; i < pstride * patch_h; i += 1024) {
        const int r = i / pstride, c = i - r * pstride - SOLO_PAD;
        s_psf[i] = (c >= 0 && c < patch_w) ? psf[(int64_t) (ppy + r) * psf_row_stride + (ppx + c)] : 0.0f;
    }
    if (tid < SOLO_MAX_BLOCKS)
        s_tkey[tid] = 0;
    __syncthreads();

    // The best two of this thread's tiles (keys; 0 = none), all (up to 16) LDS reads in flight.
    // Tiles are visited in increasing key order of equal values, so plain key comparisons do.
    key_t b1 = 0, b2 = 0;
    const bool few_tiles = tiles_x <= 128 && tiles_y <= 128;
    auto consider = [&](key_t k) {
        const key_t lo = k > b1 ? b1 : k;
        b1 = k > b1 ? k : b1;
        b2 = lo > b2 ? lo : b2;
    };
    auto rescan = [&]() {
        b1 = b2 = 0;
        if (few_tiles) {
#pragma unroll 1
            for (int h = 0; h < 4; h++) {
                const int ty = own_y + 32 * h;
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int tx = own_x + 32 * i;
                    v[i] = s_val[(ty < tiles_y && tx < tiles_x) ? ty * tiles_x + tx : 0];
                }
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int tx = own_x + 32 * i;
                    consider((ty < tiles_y && tx < tiles_x) ? solo_tile_key(v[i], ty, tx) : 0);
                }
            }
        } else {
            for (int ty = own_y; ty < tiles_y; ty += 32)
                for (int tx = own_x; tx < tiles_x; tx += 32)
                    consider(solo_tile_key(s_val[ty * tiles_x + tx], ty, tx));
        }
    };
    rescan();
    bool stale = false;             // b1 / b2 have to be read again

    const int nbx = (patch_w + TILE - 1) / TILE + 1, nby = (patch_h + TILE - 1) / TILE + 1;
    // Work split: the lattice has nby * 32 rows of nbx * 32 pixels; the waves share the rows in
    // consecutive runs.  Lane = (lattice column) * 8 + (group of four pixels).
    const int lattice_rows = nby * TILE;
    const int rows_per_wave = (lattice_rows + 15) >> 4;
    const int row_begin = __builtin_amdgcn_readfirstlane(min(wave * rows_per_wave, lattice_rows));
    const int row_end = __builtin_amdgcn_readfirstlane(min(row_begin + rows_per_wave, lattice_rows));
    const int jx = (tid & 63) >> 3, grp = tid & 7;
    const unsigned row_bytes = (unsigned) row_stride * 4u;

#ifdef KIMG_CLEAN_STAMPS
    long long sacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t = wall_clock64();
#ifndef KIMG_SOLO_STAMP_TID
#define KIMG_SOLO_STAMP_TID 0
#endif
#define SSTAMP(i) do { if (tid == KIMG_SOLO_STAMP_TID) { const long long n_ = wall_clock64(); sacc[i] += n_ - st_t; st_t = n_; } } while (0)
#else
#define SSTAMP(i) do { } while (0)
#endif
    int count = 0;
    for (;;) {
        SSTAMP(0);
        const key_t best = block_max_key_lds(b1, s_keys);
        SSTAMP(1);
        const float value = __uint_as_float((unsigned) (best >> 32));
        if (best == 0 || value < threshold || count >= limit)      // clean.py:1065-1066
            break;
        const unsigned tcode = ~(unsigned) best;
        const int pty = (int) (tcode >> 16), ptx = (int) (tcode & 0xffff);
        const unsigned rec = s_rec[pty * tiles_x + ptx];
        int py, px;
        if (rec & SOLO_NONE) {              // clean.py:950 best_pos = (x0, y0)
            py = ptx * TILE + border;
            px = pty * TILE + border;
        } else {
            py = pty * TILE + border + (int) ((rec & 1023) >> 5);
            px = ptx * TILE + border + (int) (rec & 31);
        }
        // every pixel store of the previous cycle must have landed before this cycle's loads
        __syncthreads();
        SSTAMP(2);
        const bool pos_ok = py >= 0 && py < height && px >= 0 && px < width;
        float pix;
        if (FAST && !(rec & SOLO_NONE))
            pix = (rec & SOLO_SIGN) ? -value : value;
        else
            pix = pos_ok ? dirty[(int64_t) py * row_stride + px] : 0.0f;
        const float scale = loop_gain * pix;                        // clean.py:1044
        float mod = 0.0f;
        float *mp = model + (int64_t) py * row_stride + px;
        if (tid == 0 && pos_ok)
            mod = *mp;

        const int x0 = px - patch_w / 2, y0 = py - patch_h / 2;    // clean.py:1024-1027
        // floor division: the lattice extends into the border with negative indices
        const int bx0 = (x0 - border) >= 0 ? (x0 - border) / TILE : -((border - x0 + TILE - 1) / TILE);
        const int by0 = (y0 - border) >= 0 ? (y0 - border) / TILE : -((border - y0 + TILE - 1) / TILE);

        // this lane's four pixels: the same for all rows.  "wide" lanes move them as one 16-byte
        // access (lanes with nothing to do read a clamped address and never write); lanes whose
        // group straddles the image's edge ("part") or the edge of the tile lattice (`t_part`)
        // are served pixel by pixel in passes of their own that a wave without any skips.
        const int tx = bx0 + jx;
        const int x = tx * TILE + border + 4 * grp;
        const bool lane_on = jx < nbx;
        const bool part = lane_on && !(x >= 0 && x + 3 < width) && x + 3 >= 0 && x < width;
        const bool wide_patch = lane_on && x >= 0 && x + 3 < width && x + 3 >= x0 && x < x0 + patch_w;
        const bool tx_ok = lane_on && tx >= 0 && tx < tiles_x;
        const bool t_all = tx_ok && x + 3 < width - border;
        const bool t_part = tx_ok && !t_all && x < width - border;
        const unsigned xoff = (unsigned) min(max(x, 0), width - 4) * 4u;
        const int pidx0 = SOLO_PAD + (x - x0);
        const bool any_part = __any(part), any_t_part = __any(t_part);
        char *base = reinterpret_cast<char *>(dirty);
        const int lattice_y = by0 * TILE + border;

        // best of this lane's pixels in the (at most two) tile rows of its chunk: pixel (CLEAN_I)
        // or metric, and where (row in chunk * 4 + pixel)
        auto track = [&](float d, int code, float &bs, int &bc) {
            // first strict maximum in row-major order; only positive metrics count
            // (clean.py:953-958): rows come in increasing order, the comparison is strict
            if (FAST) {
                if (fabsf(d) > fabsf(bs)) {
                    bs = d;
                    bc = code;
                }
            } else {
                const float m = 0.0f + d * d;
                if (m > bs) {
                    bs = m;
                    bc = code;
                }
            }
        };
        auto bits = [](int lo, int hi) {        // bits [lo, hi) of a SOLO_ROWS-bit mask
            lo = min(max(lo, 0), SOLO_ROWS);
            hi = min(max(hi, lo), SOLO_ROWS);
            return ((1u << hi) - 1u) & ~((1u << lo) - 1u);
        };
        for (int rb = row_begin; rb < row_end; rb += SOLO_ROWS) {
            const int nr = min(SOLO_ROWS, row_end - rb);
            const int yb = lattice_y + rb;              // image row of the chunk's first row
            // rows of the chunk (bit r = row r): inside the image; patch rows; rows that count for
            // the tiles of the chunk's first / second lattice tile row
            const unsigned m_img = bits(-yb, min(height - yb, nr));
            const unsigned m_patch = m_img & bits(y0 - yb, y0 + patch_h - yb);
            const int trow_a = rb >> 5, split = ((trow_a + 1) << 5) - rb;   // rows [split, ..) are tile row b
            const unsigned m_tile = m_img & bits(0, height - border - yb);
            const bool ta_ok = by0 + trow_a >= 0 && by0 + trow_a < tiles_y;
            const bool tb_ok = by0 + trow_a + 1 >= 0 && by0 + trow_a + 1 < tiles_y;
            const unsigned m_a = ta_ok ? m_tile & bits(0, split) : 0u;
            const unsigned m_b = tb_ok ? m_tile & bits(split, SOLO_ROWS) : 0u;
            float dv[SOLO_ROWS][4];
            // ---- loads: unconditional rows (clamped), nothing but loads in between
#pragma unroll
            for (int r = 0; r < SOLO_ROWS; r++) {
                const int y = min(max(yb + min(r, nr - 1), 0), height - 1);     // uniform
                solo_f4 v;
                __builtin_memcpy(&v, base + ((unsigned) y * row_bytes + xoff), sizeof(v));
#pragma unroll
                for (int i = 0; i < 4; i++)
                    dv[r][i] = v.v[i];
            }
            if (any_part) {
#pragma unroll
                for (int r = 0; r < SOLO_ROWS; r++) {
                    const int y = min(max(yb + min(r, nr - 1), 0), height - 1);
                    const char *row = base + (size_t) ((unsigned) y * row_bytes);
                    if (part) {
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            dv[r][i] = 0.0f;
                            if (x + i >= 0 && x + i < width)
                                dv[r][i] = *reinterpret_cast<const float *>(row + (size_t) ((unsigned) (x + i) * 4u));
                        }
                    }
                }
            }
            if (stale) {
                // in the shadow of the loads: this owner's best two, with last cycle's record in
                rescan();
                stale = false;
            }
            // (the builtin, not inline assembly: the compiler's own wait-count bookkeeping must know
            // that no load is pending any more, or it counts the stores against them)
            __builtin_amdgcn_s_waitcnt(0x0F70);         // vmcnt(0)
            SSTAMP(5);
            // ---- the patch rows: dirty -= (loop_gain * pixel) * psf, PSF rows fetched four at a time
            if (m_patch) {
#pragma unroll
                for (int h = 0; h < SOLO_ROWS; h += 4) {
                    solo_f4 pv[4];
                    if (wide_patch || part) {
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            // (rows outside the patch read a clamped row; they are not used)
                            const int pr = min(max(yb + h + r - y0, 0), patch_h - 1);
                            __builtin_memcpy(&pv[r], s_psf + pr * pstride + min(max(pidx0, 0), pstride - 4),
                                             sizeof(solo_f4));
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        if (!(m_patch & (1u << (h + r))))
                            continue;
                        const unsigned row_off = (unsigned) (yb + h + r) * row_bytes;
                        char *row = base + (size_t) row_off;
                        if (wide_patch) {
                            solo_f4 v;
#pragma unroll
                            for (int i = 0; i < 4; i++) {
                                const float t = scale * pv[r].v[i];
                                dv[h + r][i] -= t;
                                v.v[i] = dv[h + r][i];
                            }
                            __builtin_memcpy(base + (row_off + xoff), &v, sizeof(v));
                        }
                        if (any_part) {
                            if (part) {
#pragma unroll
                                for (int i = 0; i < 4; i++)
                                    if (x + i >= 0 && x + i < width && x + i >= x0 && x + i < x0 + patch_w) {
                                        const float t = scale * pv[r].v[i];
                                        dv[h + r][i] -= t;
                                        *reinterpret_cast<float *>(row + (size_t) ((unsigned) (x + i) * 4u)) = dv[h + r][i];
                                    }
                            }
                        }
                    }
                }
            }
            SSTAMP(6);
            // ---- the tiles' maxima
            float bs_a = 0.0f, bs_b = 0.0f;
            int bc_a = 0, bc_b = 0;
#pragma unroll
            for (int r = 0; r < SOLO_ROWS; r++) {
                if (m_a & (1u << r)) {
                    if (t_all) {
#pragma unroll
                        for (int i = 0; i < 4; i++)
                            track(dv[r][i], r * 4 + i, bs_a, bc_a);
                    }
                    if (any_t_part) {
                        if (t_part) {
#pragma unroll
                            for (int i = 0; i < 4; i++)
                                if (x + i < width - border)
                                    track(dv[r][i], r * 4 + i, bs_a, bc_a);
                        }
                    }
                }
                if (m_b & (1u << r)) {
                    if (t_all) {
#pragma unroll
                        for (int i = 0; i < 4; i++)
                            track(dv[r][i], r * 4 + i, bs_b, bc_b);
                    }
                    if (any_t_part) {
                        if (t_part) {
#pragma unroll
                            for (int i = 0; i < 4; i++)
                                if (x + i < width - border)
                                    track(dv[r][i], r * 4 + i, bs_b, bc_b);
                        }
                    }
                }
            }
            // ---- the tiles' best keys of this chunk: 8 lanes per tile, then one LDS atomic
            auto flush = [&](float bs, int bc, int trow) {
                const float bm = FAST ? fabsf(bs) : bs;
                const int idx = ((rb + (bc >> 2)) & 31) * TILE + 4 * grp + (bc & 3);
                key_t k = bm > 0.0f ? solo_pixel_key(bm, idx, FAST && bs < 0.0f) : 0;
                k = key_max(k, key_dpp<0xB1>(k));       // quad_perm [1,0,3,2]
                k = key_max(k, key_dpp<0x4E>(k));       // quad_perm [2,3,0,1]
                k = key_max(k, key_dpp<0x141>(k));      // row_half_mirror
                if (grp == 0 && k != 0 && lane_on)
                    atomicMax(&s_tkey[trow * nbx + jx], k);
            };
            if (m_a)
                flush(bs_a, bc_a, trow_a);
            if (m_b)
                flush(bs_b, bc_b, trow_a + 1);
        }
        SSTAMP(3);
        if (tid == 0) {
            *reinterpret_cast<float4 *>(log + (int64_t) count * 4) =
                make_float4(value, __int_as_float(py), __int_as_float(px), scale);
            if (pos_ok)
                *mp = mod + scale;                                  // clean.py:1047
        }
        count++;
        lds_barrier();          // the chunks' keys are in LDS
        SSTAMP(4);
        // The owners of the rewritten tiles (a patch spans fewer than 32 tiles either way: at most
        // one tile per owner) take the new record; their new best is the better of it and of the
        // best of their other tiles, which is the first of (b1, b2) that is not this tile.
        const int ojx = (own_x - bx0) & 31, ojy = (own_y - by0) & 31;
        if (ojx < nbx && ojy < nby) {
            const int otx = bx0 + ojx, oty = by0 + ojy;
            if (otx >= 0 && otx < tiles_x && oty >= 0 && oty < tiles_y) {
                const int t = oty * tiles_x + otx;
                const key_t kt = s_tkey[ojy * nbx + ojx];
                s_tkey[ojy * nbx + ojx] = 0;
                float nv = 0.0f;
                unsigned nrec = SOLO_NONE;
                if (kt != 0) {
                    const unsigned lo = (unsigned) kt;
                    nv = __uint_as_float((unsigned) (kt >> 32));
                    nrec = (1023 - ((lo >> 1) & 1023)) | ((lo & 1) ? SOLO_SIGN : 0);
                }
                s_val[t] = nv;
                s_rec[t] = (unsigned short) nrec;
                const unsigned code = ~(unsigned) ((oty << 16) | otx);
                const key_t others = (unsigned) b1 == code ? b2 : b1;
                b1 = key_max(others, solo_tile_key(nv, oty, otx));
                stale = true;
            }
        }
    }

    __syncthreads();
    for (int t = tid; t < tiles; t += 1024) {
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        const int ox = tx * TILE + border, oy = ty * TILE + border;
        const unsigned rec = s_rec[t];
        tile_max[t] = s_val[t];
        int2 pos;
        if (rec & SOLO_NONE)
            pos = make_int2(ox, oy);        // clean.py:950 (x0, y0)
        else
            pos = make_int2(oy + (int) ((rec & 1023) >> 5), ox + (int) (rec & 31));
        *reinterpret_cast<int2 *>(tile_pos + 2 * t) = pos;
    }
    if (tid == 0)
        *reinterpret_cast<int4 *>(&scratch->st[0]) = make_int4(count, 1, limit, __float_as_int(threshold));
#ifdef KIMG_CLEAN_STAMPS
    if (tid == KIMG_SOLO_STAMP_TID) {
        for (int i = 0; i < 8; i++)
            scratch->st[0].pad[i] = (int) (sacc[i] * 100 / max(count, 1));     // 1/100 tick (0.1 ns) per cycle
    }
#endif
#undef SSTAMP
}

// Pixel values at every tile's peak position (the part of a tile record the tile scan of
// kimg_update_tiles does not produce); once per kimg_clean_cycles call.
__global__ __launch_bounds__(256) void tile_pix_kernel(
    const float *__restrict__ dirty, int64_t row_stride, int64_t pol_stride, int width,
    int height, int P, const int32_t *__restrict__ tile_pos, int num_tiles,
    fused_scratch *scratch)
{
    float *tile_pix = reinterpret_cast<float *>(scratch + 1);
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= num_tiles)
        return;
    const int y = tile_pos[2 * t], x = tile_pos[2 * t + 1];
    const bool ok = y >= 0 && y < height && x >= 0 && x < width;
    for (int p = 0; p < 4; p++)
        tile_pix[4 * t + p] = (ok && p < P) ? dirty[p * pol_stride + (int64_t) y * row_stride + x] : 0.0f;
}

// Every owner's best three tiles (into the copy of the table the first launch's keeper reads) and
// the best tile of all with its record, the first launch's `rest` (see fused_cycle); once per
// kimg_clean_cycles call, after tile_pix_kernel.
__global__ __launch_bounds__(1024) void owner_best_kernel(const float *__restrict__ tile_max,
                                                          const int32_t *__restrict__ tile_pos,
                                                          int tiles_x, int tiles_y,
                                                          fused_scratch *scratch)
{
    __shared__ key_t s_keys[16];
    const float *tile_pix = reinterpret_cast<const float *>(scratch + 1);
    owned_tiles walk(threadIdx.x, tiles_x, tiles_y);
    const owner3_t b = walk.best(tile_max, -1, 0.0f);
    scratch->owner3[0][threadIdx.x] = b;
    const key_t key = b.v[0] >= 0.0f ? make_key(b.v[0], b.t[0]) : 0;
    const key_t best = block_max_key(key, s_keys);
    rest_t *rout = &scratch->rest[0];
    if (best == 0) {
        if (threadIdx.x == 0)
            rout->value = -1.0f;
    } else if (key == best) {
        rest_t r;
        r.value = b.v[0];
        r.tile = b.t[0];
        r.y = tile_pos[2 * b.t[0]];
        r.x = tile_pos[2 * b.t[0] + 1];
#pragma unroll
        for (int p = 0; p < 4; p++)
            r.pix[p] = tile_pix[4 * b.t[0] + p];
        *rout = r;
    }
}

// Fold the deltas of the last cycle into the base tile arrays (the state left by an even number
// of cycle launches is st[0], its pending deltas are deltas[0]).
__global__ __launch_bounds__(1024) void apply_deltas_kernel(fused_scratch *scratch, float *tile_max,
                                                            int32_t *tile_pos)
{
    float *tile_pix = reinterpret_cast<float *>(scratch + 1);
    const delta_t d = scratch->deltas[0][threadIdx.x];
    if (d.tag == scratch->st[0].count + 1)
        apply_delta(d, tile_max, tile_pos, tile_pix);
}

// ---- PSF patch bound ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void psf_patch_kernel(
    const float *__restrict__ psf, int64_t row_stride, int64_t pol_stride, int P,
    int min_x, int min_y, int max_x, int max_y, int mid_x, int mid_y, float threshold,
    int32_t *__restrict__ bound)
{
    int dx = 0, dy = 0;
    for (int y = min_y + blockIdx.y; y <= max_y; y += gridDim.y)
        for (int x = min_x + blockIdx.x * blockDim.x + threadIdx.x; x <= max_x;
             x += gridDim.x * blockDim.x) {
            bool over = false;
            for (int p = 0; p < P; p++)
                over |= fabsf(psf[p * pol_stride + (int64_t) y * row_stride + x]) >= threshold;
            if (over) {
                dx = max(dx, abs(x - mid_x));
                dy = max(dy, abs(y - mid_y));
            }
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        dx = max(dx, __shfl_xor(dx, off, WAVE));
        dy = max(dy, __shfl_xor(dy, off, WAVE));
    }
    if ((threadIdx.x & 63) == 0 && (dx | dy)) {
        atomicMax(&bound[0], dx);
        atomicMax(&bound[1], dy);
    }
}

// ---- noise estimate: radix select on the bit pattern of |x| ------------------------------
__global__ __launch_bounds__(256) void abs_histogram_kernel(
    const float *__restrict__ image, int64_t row_stride, int64_t pol_stride, int width,
    int height, int P, int border, int pass, uint32_t prefix, uint32_t *__restrict__ hist)
{
    __shared__ uint32_t local[256];
    local[threadIdx.x] = 0;
    __syncthreads();
    const int shift = 8 * pass;
    // Run-length accumulation: in the first pass (sign-less exponent byte) nearly every pixel
    // of a noise-like image falls into the same two or three bins, and one LDS atomic per
    // pixel would serialise the whole wave on them.
    uint32_t run_bin = 0, run = 0;
    // four rows per round, their loads issued together (the loop is otherwise one dependent
    // memory round trip per pixel row)
    constexpr int ROWS = 4;
    const int x = border + blockIdx.x * blockDim.x + threadIdx.x;
    const bool x_ok = x < width - border;
    for (int p = 0; p < P; p++)
        for (int y0 = border + blockIdx.y; y0 < height - border; y0 += gridDim.y * ROWS) {
            uint32_t keys[ROWS];
            bool ok[ROWS];
#pragma unroll
            for (int r = 0; r < ROWS; r++) {
                const int y = y0 + r * gridDim.y;
                ok[r] = x_ok && y < height - border;
                keys[r] = ok[r] ? __float_as_uint(image[p * pol_stride + (int64_t) y * row_stride + x])
                                      & 0x7fffffffu : 0u;
            }
#pragma unroll
            for (int r = 0; r < ROWS; r++) {
                const uint32_t key = keys[r];
                if (ok[r] && (pass == 3 || (key >> (shift + 8)) == prefix)) {
                    const uint32_t bin = (key >> shift) & 255u;
                    if (bin != run_bin) {
                        if (run)
                            atomicAdd(&local[run_bin], run);
                        run_bin = bin;
                        run = 0;
                    }
                    run++;
                }
            }
        }
    if (run)
        atomicAdd(&local[run_bin], run);
    __syncthreads();
    if (local[threadIdx.x])
        atomicAdd(&hist[threadIdx.x], local[threadIdx.x]);
}

__global__ __launch_bounds__(256) void abs_count_le_kernel(
    const float *__restrict__ image, int64_t row_stride, int64_t pol_stride, int width,
    int height, int P, int border, uint32_t value_bits, uint32_t *__restrict__ out)
{
    uint32_t count = 0, next = 0xffffffffu;
    constexpr int ROWS = 8;
    const int x = border + blockIdx.x * blockDim.x + threadIdx.x;
    const bool x_ok = x < width - border;
    for (int p = 0; p < P; p++)
        for (int y0 = border + blockIdx.y; y0 < height - border; y0 += gridDim.y * ROWS) {
            uint32_t keys[ROWS];
            bool ok[ROWS];
#pragma unroll
            for (int r = 0; r < ROWS; r++) {
                const int y = y0 + r * gridDim.y;
                ok[r] = x_ok && y < height - border;
                keys[r] = ok[r] ? __float_as_uint(image[p * pol_stride + (int64_t) y * row_stride + x])
                                      & 0x7fffffffu : 0u;
            }
#pragma unroll
            for (int r = 0; r < ROWS; r++)
                if (ok[r]) {
                    if (keys[r] <= value_bits)
                        count++;
                    else
                        next = min(next, keys[r]);
                }
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        count += __shfl_xor(count, off, WAVE);
        next = min(next, (uint32_t) __shfl_xor((int) next, off, WAVE));
    }
    // one pair of atomics per workgroup (same-address atomics from thousands of waves serialise)
    __shared__ uint32_t s_count[4], s_next[4];
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_count[wv] = count;
        s_next[wv] = next;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int) (blockDim.x >> 6); w++) {
            count += s_count[w];
            next = min(next, s_next[w]);
        }
        if (count)
            atomicAdd(&out[0], count);
        if (next != 0xffffffffu)
            atomicMin(&out[1], next);
    }
}

dim3 region_grid(int width, int height, int max_blocks = 2048)
{
    int bx = kimg_divup(width, 256);
    if (bx < 1) bx = 1;
    int by = height < max_blocks / bx ? height : max_blocks / bx;
    return dim3(bx, by > 0 ? by : 1);
}

} // namespace

extern "C" int kimg_update_tiles(const float *dirty, int64_t row_stride, int64_t pol_stride,
                                 int width, int height, int num_polarizations, int border,
                                 int mode, float *tile_max, int32_t *tile_pos, int tiles_x,
                                 int tiles_y, int tile_x0, int tile_y0, int tile_x1, int tile_y1,
                                 void *stream)
{
    KIMG_CHECK_ARG(dirty && tile_max && tile_pos && width > 0 && height > 0);
    KIMG_CHECK_ARG(num_polarizations >= 1 && num_polarizations <= 4 && border >= 0);
    KIMG_CHECK_ARG(tile_x0 >= 0 && tile_y0 >= 0 && tile_x1 <= tiles_x && tile_y1 <= tiles_y);
    if (tile_x0 >= tile_x1 || tile_y0 >= tile_y1)
        return 0;                                           // clean.py:462
    dim3 g(tile_x1 - tile_x0, tile_y1 - tile_y0);
    hipStream_t s = (hipStream_t) stream;
    if (mode == KIMG_CLEAN_I)
        update_tiles_kernel<KIMG_CLEAN_I><<<g, 256, 0, s>>>(
            dirty, row_stride, pol_stride, width, height, num_polarizations, border, tile_max,
            tile_pos, tiles_x, tile_x0, tile_y0);
    else if (mode == KIMG_CLEAN_SUMSQ)
        update_tiles_kernel<KIMG_CLEAN_SUMSQ><<<g, 256, 0, s>>>(
            dirty, row_stride, pol_stride, width, height, num_polarizations, border, tile_max,
            tile_pos, tiles_x, tile_x0, tile_y0);
    else
        return KIMG_EINVAL;
    return kimg_launch_status();
}

extern "C" int kimg_find_peak(const float *dirty, int64_t row_stride, int64_t pol_stride,
                              int num_polarizations, const float *tile_max,
                              const int32_t *tile_pos, int tiles_x, int tiles_y,
                              float *peak_value, int32_t *peak_pos, float *peak_pixel,
                              void *stream)
{
    KIMG_CHECK_ARG(dirty && tile_max && tile_pos && peak_value && peak_pos && peak_pixel);
    KIMG_CHECK_ARG(tiles_x > 0 && tiles_y > 0 && num_polarizations >= 1);
    find_peak_kernel<<<1, 1024, 0, (hipStream_t) stream>>>(
        dirty, row_stride, pol_stride, num_polarizations, tile_max, tile_pos, tiles_x * tiles_y,
        peak_value, peak_pos, peak_pixel);
    return kimg_launch_status();
}

extern "C" int kimg_subtract_psf(float *dirty, float *model, int64_t row_stride,
                                 int64_t pol_stride, int width, int height, int num_polarizations,
                                 const float *psf, int64_t psf_row_stride, int64_t psf_pol_stride,
                                 int psf_width, int psf_height, int patch_width, int patch_height,
                                 const float *peak_pixel, int pos_x, int pos_y, float loop_gain,
                                 void *stream)
{
    KIMG_CHECK_ARG(dirty && model && psf && peak_pixel);
    KIMG_CHECK_ARG(num_polarizations >= 1 && num_polarizations <= 4);
    KIMG_CHECK_ARG(patch_width > 0 && patch_height > 0 && patch_width <= psf_width
                   && patch_height <= psf_height);
    KIMG_CHECK_ARG(pos_x >= 0 && pos_x < width && pos_y >= 0 && pos_y < height);
    const int psf_x0 = psf_width / 2 - patch_width / 2;     // clean.py:699-700
    const int psf_y0 = psf_height / 2 - patch_height / 2;
    dim3 g(kimg_divup(patch_width, 64), kimg_divup(patch_height, 4));
    subtract_psf_kernel<<<g, 256, 0, (hipStream_t) stream>>>(
        dirty, model, row_stride, pol_stride, width, height, num_polarizations, psf,
        psf_row_stride, psf_pol_stride, psf_x0, psf_y0, patch_width, patch_height, peak_pixel,
        pos_x, pos_y, pos_x - patch_width / 2, pos_y - patch_height / 2, loop_gain);
    return kimg_launch_status();
}

// byte offset of the persistent form's header in the state buffer (behind fused_scratch + tile_pix)
static size_t persist_offset(int tiles_x, int tiles_y)
{
    const size_t n = sizeof(fused_scratch) + (size_t) tiles_x * tiles_y * 4 * sizeof(float);
    return (n + 255) / 256 * 256;
}

extern "C" size_t kimg_clean_state_bytes(int num_polarizations, int tiles_x, int tiles_y)
{
    (void) num_polarizations;
    static_assert(sizeof(fused_scratch) >= sizeof(clean_state), "the two forms share the scratch");
    if (tiles_x <= 0 || tiles_y <= 0)
        return 0;
    const size_t n = persist_offset(tiles_x, tiles_y) + sizeof(persist_header)
                     + (size_t) (PERSIST_MAX_WGS - 1) * tiles_x * tiles_y * sizeof(replica_t);
    const size_t m = kimg_clean_multi_state_bytes(tiles_x, tiles_y);
    return n > m ? n : m;
}

namespace {

__global__ void init_state_kernel(clean_state *state, int limit, float threshold)
{
    state->limit = limit;
    state->threshold = threshold;
}

struct cycle_args {
    float *dirty, *model;
    int64_t row_stride, pol_stride;
    int width, height, P;
    const float *psf;
    int64_t psf_row_stride, psf_pol_stride;
    int psf_width, psf_height, patch_width, patch_height, border, mode;
    float loop_gain;
    float *tile_max;
    int32_t *tile_pos;
    int tiles_x, tiles_y;
    clean_state *state;
    float *log;
    int fused;              // one launch per cycle (state is then a fused_scratch)
    int batch;              // > 0: that many channels per launch, described by `tab` (the fields
                            // dirty .. log, patch_width / patch_height above are then unused)
    int batch_blocks;       // largest number of lattice blocks over the batch's channels
    batch_table tab;
};

// One minor cycle = two dependent launches (peak + threshold test, then subtract + tile update).
int enqueue_cycle(const cycle_args &a, hipStream_t s, int index)
{
    if (a.batch > 0) {
        const dim3 gb(a.batch_blocks + 2, 1, a.batch);          // + the two bookkeeping workgroups
        if (a.mode == KIMG_CLEAN_I)
            cycle_fused_batch_kernel<KIMG_CLEAN_I><<<gb, 1024, 0, s>>>(
                a.tab, a.row_stride, a.pol_stride, a.width, a.height, a.P, a.psf_row_stride,
                a.psf_pol_stride, a.psf_width, a.psf_height, a.border, a.tiles_x, a.tiles_y,
                a.loop_gain, index & 1);
        else
            cycle_fused_batch_kernel<KIMG_CLEAN_SUMSQ><<<gb, 1024, 0, s>>>(
                a.tab, a.row_stride, a.pol_stride, a.width, a.height, a.P, a.psf_row_stride,
                a.psf_pol_stride, a.psf_width, a.psf_height, a.border, a.tiles_x, a.tiles_y,
                a.loop_gain, index & 1);
        return kimg_launch_status();
    }
    dim3 g(kimg_divup(a.patch_width, TILE) + 1, kimg_divup(a.patch_height, TILE) + 1);
    const int num_tiles = a.tiles_x * a.tiles_y;
    if (a.fused) {
        fused_scratch *fs = reinterpret_cast<fused_scratch *>(a.state);
        g.y += 1;               // the bookkeeping workgroup's row
        if (a.mode == KIMG_CLEAN_I)
            cycle_fused_kernel<KIMG_CLEAN_I><<<g, 1024, 0, s>>>(
                a.dirty, a.model, a.row_stride, a.pol_stride, a.width, a.height, a.P, a.psf,
                a.psf_row_stride, a.psf_pol_stride, a.psf_width, a.psf_height, a.patch_width,
                a.patch_height, a.border, a.tile_max, a.tile_pos, a.tiles_x, a.tiles_y,
                a.loop_gain, fs, index & 1, a.log);
        else
            cycle_fused_kernel<KIMG_CLEAN_SUMSQ><<<g, 1024, 0, s>>>(
                a.dirty, a.model, a.row_stride, a.pol_stride, a.width, a.height, a.P, a.psf,
                a.psf_row_stride, a.psf_pol_stride, a.psf_width, a.psf_height, a.patch_width,
                a.patch_height, a.border, a.tile_max, a.tile_pos, a.tiles_x, a.tiles_y,
                a.loop_gain, fs, index & 1, a.log);
        return kimg_launch_status();
    }
    if (a.mode == KIMG_CLEAN_I) {
        cycle_find_peak_kernel<KIMG_CLEAN_I><<<1, 1024, 0, s>>>(
            a.dirty, a.model, a.row_stride, a.pol_stride, a.P, a.tile_max, a.tile_pos, num_tiles,
            a.loop_gain, a.state, a.log);
        cycle_subtract_update_kernel<KIMG_CLEAN_I><<<g, 256, 0, s>>>(
            a.dirty, a.row_stride, a.pol_stride, a.width, a.height, a.P, a.psf, a.psf_row_stride,
            a.psf_pol_stride, a.psf_width, a.psf_height, a.patch_width, a.patch_height, a.border,
            a.tile_max, a.tile_pos, a.tiles_x, a.tiles_y, a.state);
    } else {
        cycle_find_peak_kernel<KIMG_CLEAN_SUMSQ><<<1, 1024, 0, s>>>(
            a.dirty, a.model, a.row_stride, a.pol_stride, a.P, a.tile_max, a.tile_pos, num_tiles,
            a.loop_gain, a.state, a.log);
        cycle_subtract_update_kernel<KIMG_CLEAN_SUMSQ><<<g, 256, 0, s>>>(
            a.dirty, a.row_stride, a.pol_stride, a.width, a.height, a.P, a.psf, a.psf_row_stride,
            a.psf_pol_stride, a.psf_width, a.psf_height, a.patch_width, a.patch_height, a.border,
            a.tile_max, a.tile_pos, a.tiles_x, a.tiles_y, a.state);
    }
    return kimg_launch_status();
}

// hipGraph of GRAPH_CYCLES minor cycles, cached per argument set: the minor-cycle loop is
// launch-bound, and replaying a captured graph costs far less host time than 2 launches
// per cycle.  The device-side `limit` makes surplus cycles of the last replay no-ops.
#ifndef KIMG_GRAPH_CYCLES
#define KIMG_GRAPH_CYCLES 64
#endif
constexpr int GRAPH_CYCLES = KIMG_GRAPH_CYCLES;
// the one-launch form alternates two state / delta buffers by launch parity and must leave the
// final state in st[0]: a replay has to be an even number of launches
static_assert(GRAPH_CYCLES >= 2 && GRAPH_CYCLES % 2 == 0, "KIMG_GRAPH_CYCLES must be even");
constexpr int GRAPH_CACHE = 32;     // argument sets (channels in flight x patch sizes)

struct graph_entry {
    bool valid;
    int users;              // calls that hold `exec` and have not finished enqueuing its replays
    cycle_args args;
    hipGraphExec_t exec;
    hipEvent_t last_use;    // recorded after the last replay enqueued by a finished call
    bool used;
    int device;             // the device `last_use` (and the graph) belongs to
};
graph_entry graph_cache[GRAPH_CACHE];
std::mutex graph_mutex;         // channels imaged concurrently share the cache

// The cached (or newly captured) graph for `a`; the entry stays pinned until graph_release().
// An entry is only evicted when no call is using it and the replays enqueued from it have
// completed (its event has fired), so a graph is never destroyed while it is in flight.
graph_entry *cycles_graph(const cycle_args &a, hipStream_t s)
{
    std::lock_guard<std::mutex> lock(graph_mutex);
    for (int i = 0; i < GRAPH_CACHE; i++)
        if (graph_cache[i].valid && memcmp(&graph_cache[i].args, &a, sizeof(a)) == 0) {
            graph_cache[i].users++;
            return &graph_cache[i];
        }
    graph_entry *slot = nullptr;
    for (int i = 0; i < GRAPH_CACHE && !slot; i++)
        if (!graph_cache[i].valid)
            slot = &graph_cache[i];
    for (int i = 0; i < GRAPH_CACHE && !slot; i++)
        if (graph_cache[i].users == 0
            && (!graph_cache[i].used || hipEventQuery(graph_cache[i].last_use) == hipSuccess))
            slot = &graph_cache[i];
    if (!slot)
        return nullptr;             // every entry busy: the caller enqueues plain launches
    hipGraph_t graph = nullptr;
    // (captured on the library's own stream of this thread, launched on the caller's: see
    // kimg_capture_stream)
    hipStream_t cs = kimg_capture_stream();
    if (cs == nullptr || hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal) != hipSuccess)
        return nullptr;
    int rc = 0;
    for (int i = 0; i < GRAPH_CYCLES && rc == 0; i++)
        rc = enqueue_cycle(a, cs, i);
    const hipError_t ended = hipStreamEndCapture(cs, &graph);
    if (ended != hipSuccess || rc != 0) {
        if (ended == hipSuccess && graph != nullptr)
            (void) hipGraphDestroy(graph);      // (a launch failed during the capture)
        return nullptr;
    }
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void) hipGraphDestroy(graph);
    if (e != hipSuccess)
        return nullptr;
    // (imagers on several devices may share this process: an event is recorded on streams of the
    // device it was created on, so a slot taken over from another device gets a new one)
    int device = 0;
    (void) hipGetDevice(&device);
    if (slot->valid) {
        (void) hipGraphExecDestroy(slot->exec);
        if (slot->device != device) {
            (void) hipEventDestroy(slot->last_use);
            slot->valid = false;
        }
    }
    if (!slot->valid && hipEventCreateWithFlags(&slot->last_use, hipEventDisableTiming) != hipSuccess) {
        (void) hipGraphExecDestroy(exec);
        return nullptr;
    }
    slot->device = device;
    slot->valid = true;
    slot->used = false;
    slot->users = 1;
    slot->args = a;
    slot->exec = exec;
    return slot;
}

void graph_release(graph_entry *entry, hipStream_t s)
{
    std::lock_guard<std::mutex> lock(graph_mutex);
    (void) hipEventRecord(entry->last_use, s);
    entry->used = true;
    entry->users--;
}

} // namespace

// The minor cycles of one major cycle in one call (frontend.py:560-585): the first cycle runs without
// a threshold, the threshold of the rest follows from its peak on the device.  Only where the
// multi-component form runs (KIMG_EUNSUPPORTED otherwise: the caller takes kimg_clean_cycles then).
extern "C" int kimg_clean_major_cycles(float *dirty, float *model, int64_t row_stride,
                                       int64_t pol_stride, int width, int height, int num_polarizations,
                                       const float *psf, int64_t psf_row_stride, int64_t psf_pol_stride,
                                       int psf_width, int psf_height, int patch_width, int patch_height,
                                       int border, int mode, float loop_gain, double noise_threshold,
                                       double left_for_next, float *tile_max, int32_t *tile_pos,
                                       int tiles_x, int tiles_y, int max_cycles, int form, void *state,
                                       float *log, void *stream, int *cycles_done, float *first_peak)
{
    KIMG_CHECK_ARG(dirty && model && psf && tile_max && tile_pos && state && log);
    KIMG_CHECK_ARG(num_polarizations >= 1 && num_polarizations <= 4 && max_cycles >= 1);
    KIMG_CHECK_ARG(patch_width > 0 && patch_height > 0 && patch_width <= psf_width
                   && patch_height <= psf_height && tiles_x > 0 && tiles_y > 0);
    KIMG_CHECK_ARG(mode == KIMG_CLEAN_I || mode == KIMG_CLEAN_SUMSQ);
    KIMG_CHECK_ARG(noise_threshold == noise_threshold && left_for_next == left_for_next);      // (not NaN)
    const int kind = form & 0xff;
    KIMG_CHECK_ARG(kind == KIMG_CLEAN_FORM_AUTO || kind == KIMG_CLEAN_FORM_MULTI);
    if (kimg_clean_multi_components(patch_width, patch_height, tiles_x, tiles_y) < (kind == KIMG_CLEAN_FORM_MULTI ? 1 : 2))
        return KIMG_EUNSUPPORTED;
    return kimg_clean_multi_run(dirty, model, row_stride, pol_stride, width, height, num_polarizations, psf,
                                psf_row_stride, psf_pol_stride, psf_width, psf_height, patch_width,
                                patch_height, border, mode, loop_gain, 0.0f, tile_max, tile_pos, tiles_x,
                                tiles_y, max_cycles, (form >> 8) & 0xff, (form >> 16) & 0x1f, true,
                                noise_threshold, left_for_next, state, log, (hipStream_t) stream, cycles_done,
                                first_peak);
}

extern "C" int kimg_clean_cycles(float *dirty, float *model, int64_t row_stride,
                                 int64_t pol_stride, int width, int height, int num_polarizations,
                                 const float *psf, int64_t psf_row_stride, int64_t psf_pol_stride,
                                 int psf_width, int psf_height, int patch_width, int patch_height,
                                 int border, int mode, float loop_gain, float threshold,
                                 float *tile_max, int32_t *tile_pos, int tiles_x, int tiles_y,
                                 int max_cycles, int form, void *state, float *log, void *stream)
{
    KIMG_CHECK_ARG(dirty && model && psf && tile_max && tile_pos && state && log);
    KIMG_CHECK_ARG(num_polarizations >= 1 && num_polarizations <= 4 && max_cycles >= 0);
    KIMG_CHECK_ARG(patch_width > 0 && patch_height > 0 && patch_width <= psf_width
                   && patch_height <= psf_height && tiles_x > 0 && tiles_y > 0);
    KIMG_CHECK_ARG(mode == KIMG_CLEAN_I || mode == KIMG_CLEAN_SUMSQ);
    const int components = (form >> 8) & 0xff;      // KIMG_CLEAN_FORM_MULTI: lattices per launch
    // ... and steps per lattice (0: as many as the form takes); bit 4: the repeated-steps kernel from
    // the first launch on, whatever the field looks like (tests)
    const int repeats = (form >> 16) & 0x1f;
    form &= 0xff;
    KIMG_CHECK_ARG(form == KIMG_CLEAN_FORM_AUTO || form == KIMG_CLEAN_FORM_TWO_LAUNCH
                   || form == KIMG_CLEAN_FORM_ONE_LAUNCH || form == KIMG_CLEAN_FORM_PERSISTENT
                   || form == KIMG_CLEAN_FORM_ONE_WORKGROUP || form == KIMG_CLEAN_FORM_MULTI);
    hipStream_t s = (hipStream_t) stream;
    // several components per launch where the patch leaves room for at least two lattices among
    // the 256 records of a launch (a call of a few cycles is not worth the host-paced loop)
    {
        const int m = kimg_clean_multi_components(patch_width, patch_height, tiles_x, tiles_y);
        if (max_cycles > 0 && (form == KIMG_CLEAN_FORM_MULTI ? m >= 1
                                                             : form == KIMG_CLEAN_FORM_AUTO && m >= 2
                                                                   && max_cycles >= 4)) {
            const int rc = kimg_clean_multi_run(
                dirty, model, row_stride, pol_stride, width, height, num_polarizations, psf,
                psf_row_stride, psf_pol_stride, psf_width, psf_height, patch_width, patch_height,
                border, mode, loop_gain, threshold, tile_max, tile_pos, tiles_x, tiles_y,
                max_cycles, components, repeats, false, 0.0, 0.0, state, log, s);
            if (rc != KIMG_EUNSUPPORTED)
                return rc;
        }
        if (form == KIMG_CLEAN_FORM_MULTI)
            form = KIMG_CLEAN_FORM_AUTO;
    }
    // one launch per cycle when the patch touches few lattice blocks (every workgroup then
    // repeats the global peak search)
    const int bx = kimg_divup(patch_width, TILE) + 1, by = kimg_divup(patch_height, TILE) + 1;
    const bool fused = bx * (by + 1) <= FUSED_MAX_BLOCKS && bx <= 32 && by <= 32
                       && kimg_divup(tiles_x, 32) * kimg_divup(tiles_y, 32) <= FUSED_MAX_SLOTS
                       && form != KIMG_CLEAN_FORM_TWO_LAUNCH;
    // the whole loop in one workgroup when the patch is small and the tile records and the PSF
    // patch fit LDS
    const size_t solo_lds = (((size_t) tiles_x * tiles_y * 6 + 15) & ~(size_t) 15)
                            + (size_t) (patch_width + 2 * SOLO_PAD) * patch_height * sizeof(float);
    const bool solo_ok = num_polarizations == 1 && bx <= SOLO_MAX_BX && by <= 32
                         && bx * by <= SOLO_MAX_BLOCKS
                         && (uint64_t) height * (uint64_t) row_stride < (1u << 30)
                         && solo_lds <= SOLO_LDS_LIMIT && max_cycles > 0;
    if (solo_ok && (form == KIMG_CLEAN_FORM_ONE_WORKGROUP
                    || (form == KIMG_CLEAN_FORM_AUTO && bx * by <= SOLO_AUTO_BLOCKS))) {
#define SOLO(MODE) do { \
        { \
            const int lds_rc = kimg_dynamic_lds(reinterpret_cast<const void *>(&cycle_solo_kernel<MODE>), \
                                                SOLO_LDS_LIMIT); \
            if (lds_rc) \
                return lds_rc; \
        } \
        cycle_solo_kernel<MODE><<<1, 1024, solo_lds, s>>>( \
            dirty, model, row_stride, width, height, psf, psf_row_stride, psf_width, psf_height, \
            patch_width, patch_height, border, tile_max, tile_pos, tiles_x, tiles_y, loop_gain, \
            threshold, max_cycles, static_cast<fused_scratch *>(state), log); } while (0)
        if (mode == KIMG_CLEAN_I)
            SOLO(KIMG_CLEAN_I);
        else
            SOLO(KIMG_CLEAN_SUMSQ);
#undef SOLO
        return kimg_launch_status();
    }
    KIMG_HIP(hipMemsetAsync(state, 0, sizeof(fused_scratch), s));
    init_state_kernel<<<1, 1, 0, s>>>(static_cast<clean_state *>(state), max_cycles, threshold);
    if (fused)
        tile_pix_kernel<<<kimg_divup(tiles_x * tiles_y, 256), 256, 0, s>>>(
            dirty, row_stride, pol_stride, width, height, num_polarizations, tile_pos,
            tiles_x * tiles_y, static_cast<fused_scratch *>(state));
    // the whole loop in one launch when the patch's lattice blocks can all be resident and the
    // tile maxima fit LDS
    const size_t persist_lds = (((size_t) tiles_x * tiles_y * 4 + 15) & ~(size_t) 15)
                               + 1024 * sizeof(delta_t);
    const bool persistent = fused && bx * by <= PERSIST_MAX_WGS && persist_lds <= PERSIST_LDS_LIMIT
                            && form == KIMG_CLEAN_FORM_PERSISTENT && max_cycles > 0;
    if (persistent) {
        unsigned char *base = static_cast<unsigned char *>(state) + persist_offset(tiles_x, tiles_y);
        persist_header *hdr = reinterpret_cast<persist_header *>(base);
        replica_t *replicas = reinterpret_cast<replica_t *>(base + sizeof(persist_header));
        KIMG_HIP(hipMemsetAsync(hdr, 0, sizeof(persist_header), s));
        const dim3 g(bx, by);
#define PERSIST(MODE) do { \
        { \
            const int lds_rc = kimg_dynamic_lds(reinterpret_cast<const void *>(&cycle_persistent_kernel<MODE>), \
                                                PERSIST_LDS_LIMIT); \
            if (lds_rc) \
                return lds_rc; \
        } \
        cycle_persistent_kernel<MODE><<<g, 1024, persist_lds, s>>>( \
            dirty, model, row_stride, pol_stride, width, height, num_polarizations, psf, \
            psf_row_stride, psf_pol_stride, psf_width, psf_height, patch_width, patch_height, border, \
            tile_max, tile_pos, tiles_x, tiles_y, loop_gain, threshold, max_cycles, \
            static_cast<fused_scratch *>(state), hdr, replicas, log); } while (0)
        if (mode == KIMG_CLEAN_I)
            PERSIST(KIMG_CLEAN_I);
        else
            PERSIST(KIMG_CLEAN_SUMSQ);
#undef PERSIST
        persist_status_kernel<<<1, 1, 0, s>>>(static_cast<fused_scratch *>(state), hdr);
        return kimg_launch_status();
    }
    if (fused)
        owner_best_kernel<<<1, 1024, 0, s>>>(tile_max, tile_pos, tiles_x, tiles_y,
                                             static_cast<fused_scratch *>(state));
    cycle_args a;
    memset(&a, 0, sizeof(a));       // padding bytes take part in the cache key comparison
    a.dirty = dirty; a.model = model; a.row_stride = row_stride; a.pol_stride = pol_stride;
    a.width = width; a.height = height; a.P = num_polarizations; a.psf = psf;
    a.psf_row_stride = psf_row_stride; a.psf_pol_stride = psf_pol_stride;
    a.psf_width = psf_width; a.psf_height = psf_height; a.patch_width = patch_width;
    a.patch_height = patch_height; a.border = border; a.mode = mode; a.loop_gain = loop_gain;
    a.tile_max = tile_max; a.tile_pos = tile_pos; a.tiles_x = tiles_x;
    a.tiles_y = tiles_y; a.state = static_cast<clean_state *>(state); a.log = log;
    a.fused = fused;
    int done = 0;
    if (max_cycles >= GRAPH_CYCLES / 2) {
        graph_entry *entry = cycles_graph(a, s);
        if (entry) {
            hipError_t e = hipSuccess;
            for (; done < max_cycles && e == hipSuccess; done += GRAPH_CYCLES)
                e = hipGraphLaunch(entry->exec, s);
            graph_release(entry, s);
            if (e != hipSuccess)
                return -(int) e;
            done = max_cycles;
        }
    }
    // (an even number of launches, so that the fused form leaves its state in st[0]; the
    // device-side limit makes the surplus one a no-op)
    for (int i = 0; done < max_cycles || (i & 1); done++, i++) {
        int rc = enqueue_cycle(a, s, i);
        if (rc)
            return rc;
    }
    if (fused)
        apply_deltas_kernel<<<1, 1024, 0, s>>>(static_cast<fused_scratch *>(state), tile_max,
                                               tile_pos);
    return kimg_launch_status();
}

extern "C" int kimg_clean_cycles_batch(const kimg_clean_channel *channels_in, int num_channels,
                                       int64_t row_stride, int64_t pol_stride, int width,
                                       int height, int num_polarizations, int64_t psf_row_stride,
                                       int64_t psf_pol_stride, int psf_width, int psf_height,
                                       int border, int mode, float loop_gain, int tiles_x,
                                       int tiles_y, void *stream)
{
    KIMG_CHECK_ARG(channels_in && num_channels >= 1 && num_channels <= KIMG_CLEAN_BATCH_MAX);
    KIMG_CHECK_ARG(num_polarizations >= 1 && num_polarizations <= 4 && tiles_x > 0 && tiles_y > 0);
    KIMG_CHECK_ARG(mode == KIMG_CLEAN_I || mode == KIMG_CLEAN_SUMSQ);
    hipStream_t s = (hipStream_t) stream;
    cycle_args a;
    memset(&a, 0, sizeof(a));       // padding bytes take part in the cache key comparison
    int max_cycles = 0;
    if (kimg_divup(tiles_x, 32) * kimg_divup(tiles_y, 32) > FUSED_MAX_SLOTS)
        return KIMG_EUNSUPPORTED;
    // The order of the channels does not matter to the results, but the captured graph is cached
    // per argument set: channels that meet in another order (threads arriving at a rendezvous)
    // must find the same graph, so the table is kept sorted by image address.
    kimg_clean_channel sorted[KIMG_CLEAN_BATCH_MAX];
    for (int c = 0; c < num_channels; c++) {
        int at = c;
        while (at > 0 && sorted[at - 1].dirty > channels_in[c].dirty) {
            sorted[at] = sorted[at - 1];
            at--;
        }
        sorted[at] = channels_in[c];
    }
    const kimg_clean_channel *channels = sorted;
    for (int c = 0; c < num_channels; c++) {
        const kimg_clean_channel &ch = channels[c];
        KIMG_CHECK_ARG(ch.dirty && ch.model && ch.psf && ch.tile_max && ch.tile_pos && ch.state
                       && ch.log && ch.max_cycles >= 0);
        KIMG_CHECK_ARG(ch.patch_width > 0 && ch.patch_height > 0 && ch.patch_width <= psf_width
                       && ch.patch_height <= psf_height);
        for (int o = 0; o < c; o++)     // channels are cleaned concurrently: no shared buffers
            KIMG_CHECK_ARG(channels[o].dirty != ch.dirty && channels[o].state != ch.state
                           && channels[o].tile_max != ch.tile_max && channels[o].log != ch.log);
        const int bx = kimg_divup(ch.patch_width, TILE) + 1, by = kimg_divup(ch.patch_height, TILE) + 1;
        if (!(bx * (by + 1) <= FUSED_MAX_BLOCKS && bx <= 32 && by <= 32))
            return KIMG_EUNSUPPORTED;
        a.batch_blocks = bx * by > a.batch_blocks ? bx * by : a.batch_blocks;
        max_cycles = ch.max_cycles > max_cycles ? ch.max_cycles : max_cycles;
        batch_channel &b = a.tab.ch[c];
        b.dirty = ch.dirty;
        b.model = ch.model;
        b.psf = ch.psf;
        b.tile_max = ch.tile_max;
        b.tile_pos = ch.tile_pos;
        b.scratch = static_cast<fused_scratch *>(ch.state);
        b.log = ch.log;
        b.patch_w = ch.patch_width;
        b.patch_h = ch.patch_height;
    }
    a.batch = num_channels;
    a.row_stride = row_stride; a.pol_stride = pol_stride; a.width = width; a.height = height;
    a.P = num_polarizations; a.psf_row_stride = psf_row_stride; a.psf_pol_stride = psf_pol_stride;
    a.psf_width = psf_width; a.psf_height = psf_height; a.border = border; a.mode = mode;
    a.loop_gain = loop_gain; a.tiles_x = tiles_x; a.tiles_y = tiles_y; a.fused = 1;
    // per channel, as kimg_clean_cycles does for the one-launch form: state, peak pixels of every
    // tile, every owner's best two tiles
    for (int c = 0; c < num_channels; c++) {
        const kimg_clean_channel &ch = channels[c];
        KIMG_HIP(hipMemsetAsync(ch.state, 0, sizeof(fused_scratch), s));
        init_state_kernel<<<1, 1, 0, s>>>(static_cast<clean_state *>(ch.state), ch.max_cycles,
                                          ch.threshold);
        tile_pix_kernel<<<kimg_divup(tiles_x * tiles_y, 256), 256, 0, s>>>(
            ch.dirty, row_stride, pol_stride, width, height, num_polarizations, ch.tile_pos,
            tiles_x * tiles_y, static_cast<fused_scratch *>(ch.state));
        owner_best_kernel<<<1, 1024, 0, s>>>(ch.tile_max, ch.tile_pos, tiles_x, tiles_y,
                                             static_cast<fused_scratch *>(ch.state));
    }
    int rc = kimg_launch_status();
    if (rc)
        return rc;
    int done = 0;
    if (max_cycles >= GRAPH_CYCLES / 2) {
        graph_entry *entry = cycles_graph(a, s);
        if (entry) {
            hipError_t e = hipSuccess;
            for (; done < max_cycles && e == hipSuccess; done += GRAPH_CYCLES)
                e = hipGraphLaunch(entry->exec, s);
            graph_release(entry, s);
            if (e != hipSuccess)
                return -(int) e;
            done = max_cycles;
        }
    }
    // (an even number of launches: the final state is then in st[0], see kimg_clean_cycles)
    for (int i = 0; done < max_cycles || (i & 1); done++, i++) {
        rc = enqueue_cycle(a, s, i);
        if (rc)
            return rc;
    }
    for (int c = 0; c < num_channels; c++)
        apply_deltas_kernel<<<1, 1024, 0, s>>>(static_cast<fused_scratch *>(channels[c].state),
                                               channels[c].tile_max, channels[c].tile_pos);
    return kimg_launch_status();
}

extern "C" int kimg_psf_patch(const float *psf, int64_t row_stride, int64_t pol_stride,
                              int num_polarizations, int min_x, int min_y, int max_x, int max_y,
                              int mid_x, int mid_y, float threshold, int32_t *bound, void *stream)
{
    KIMG_CHECK_ARG(psf && bound && num_polarizations >= 1 && max_x >= min_x && max_y >= min_y);
    hipStream_t s = (hipStream_t) stream;
    KIMG_HIP(hipMemsetAsync(bound, 0, 2 * sizeof(int32_t), s));
    psf_patch_kernel<<<region_grid(max_x - min_x + 1, max_y - min_y + 1), 256, 0, s>>>(
        psf, row_stride, pol_stride, num_polarizations, min_x, min_y, max_x, max_y, mid_x, mid_y,
        threshold, bound);
    return kimg_launch_status();
}

extern "C" int kimg_abs_histogram(const float *image, int64_t row_stride, int64_t pol_stride,
                                  int width, int height, int num_polarizations, int border,
                                  int pass, uint32_t prefix, uint32_t *hist, void *stream)
{
    KIMG_CHECK_ARG(image && hist && pass >= 0 && pass <= 3 && border >= 0);
    KIMG_CHECK_ARG(width > 2 * border && height > 2 * border && num_polarizations >= 1);
    hipStream_t s = (hipStream_t) stream;
    KIMG_HIP(hipMemsetAsync(hist, 0, 256 * sizeof(uint32_t), s));
    abs_histogram_kernel<<<region_grid(width - 2 * border, height - 2 * border, 2048), 256, 0, s>>>(
        image, row_stride, pol_stride, width, height, num_polarizations, border, pass, prefix, hist);
    return kimg_launch_status();
}

extern "C" int kimg_abs_count_le(const float *image, int64_t row_stride, int64_t pol_stride,
                                 int width, int height, int num_polarizations, int border,
                                 float value, uint32_t *out, void *stream)
{
    KIMG_CHECK_ARG(image && out && border >= 0 && num_polarizations >= 1);
    KIMG_CHECK_ARG(width > 2 * border && height > 2 * border);
    hipStream_t s = (hipStream_t) stream;
    static const uint32_t init[2] = {0u, 0xffffffffu};
    KIMG_HIP(hipMemcpyAsync(out, init, sizeof(init), hipMemcpyHostToDevice, s));
    union { float f; uint32_t u; } conv;
    conv.f = value;
    const uint32_t bits = conv.u;
    abs_count_le_kernel<<<region_grid(width - 2 * border, height - 2 * border, 4096), 256, 0, s>>>(
        image, row_stride, pol_stride, width, height, num_polarizations, border,
        bits & 0x7fffffffu, out);
    return kimg_launch_status();
}

// ---- whole noise estimate without host round trips ------------------------------------------
namespace {

// The two middle ranks ((n-1)/2 and n/2: the same element for odd n) are resolved side by side,
// so the upper one costs no pass of its own: while both still lie behind the same bytes one
// histogram serves the pair, and from the pass after they part each has its own.
// Workgroups add their histogram into one of NOISE_SLOTS copies: atomics on one address are
// served one after the other (~30 ns each), and 2048 workgroups behind the same few bins of the
// exponent byte were most of a pass.
constexpr int NOISE_SLOTS = 32;
struct noise_state {
    uint32_t prefix[2];     // bytes of the rank's |x| chosen so far
    uint32_t k[2];          // rank still to resolve inside the current prefix
    uint32_t hist[2][NOISE_SLOTS][256];
};

// One radix-select pass over the interior: histogram of byte `pass` of |x|'s bit pattern among
// the pixels whose higher bytes equal the rank's prefix.  TOP: the first pass (the sign-less
// exponent byte, every pixel counts).
template<bool TOP>
__global__ __launch_bounds__(256) void abs_histogram2_kernel(
    const float *__restrict__ image, int64_t row_stride, int64_t pol_stride, int width,
    int height, int P, int border, int pass, noise_state *__restrict__ st)
{
    __shared__ uint32_t local[2][256];
    local[0][threadIdx.x] = 0;
    local[1][threadIdx.x] = 0;
    const uint32_t prefix0 = st->prefix[0], prefix1 = st->prefix[1];
    const bool split = !TOP && prefix0 != prefix1;
    __syncthreads();
    const int shift = TOP ? 24 : 8 * pass;
    // TOP: the exponent byte of a noise-like image takes three or four neighbouring values, and
    // 64 LDS atomics on one address are served one after the other; each thread counts the four
    // values up to the largest of its wave's first row in registers instead (anything else --
    // an image with a wide dynamic range -- still goes to the LDS histogram directly).
    uint32_t base = 0, near[4] = {0, 0, 0, 0};
    bool have_base = false;
    constexpr int ROWS = 4;
    const int x = border + blockIdx.x * blockDim.x + threadIdx.x;
    const bool x_ok = x < width - border;
    for (int p = 0; p < P; p++)
        for (int y0 = border + blockIdx.y; y0 < height - border; y0 += gridDim.y * ROWS) {
            uint32_t keys[ROWS];
            bool ok[ROWS];
#pragma unroll
            for (int r = 0; r < ROWS; r++) {
                const int y = y0 + r * gridDim.y;
                ok[r] = x_ok && y < height - border;
                keys[r] = ok[r] ? __float_as_uint(image[p * pol_stride + (int64_t) y * row_stride + x])
                                      & 0x7fffffffu : 0u;
            }
            if (TOP && !have_base) {
                base = ok[0] ? keys[0] >> 24 : 0u;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1)
                    base = max(base, (uint32_t) __shfl_xor((int) base, off, WAVE));
                have_base = true;
            }
#pragma unroll
            for (int r = 0; r < ROWS; r++) {
                const uint32_t key = keys[r];
                const uint32_t bin = (key >> shift) & 255u;
                if (TOP) {
                    const uint32_t d = base - bin;
                    if (ok[r]) {
#pragma unroll
                        for (int j = 0; j < 4; j++)
                            near[j] += d == (uint32_t) j;
                        if (d > 3u)
                            atomicAdd(&local[0][bin], 1u);
                    }
                } else {
                    const uint32_t upper = key >> (shift + 8);
                    // a mantissa byte: the bins of neighbouring pixels differ
                    if (ok[r] && upper == prefix0)
                        atomicAdd(&local[0][bin], 1u);
                    if (split && ok[r] && upper == prefix1)
                        atomicAdd(&local[1][bin], 1u);
                }
            }
        }
    if (TOP) {
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (near[j])
                atomicAdd(&local[0][base - j], near[j]);
    }
    __syncthreads();
    const int slot = (blockIdx.y * gridDim.x + blockIdx.x) % NOISE_SLOTS;
    if (local[0][threadIdx.x])
        atomicAdd(&st->hist[0][slot][threadIdx.x], local[0][threadIdx.x]);
    if (local[1][threadIdx.x])
        atomicAdd(&st->hist[1][slot][threadIdx.x], local[1][threadIdx.x]);
}

// Pick, for each of the two ranks, the byte whose bin holds it, descend into it, clear the
// histograms for the next pass.
__global__ __launch_bounds__(256) void radix_select2_kernel(noise_state *st, bool last,
                                                            float median_to_rms, float *out)
{
    __shared__ uint32_t cum[256];
    const int t = threadIdx.x;
    const bool split = st->prefix[0] != st->prefix[1];
    const uint32_t k[2] = {st->k[0], st->k[1]};
    const uint32_t prefix[2] = {st->prefix[0], st->prefix[1]};
    __syncthreads();
    uint32_t total[2] = {0, 0};
    for (int r = 0; r < 2; r++)
        for (int slot = 0; slot < NOISE_SLOTS; slot++) {
            total[r] += st->hist[r][slot][t];
            st->hist[r][slot][t] = 0;
        }
    for (int r = 0; r < 2; r++) {
        if (r == 0 || split) {
            const uint32_t mine = total[r];
            __syncthreads();
            cum[t] = mine;
            __syncthreads();
            for (int off = 1; off < 256; off <<= 1) {
                const uint32_t add = t >= off ? cum[t - off] : 0;
                __syncthreads();
                cum[t] += add;
                __syncthreads();
            }
        }
        const uint32_t mine = total[split ? r : 0];
        const uint32_t below = cum[t] - mine;
        if (below <= k[r] && k[r] < cum[t]) {       // exactly one bin
            st->k[r] = k[r] - below;
            st->prefix[r] = (prefix[r] << 8) | (uint32_t) t;
        }
    }
    if (last) {
        __threadfence_block();
        __syncthreads();
        if (t == 0) {
            const volatile uint32_t *chosen = st->prefix;
            const float lo = __uint_as_float(chosen[0]);
            const float hi = __uint_as_float(chosen[1]);
            const float median = (lo + hi) / 2.0f;      // np.median of float32 data (clean.py:942)
            *out = median * median_to_rms;
        }
    }
}

// Clear the histograms and set the two ranks.
__global__ __launch_bounds__(1024) void noise_init_kernel(noise_state *st, uint32_t k0, uint32_t k1)
{
    uint32_t *words = &st->hist[0][0][0];
    for (int i = threadIdx.x; i < 2 * NOISE_SLOTS * 256; i += blockDim.x)
        words[i] = 0;
    if (threadIdx.x == 0) {
        st->prefix[0] = st->prefix[1] = 0;
        st->k[0] = k0;
        st->k[1] = k1;
    }
}

} // namespace

extern "C" size_t kimg_noise_est_scratch_bytes(void) { return sizeof(noise_state); }

extern "C" int kimg_noise_est(const float *image, int64_t row_stride, int64_t pol_stride,
                              int width, int height, int num_polarizations, int border,
                              float median_to_rms, void *scratch, float *out, void *stream)
{
    KIMG_CHECK_ARG(image && scratch && out && border >= 0 && num_polarizations >= 1);
    KIMG_CHECK_ARG(width > 2 * border && height > 2 * border);
    const int64_t n64 = (int64_t) (width - 2 * border) * (height - 2 * border) * num_polarizations;
    KIMG_CHECK_ARG(n64 < ((int64_t) 1 << 32));
    const uint32_t n = (uint32_t) n64;
    hipStream_t s = (hipStream_t) stream;
    noise_state *st = static_cast<noise_state *>(scratch);
    noise_init_kernel<<<1, 1024, 0, s>>>(st, (n - 1) / 2, n / 2);    // the middle pair, clean.py:938-943
    const dim3 gh = region_grid(width - 2 * border, height - 2 * border, 2048);
    for (int pass = 3; pass >= 0; pass--) {
        if (pass == 3)
            abs_histogram2_kernel<true><<<gh, 256, 0, s>>>(
                image, row_stride, pol_stride, width, height, num_polarizations, border, pass, st);
        else
            abs_histogram2_kernel<false><<<gh, 256, 0, s>>>(
                image, row_stride, pol_stride, width, height, num_polarizations, border, pass, st);
        radix_select2_kernel<<<1, 256, 0, s>>>(st, pass == 0, median_to_rms, out);
    }
    return kimg_launch_status();
}

// (kimg_preload, api.hip)
KIMG_PRELOAD_THIS_UNIT(subtract_psf_kernel)
