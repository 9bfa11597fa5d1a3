// Hogbom CLEAN, several components per launch (KIMG_CLEAN_FORM_MULTI).
//
// The minor cycle of the reference (CleanHost.__call__, clean.py:1060-1075; GPU form clean.py:848-891)
// is strictly sequential: peak -> subtract -> rescan -> peak ...  On the device one such step is a
// chain of dependent memory round trips behind a kernel boundary, about 5.5 us whatever the patch
// size (cycle_fused_kernel in clean.hip), i.e. 180 K components per second, on a tenth of the CUs.
// What makes several steps per launch possible is that the components of a few consecutive cycles
// are nearly always far apart: cycle k + 1 picks the next-best tile of the image unless subtraction
// k left something even larger behind, and a subtraction only touches the "lattice" of 32 x 32 blocks
// around its peak.  So a launch
//
//   VERIFIES the last launch's plan (c_1 .. c_M): component i + 1 really was c_{i+1} iff every tile
//       record ("delta") the lattice workgroups of c_1 .. c_i produced has a smaller key than
//       c_{i+1} -- key = (metric, lowest tile index first), the reference's tie-break (clean.py:
//       953-958, np.argmax at :1062).  j = the longest prefix that holds; components j+1 .. M were
//       evaluated for nothing, and nothing of theirs was written anywhere that counts;
//   COMMITS c_1 .. c_j: log entries and model pixels (keeper workgroup), their deltas go to the base
//       tile arrays (folder workgroup), their pixel subtractions are written now ("pending": the
//       image in memory lags the committed components by exactly one launch);
//   PLANS the next components from the per-lattice best of the committed deltas and the keeper's
//       sorted list R of the best tiles outside the last lattices -- a candidate is taken only while
//       order and independence are PROVEN: it beats every tile outside the candidate pool (the
//       second-best delta of any lattice, the floor of R, the pool's overflow), passes threshold and
//       cycle limit, and its lattice is disjoint from those taken so far (a candidate whose tile lies
//       inside a taken lattice is skipped: its value is about to change and the next verification
//       covers it).  The first candidate is the exact global maximum in every case, so a launch
//       always makes the progress of one reference cycle;
//   EXECUTES the plan: one workgroup per 32 x 32 block of each planned lattice computes the block's
//       pixels after the subtraction IN REGISTERS ONLY and publishes the block's new tile record;
//       blocks that also belong to a committed lattice get that pending subtraction first (and
//       written back); pending blocks no new lattice covers are written by workgroups of their own.
//
// Every workgroup derives verification and plan itself from the same 8 KB of records (one record
// per thread, two LDS exchanges, one wave-level sort of <= 16 candidates): no workgroup waits for
// another one, so launches of several channels can share the device in any interleaving.
// Arithmetic and selection are those of the reference, bit for bit: the executable specification is
// oracle/clean_multi_model.py, checked against the restated CleanHost on the CPU.
//
// On the 4096^2 bench image (200 sources, 133 x 111 patch) the plan holds 7.1 of 8 components on
// average and no lattice is evaluated in vain.
#include "kimg_common.h"
#include <limits.h>
#include <string.h>
#include <sched.h>
#include <time.h>
#include <mutex>

namespace {

typedef unsigned long long mkey_t;

constexpr int TILE = 32;                // clean.py:996
constexpr int MC_MAX = 8;               // components per launch
constexpr int MC_THREADS = 256;         // threads of a workgroup = records of a launch
constexpr int MC_REST = 8;              // entries of the keeper's list that the next launch reads
constexpr int MC_CAP = 256;             // tiles the keeper sorts: one per thread
constexpr mkey_t MC_REAL = 1024;        // smaller keys stand for "nothing" (distinct fillers)
constexpr int MC_MAX_TILES = 2047;      // tiles per axis (11 bits each in a key)

// ---- keys ------------------------------------------------------------------------------------
// (metric value, tile, best pixel within the tile) in one word that orders like the reference's
// selection: larger value first (values are non-negative floats, never NaN: a NaN metric never
// replaces a tile's best), then the lower tile index in row-major order; the pixel bits only ride
// along (keys of one tile are never compared).  Everything a component needs but its pixel values.
__device__ inline mkey_t mc_key(float value, int tx, int ty, int yy, int xx)
{
    const unsigned lo = (((unsigned) ~((ty << 11) | tx)) & 0x3FFFFFu) << 10 | (unsigned) (yy << 5 | xx);
    return ((mkey_t) __float_as_uint(value) << 32) | lo;
}

struct mc_cand {
    float value;
    int tx, ty, y, x;
};

__device__ inline mc_cand mc_decode(mkey_t k, int border)
{
    mc_cand c;
    const unsigned lo = (unsigned) k;
    const unsigned t = ~(lo >> 10) & 0x3FFFFFu;
    c.tx = (int) (t & 2047u);
    c.ty = (int) (t >> 11);
    c.value = __uint_as_float((unsigned) (k >> 32));
    if (c.value == 0.0f) {
        // a tile without any positive metric keeps the (x0, y0) start position of clean.py:950,
        // stored the way the reference stores it: best_pos[0] = x0, best_pos[1] = y0
        c.y = c.tx * TILE + border;
        c.x = c.ty * TILE + border;
    } else {
        c.y = c.ty * TILE + border + (int) ((lo >> 5) & 31u);
        c.x = c.tx * TILE + border + (int) (lo & 31u);
    }
    return c;
}

__device__ inline mkey_t kmax(mkey_t a, mkey_t b) { return a > b ? a : b; }

template <int CTRL>
__device__ inline mkey_t kdpp(mkey_t k)         // lanes without a source read 0
{
    const unsigned lo = __builtin_amdgcn_mov_dpp((unsigned) k, CTRL, 0xf, 0xf, true);
    const unsigned hi = __builtin_amdgcn_mov_dpp((unsigned) (k >> 32), CTRL, 0xf, 0xf, true);
    return ((mkey_t) hi << 32) | lo;
}

// maximum over each 16-lane row, in every lane of the row
__device__ inline mkey_t row_max(mkey_t k)
{
    k = kmax(k, kdpp<0xB1>(k));         // quad_perm [1,0,3,2]
    k = kmax(k, kdpp<0x4E>(k));         // quad_perm [2,3,0,1]
    k = kmax(k, kdpp<0x141>(k));        // row_half_mirror
    k = kmax(k, kdpp<0x140>(k));        // row_mirror
    return k;
}

__device__ inline mkey_t lane_key(mkey_t k, int lane)
{
    return ((mkey_t) (unsigned) __builtin_amdgcn_readlane((int) (k >> 32), lane) << 32)
           | (unsigned) __builtin_amdgcn_readlane((int) k, lane);
}

// workgroup barrier that orders LDS traffic only (a __syncthreads() also waits for the wave's
// global stores, which nobody here is waiting for)
__device__ inline void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// number of keys of the 16-lane row that are larger than this lane's (all keys distinct)
template <int N>
__device__ inline int row_rank(mkey_t k)
{
    const mkey_t o = kdpp<0x120 + N>(k);        // row_ror:N
    int r = o > k ? 1 : 0;
    if constexpr (N < 15)
        r += row_rank<N + 1>(k);
    return r;
}

// What the keeper tells the host after every launch: launches done (24 bits), the call's tag (8),
// done (1), components committed (31).
__host__ __device__ inline unsigned long long progress_word(int launches, int gen, bool done, int count)
{
    return ((unsigned long long) ((unsigned) launches & 0xffffffu) << 40)
           | ((unsigned long long) ((unsigned) gen & 0xffu) << 32) | (done ? 0x80000000ull : 0ull)
           | ((unsigned) count & 0x7fffffffu);
}

// ---- state -----------------------------------------------------------------------------------
struct __attribute__((aligned(16))) mc_record {
    mkey_t key;
    float pix[4];           // pixel values at the tile's best pixel
    int pad[2];
};
static_assert(sizeof(mc_record) == 32, "two 16-byte accesses");

struct __attribute__((aligned(16))) mc_state {      // written by the keeper of a launch for the next one
    int count, done, limit;
    float threshold;
    int planned;            // components planned by the launch that wrote this
    int rest_n;             // entries of `rest`
    int tau;                // value bits above which the keeper lists a tile (-1: every tile)
    int launches;
    mkey_t rest_floor;      // every tile outside the planned lattices that is not in `rest` has a key <= this
    int gen;                // the call's tag in the progress word
    int pad;
    mc_record plan[MC_MAX]; // the planned components: key + pixel values at the peak
    mc_record rest[MC_REST];
};

static_assert(sizeof(mc_state) == 560, "Clean.last_launches reads `launches` at fixed offsets");

struct mc_scratch {
    int head[4];            // count, done, limit, threshold bits: what the host reads, as the other forms'
    int pad[12];
    mc_state st[2];         // by launch parity
    mc_record deltas[2][MC_THREADS];    // slot = lattice * seg + block
    // float tile_pix[tiles][4] follows
};

struct mc_geom {
    int64_t row_stride, pol_stride;
    int width, height, P;
    int64_t psf_row_stride, psf_pol_stride;
    int psf_w, psf_h, patch_w, patch_h, border;
    int tiles_x, tiles_y;
    int lat_x, lat_y;       // lattice blocks of a patch
    int seg;                // record slots per lattice: a power of two >= 16
    int mmax;               // components per launch
    float loop_gain;
};

__device__ inline int lat_origin(int pos, int patch, int border)
{
    return (pos - patch / 2 - border) >> 5;      // floor: the lattice extends into the border
}

__device__ inline int lower_tau(int tau)
{
    return tau > 0x00800000 ? tau - 0x00400000 : -1;        // about 0.7 x the value
}

// In-kernel time stamps (test build -DKIMG_MC_STAMPS, tools/exp_clean_multi_stamps.py): shader-clock
// cycles since the workgroup's first instruction, summed over launches, for the keeper (row 0) and
// for the workgroup of block (0, 0) of the first planned lattice (row 1), behind the tile pixels.
#ifdef KIMG_MC_STAMPS
#define MC_STAMP(i) do { dbg_v[i] = clock64() - dbg_t0; } while (0)
#define MC_COUNT(i, v) do { dbg_v[i] += (v); } while (0)
#define MC_FLUSH() do { if (dbg_row >= 0 && threadIdx.x == 0) for (int i_ = 0; i_ < 20; i_++) dbg[dbg_row * 32 + i_] += dbg_v[i_]; } while (0)
#else
#define MC_STAMP(i) do { } while (0)
#define MC_COUNT(i, v) do { } while (0)
#define MC_FLUSH() do { } while (0)
#endif

// ---- the keeper's list -----------------------------------------------------------------------
// The best tiles outside the rectangles `excl` (this launch's lattices), sorted, into next->rest,
// with a floor: every tile outside them that is not listed has a key <= floor.  Values: the base tile
// arrays -- exact for every tile outside the rectangles `flux` (the lattices being folded by this
// launch's folder), whose tiles come from the delta records the threads hold instead.
//   pass A  every thread filters its share of the tile maxima by value alone (bits > tau) and
//           appends the indices that pass to a list in LDS; threads that hold a committed delta
//           above tau append it too.  tau follows the image down from launch to launch so that
//           some tens of tiles pass;
//   pass B  one listed tile per thread: tile coordinates, the rectangle tests, its key;
//   rank    one listed tile per thread: the number of larger keys; the best MC_REST go out with their
//           best pixel and the values there (base arrays or delta table).
// If more than MC_CAP tiles pass, or none survives pass B, tau is moved (bisection, bounded) and the
// passes are repeated: rare.  One workgroup; all threads call.
struct rest_lds {
    mkey_t key[MC_CAP];
    int idx[MC_CAP];            // tile index in the base arrays, or -1 - (slot of the delta record)
    int na, hi, nvalid;
    mkey_t best;
};

__device__ __attribute__((always_inline)) inline void mc_build_rest(
    const float *tile_max, const int32_t *tile_pos, const float *tile_pix, const mc_record *deltas,
    const mc_geom &g, int tau, int nmin, const int (*flux)[2], int nflux, const int (*excl)[2],
    int nexcl, mkey_t dkey, mc_state *next, rest_lds &s, long long *dbg_v, long long dbg_t0)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int nt = g.tiles_x * g.tiles_y;
    constexpr int CH = 8;
    // lane r of every wave keeps rectangle r: the folder's first, then this launch's
    int myrx = INT_MIN / 2, myry = INT_MIN / 2;
    if (lane < nflux) {
        myrx = flux[lane][0];
        myry = flux[lane][1];
    } else if (lane < nflux + nexcl) {
        myrx = excl[lane - nflux][0];
        myry = excl[lane - nflux][1];
    }
    const unsigned flux_mask = (1u << nflux) - 1u;
    const unsigned excl_mask = ((1u << (nflux + nexcl)) - 1u) & ~flux_mask;
    auto rect_hits = [&](int tx, int ty) {
        unsigned hits = 0;
#pragma unroll
        for (int r = 0; r < 2 * MC_MAX; r++) {
            const int rx = __builtin_amdgcn_readlane(myrx, r), ry = __builtin_amdgcn_readlane(myry, r);
            hits |= ((unsigned) (tx - rx) < (unsigned) g.lat_x && (unsigned) (ty - ry) < (unsigned) g.lat_y)
                        ? 1u << r : 0u;
        }
        return hits;
    };
    int lo = -2, up = -2;           // tau known to list too many / known to list nothing (-2: none)
    bool single = false;
    int na = 0;
    for (int attempt = 0; attempt < 100; attempt++) {
        __syncthreads();
        if (tid == 0) {
            s.na = 0;
            s.hi = -1;
            s.nvalid = 0;
            s.best = 0;
        }
        __syncthreads();
        // ---- pass A (single: the tiles whose value is exactly tau + 1, tested at once) -----------
        int my_hi = -1;
        if (!single) {
            for (int base = 0; base < nt; base += 4 * MC_THREADS * CH) {
                float4 v[CH];
#pragma unroll
                for (int c = 0; c < CH; c++) {
                    const int i4 = base + 4 * (tid + MC_THREADS * c);
                    if (i4 + 3 < nt) {
                        v[c] = *reinterpret_cast<const float4 *>(tile_max + i4);
                    } else {
                        v[c].x = i4 < nt ? tile_max[i4] : -1.0f;
                        v[c].y = i4 + 1 < nt ? tile_max[i4 + 1] : -1.0f;
                        v[c].z = i4 + 2 < nt ? tile_max[i4 + 2] : -1.0f;
                        v[c].w = -1.0f;
                    }
                }
#pragma unroll
                for (int c = 0; c < CH; c++) {
                    // (non-negative floats order like their bit patterns; -1 = no tile)
                    const int top = max(max(__float_as_int(v[c].x), __float_as_int(v[c].y)),
                                        max(__float_as_int(v[c].z), __float_as_int(v[c].w)));
                    if (top > tau) {
                        const float e[4] = {v[c].x, v[c].y, v[c].z, v[c].w};
                        for (int k = 0; k < 4; k++) {
                            const int bits = __float_as_int(e[k]);
                            if (bits > tau) {
                                const int slot = atomicAdd(&s.na, 1);
                                if (slot < MC_CAP)
                                    s.idx[slot] = base + 4 * (tid + MC_THREADS * c) + k;
                            }
                        }
                        my_hi = max(my_hi, top);
                    }
                }
            }
            if (dkey >= MC_REAL && (int) (dkey >> 32) > tau) {
                const int slot = atomicAdd(&s.na, 1);
                my_hi = max(my_hi, (int) (dkey >> 32));
                if (slot < MC_CAP) {
                    s.idx[slot] = -1 - tid;
                    s.key[slot] = dkey;
                }
            }
            if (my_hi >= 0)
                atomicMax(&s.hi, my_hi);
        } else {
            for (int t = tid; t < nt; t += MC_THREADS) {
                const float value = tile_max[t];
                if (__float_as_int(value) == tau + 1) {
                    const int ty = t / g.tiles_x, tx = t - ty * g.tiles_x;
                    if (!(rect_hits(tx, ty) & (flux_mask | excl_mask)))
                        atomicMax(&s.best, mc_key(value, tx, ty, 0, 0));
                }
            }
            if (dkey >= MC_REAL && (int) (dkey >> 32) == tau + 1) {
                const mc_cand c = mc_decode(dkey, g.border);
                if (!(rect_hits(c.tx, c.ty) & excl_mask))
                    atomicMax(&s.best, dkey);
            }
        }
        __syncthreads();
        if (single) {
            // the one tile listed: the lowest tile index among those that share the largest value
            const mkey_t best = s.best;
            __syncthreads();
            if (tid == 0) {
                s.na = best >= MC_REAL ? 1 : 0;
                s.key[0] = best;
                const mc_cand c = mc_decode(best, g.border);
                s.idx[0] = c.ty * g.tiles_x + c.tx;
            }
            __syncthreads();
            if (dkey >= MC_REAL && (dkey >> 10) == (best >> 10)) {
                s.idx[0] = -1 - tid;
                s.key[0] = dkey;
            }
            __syncthreads();
            na = s.na;
            MC_COUNT(15, 1);
            break;
        }
        na = s.na;
        const int hi = s.hi;
        MC_COUNT(15, 1);
        if (attempt == 0)
            MC_STAMP(8);
        if (na > MC_CAP) {
            lo = tau;
            if (up >= 0 ? up - lo <= 1 : hi - lo <= 1 || attempt >= 90)
                single = true;          // more than MC_CAP tiles share the largest value
            else
                tau = lo + ((up >= 0 ? up : hi) - lo) / 2;
            continue;
        }
        // ---- pass B --------------------------------------------------------------------------------
        if (tid < na) {
            const int t = s.idx[tid];
            mkey_t key;
            if (t >= 0) {
                const float value = tile_max[t];
                const int ty = t / g.tiles_x, tx = t - ty * g.tiles_x;
                key = (rect_hits(tx, ty) & (flux_mask | excl_mask)) ? 0 : mc_key(value, tx, ty, 0, 0);
            } else {
                key = s.key[tid];
                const mc_cand c = mc_decode(key, g.border);
                if (rect_hits(c.tx, c.ty) & excl_mask)
                    key = 0;
            }
            s.key[tid] = key;
            if (key)
                atomicAdd(&s.nvalid, 1);
        }
        __syncthreads();
        const int nvalid = s.nvalid;
        if ((nvalid == 0 || (nvalid < nmin && lo == -2 && up == -2)) && tau >= 0) {
            if (nvalid == 0)
                up = tau;
            if (lo == -2)
                tau = lower_tau(tau);
            else if (up - lo <= 1)
                single = true, tau = lo;
            else
                tau = lo + (up - lo) / 2;
            continue;
        }
        break;
    }
    MC_STAMP(9);
    // ---- rank and output ----------------------------------------------------------------------------
    const mkey_t k = tid < na ? s.key[tid] : 0;
    int rank = 0;
    for (int i = 0; i < na; i++)
        rank += s.key[i] > k ? 1 : 0;
    const int nvalid = single ? na : s.nvalid;
    MC_COUNT(14, nvalid);
    const mkey_t floor_tau = single ? (na ? (s.key[0] >> 10 << 10) - 1 : 0)
                                    : (tau < 0 ? 0 : ((mkey_t) (unsigned) tau << 32 | 0xffffffffu));
    if (k >= MC_REAL && rank < MC_REST) {
        mc_record r;
        r.key = k;
        r.pad[0] = r.pad[1] = 0;
        const int t = s.idx[tid];
        if (t >= 0) {
            // a tile of the base arrays: its best pixel and the values there
            const mc_cand c = mc_decode(k, g.border);
            const int2 pos = *reinterpret_cast<const int2 *>(tile_pos + 2 * t);
            const float4 px = *reinterpret_cast<const float4 *>(tile_pix + 4 * t);
            if (c.value != 0.0f)
                r.key = mc_key(c.value, c.tx, c.ty, pos.x - (c.ty * TILE + g.border),
                               pos.y - (c.tx * TILE + g.border));
            r.pix[0] = px.x;
            r.pix[1] = px.y;
            r.pix[2] = px.z;
            r.pix[3] = px.w;
        } else {
            const float4 px = *reinterpret_cast<const float4 *>(deltas[-1 - t].pix);
            r.pix[0] = px.x;
            r.pix[1] = px.y;
            r.pix[2] = px.z;
            r.pix[3] = px.w;
        }
        next->rest[rank] = r;
    }
    if (k >= MC_REAL && rank == MC_REST)
        next->rest_floor = k | 0x3ffu;
    if (tid == 0) {
        next->rest_n = nvalid < MC_REST ? nvalid : MC_REST;
        if (nvalid <= MC_REST)
            next->rest_floor = floor_tau;
    }
    // where the filter stands for the next launch: some tens of tiles should pass
    if (tid == 0 && (single || nvalid <= 40))
        next->tau = !single && nvalid < 14 ? lower_tau(tau) : tau;
    if (k >= MC_REAL && nvalid > 40 && rank == 28)
        next->tau = (int) (k >> 32) - 1;
    MC_STAMP(10);
}

// ---- what every workgroup of a launch works out for itself --------------------------------------
struct mc_lds {
    mkey_t row[16], row2[16];           // per 16 records: best key; best of the non-best
    mkey_t plan[MC_MAX];                // the planned components' keys
    float prev_pix[MC_MAX][4];          // pixel values of the previous plan's components
    float new_pix[MC_MAX][4];           // ... of this launch's
    int prev_lat[MC_MAX][2];            // lattice origins (tile coordinates) of the previous plan
    int new_lat[MC_MAX][2];
    int new_pos[MC_MAX][2];             // (y, x) of this launch's components
    int prev_pos[MC_MAX][2];
    mkey_t keys[MC_THREADS / 64];       // block reduction of the pixel phase
};

constexpr int ROLE_KEEPER = 0, ROLE_FOLDER = 1, ROLE_NEW = 2, ROLE_COMMIT = 3;


// One block of pixels: dirty (+ a pending subtraction, written back) (- a planned subtraction, in
// registers), and the block's tile record after it.  256 threads, four pixels each in row-major
// order (first strict maximum in that order, clean.py:953-958).
template <int MODE>
__device__ __attribute__((always_inline)) inline void mc_block(float *dirty, const float *__restrict__ psf, const mc_geom &g,
                                int tx, int ty, bool has_pend, int pend_y, int pend_x,
                                const float *pend_scale, bool has_new, int new_y, int new_x,
                                const float *new_scale, mc_record *out, mc_lds &s,
                                long long *dbg_v, long long dbg_t0)
{
    const int tid = threadIdx.x;
    const int ox = tx * TILE + g.border, oy = ty * TILE + g.border;
    const bool is_tile = tx >= 0 && tx < g.tiles_x && ty >= 0 && ty < g.tiles_y;
    const int ax0 = pend_x - g.patch_w / 2, ay0 = pend_y - g.patch_h / 2;      // clean.py:1024-1027
    const int bx0 = new_x - g.patch_w / 2, by0 = new_y - g.patch_h / 2;
    const int adx = g.psf_w / 2 - pend_x, ady = g.psf_h / 2 - pend_y;          // psf index = image index + d
    const int bdx = g.psf_w / 2 - new_x, bdy = g.psf_h / 2 - new_y;
    float dv[4][4], pa[4][4], pb[4][4];
    bool inside[4], in_a[4], in_b[4];
    const int x = ox + (tid & 31);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int y = oy + (tid >> 5) + 8 * k;
        inside[k] = x >= 0 && x < g.width && y >= 0 && y < g.height;
        in_a[k] = has_pend && inside[k] && x >= ax0 && x < ax0 + g.patch_w && y >= ay0 && y < ay0 + g.patch_h;
        in_b[k] = has_new && inside[k] && x >= bx0 && x < bx0 + g.patch_w && y >= by0 && y < by0 + g.patch_h;
        const int64_t ia = (int64_t) y * g.row_stride + x;
#pragma unroll
        for (int p = 0; p < 4; p++) {
            dv[k][p] = 0.0f;
            pa[k][p] = 0.0f;
            pb[k][p] = 0.0f;
            if (p < g.P) {
                if (inside[k])
                    dv[k][p] = dirty[p * g.pol_stride + ia];
                if (in_a[k])
                    pa[k][p] = psf[p * g.psf_pol_stride + (int64_t) (y + ady) * g.psf_row_stride + (x + adx)];
                if (in_b[k])
                    pb[k][p] = psf[p * g.psf_pol_stride + (int64_t) (y + bdy) * g.psf_row_stride + (x + bdx)];
            }
        }
    }
    float best = 0.0f;
    int best_k = -1;
#ifdef KIMG_MC_STAMPS
    if (dv[0][0] + pa[0][0] + pb[0][0] + dv[3][0] + pb[3][0] == 12345.678f)
        best = 1.0f;                    // (the loads have to complete before the stamp)
    MC_STAMP(6);
#endif
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int y = oy + (tid >> 5) + 8 * k;
        const int64_t ia = (int64_t) y * g.row_stride + x;
        float metric = 0.0f;
#pragma unroll
        for (int p = 0; p < 4; p++) {
            if (p < g.P) {
                if (in_a[k]) {
                    const float t = pend_scale[p] * pa[k][p];      // clean.py:1044-1046: two roundings
                    dv[k][p] -= t;
                    dirty[p * g.pol_stride + ia] = dv[k][p];
                }
                if (in_b[k]) {
                    const float t = new_scale[p] * pb[k][p];
                    dv[k][p] -= t;
                }
                if (MODE == KIMG_CLEAN_I) {
                    if (p == 0)
                        metric = fabsf(dv[k][0]);
                } else {
                    metric += dv[k][p] * dv[k][p];                 // clean.py:962-964
                }
            }
        }
        const bool in_tile = inside[k] && is_tile && x < g.width - g.border && y < g.height - g.border;
        if (in_tile && metric > best) {
            best = metric;
            best_k = k;
        }
    }
    if (!out)
        return;
    if (!is_tile) {
        if (tid == 0) {
            mc_record o;
            o.key = 0;
            o.pix[0] = o.pix[1] = o.pix[2] = o.pix[3] = 0.0f;
            o.pad[0] = o.pad[1] = 0;
            *out = o;
        }
        return;
    }
    // (metric, lowest pixel index first)
    mkey_t k = best_k >= 0 ? ((mkey_t) __float_as_uint(best) << 32) | (unsigned) ~(tid + 256 * best_k) : 0;
    k = row_max(k);
    const mkey_t w = kmax(kmax(lane_key(k, 0), lane_key(k, 16)), kmax(lane_key(k, 32), lane_key(k, 48)));
    if ((tid & 63) == 0)
        s.keys[tid >> 6] = w;
    lds_barrier();
    const mkey_t tb = kmax(kmax(s.keys[0], s.keys[1]), kmax(s.keys[2], s.keys[3]));
    const int widx = ~(int) (unsigned) tb;
    if (tb == 0 ? tid == 0 : tid == (widx & 255)) {
        mc_record o;
        o.pad[0] = o.pad[1] = 0;
        if (tb == 0) {
            // no positive metric: value 0 and the (x0, y0) position of clean.py:950, which the key
            // implies; the pixel there is read when (if ever) this tile wins
            o.key = mc_key(0.0f, tx, ty, 0, 0);
            o.pix[0] = o.pix[1] = o.pix[2] = o.pix[3] = 0.0f;
        } else {
            const int kk = widx >> 8;
            o.key = mc_key(__uint_as_float((unsigned) (tb >> 32)), tx, ty, widx >> 5, widx & 31);
#pragma unroll
            for (int p = 0; p < 4; p++)
                o.pix[p] = kk == 0 ? dv[0][p] : kk == 1 ? dv[1][p] : kk == 2 ? dv[2][p] : dv[3][p];
        }
        *out = o;
    }
    MC_STAMP(7);
}

template <int MODE>
__global__ __launch_bounds__(MC_THREADS) void cycle_multi_kernel(
    float *dirty, float *model, const float *__restrict__ psf, float *tile_max, int32_t *tile_pos,
    mc_geom g, mc_scratch *scratch, int parity, float *log, unsigned long long *progress)
{
    __shared__ mc_lds s;
    __shared__ rest_lds sr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, e = lane & 15;
    // grid = (lat_x, lat_y, 1 + 2 mmax): plane 0 holds the two bookkeeping workgroups, planes
    // 1 .. mmax the blocks of the planned lattices, the rest the blocks of the committed ones
    int role, comp = 0;
    if (blockIdx.z == 0) {
        if (blockIdx.y != 0 || blockIdx.x > 1)
            return;
        role = blockIdx.x == 0 ? ROLE_KEEPER : ROLE_FOLDER;
    } else if ((int) blockIdx.z <= g.mmax) {
        role = ROLE_NEW;
        comp = blockIdx.z - 1;
    } else {
        role = ROLE_COMMIT;
        comp = blockIdx.z - 1 - g.mmax;
    }
    const mc_state *cur = &scratch->st[parity];
    mc_state *next = &scratch->st[parity ^ 1];
    float *tile_pix = reinterpret_cast<float *>(scratch + 1);
#ifdef KIMG_MC_STAMPS
    long long *dbg = reinterpret_cast<long long *>(tile_pix + 4 * (size_t) g.tiles_x * g.tiles_y);
    const long long dbg_t0 = clock64();
    const int dbg_row = role == ROLE_KEEPER ? 0 : (role == ROLE_NEW && comp == 0 && blockIdx.x == 0
                                                    && blockIdx.y == 0) ? 1 : -1;
    long long dbg_v[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    MC_COUNT(16, 1);
#else
    long long *dbg_v = nullptr;
    const long long dbg_t0 = 0;
#endif

    // ---- round trip 1: state, this thread's record, the previous plan and the list ----------
    const int4 st = *reinterpret_cast<const int4 *>(cur);                  // count, done, limit, threshold
    const int4 st2 = *reinterpret_cast<const int4 *>(&cur->planned);       // planned, rest_n, tau, launches
    const int4 st3 = *reinterpret_cast<const int4 *>(&cur->rest_floor);    // floor, gen
    const mkey_t rest_floor = ((mkey_t) (unsigned) st3.y << 32) | (unsigned) st3.x;
    const int gen = st3.z;
    const mc_record *dp = &scratch->deltas[parity][tid];
    const int4 d0 = reinterpret_cast<const int4 *>(dp)[0];
    const int4 d1 = reinterpret_cast<const int4 *>(dp)[1];
    const mc_record *pp = e < MC_MAX ? &cur->plan[e] : &cur->rest[e - MC_MAX];
    const int4 p0 = reinterpret_cast<const int4 *>(pp)[0];
    const int4 p1 = reinterpret_cast<const int4 *>(pp)[1];
    const int count0 = st.x, done = st.y, limit = st.z;
    const float threshold = __int_as_float(st.w);
    const int Mp = st2.x, rest_n = st2.y, tau = st2.z;
    if (done) {
        if (role == ROLE_KEEPER && tid == 0) {
            *reinterpret_cast<int4 *>(next) = st;
            *reinterpret_cast<int4 *>(&next->planned) = make_int4(0, 0, tau, st2.w + 1);
            next->gen = gen;
            if (progress)
                __hip_atomic_store(progress, progress_word(st2.w + 1, gen, true, count0),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    const int shift = 31 - __builtin_clz(g.seg);                // seg is a power of two
    const int li = tid >> shift, lb = tid & (g.seg - 1);
    const int nb = g.lat_x * g.lat_y, q = g.seg >> 4;           // rows of 16 records per lattice
    const mkey_t dkey_raw = ((mkey_t) (unsigned) d0.y << 32) | (unsigned) d0.x;
    const bool live = li < Mp && lb < nb && dkey_raw >= MC_REAL;
    const mkey_t dkey = live ? dkey_raw : 0;
    const float dpix[4] = {__int_as_float(d0.z), __int_as_float(d0.w), __int_as_float(d1.x),
                           __int_as_float(d1.y)};
    const mkey_t pkey = ((mkey_t) (unsigned) p0.y << 32) | (unsigned) p0.x;    // plan[e] or rest[e - 8]
    if (wave == 0 && lane < MC_MAX) {
        s.prev_pix[lane][0] = __int_as_float(p0.z);
        s.prev_pix[lane][1] = __int_as_float(p0.w);
        s.prev_pix[lane][2] = __int_as_float(p1.x);
        s.prev_pix[lane][3] = __int_as_float(p1.y);
        const mc_cand c = mc_decode(pkey, g.border);
        s.prev_lat[lane][0] = lat_origin(c.x, g.patch_w, g.border);
        s.prev_lat[lane][1] = lat_origin(c.y, g.patch_h, g.border);
        s.prev_pos[lane][0] = c.y;
        s.prev_pos[lane][1] = c.x;
    }
    {
        const mkey_t r = row_max(dkey);
        if ((lane & 15) == 0)
            s.row[tid >> 4] = r;
    }
    lds_barrier();
    MC_STAMP(1);

    // ---- verify: the longest prefix of the previous plan that held ------------------------------
    mkey_t a = 0;                       // lane e < 8: best record of lattice e
    for (int r = 0; r < q; r++)
        a = kmax(a, e < MC_MAX && e * q + r < 16 ? s.row[(e * q + r) & 15] : 0);
    mkey_t mine = 0;                    // best record of this thread's lattice
    for (int r = 0; r < q; r++)
        mine = kmax(mine, s.row[(li * q + r) & 15]);
    int j;
    {
        mkey_t pre = a;                 // inclusive prefix maximum over lanes 0 .. e of the row
        pre = kmax(pre, kdpp<0x111>(pre));          // row_shr:1
        pre = kmax(pre, kdpp<0x112>(pre));          // row_shr:2
        pre = kmax(pre, kdpp<0x114>(pre));          // row_shr:4
        const mkey_t before = kdpp<0x111>(pre);     // lattices 0 .. e - 1
        const bool ok = e < Mp && e < MC_MAX && (e == 0 || before < pkey);
        const unsigned held = (unsigned) __builtin_amdgcn_ballot_w64(ok) & 0xffu;
        j = __builtin_ctz(~held);       // leading run of ones
    }
    const int count = count0 + j;
    MC_STAMP(2);
    if (role == ROLE_FOLDER) {
        // the committed records go to the base arrays (nobody in this launch reads them there)
        if (live && li < j) {
            const mc_cand c = mc_decode(dkey, g.border);
            const int t = c.ty * g.tiles_x + c.tx;
            tile_max[t] = c.value;
            *reinterpret_cast<int2 *>(tile_pos + 2 * t) = make_int2(c.y, c.x);
            *reinterpret_cast<float4 *>(tile_pix + 4 * t) = make_float4(dpix[0], dpix[1], dpix[2], dpix[3]);
        }
        return;
    }
    if (role == ROLE_COMMIT && comp >= j)
        return;
    {
        const mkey_t second = (live && li < j && dkey != mine) ? dkey : 0;
        const mkey_t r = row_max(second);
        if ((lane & 15) == 0)
            s.row2[tid >> 4] = r;
    }
    lds_barrier();

    // ---- plan -------------------------------------------------------------------------------------
    MC_STAMP(3);
    const bool mispredicted = j < Mp;
    mkey_t bound;
    {
        const mkey_t r = row_max(s.row2[e]);
        bound = kmax(lane_key(r, 0), rest_floor);
    }
    // the pool, one candidate per lane of a row: the committed lattices' best records and the list
    // (after a misprediction: those records and the first component that was not committed, which
    // was the best tile outside the committed lattices; exactly one component is then planned)
    mkey_t cand;
    if (e < MC_MAX)
        cand = e < j ? a : 0;
    else if (mispredicted)
        cand = e == MC_MAX ? lane_key(pkey, j & 7) : 0;
    else
        cand = e - MC_MAX < rest_n ? pkey : 0;
    if (cand < MC_REAL)
        cand = 1 + e;                   // distinct fillers below every real key
    const int rank = row_rank<1>(cand);
    // sorted: lane p of the row gets the p-th largest
    mkey_t sk;
    {
        const unsigned lo = __builtin_amdgcn_ds_permute(rank << 2, (int) (unsigned) cand);
        const unsigned hi = __builtin_amdgcn_ds_permute(rank << 2, (int) (unsigned) (cand >> 32));
        sk = ((mkey_t) hi << 32) | lo;
    }
    const mc_cand c = mc_decode(sk, g.border);
    const int cbx = lat_origin(c.x, g.patch_w, g.border), cby = lat_origin(c.y, g.patch_h, g.border);
    // Everything the walk below needs, as masks over the sorted positions (bit p = candidate p):
    // per-candidate properties by one compare each, and per pivot k the candidates whose tile lies
    // inside lattice k / whose lattice meets lattice k.  The walk itself is scalar.
    const mkey_t overflow = lane_key(sk, MC_MAX);
    const mkey_t bound2 = kmax(bound, overflow >= MC_REAL ? overflow | 0x3ffu : 0);
    const unsigned m_real = (unsigned) __builtin_amdgcn_ballot_w64(sk >= MC_REAL) & 0xffu;
    const unsigned m_above = (unsigned) __builtin_amdgcn_ballot_w64(sk > bound2) & 0xffu;
    const unsigned m_thr = (unsigned) __builtin_amdgcn_ballot_w64(!(c.value < threshold)) & 0xffu;
    const unsigned m_zero = (unsigned) __builtin_amdgcn_ballot_w64(c.value == 0.0f) & 0xffu;
    unsigned m_ins[MC_MAX], m_ovl[MC_MAX];
#pragma unroll
    for (int k = 0; k < MC_MAX; k++) {
        const int pbx = __builtin_amdgcn_readlane(cbx, k), pby = __builtin_amdgcn_readlane(cby, k);
        m_ins[k] = (unsigned) __builtin_amdgcn_ballot_w64(
            (unsigned) (c.tx - pbx) < (unsigned) g.lat_x && (unsigned) (c.ty - pby) < (unsigned) g.lat_y);
        m_ovl[k] = (unsigned) __builtin_amdgcn_ballot_w64(
            (unsigned) (cbx - pbx + g.lat_x - 1) < (unsigned) (2 * g.lat_x - 1)
            && (unsigned) (cby - pby + g.lat_y - 1) < (unsigned) (2 * g.lat_y - 1));
    }
    unsigned picked = 0, skip = 0, stop = 0;
    int M = 0;
    bool done_now = false, zero_special = false;
    {
        const int mcap = mispredicted ? 1 : g.mmax;
        // (the first candidate is the largest tile of the image if it beats every tile that is not
        // listed; with an entry of the list in the pool it does)
        const bool first_proven = rest_n > 0 || mispredicted || lane_key(sk, 0) > rest_floor;
#pragma unroll
        for (int p = 0; p < MC_MAX; p++) {
            const unsigned bit = 1u << p;
            if (!(m_real & bit))
                break;
            const bool first = M == 0;
            if (first ? !first_proven : !(m_above & bit))
                break;
            if (!(m_thr & bit) || count + M >= limit) {         // clean.py:1065-1066
                done_now = first;
                break;
            }
            if (skip & bit)
                continue;               // inside a planned lattice: its value is about to change
            if (stop & bit)
                break;
            if (m_zero & bit) {
                // its pixel is read from the image, which must be up to date
                if (first && j == 0) {
                    picked |= bit;
                    M = 1;
                    zero_special = true;
                }
                break;
            }
            picked |= bit;
            skip |= m_ins[p];
            stop |= m_ovl[p];
            M++;
            if (M == mcap)
                break;
        }
        if (!(m_real & 1u) && j == 0 && rest_floor == 0)
            done_now = true;            // no tiles at all
    }
    MC_STAMP(4);
    MC_COUNT(17, M);
    MC_COUNT(18, j);
    const int my_m = __builtin_popcount(picked & ((1u << (lane & 31)) - 1u));
    if (wave == 0 && lane < MC_MAX && (picked >> lane & 1u)) {
        s.plan[my_m] = sk;
        s.new_lat[my_m][0] = cbx;
        s.new_lat[my_m][1] = cby;
        s.new_pos[my_m][0] = c.y;
        s.new_pos[my_m][1] = c.x;
    }
    // the pixel values of the planned components: whoever holds a planned record says so
    {
        const mkey_t held = (live && li < j) ? dkey : 0;
        // (after a misprediction lane 8 speaks for plan[j], whose pixel values lane j loaded)
        const mkey_t held2 = wave == 0 && lane >= MC_MAX && lane < 2 * MC_MAX
                                     && (mispredicted ? lane == MC_MAX : lane - MC_MAX < rest_n)
                                 ? (mispredicted ? lane_key(pkey, j & 7) : pkey) : 0;
        float h2[4] = {__int_as_float(p0.z), __int_as_float(p0.w), __int_as_float(p1.x), __int_as_float(p1.y)};
        if (mispredicted) {
#pragma unroll
            for (int p = 0; p < 4; p++)
                h2[p] = s.prev_pix[j & 7][p];
        }
#pragma unroll
        for (int p = 0; p < MC_MAX; p++) {
            const mkey_t kp = lane_key(sk, p);
            const int m = __builtin_popcount(picked & ((1u << p) - 1u));
            if ((picked >> p & 1u) && held == kp) {
#pragma unroll
                for (int i = 0; i < 4; i++)
                    s.new_pix[m][i] = dpix[i];
            }
            if ((picked >> p & 1u) && held2 == kp) {
#pragma unroll
                for (int i = 0; i < 4; i++)
                    s.new_pix[m][i] = h2[i];
            }
        }
    }
    lds_barrier();
    MC_STAMP(5);
    if (zero_special) {
        // a tile without any positive metric won: its record holds the (x0, y0) start position of
        // clean.py:950, whose pixel is read now (nothing is pending: the image is up to date)
        if (tid < 4) {
            const int py = s.new_pos[0][0], px = s.new_pos[0][1];
            const bool ok = py >= 0 && py < g.height && px >= 0 && px < g.width && tid < g.P;
            s.new_pix[0][tid] = ok ? dirty[tid * g.pol_stride + (int64_t) py * g.row_stride + px] : 0.0f;
        }
        lds_barrier();
    }

    if (role == ROLE_NEW) {
        if (comp >= M)
            return;
        const int tx = s.new_lat[comp][0] + (int) blockIdx.x, ty = s.new_lat[comp][1] + (int) blockIdx.y;
        int pend = -1;
        for (int i = 0; i < j; i++)
            if ((unsigned) (tx - s.prev_lat[i][0]) < (unsigned) g.lat_x
                && (unsigned) (ty - s.prev_lat[i][1]) < (unsigned) g.lat_y)
                pend = i;
        float ps[4], ns[4];
#pragma unroll
        for (int p = 0; p < 4; p++) {
            ps[p] = g.loop_gain * s.prev_pix[pend < 0 ? 0 : pend][p];      // clean.py:1044
            ns[p] = g.loop_gain * s.new_pix[comp][p];
        }
        mc_block<MODE>(dirty, psf, g, tx, ty, pend >= 0, s.prev_pos[pend < 0 ? 0 : pend][0],
                       s.prev_pos[pend < 0 ? 0 : pend][1], ps, true, s.new_pos[comp][0],
                       s.new_pos[comp][1], ns,
                       &scratch->deltas[parity ^ 1][comp * g.seg + (int) blockIdx.y * g.lat_x + (int) blockIdx.x], s,
                       dbg_v, dbg_t0);
        MC_FLUSH();
        return;
    }
    if (role == ROLE_COMMIT) {
        const int tx = s.prev_lat[comp][0] + (int) blockIdx.x, ty = s.prev_lat[comp][1] + (int) blockIdx.y;
        for (int m = 0; m < M; m++)
            if ((unsigned) (tx - s.new_lat[m][0]) < (unsigned) g.lat_x
                && (unsigned) (ty - s.new_lat[m][1]) < (unsigned) g.lat_y)
                return;                 // a workgroup of the new lattice writes this block
        float ps[4];
#pragma unroll
        for (int p = 0; p < 4; p++)
            ps[p] = g.loop_gain * s.prev_pix[comp][p];
        mc_block<MODE>(dirty, psf, g, tx, ty, true, s.prev_pos[comp][0], s.prev_pos[comp][1], ps, false,
                       0, 0, ps, nullptr, s, dbg_v, dbg_t0);
        return;
    }

    // ---- keeper: the committed components, the next state, the list for the next launch ---------
    if (tid < j) {
        const int py = s.prev_pos[tid][0], px = s.prev_pos[tid][1];
        float *entry = log + (int64_t) (count0 + tid) * (3 + g.P);
        entry[0] = __uint_as_float((unsigned) (pkey >> 32));    // (lane tid of wave 0 holds plan[tid])
        entry[1] = __int_as_float(py);
        entry[2] = __int_as_float(px);
        for (int p = 0; p < g.P; p++) {
            const float sc = g.loop_gain * s.prev_pix[tid][p];
            float *mp = model + p * g.pol_stride + (int64_t) py * g.row_stride + px;
            entry[3 + p] = sc;
            *mp += sc;                                          // clean.py:1047
        }
    }
    if (tid < M) {
        mc_record r;
        r.key = s.plan[tid];
#pragma unroll
        for (int p = 0; p < 4; p++)
            r.pix[p] = s.new_pix[tid][p];
        r.pad[0] = r.pad[1] = 0;
        next->plan[tid] = r;
    }
    if (tid == 0) {
        *reinterpret_cast<int4 *>(next) = make_int4(count, done_now ? 1 : 0, limit, st.w);
        next->planned = M;
        next->launches = st2.w + 1;
        next->gen = gen;
    }
    // (what the host reads goes out last: the word in host memory is a long way off, and the
    // barriers of the list below would wait for it)
    auto finish = [&]() {
        if (tid == 0) {
            *reinterpret_cast<int4 *>(scratch->head) = make_int4(count, done_now ? 1 : 0, limit, st.w);
            if (progress)
                __hip_atomic_store(progress, progress_word(st2.w + 1, gen, done_now, count),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    };
    MC_STAMP(6);
    if (done_now) {
        if (tid == 0) {
            next->rest_n = 0;
            next->tau = tau;
            next->rest_floor = 0;
        }
        finish();
        return;
    }
    mc_build_rest(tile_max, tile_pos, tile_pix, scratch->deltas[parity], g, tau, 1, s.prev_lat, j,
                  s.new_lat, M, (live && li < j) ? dkey : 0, next, sr, dbg_v, dbg_t0);
    finish();
    MC_STAMP(11);
    MC_FLUSH();
}

// ---- per call ----------------------------------------------------------------------------------
// pixel values at every tile's best pixel (the part of a tile record kimg_update_tiles leaves out)
__global__ __launch_bounds__(256) void mc_tile_pix_kernel(
    const float *__restrict__ dirty, int64_t row_stride, int64_t pol_stride, int width, int height,
    int P, const int32_t *__restrict__ tile_pos, int num_tiles, mc_scratch *scratch)
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

// the state the first launch reads: nothing planned, the list of the best tiles of the whole image
__global__ __launch_bounds__(MC_THREADS) void mc_init_kernel(const float *tile_max,
                                                            const int32_t *tile_pos, mc_geom g,
                                                            mc_scratch *scratch, int limit,
                                                            float threshold, int gen,
                                                            unsigned long long *progress)
{
    __shared__ rest_lds sr;
    __shared__ int s_max;
    const int tid = threadIdx.x;
    mc_state *st = &scratch->st[0];
    const float *tile_pix = reinterpret_cast<const float *>(scratch + 1);
    if (tid == 0)
        s_max = -1;
    __syncthreads();
    int best = -1;
    for (int t = tid; t < g.tiles_x * g.tiles_y; t += MC_THREADS)
        best = max(best, __float_as_int(tile_max[t]));
    atomicMax(&s_max, best);
    __syncthreads();
    if (tid == 0) {
        *reinterpret_cast<int4 *>(st) = make_int4(0, 0, limit, __float_as_int(threshold));
        st->planned = 0;
        st->launches = 0;
        st->gen = gen;
        *reinterpret_cast<int4 *>(scratch->head) = make_int4(0, 0, limit, __float_as_int(threshold));
        scratch->pad[0] = 0x4d554c54;   // "MULT": which form the buffer holds (Clean.last_launches)
        if (progress)
            __hip_atomic_store(progress, progress_word(0, gen, false, 0), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
    }
#ifdef KIMG_MC_STAMPS
    long long dbg_v[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#else
    long long *dbg_v = nullptr;
#endif
    mc_build_rest(tile_max, tile_pos, tile_pix, nullptr, g, lower_tau(s_max), 32, nullptr, 0, nullptr,
                  0, 0, st, sr, dbg_v, 0);
}

// ---- host --------------------------------------------------------------------------------------
struct multi_args {
    float *dirty, *model;
    const float *psf;
    float *tile_max;
    int32_t *tile_pos;
    mc_geom g;
    mc_scratch *scratch;
    float *log;
    unsigned long long *progress;
    int mode;
};

int enqueue_launch(const multi_args &a, hipStream_t s, int parity)
{
    const dim3 grid(a.g.lat_x, a.g.lat_y, 1 + 2 * a.g.mmax);
    if (a.mode == KIMG_CLEAN_I)
        cycle_multi_kernel<KIMG_CLEAN_I><<<grid, MC_THREADS, 0, s>>>(
            a.dirty, a.model, a.psf, a.tile_max, a.tile_pos, a.g, a.scratch, parity, a.log, a.progress);
    else
        cycle_multi_kernel<KIMG_CLEAN_SUMSQ><<<grid, MC_THREADS, 0, s>>>(
            a.dirty, a.model, a.psf, a.tile_max, a.tile_pos, a.g, a.scratch, parity, a.log, a.progress);
    return kimg_launch_status();
}

// hipGraphs of MULTI_GRAPH launches, cached per argument set (as cycles_graph in clean.hip)
constexpr int MULTI_GRAPH = 8;
static_assert(MULTI_GRAPH % 2 == 0, "launches alternate two state buffers");
constexpr int MULTI_CACHE = 32;

struct multi_graph {
    bool valid, used;
    int users, device;
    multi_args args;
    hipGraphExec_t exec;
    hipEvent_t last_use;
};
multi_graph multi_cache[MULTI_CACHE];
std::mutex multi_mutex;

multi_graph *multi_graph_for(const multi_args &a, hipStream_t s)
{
    std::lock_guard<std::mutex> lock(multi_mutex);
    for (int i = 0; i < MULTI_CACHE; i++)
        if (multi_cache[i].valid && memcmp(&multi_cache[i].args, &a, sizeof(a)) == 0) {
            multi_cache[i].users++;
            return &multi_cache[i];
        }
    multi_graph *slot = nullptr;
    for (int i = 0; i < MULTI_CACHE && !slot; i++)
        if (!multi_cache[i].valid)
            slot = &multi_cache[i];
    for (int i = 0; i < MULTI_CACHE && !slot; i++)
        if (multi_cache[i].users == 0
            && (!multi_cache[i].used || hipEventQuery(multi_cache[i].last_use) == hipSuccess))
            slot = &multi_cache[i];
    if (!slot)
        return nullptr;                 // every entry busy: the caller enqueues plain launches
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess)
        return nullptr;
    int rc = 0;
    for (int i = 0; i < MULTI_GRAPH && rc == 0; i++)
        rc = enqueue_launch(a, s, i & 1);
    const hipError_t ended = hipStreamEndCapture(s, &graph);
    if (ended != hipSuccess || rc != 0) {
        if (ended == hipSuccess && graph != nullptr)
            (void) hipGraphDestroy(graph);
        return nullptr;
    }
    hipGraphExec_t exec = nullptr;
    const hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void) hipGraphDestroy(graph);
    if (e != hipSuccess)
        return nullptr;
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

void multi_graph_release(multi_graph *entry, hipStream_t s)
{
    std::lock_guard<std::mutex> lock(multi_mutex);
    (void) hipEventRecord(entry->last_use, s);
    entry->used = true;
    entry->users--;
}

// Progress words the keeper writes for the host: pinned, host-coherent memory, one word (on a
// cache line of its own) per state buffer, so that a call on the same buffers finds the same word
// (the word's address is a kernel argument, hence part of a cached graph's identity).  Launches a
// call left behind on its stream (those enqueued ahead of the one that ended the loop) still write
// their word when the next call has begun: every call tags its words with a generation.
constexpr int PROGRESS_SLOTS = 128;
struct progress_slot {
    const void *owner;
    unsigned gen;
    unsigned long long stamp;
    bool busy;
};
unsigned long long *progress_pool = nullptr, *progress_pool_dev = nullptr;
progress_slot progress_slots[PROGRESS_SLOTS];
unsigned long long progress_clock = 0;
std::mutex progress_mutex;

int progress_acquire(const void *owner, unsigned *gen)
{
    std::lock_guard<std::mutex> lock(progress_mutex);
    if (!progress_pool) {
        void *p = nullptr, *d = nullptr;
        if (hipHostMalloc(&p, PROGRESS_SLOTS * 64, hipHostMallocPortable | hipHostMallocMapped
                                                       | hipHostMallocCoherent) != hipSuccess)
            return -1;
        memset(p, 0xff, PROGRESS_SLOTS * 64);
        if (hipHostGetDevicePointer(&d, p, 0) != hipSuccess)
            d = p;
        progress_pool = static_cast<unsigned long long *>(p);
        progress_pool_dev = static_cast<unsigned long long *>(d);
    }
    int at = -1;
    for (int i = 0; i < PROGRESS_SLOTS && at < 0; i++)
        if (progress_slots[i].owner == owner && !progress_slots[i].busy)
            at = i;
    for (int i = 0; i < PROGRESS_SLOTS && at < 0; i++)
        if (progress_slots[i].owner == owner)
            return -1;                  // (two calls on one state buffer at once)
    if (at < 0)
        for (int i = 0; i < PROGRESS_SLOTS; i++)
            if (!progress_slots[i].busy && (at < 0 || progress_slots[i].stamp < progress_slots[at].stamp))
                at = i;
    if (at < 0)
        return -1;
    progress_slots[at].owner = owner;
    progress_slots[at].busy = true;
    progress_slots[at].stamp = ++progress_clock;
    progress_slots[at].gen = (progress_slots[at].gen + 1) & 0xffu;
    *gen = progress_slots[at].gen;
    return at;
}

void progress_release(int at)
{
    std::lock_guard<std::mutex> lock(progress_mutex);
    progress_slots[at].busy = false;
}

double now_s()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

} // namespace

// Is the multi-component form available for this geometry, and with how many components per launch?
int kimg_clean_multi_components(int patch_width, int patch_height, int tiles_x, int tiles_y)
{
    const int lat_x = kimg_divup(patch_width, TILE) + 1, lat_y = kimg_divup(patch_height, TILE) + 1;
    const int64_t nb = (int64_t) lat_x * lat_y;
    if (nb > MC_THREADS || tiles_x > MC_MAX_TILES || tiles_y > MC_MAX_TILES)
        return 0;
    int seg = 16;
    while (seg < nb)
        seg *= 2;
    const int m = MC_THREADS / seg;
    return m < MC_MAX ? m : MC_MAX;
}

size_t kimg_clean_multi_state_bytes(int tiles_x, int tiles_y)
{
    return sizeof(mc_scratch) + (size_t) tiles_x * tiles_y * 4 * sizeof(float) + 1024;     // (+ stamps of a test build)
}

// The loop of kimg_clean_cycles in this form.  Unlike the other forms it is HOST-PACED: how many
// launches a call needs depends on the data (1 to 8 components each), so the host enqueues them in
// short graphs while it watches the progress word, and the call returns when the loop is done (or
// the last launches it needs are enqueued) -- it blocks for about as long as the loop runs and cannot be
// captured into a caller's graph.
int kimg_clean_multi_run(float *dirty, float *model, int64_t row_stride, int64_t pol_stride,
                         int width, int height, int num_polarizations, const float *psf,
                         int64_t psf_row_stride, int64_t psf_pol_stride, int psf_width,
                         int psf_height, int patch_width, int patch_height, int border, int mode,
                         float loop_gain, float threshold, float *tile_max, int32_t *tile_pos,
                         int tiles_x, int tiles_y, int max_cycles, int components, void *state,
                         float *log, hipStream_t s)
{
    int mmax = kimg_clean_multi_components(patch_width, patch_height, tiles_x, tiles_y);
    if (mmax < 1 || ((uintptr_t) tile_max & 15) || ((uintptr_t) tile_pos & 7))
        return KIMG_EUNSUPPORTED;
    if (components > 0 && components < mmax)
        mmax = components;
    multi_args a;
    memset(&a, 0, sizeof(a));           // padding bytes take part in the cache key comparison
    a.dirty = dirty; a.model = model; a.psf = psf; a.tile_max = tile_max; a.tile_pos = tile_pos;
    a.g.row_stride = row_stride; a.g.pol_stride = pol_stride; a.g.width = width; a.g.height = height;
    a.g.P = num_polarizations; a.g.psf_row_stride = psf_row_stride; a.g.psf_pol_stride = psf_pol_stride;
    a.g.psf_w = psf_width; a.g.psf_h = psf_height; a.g.patch_w = patch_width; a.g.patch_h = patch_height;
    a.g.border = border; a.g.tiles_x = tiles_x; a.g.tiles_y = tiles_y;
    a.g.lat_x = kimg_divup(patch_width, TILE) + 1; a.g.lat_y = kimg_divup(patch_height, TILE) + 1;
    a.g.seg = 16;
    while (a.g.seg < a.g.lat_x * a.g.lat_y)
        a.g.seg *= 2;
    a.g.mmax = mmax; a.g.loop_gain = loop_gain;
    a.scratch = static_cast<mc_scratch *>(state); a.log = log; a.mode = mode;
    unsigned gen = 0;
    const int slot = progress_acquire(state, &gen);
    if (slot < 0)
        return KIMG_EUNSUPPORTED;
    a.progress = progress_pool_dev + 8 * slot;
    volatile unsigned long long *seen = progress_pool + 8 * slot;
    int rc = 0;
    hipError_t he = hipMemsetAsync(state, 0, sizeof(mc_scratch), s);
    if (he != hipSuccess)
        rc = -(int) he;
#ifdef KIMG_MC_STAMPS
    (void) hipMemsetAsync(reinterpret_cast<char *>(state) + sizeof(mc_scratch)
                              + (size_t) tiles_x * tiles_y * 4 * sizeof(float), 0, 1024, s);
#endif
    if (rc == 0) {
        mc_tile_pix_kernel<<<kimg_divup(tiles_x * tiles_y, 256), 256, 0, s>>>(
            dirty, row_stride, pol_stride, width, height, num_polarizations, tile_pos,
            tiles_x * tiles_y, a.scratch);
        mc_init_kernel<<<1, MC_THREADS, 0, s>>>(tile_max, tile_pos, a.g, a.scratch, max_cycles,
                                                threshold, (int) gen, a.progress);
        rc = kimg_launch_status();
    }
    // Pace: keep the device a graph or two ahead of what has been seen to complete, and stop when
    // the keeper says done.  Near the end the number of launches still needed is estimated from the
    // components per launch so far, so that few launches run after the loop has ended (each costs a
    // kernel boundary).
    multi_graph *graph = nullptr;
    int enqueued = 0;
    const double t_start = now_s();
    double t_progress = t_start;
    unsigned long long last = ~0ull;
    while (rc == 0) {
        const unsigned long long word = *seen;
        if (word != last) {
            last = word;
            t_progress = now_s();
        }
        const bool started = ((word >> 32) & 0xffu) == gen;     // (else: a word of an earlier call)
        const int launches = started ? (int) (word >> 40) : 0;
        const int count = started ? (int) (word & 0x7fffffffu) : 0;
        if (started && (word & 0x80000000u))
            break;
        const int in_flight = enqueued - launches;
        // components per launch so far (at least 1, optimistic before anything is known)
        const double per = launches > 0 && count > 0 ? (double) count / launches : (double) mmax;
        int need = (int) ((max_cycles - count) / per) + 2 - in_flight;
        if (need > 2 * MULTI_GRAPH - in_flight)
            need = 2 * MULTI_GRAPH - in_flight;
        if (in_flight == 0 && need < 2)
            need = 2;
        if (need >= MULTI_GRAPH && !graph)
            graph = multi_graph_for(a, s);
        if (need >= MULTI_GRAPH && graph) {
            he = hipGraphLaunch(graph->exec, s);
            if (he != hipSuccess)
                rc = -(int) he;
            enqueued += MULTI_GRAPH;
            continue;
        }
        if (need >= 2 || (need > 0 && in_flight == 0)) {
            for (int i = 0; i < 2 && rc == 0; i++)
                rc = enqueue_launch(a, s, i);
            enqueued += 2;
            continue;
        }
        // nothing to enqueue: wait for the device
        if (now_s() - t_progress > 2e-3 && hipStreamQuery(s) == hipSuccess) {
            // the stream is idle but the word has not moved: read the state itself
            int head[8];
            mc_state st0;
            he = hipMemcpy(head, a.scratch->head, sizeof(int) * 4, hipMemcpyDeviceToHost);
            if (he == hipSuccess)
                he = hipMemcpy(&st0, &a.scratch->st[enqueued & 1], 32, hipMemcpyDeviceToHost);
            if (he != hipSuccess) {
                rc = -(int) he;
                break;
            }
            *seen = progress_word(st0.launches, (int) gen, st0.done != 0, st0.count);
            if (now_s() - t_progress > 30.0) {
                rc = KIMG_ETIMEOUT;
                break;
            }
            continue;
        }
        sched_yield();
    }
    if (graph)
        multi_graph_release(graph, s);
    progress_release(slot);
    return rc;
}
