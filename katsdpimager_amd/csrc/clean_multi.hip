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
//
// REPEATED STEPS.  While one source stands far above the rest every cycle goes to the same peak,
// and the lattices of a launch are all the same one.  But a subtraction changes the value at its own
// peak by a scalar recursion -- v <- v - fl(fl(gain v) psf_centre), per polarization -- that every
// workgroup can evaluate for itself, so a planned lattice may carry SEVERAL steps: the plan is the
// merge of the planned lattices' decreasing sequences kappa_i(0) > kappa_i(1) > ..., cut at a level
// below which nothing is proven (the bound of the walk, the first candidate that is neither planned
// nor inside a planned lattice, the value a peak has when it may not be stepped again).  A block
// applies its lattice's steps one after the other in registers and notes, per step, whether any of
// its pixels comes before the peak in the reference's order by then ("fail" bits in its record);
// its record is the tile after all steps.  Verification: lattice i allows the steps above
// L_i = kappa_i(k - 1) - 1 if a block failed after k steps, else above min(best record of the
// lattice, kappa_i(r_i - 1) - 1); the steps above max L_i are committed -- a prefix of the merged
// order, which nobody has to sort but the keeper for its log.  A lattice of which only some steps
// were committed has pixels to write but no valid records: the next launch plans exactly those
// lattices again WITHOUT steps (its workgroups write the pending pixels and publish the records),
// and the plans take single steps for a while after that (8 launches, doubling up to 64 while it
// keeps happening), so that a field where the repeated steps do not hold (a peak whose neighbours
// overtake it) costs a few per cent, not a launch per component.
// The second-best records of a committed lattice bound a proof -- except where the lattice's best
// record is planned again with the very same lattice (the common case of a repeated peak): they
// then lie inside a planned lattice and the next verification covers them.
// All of that costs a launch 1.5 us of instruction issue, so two instances of the kernel are built
// (STEPS = 1 and 8; 4 with several polarizations) and the host picks: single steps, unless the
// start-up kernel's estimate (how many steps the eight best tiles could take before they are down
// to the ninth) or the components per launch it sees say that the field is a dominated one.
#include "kimg_common.h"
#include <limits.h>
#include <string.h>
#include <sched.h>
#include <time.h>
#include <mutex>

namespace {

typedef unsigned long long mkey_t;

constexpr int TILE = 32;                // clean.py:996
constexpr int MC_MAX = 8;               // lattices per launch
constexpr int MC_STEPS = 8;             // subtractions at one peak per launch (4 with several polarizations)
constexpr int MC_COOL = 4;              // launches with single steps after repeated ones went wrong: twice this, ...
constexpr int MC_COOL_MAX = 64;         // ... doubled each time in a row up to this
constexpr int MC_THREADS = 256;         // threads of a workgroup = records of a launch
constexpr int MC_REST = 24;             // entries of the list (lanes 8 .. 31 of a wave load one each)
constexpr int MC_POOL_REST = 8;         // of which the first that survive enter the candidate pool
constexpr int MC_TOP = 192;             // entries of the list the lister maintains
constexpr int MC_LOW = 24;              // fewer than this left: the lister scans the tile maxima again
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
    // (the pixel's five bits each way are cut to size: a position that is not its tile's -- tile arrays
    // somebody else wrote -- must not reach into the tile's bits and send a record to a tile that
    // does not exist)
    const unsigned lo = (((unsigned) ~((ty << 11) | tx)) & 0x3FFFFFu) << 10 | (unsigned) ((yy & 31) << 5 | (xx & 31));
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
    int top_n;              // entries of `top`
    int tau;                // value bits above which the last scan listed a tile (-1: every tile)
    int launches;
    mkey_t rest_floor;      // for a reader of the first MC_REST entries: every other tile has a key <= this
    int gen;                // the call's tag in the progress word
    int aux;                // most steps of a planned lattice | launches of single steps left << 8 | penalty << 16
    mkey_t top_floor;       // every tile that is not in `top` has a key <= this
    int repeated;           // steps committed so far beyond the first of their lattice and launch
    int pad2;
    mc_record plan[MC_MAX]; // the planned components: key + pixel values at the peak
    mc_record top[MC_TOP];  // the best tiles of the image, sorted: key + pixel values at the tile's peak
};

static_assert(sizeof(mc_state) == 6464, "Clean.last_launches reads `launches` at fixed offsets");

struct mc_scratch {
    int head[4];            // count, done, limit, threshold bits: what the host reads, as the other forms'
    int pad[12];            // [0] "MULT"; [1] the threshold follows from the call's first component;
                            // [2..3], [4..5] two doubles: the noise threshold and the share of the first
                            // peak that is left for the next major cycle (kimg_clean_major_cycles)
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
    int mmax;               // lattices per launch
    int rmax;               // steps per lattice and launch
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

// ---- the lister's list ----------------------------------------------------------------------
// The best tiles of the image as the commits so far leave it, sorted, into next->top, with a
// floor: every tile that is not listed has a key <= floor.  (The list does not know this launch's
// plan: the next launch drops the entries inside this launch's lattices itself.)
//
// From launch to launch the list is MAINTAINED, not rebuilt: the tiles whose values changed are
// exactly those of the lattices the folder is writing, so the entries inside those go, the delta
// records above the floor come in, everything else stands, and the floor holds.  One workgroup
// reading all the tile maxima in every launch was the longest chain of the launch (60 KB through one
// CU's memory pipe: 3 us).  Only when fewer than MC_LOW entries are left (the image has been
// cleaned down to the floor) are the tile maxima scanned again, with a lower floor:
//   pass A  every thread filters its share of the tile maxima by value alone (bits > tau) and
//           appends the indices that pass to a list in LDS (values: the base arrays -- exact
//           outside the lattices being folded, whose tiles come from the delta records instead);
//   pass B  one listed tile per thread: tile coordinates, the rectangle tests, its key.
//   If more than MC_CAP tiles pass, or none survives pass B, tau is moved (bisection, bounded)
//   and the passes are repeated.
// Then, either way, one listed tile per thread: its rank = the number of larger keys; the best
// MC_TOP go out with their best pixel and the values there.  One workgroup; all threads call.
struct rest_lds {
    __attribute__((aligned(16))) mkey_t key[MC_CAP];
    int idx[MC_CAP];            // tile index in the base arrays, -1 - (slot of the delta record), or
                                // INT_MIN: an entry of the old list, pixel values in `pix`
    float pix[MC_CAP][4];
    int na, hi, nvalid;
    mkey_t best;
};

__device__ __attribute__((always_inline)) inline void mc_build_rest(
    const float *tile_max, const int32_t *tile_pos, const float *tile_pix, const mc_record *deltas,
    const mc_geom &g, int tau, int nmin, int myrx, int myry, int nflux, mkey_t dkey, float4 dpix,
    bool have_old, int4 old0, int4 old1, int old_n, mkey_t old_floor, mc_state *next, rest_lds &s,
    long long *dbg_v, long long dbg_t0)
{
    // myrx, myry: lane r < nflux of every wave holds the tile origin of rectangle r (the lattices
    // the folder is writing: their tiles come from the delta records, `dkey`)
    const int tid = threadIdx.x;
    const int nt = g.tiles_x * g.tiles_y;
    constexpr int CH = 8;
    const unsigned flux_mask = (1u << nflux) - 1u;
    auto rect_hits = [&](int tx, int ty) {
        unsigned hits = 0;
#pragma unroll
        for (int r = 0; r < MC_MAX; r++) {
            const int rx = __builtin_amdgcn_readlane(myrx, r), ry = __builtin_amdgcn_readlane(myry, r);
            hits |= ((unsigned) (tx - rx) < (unsigned) g.lat_x && (unsigned) (ty - ry) < (unsigned) g.lat_y)
                        ? 1u << r : 0u;
        }
        return hits & flux_mask;
    };
    int lo = -2, up = -2;           // tau known to list too many / known to list nothing (-2: none)
    bool single = false;
    int na = 0;
    bool scan = !have_old;
    if (!scan) {
        // ---- the old list without the lattices being folded, plus their delta records ----------
        __syncthreads();
        if (tid == 0) {
            s.na = 0;
            s.nvalid = 0;
        }
        // (this thread's entry of the old list was fetched with the launch's first loads)
        mc_record o;
        o.key = tid < old_n ? ((mkey_t) (unsigned) old0.y << 32) | (unsigned) old0.x : 0;
        o.pix[0] = __int_as_float(old0.z);
        o.pix[1] = __int_as_float(old0.w);
        o.pix[2] = __int_as_float(old1.x);
        o.pix[3] = __int_as_float(old1.y);
        const mc_cand oc = mc_decode(o.key, g.border);
        const bool keep = o.key >= MC_REAL && !rect_hits(oc.tx, oc.ty);
        const bool fresh = dkey >= MC_REAL && dkey > old_floor;
        // The old list is sorted, and few entries change: an old entry's new rank is its old one
        // minus the entries dropped before it plus the fresh ones above it; a fresh one finds its
        // place among the old by bisection.  (Ranking everything against everything was the longest
        // step of the launch.)  LDS: key[t] = old key t as it was, idx[t] = entries dropped before
        // t, key[MC_TOP + f] = fresh key f.
        const bool dropped = tid < old_n && !keep;
        const unsigned long long drops = __builtin_amdgcn_ballot_w64(dropped);
        const int wave = tid >> 6, lane = tid & 63;
        __shared__ int s_wave_drops[MC_THREADS / 64];
        if (lane == 0)
            s_wave_drops[wave] = __builtin_popcountll(drops);
        if (tid < MC_TOP)
            s.key[tid] = tid < old_n ? o.key : 0;
        __syncthreads();
        int before = __builtin_popcountll(drops & ((1ull << lane) - 1ull));
        for (int w = 0; w < wave; w++)
            before += s_wave_drops[w];
        if (tid < MC_TOP)
            s.idx[tid] = before;
        int fslot = -1;
        if (fresh) {
            fslot = atomicAdd(&s.na, 1);
            if (fslot < MC_CAP - MC_TOP)
                s.key[MC_TOP + fslot] = dkey;
        }
        __syncthreads();
        const int nf = s.na;
        int total_drops = 0;
        for (int w = 0; w < MC_THREADS / 64; w++)
            total_drops += s_wave_drops[w];
        na = old_n - total_drops + nf;
        MC_STAMP(12);
        // (more fresh entries than there is room to rank, or too few entries left: scan)
        scan = nf > MC_CAP - MC_TOP || (na < MC_LOW && old_floor != 0);
#ifdef KIMG_MC_STAMPS
        MC_COUNT(5, nf);
        MC_COUNT(6, na);
        MC_COUNT(7, old_n);
        MC_COUNT(17, (long long) __uint_as_float((unsigned) (old_floor >> 32)) * 1000);
        MC_COUNT(18, nflux);
#endif
        if (!scan) {
            auto fresh_above = [&](mkey_t k) {
                int c = 0;
                for (int f = 0; f < nf; f++)
                    c += s.key[MC_TOP + f] > k ? 1 : 0;
                return c;
            };
            auto put = [&](int rank, mkey_t k, const float *pix) {
                if (rank < MC_TOP) {
                    mc_record r;
                    r.key = k;
#pragma unroll
                    for (int p = 0; p < 4; p++)
                        r.pix[p] = pix[p];
                    r.pad[0] = r.pad[1] = 0;
                    next->top[rank] = r;
                }
                // (an entry that is dropped is above everything that stays out)
                if (rank == MC_TOP)
                    next->top_floor = k | 0x3ffu;
                if (rank == MC_REST)
                    next->rest_floor = k | 0x3ffu;
            };
            if (keep)
                put(tid - before + fresh_above(o.key), o.key, o.pix);
            if (fresh) {
                // old keys above this one: the list is sorted, largest first
                int lo_i = 0, hi_i = old_n;
                while (lo_i < hi_i) {
                    const int mid = (lo_i + hi_i) >> 1;
                    if (s.key[mid] > dkey)
                        lo_i = mid + 1;
                    else
                        hi_i = mid;
                }
                const int dropped_above = lo_i < old_n ? s.idx[lo_i] : total_drops;
                const float fp[4] = {dpix.x, dpix.y, dpix.z, dpix.w};
                put(lo_i - dropped_above + fresh_above(dkey), dkey, fp);
            }
            if (tid == 0) {
                next->top_n = na < MC_TOP ? na : MC_TOP;
                if (na <= MC_TOP)
                    next->top_floor = old_floor;
                if (na <= MC_REST)
                    next->rest_floor = old_floor;
                next->tau = tau;
            }
            MC_STAMP(9);
            MC_COUNT(14, na);
            MC_STAMP(10);
            return;
        }
        __syncthreads();
        if (tid == 0)
            s.na = 0;
        if (scan) {
            // (about 0.7 of the floor so far; the first answer with enough entries to last a while is
            // taken: every attempt is a pass over the tile maxima, 60 KB through one CU's memory
            // pipe.  A deep CLEAN brings ever more sources down to one common level, just above
            // the floor of a list that holds MC_TOP of them: the list then loses what a launch
            // commits and is rebuilt every (MC_TOP - MC_LOW) / 8 launches -- 7 % of the launches
            // of the bench image's 1000 cycles, which take 26 us instead of 8.)
            tau = lower_tau(tau);
            nmin = 64;
        }
    }
    for (int attempt = 0; scan && attempt < 100; attempt++) {
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
                    // (all loads of a round in flight together: no branches here; a group that
                    // reaches past the end is read from the last four tiles instead)
                    const int i4 = base + 4 * (tid + MC_THREADS * c);
                    v[c] = *reinterpret_cast<const float4 *>(tile_max + min(i4, nt - 4));
                }
                // which of this round's 32 tiles pass: one bit each, then ONE reservation of list
                // slots per thread (an LDS atomic per tile would be a round trip per tile)
                unsigned pass = 0;
#pragma unroll
                for (int c = 0; c < CH; c++) {
                    const int i4 = base + 4 * (tid + MC_THREADS * c), b4 = min(i4, nt - 4);
                    // (non-negative floats order like their bit patterns; -1 = not this group's tile)
                    const int e[4] = {b4 >= i4 ? __float_as_int(v[c].x) : -1,
                                      b4 + 1 >= i4 ? __float_as_int(v[c].y) : -1,
                                      b4 + 2 >= i4 ? __float_as_int(v[c].z) : -1,
                                      b4 + 3 >= i4 ? __float_as_int(v[c].w) : -1};
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        pass |= e[k] > tau ? 1u << (4 * c + k) : 0u;
                    my_hi = max(my_hi, max(max(e[0], e[1]), max(e[2], e[3])));
                }
                if (pass) {
                    int slot = atomicAdd(&s.na, __builtin_popcount(pass));
                    while (pass) {
                        const int bit = __builtin_ctz(pass);
                        pass &= pass - 1u;
                        const int c = bit >> 2, i4 = base + 4 * (tid + MC_THREADS * c);
                        if (slot < MC_CAP)
                            s.idx[slot] = min(i4, nt - 4) + (bit & 3);
                        slot++;
                    }
                }
            }
            if (attempt == 0)
                MC_STAMP(13);
            if (dkey >= MC_REAL && (int) (dkey >> 32) > tau) {
                const int slot = atomicAdd(&s.na, 1);
                my_hi = max(my_hi, (int) (dkey >> 32));
                if (slot < MC_CAP) {
                    s.idx[slot] = -1 - tid;
                    s.key[slot] = dkey;
                }
            }
            if (my_hi > tau)
                atomicMax(&s.hi, my_hi);
        } else {
            for (int t = tid; t < nt; t += MC_THREADS) {
                const float value = tile_max[t];
                if (__float_as_int(value) == tau + 1) {
                    const int ty = t / g.tiles_x, tx = t - ty * g.tiles_x;
                    if (!rect_hits(tx, ty))
                        atomicMax(&s.best, mc_key(value, tx, ty, 0, 0));
                }
            }
            if (dkey >= MC_REAL && (int) (dkey >> 32) == tau + 1)
                atomicMax(&s.best, dkey);
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
                key = rect_hits(tx, ty) ? 0 : mc_key(value, tx, ty, 0, 0);
            } else {
                key = s.key[tid];
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
    MC_COUNT(19, scan ? 1 : 0);
    // ---- rank and output ----------------------------------------------------------------------------
    const mkey_t k = tid < na ? s.key[tid] : 0;
    int rank = 0;
    {
        // (two keys per LDS read; a key past the end of the list never counts)
        const ulonglong2 *pairs = reinterpret_cast<const ulonglong2 *>(s.key);
#pragma unroll 4
        for (int i = 0; i < na; i += 2) {
            const ulonglong2 two = pairs[i >> 1];
            rank += two.x > k ? 1 : 0;
            rank += i + 1 < na && two.y > k ? 1 : 0;
        }
    }
    const int nvalid = !scan || single ? na : s.nvalid;
    MC_COUNT(14, nvalid);
    // the floor of what was gathered: the old one if the list was only maintained, else the scan's
    const mkey_t gathered = !scan ? old_floor
                            : single ? (na ? (s.key[0] >> 10 << 10) - 1 : ((mkey_t) (unsigned) (tau + 1) << 32 | 0xffffffffu))
                            : (tau < 0 ? 0 : ((mkey_t) (unsigned) tau << 32 | 0xffffffffu));
    if (k >= MC_REAL && rank < MC_TOP) {
        mc_record r;
        r.key = k;
        r.pad[0] = r.pad[1] = 0;
        const int t = s.idx[tid];
        float4 px;
        if (t >= 0) {
            // a tile of the base arrays: its best pixel and the values there
            const mc_cand c = mc_decode(k, g.border);
            const int2 pos = *reinterpret_cast<const int2 *>(tile_pos + 2 * t);
            px = *reinterpret_cast<const float4 *>(tile_pix + 4 * t);
            if (c.value != 0.0f)
                r.key = mc_key(c.value, c.tx, c.ty, pos.x - (c.ty * TILE + g.border),
                               pos.y - (c.tx * TILE + g.border));
        } else if (t != INT_MIN) {
            px = *reinterpret_cast<const float4 *>(deltas[-1 - t].pix);
        } else {
            px = *reinterpret_cast<const float4 *>(s.pix[tid]);
        }
        r.pix[0] = px.x;
        r.pix[1] = px.y;
        r.pix[2] = px.z;
        r.pix[3] = px.w;
        next->top[rank] = r;
    }
    // (an entry that is dropped is above everything that stays out)
    if (k >= MC_REAL && rank == MC_TOP)
        next->top_floor = k | 0x3ffu;
    if (k >= MC_REAL && rank == MC_REST)
        next->rest_floor = k | 0x3ffu;
    if (tid == 0) {
        next->top_n = nvalid < MC_TOP ? nvalid : MC_TOP;
        if (nvalid <= MC_TOP)
            next->top_floor = gathered;
        if (nvalid <= MC_REST)
            next->rest_floor = gathered;
        next->tau = tau;
    }
    MC_STAMP(10);
}

// ---- what every workgroup of a launch works out for itself --------------------------------------
struct mc_lds {
    mkey_t row[16], row2[16];           // per 16 records: best key; best of the non-best
    unsigned rowf[16];                  // per 16 records: the fail bits of their blocks
    unsigned fails[MC_THREADS / 64];    // block reduction of the pixel phase: fail bits
    __attribute__((aligned(16))) mkey_t seqk[MC_MAX * MC_STEPS];   // keeper: the committed steps' keys
    float seqp[MC_MAX * MC_STEPS][4];   // ... and the pixel values at the peak before each step
    float row_pix[16][4];               // pixel values of a row's best record
    mc_record pool[MC_THREADS / 64][MC_POOL_REST];      // per wave: the list's surviving entries
    mkey_t keys[MC_THREADS / 64];       // block reduction of the pixel phase
};

constexpr int ROLE_KEEPER = 0, ROLE_FOLDER = 1, ROLE_LISTER = 2, ROLE_NEW = 3, ROLE_COMMIT = 4;


// One subtraction at a peak, as the lattice workgroups compute it for that pixel (clean.py:1044-1046:
// scale = gain * value, separately rounded product with the PSF's centre): v becomes the values
// after it; returns the metric there.
template <int MODE, int PMAX>
__device__ inline float mc_peak_step(float (&v)[PMAX], const float (&centre)[PMAX], float gain, int P)
{
    float metric = 0.0f;
#pragma unroll
    for (int p = 0; p < PMAX; p++) {
        if (PMAX == 1 || p < P) {
            const float scale = gain * v[p];
            const float t = scale * centre[p];
            v[p] -= t;
            if (MODE == KIMG_CLEAN_I) {
                if (p == 0)
                    metric = fabsf(v[0]);
            } else {
                metric += v[p] * v[p];
            }
        }
    }
    return metric;
}

// One block of pixels: dirty (+ the pending subtractions of a committed lattice, written back) (- the
// steps of a planned lattice, in registers), and the block's tile record after them.  256 threads,
// four pixels each in row-major order (first strict maximum in that order, clean.py:953-958).
// Between the steps of the planned lattice every pixel is held against the value the peak has by
// then (`peak_bits`, metric bits after k steps): a pixel that comes before the peak in the
// reference's order (larger metric, lower tile, earlier pixel) sets fail bit k of the record.
// PMAX: 1 when there is one polarization (no loop over planes, a third of the instructions), else 4.
template <int MODE, int PMAX, int STEPS>
__device__ __attribute__((always_inline)) inline void mc_block(float *dirty, const float *__restrict__ psf, const mc_geom &g,
                                int tx, int ty, int pend_n, int pend_y, int pend_x,
                                const float (&pend_scale)[STEPS][PMAX], bool has_new, int new_n, int new_y,
                                int new_x, const float (&new_scale)[STEPS][PMAX],
                                const unsigned (&peak_bits)[STEPS], int peak_tile, int peak_idx,
                                mc_record *out, mc_lds &s, long long *dbg_v, long long dbg_t0)
{
    const int tid = threadIdx.x;
    const int ox = tx * TILE + g.border, oy = ty * TILE + g.border;
    const bool is_tile = tx >= 0 && tx < g.tiles_x && ty >= 0 && ty < g.tiles_y;
    const bool has_pend = pend_n > 0;
    const int ax0 = pend_x - g.patch_w / 2, ay0 = pend_y - g.patch_h / 2;      // clean.py:1024-1027
    const int bx0 = new_x - g.patch_w / 2, by0 = new_y - g.patch_h / 2;
    const int adx = g.psf_w / 2 - pend_x, ady = g.psf_h / 2 - pend_y;          // psf index = image index + d
    const int bdx = g.psf_w / 2 - new_x, bdy = g.psf_h / 2 - new_y;
    float dv[4][PMAX], pa[4][PMAX], pb[4][PMAX];
    bool inside[4], in_a[4], in_b[4];
    const int x = ox + (tid & 31);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int y = oy + (tid >> 5) + 8 * k;
        inside[k] = x >= 0 && x < g.width && y >= 0 && y < g.height;
        in_a[k] = has_pend && inside[k] && x >= ax0 && x < ax0 + g.patch_w && y >= ay0 && y < ay0 + g.patch_h;
        in_b[k] = has_new && new_n > 0 && inside[k] && x >= bx0 && x < bx0 + g.patch_w && y >= by0
                  && y < by0 + g.patch_h;
        const int64_t ia = (int64_t) y * g.row_stride + x;
#pragma unroll
        for (int p = 0; p < PMAX; p++) {
            dv[k][p] = 0.0f;
            pa[k][p] = 0.0f;
            pb[k][p] = 0.0f;
            if (PMAX == 1 || p < g.P) {
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
    // the pending steps, one after the other (each was a cycle of its own: clean.py:1044-1046, two
    // roundings per step), written back
#pragma unroll
    for (int st = 0; st < STEPS; st++) {
        if (st < pend_n) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
#pragma unroll
                for (int p = 0; p < PMAX; p++) {
                    if ((PMAX == 1 || p < g.P) && in_a[k]) {
                        const float t = pend_scale[st][p] * pa[k][p];
                        dv[k][p] -= t;
                    }
                }
            }
        }
    }
    if (has_pend) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int y = oy + (tid >> 5) + 8 * k;
            const int64_t ia = (int64_t) y * g.row_stride + x;
#pragma unroll
            for (int p = 0; p < PMAX; p++)
                if ((PMAX == 1 || p < g.P) && in_a[k])
                    dirty[p * g.pol_stride + ia] = dv[k][p];
        }
    }
    // the planned steps, in registers; between them, is the peak still the first pixel of its lattice?
    // A pixel comes before the peak if its metric is larger, or equal and it is the earlier one (lower
    // tile, earlier pixel of the same tile): metric bits + 1 for those, against the peak's bits
    // (non-negative floats order like their bit patterns); pixels that are no tile's count as 0.
    unsigned fail = 0;
    unsigned keep[4], early[4];
    if (STEPS > 1 && new_n > 1) {
        const int t = ty * g.tiles_x + tx;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int y = oy + (tid >> 5) + 8 * k;
            const bool in_tile = inside[k] && is_tile && x < g.width - g.border && y < g.height - g.border;
            keep[k] = in_tile ? 0xffffffffu : 0u;
            early[k] = (in_tile && (t < peak_tile || (t == peak_tile && tid + 256 * k < peak_idx))) ? 1u : 0u;
        }
    }
#pragma unroll
    for (int st = 0; st < STEPS; st++) {
        if (st < new_n) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
#pragma unroll
                for (int p = 0; p < PMAX; p++) {
                    if ((PMAX == 1 || p < g.P) && in_b[k]) {
                        const float t = new_scale[st][p] * pb[k][p];
                        dv[k][p] -= t;
                    }
                }
            }
            if (STEPS > 1 && st + 1 < new_n) {
                bool beaten = false;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    float metric = 0.0f;
#pragma unroll
                    for (int p = 0; p < PMAX; p++) {
                        if (PMAX == 1 || p < g.P) {
                            if (MODE == KIMG_CLEAN_I) {
                                if (p == 0)
                                    metric = fabsf(dv[k][0]);
                            } else {
                                metric += dv[k][p] * dv[k][p];
                            }
                        }
                    }
                    beaten = beaten || (__float_as_uint(metric) & keep[k]) + early[k] > peak_bits[st + 1];
                }
                if (__builtin_amdgcn_ballot_w64(beaten))
                    fail |= 1u << (st + 1);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        float metric = 0.0f;
#pragma unroll
        for (int p = 0; p < PMAX; p++) {
            if (PMAX == 1 || p < g.P) {
                if (MODE == KIMG_CLEAN_I) {
                    if (p == 0)
                        metric = fabsf(dv[k][0]);
                } else {
                    metric += dv[k][p] * dv[k][p];                 // clean.py:962-964
                }
            }
        }
        const int y = oy + (tid >> 5) + 8 * k;
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
    if ((tid & 63) == 0) {
        s.keys[tid >> 6] = w;
        s.fails[tid >> 6] = fail;       // (wave-uniform)
    }
    lds_barrier();
    const mkey_t tb = kmax(kmax(s.keys[0], s.keys[1]), kmax(s.keys[2], s.keys[3]));
    const int widx = ~(int) (unsigned) tb;
    if (tb == 0 ? tid == 0 : tid == (widx & 255)) {
        mc_record o;
        o.pad[0] = (int) (s.fails[0] | s.fails[1] | s.fails[2] | s.fails[3]);
        o.pad[1] = 0;
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
                o.pix[p] = p >= PMAX ? 0.0f
                           : kk == 0 ? dv[0][p < PMAX ? p : 0] : kk == 1 ? dv[1][p < PMAX ? p : 0]
                           : kk == 2 ? dv[2][p < PMAX ? p : 0] : dv[3][p < PMAX ? p : 0];
        }
        *out = o;
    }
    MC_STAMP(7);
}

// STEPS: the most steps a lattice takes per launch.  Two instances are built: STEPS = 1 plans and
// verifies single steps only and is what a field of many comparable sources needs (everything that
// serves the repeated steps costs a launch 1.5 us of instruction issue); STEPS = 8 (4 with several
// polarizations) is for fields with a few dominant sources.  Either reads what the other leaves behind
// as long as the plan it finds has single steps (the host sees to that: kimg_clean_multi_run).
template <int MODE, int PMAX, int STEPS>
__global__ __launch_bounds__(MC_THREADS) void cycle_multi_kernel(
    float *dirty, float *model, const float *__restrict__ psf, float *tile_max, int32_t *tile_pos,
    mc_geom g, mc_scratch *scratch, int parity, float *log, unsigned long long *progress)
{
    __shared__ mc_lds s;
    __shared__ rest_lds sr;
    // Lanes: both halves of a wave do the same (l), and the candidate pool lives in the first row of
    // sixteen lanes (e): every wave works the plan out for itself, so nothing about it has to pass
    // through LDS or a barrier.
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l = lane & 31, e = lane & 15;
    // grid = (lat_x, lat_y, 1 + 2 mmax): plane 0 holds the three bookkeeping workgroups, planes
    // 1 .. mmax the blocks of the planned lattices, the rest the blocks of the committed ones
    int role, comp = 0;
    if (blockIdx.z == 0) {
        const int b = blockIdx.y * g.lat_x + blockIdx.x;
        if (b > 2)
            return;
        role = b == 0 ? ROLE_KEEPER : b == 1 ? ROLE_FOLDER : ROLE_LISTER;
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
    const int dbg_row = role == ROLE_KEEPER ? 0 : role == ROLE_LISTER ? 2
                        : (role == ROLE_NEW && comp == 0 && blockIdx.x == 0 && blockIdx.y == 0) ? 1 : -1;
    long long dbg_v[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    MC_COUNT(16, 1);
#else
    long long *dbg_v = nullptr;
    const long long dbg_t0 = 0;
#endif

    // ---- round trip 1: state, this thread's record, the previous plan and the list ----------
    const int4 st = *reinterpret_cast<const int4 *>(cur);                  // count, done, limit, threshold
    const int4 st2 = *reinterpret_cast<const int4 *>(&cur->planned);       // planned, rest_n, tau, launches
    const int4 st3 = *reinterpret_cast<const int4 *>(&cur->rest_floor);    // floor, gen, aux
    const mkey_t rest_floor = ((mkey_t) (unsigned) st3.y << 32) | (unsigned) st3.x;
    const int gen = st3.z;
    const int steps_max = st3.w & 0xff, cool = (st3.w >> 8) & 0xff, penalty = (st3.w >> 16) & 0xff;
    const mc_record *dp = &scratch->deltas[parity][tid];
    const int4 d0 = reinterpret_cast<const int4 *>(dp)[0];
    const int4 d1 = reinterpret_cast<const int4 *>(dp)[1];
    // lane l < 8: component l of the previous plan; lane l >= 8: entry l - 8 of the list
    const mc_record *pp = l < MC_MAX ? &cur->plan[l] : &cur->top[l - MC_MAX];
    const int4 p0 = reinterpret_cast<const int4 *>(pp)[0];
    const int4 p1 = reinterpret_cast<const int4 *>(pp)[1];
    // the PSF's centre: what a subtraction does to the value at its own peak
    float centre[PMAX];
#pragma unroll
    for (int p = 0; p < PMAX; p++)
        centre[p] = ((STEPS > 1 || role == ROLE_KEEPER) && (PMAX == 1 || p < g.P))
                        ? psf[p * g.psf_pol_stride + (int64_t) (g.psf_h / 2) * g.psf_row_stride + g.psf_w / 2]
                        : 0.0f;
    int4 old0 = make_int4(0, 0, 0, 0), old1 = old0, st4 = old0;
    // (the keeper's own share: the count it carries on)
    const int repeated0 = (role == ROLE_KEEPER && tid == 0) ? cur->repeated : 0;
    if (role == ROLE_LISTER) {
        // the lister's own share of the round trip: its entry of the list, the list's floor
        if (tid < MC_TOP) {
            old0 = reinterpret_cast<const int4 *>(&cur->top[tid])[0];
            old1 = reinterpret_cast<const int4 *>(&cur->top[tid])[1];
        }
        st4 = *reinterpret_cast<const int4 *>(&cur->top_floor);
    }
    const int count0 = st.x, done = st.y, limit = st.z;
    float threshold = __int_as_float(st.w);
    const int Mp = st2.x, top_n = st2.y, tau = st2.z;
    // A call whose threshold follows from its first component (frontend.py:560-575: one cycle
    // without a threshold, then threshold = max(noise threshold, (1 - major gain) x that peak) for
    // the rest -- or no further cycle if the peak itself is not above it): the first launch plans
    // that one component, the second commits it and works the threshold out, in the host's
    // arithmetic (doubles; clean.py:166-184 for what a metric is as a flux).
    bool first_alone = false, stop_after_first = false;
    if (count0 == 0 && !done && scratch->pad[1]) {
        if (Mp == 0) {
            first_alone = true;
            threshold = 0.0f;
        } else {
            const double noise_threshold = *reinterpret_cast<const double *>(&scratch->pad[2]);
            const double left = *reinterpret_cast<const double *>(&scratch->pad[4]);
            const mc_record *first = &cur->plan[0];
            const double metric = (double) __uint_as_float((unsigned) (first->key >> 32));
            const double power = MODE == KIMG_CLEAN_I ? metric : sqrt(metric);
            const double mgain = left * power;
            const double t = noise_threshold > mgain ? noise_threshold : mgain;
            stop_after_first = power <= t;
            threshold = (float) (MODE == KIMG_CLEAN_I ? t : t * t);
        }
    }
    const int threshold_bits = __float_as_int(threshold);
    if (count0 == 0 && !done && Mp > 0 && role == ROLE_KEEPER && tid == 0 && progress && scratch->pad[1])
        __hip_atomic_store(progress + 2, ((unsigned long long) (unsigned) gen << 32) | (unsigned) (cur->plan[0].key >> 32),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const int rest_n = min(top_n, MC_REST);
    if (done) {
        if (role == ROLE_KEEPER && tid == 0) {
            *reinterpret_cast<int4 *>(next) = st;
            *reinterpret_cast<int4 *>(&next->planned) = make_int4(0, 0, tau, st2.w + 1);
            next->gen = gen;
            next->aux = 0;
            next->repeated = repeated0;
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
    const mkey_t pkey = ((mkey_t) (unsigned) p0.y << 32) | (unsigned) p0.x;
    const float ppix[4] = {__int_as_float(p0.z), __int_as_float(p0.w), __int_as_float(p1.x),
                           __int_as_float(p1.y)};
    const bool planned_lane = lane < MC_MAX && lane < Mp;       // this lane holds a component of the plan
    const int pr = planned_lane ? p1.z : 0;                     // ... of so many steps
    // this lane's record as a candidate: its tile and the lattice of the component it would make
    const mc_cand pc = mc_decode(pkey, g.border);
    const int plx = lat_origin(pc.x, g.patch_w, g.border), ply = lat_origin(pc.y, g.patch_h, g.border);
    {
        // first exchange: the best record of every 16, with its pixel values; the fail bits
        const mkey_t r = row_max(dkey);
        if (e == 0)
            s.row[tid >> 4] = r;
        if (dkey != 0 && dkey == r)
            *reinterpret_cast<float4 *>(s.row_pix[tid >> 4]) = make_float4(dpix[0], dpix[1], dpix[2], dpix[3]);
        if (STEPS > 1 && steps_max > 1) {
            unsigned f = live ? (unsigned) d1.z : 0u;
            f |= (unsigned) __builtin_amdgcn_mov_dpp((int) f, 0xB1, 0xf, 0xf, true);
            f |= (unsigned) __builtin_amdgcn_mov_dpp((int) f, 0x4E, 0xf, 0xf, true);
            f |= (unsigned) __builtin_amdgcn_mov_dpp((int) f, 0x141, 0xf, 0xf, true);
            f |= (unsigned) __builtin_amdgcn_mov_dpp((int) f, 0x140, 0xf, 0xf, true);
            if (e == 0)
                s.rowf[tid >> 4] = f;
        }
    }
    // (meanwhile) the value at each planned peak before its step s: kq[s], with the pixel values
    // pq[s] -- the scalar recursion of mc_peak_step, as far as any lattice of the plan goes
    mkey_t kq[STEPS + 1];
    float pq[STEPS][PMAX];
#pragma unroll
    for (int st_ = 0; st_ < STEPS; st_++) {
        kq[st_ + 1] = 0;
#pragma unroll
        for (int p = 0; p < PMAX; p++)
            pq[st_][p] = ppix[p];
    }
    kq[0] = pkey;
    if (STEPS > 1 && steps_max > 1) {
        float v[PMAX];
#pragma unroll
        for (int p = 0; p < PMAX; p++)
            v[p] = ppix[p];
#pragma unroll
        for (int st_ = 1; st_ < STEPS; st_++) {
            if (st_ < steps_max) {
                const float m = mc_peak_step<MODE, PMAX>(v, centre, g.loop_gain, g.P);
                kq[st_] = ((mkey_t) __float_as_uint(m) << 32) | (unsigned) pkey;
#pragma unroll
                for (int p = 0; p < PMAX; p++)
                    pq[st_][p] = v[p];
            }
        }
    }
    lds_barrier();
    MC_STAMP(1);

    // ---- verify: the steps of the previous plan that held -------------------------------------------
    mkey_t a = 0;                       // lane e < 8: best record of lattice e ...
    int arow = 0;
    unsigned afail = 0;                 // ... and the fail bits of its blocks
    mkey_t mine = 0;                    // best record of this thread's lattice
    // (q = 1, 2, 4, 8 or 16 rows per lattice; four reads in flight at a time, none of them conditional)
    for (int r0 = 0; r0 < q; r0 += 4) {
        mkey_t v[4], w[4];
        unsigned f[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            v[i] = s.row[(e * q + r0 + i) & 15];
            w[i] = s.row[(li * q + r0 + i) & 15];
            f[i] = (STEPS > 1 && steps_max > 1) ? s.rowf[(e * q + r0 + i) & 15] : 0u;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int row = e * q + r0 + i;
            const bool valid = e < MC_MAX && r0 + i < q && row < 16;
            if (valid && v[i] > a) {
                a = v[i];
                arow = row;
            }
            afail |= valid ? f[i] : 0u;
            mine = kmax(mine, r0 + i < q ? w[i] : 0);
        }
    }
    const float4 apix = *reinterpret_cast<const float4 *>(s.row_pix[arow & 15]);   // ... and its pixel values
    // Lattice i allows the steps whose keys are above its level: the key before the step after
    // which one of its blocks saw the peak beaten (that step itself still held), else whatever
    // its blocks hold after all of its steps (its own steps are never cut by that).  The steps
    // above every lattice's level are committed: a prefix of the plan's order.
    int held;                           // lanes 0 .. 7: steps of this lane's lattice that are committed
    unsigned fullm, pendm, partm;       // lattices with all / at least one / some but not all steps committed
    int committed = 0;
    if (STEPS == 1 || steps_max <= 1) {
        // Single steps, planned in the order of their keys: component i + 1 held iff the records of
        // lattices 1 .. i are all below its key (the same as the levels below say: the committed
        // steps are a prefix).  A plan without steps (lattices evaluated again) is all there.
        mkey_t pre = a;                 // inclusive prefix maximum over lanes 0 .. e of the row
        pre = kmax(pre, kdpp<0x111>(pre));          // row_shr:1
        pre = kmax(pre, kdpp<0x112>(pre));          // row_shr:2
        pre = kmax(pre, kdpp<0x114>(pre));          // row_shr:4
        const mkey_t before = kdpp<0x111>(pre);     // lattices 0 .. e - 1
        const bool ok = planned_lane && (lane == 0 || before < pkey);
        const unsigned run = (unsigned) __builtin_amdgcn_ballot_w64(ok) & 0xffu;
        const int j = steps_max == 0 ? 0 : __builtin_ctz(~run);       // leading run of ones
        held = (planned_lane && lane < j) ? 1 : 0;
        pendm = (1u << j) - 1u;
        fullm = steps_max == 0 ? (Mp >= 8 ? 0xffu : (1u << Mp) - 1u) : pendm;
        partm = 0;
        committed = j;
    } else {
        mkey_t level = 0;
        if (planned_lane && pr > 0) {
            const unsigned f = afail & ((1u << pr) - 2u);       // bits 1 .. pr - 1
            mkey_t last = kq[0], cut = kq[0];
#pragma unroll
            for (int st_ = 1; st_ < STEPS; st_++) {
                if (st_ < pr)
                    last = kq[st_];
                if (f && st_ == __builtin_ctz(f | 0x80000000u) - 1)
                    cut = kq[st_];
            }
            level = f ? cut - 1 : (a < last - 1 ? a : last - 1);
        }
        level = row_max(level);
        level = lane_key(level, 0);
        held = 0;
#pragma unroll
        for (int st_ = 0; st_ < STEPS; st_++) {
            if (st_ < steps_max) {
                const bool ok = st_ < pr && kq[st_] > level;
                held += ok ? 1 : 0;
                committed += __builtin_popcountll(__builtin_amdgcn_ballot_w64(planned_lane && ok));
            }
        }
        fullm = (unsigned) __builtin_amdgcn_ballot_w64(planned_lane && held == pr) & 0xffu;
        pendm = (unsigned) __builtin_amdgcn_ballot_w64(planned_lane && held > 0) & 0xffu;
        partm = (unsigned) __builtin_amdgcn_ballot_w64(planned_lane && held > 0 && held < pr) & 0xffu;
    }
    const int count = count0 + committed;
    const bool mispredicted = fullm != (Mp >= 8 ? 0xffu : (1u << Mp) - 1u);
    const bool repair = STEPS > 1 && partm != 0;        // (a plan of single steps is never partly committed)
    MC_STAMP(2);
    if (role == ROLE_FOLDER) {
        // the records of wholly committed lattices go to the base arrays (nobody in this launch
        // reads them there)
        if (live && (fullm >> li & 1u)) {
            const mc_cand c = mc_decode(dkey, g.border);
            const int t = c.ty * g.tiles_x + c.tx;
            if (c.tx >= g.tiles_x || c.ty >= g.tiles_y)
                return;                 // (no such tile: a record nobody of this call made)
            tile_max[t] = c.value;
            *reinterpret_cast<int2 *>(tile_pos + 2 * t) = make_int2(c.y, c.x);
            *reinterpret_cast<float4 *>(tile_pix + 4 * t) = make_float4(dpix[0], dpix[1], dpix[2], dpix[3]);
        }
        return;
    }
    if (role == ROLE_LISTER) {
        // the list for the next launch, next to (not behind) everybody's planning: entries inside
        // lattices whose pixels change go (wholly committed: their records come in; partly: their
        // records come with the next launch)
        const bool changing = planned_lane && ((fullm | partm) >> lane & 1u);
        mc_build_rest(tile_max, tile_pos, tile_pix, scratch->deltas[parity], g, tau, 1,
                      changing ? plx : INT_MIN / 2, changing ? ply : INT_MIN / 2, Mp,
                      (live && (fullm >> li & 1u)) ? dkey : 0, make_float4(dpix[0], dpix[1], dpix[2], dpix[3]), true,
                      old0, old1, top_n,
                      ((mkey_t) (unsigned) st4.y << 32) | (unsigned) st4.x, next, sr, dbg_v, dbg_t0);
        MC_STAMP(11);
        MC_FLUSH();
        return;
    }
    if (role == ROLE_COMMIT && !(pendm >> comp & 1u))
        return;
    {
        const mkey_t second = (live && (fullm >> li & 1u) && dkey != mine) ? dkey : 0;
        const mkey_t r = row_max(second);
        if (e == 0)
            s.row2[tid >> 4] = r;
    }
    // (meanwhile) which entries of the list survive: those outside every lattice of the previous
    // plan, committed or not -- their values stand; the first MC_POOL_REST of them enter the pool
    bool surv = l >= MC_MAX && l - MC_MAX < rest_n && pkey >= MC_REAL && !mispredicted;
#pragma unroll
    for (int i = 0; i < MC_MAX; i++) {
        const int rx = __builtin_amdgcn_readlane(plx, i), ry = __builtin_amdgcn_readlane(ply, i);
        const bool inside = (unsigned) (pc.tx - rx) < (unsigned) g.lat_x
                            && (unsigned) (pc.ty - ry) < (unsigned) g.lat_y;
        surv = surv && !(inside && i < Mp);
    }
    const unsigned smask = (unsigned) __builtin_amdgcn_ballot_w64(surv);    // (lanes 0 .. 31)
    const int srank = __builtin_popcount(smask & ((1u << l) - 1u));
    if (lane < 32 && surv && srank < MC_POOL_REST) {
        mc_record r;
        r.key = pkey;
#pragma unroll
        for (int i = 0; i < 4; i++)
            r.pix[i] = ppix[i];
        r.pad[0] = r.pad[1] = 0;
        s.pool[wave][srank] = r;
    }
    mkey_t list_floor = rest_floor;     // every tile outside the pool and the last lattices is below
    {
        const unsigned ninth = (unsigned) __builtin_amdgcn_ballot_w64(surv && srank == MC_POOL_REST);
        if (ninth)
            list_floor = lane_key(pkey, __builtin_ctz(ninth)) | 0x3ffu;
    }
    const int nsurv = min(__builtin_popcount(smask), MC_POOL_REST);
    lds_barrier();
    MC_STAMP(3);

    // ---- plan -------------------------------------------------------------------------------------
    // lane e < 8: the best of the records of lattice e that are not its best one (they bound what can
    // be proven unless the lattice is planned again as it is, see below)
    mkey_t sec = 0;
    for (int r0 = 0; STEPS > 1 && r0 < q; r0 += 4) {
        mkey_t v[4];
#pragma unroll
        for (int i = 0; i < 4; i++)
            v[i] = s.row2[(e * q + r0 + i) & 15];
#pragma unroll
        for (int i = 0; i < 4; i++)
            sec = kmax(sec, (e < MC_MAX && r0 + i < q && e * q + r0 + i < 16) ? v[i] : 0);
    }
    // the pool, one candidate per lane of the first row: the committed lattices' best records and
    // the surviving entries of the list.  After a misprediction: those records and the best
    // component that was not committed, which was the best tile outside the committed lattices;
    // exactly one component is then planned.  With partly committed lattices: those, to be
    // evaluated again without steps, and nothing else.
    mkey_t cand = 0;
    float cpix[4] = {apix.x, apix.y, apix.z, apix.w};
    // (lanes 0 .. 7 of the wave hold the plan: the best of its lattices without any committed step)
    mkey_t left = 0;
    unsigned left_at = 0;
    float left_pix[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (mispredicted && !repair) {
        left = (planned_lane && held == 0 && pr > 0) ? pkey : 0;
        left = lane_key(row_max(left), 0);
        left_at = (unsigned) __builtin_amdgcn_ballot_w64(planned_lane && left >= MC_REAL && pkey == left);
#pragma unroll
        for (int i = 0; i < 4; i++)
            left_pix[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ppix[i]),
                                                                   left_at ? __builtin_ctz(left_at) : 0));
    }
    if (repair) {
        if (lane < MC_MAX && (partm >> lane & 1u)) {
            cand = pkey;
#pragma unroll
            for (int i = 0; i < 4; i++)
                cpix[i] = ppix[i];
        }
    } else if (e < MC_MAX) {
        cand = (fullm >> e & 1u) ? a : 0;
    } else if (mispredicted) {
        if (e == MC_MAX && left_at) {
            cand = left;
#pragma unroll
            for (int i = 0; i < 4; i++)
                cpix[i] = left_pix[i];
        }
    } else if (e - MC_MAX < nsurv) {
        const mc_record *r = &s.pool[wave][e - MC_MAX];
        cand = r->key;
#pragma unroll
        for (int i = 0; i < 4; i++)
            cpix[i] = r->pix[i];
    }
    if (cand < MC_REAL)
        cand = 1 + e;                   // distinct fillers below every real key
    const int rank = row_rank<1>(cand);
    // sorted: lane p of the first row gets the p-th largest (the other rows keep their own)
    const int to = (lane < 16 ? rank : lane) << 2;
    mkey_t sk;
    float spix[4];
    {
        const unsigned lo = __builtin_amdgcn_ds_permute(to, (int) (unsigned) cand);
        const unsigned hi = __builtin_amdgcn_ds_permute(to, (int) (unsigned) (cand >> 32));
        sk = ((mkey_t) hi << 32) | lo;
#pragma unroll
        for (int i = 0; i < 4; i++)
            spix[i] = __int_as_float(__builtin_amdgcn_ds_permute(to, __float_as_int(cpix[i])));
    }
    const mc_cand c = mc_decode(sk, g.border);
    const int cbx = lat_origin(c.x, g.patch_w, g.border), cby = lat_origin(c.y, g.patch_h, g.border);
    if (role == ROLE_NEW)
        MC_STAMP(8);
    // Everything the walk below needs, as masks over the sorted positions (bit p = candidate p):
    // per-candidate properties by one compare each, and per pivot k the candidates whose tile lies
    // inside lattice k / whose lattice meets lattice k.  The walk itself is scalar and branch-free.
    unsigned m_ins[MC_MAX], m_ovl[MC_MAX];
#pragma unroll
    for (int k = 0; k < MC_MAX; k++) {
        const int pbx = __builtin_amdgcn_readlane(cbx, k), pby = __builtin_amdgcn_readlane(cby, k);
        m_ins[k] = (unsigned) __builtin_amdgcn_ballot_w64((unsigned) (c.tx - pbx) < (unsigned) g.lat_x)
                   & (unsigned) __builtin_amdgcn_ballot_w64((unsigned) (c.ty - pby) < (unsigned) g.lat_y);
        m_ovl[k] = (unsigned) __builtin_amdgcn_ballot_w64(
                       (unsigned) (cbx - pbx + g.lat_x - 1) < (unsigned) (2 * g.lat_x - 1))
                   & (unsigned) __builtin_amdgcn_ballot_w64(
                       (unsigned) (cby - pby + g.lat_y - 1) < (unsigned) (2 * g.lat_y - 1));
    }
    // What bounds a proof: the list's floor, the pool's overflow, and the second-best records of
    // the committed lattices -- except where a lattice's best record is a candidate that would be
    // planned with the very same lattice and that no earlier candidate's lattice holds: it is then
    // planned (the other records lie inside a planned lattice, and the next verification covers
    // them), or the walk ends at or before it (and what ends the walk bounds everything after).
    if (role == ROLE_NEW)
        MC_STAMP(9);
    const mkey_t overflow = lane_key(sk, MC_MAX);
    mkey_t bound2 = kmax(list_floor, overflow >= MC_REAL ? overflow | 0x3ffu : 0);
    if (STEPS == 1) {
        // (single steps: every second-best record bounds)
        bound2 = kmax(bound2, lane_key(row_max(s.row2[e]), 0));
    } else {
        mkey_t second = lane_key(row_max(lane < MC_MAX ? sec : 0), 0);
        if (second > bound2 && !mispredicted) {
            bool lifted = false;
            if (lane < MC_MAX && (fullm >> lane & 1u) && rank < MC_MAX) {
                const mc_cand ac = mc_decode(a, g.border);
                const bool same = lat_origin(ac.x, g.patch_w, g.border) == plx
                                  && lat_origin(ac.y, g.patch_h, g.border) == ply;
                bool held_by_earlier = false;
#pragma unroll
                for (int k = 0; k < MC_MAX; k++)
                    held_by_earlier = held_by_earlier || (k < rank && (m_ins[k] >> rank & 1u));
                lifted = same && !held_by_earlier;
            }
            second = lane_key(row_max((lane < MC_MAX && !lifted) ? sec : 0), 0);
        }
        bound2 = kmax(bound2, second);
    }
    if (role == ROLE_NEW)
        MC_STAMP(10);
    const unsigned m_real = (unsigned) __builtin_amdgcn_ballot_w64(sk >= MC_REAL) & 0xffu;
    const unsigned m_above = (unsigned) __builtin_amdgcn_ballot_w64(sk > bound2) & 0xffu;
    const unsigned m_thr = (unsigned) __builtin_amdgcn_ballot_w64(!(c.value < threshold)) & 0xffu;
    const unsigned m_zero = (unsigned) __builtin_amdgcn_ballot_w64(c.value == 0.0f) & 0xffu;
    unsigned picked = 0;
    int M = 0;
    bool done_now, zero_special;
    unsigned skipped = 0;
    if (repair) {
        picked = m_real;
        M = __builtin_popcount(picked);
        done_now = false;
        zero_special = false;
    } else {
        // (the first candidate is the largest tile of the image if it beats every tile that is not
        // listed; with an entry of the list in the pool it does)
        const bool first_proven = nsurv > 0 || mispredicted || lane_key(sk, 0) > list_floor;
        const unsigned room = (unsigned) min(mispredicted || first_alone ? 1 : g.mmax, limit - count);
        // the walk considers the candidates in order while each is real, proven to be next (above
        // everything outside the pool) and passes the threshold: a prefix of the positions
        const unsigned elig = m_real & m_thr & (m_above | (first_proven ? 1u : 0u))
                              & (limit > count ? 0xffu : 0u);
        const unsigned pref = (1u << __builtin_ctz(~elig)) - 1u;       // its lowest run of ones
        // clean.py:1065-1066: the loop ends when the largest tile of all fails threshold or limit
        done_now = (m_real & 1u) && first_proven && !(elig & 1u);
        unsigned skip = 0, stop = 0, alive = 1, zs = 0;
#pragma unroll
        for (int p = 0; p < MC_MAX; p++) {
            const unsigned c1 = (pref >> p) & alive;                        // candidate p is considered
            const unsigned sk1 = (skip >> p) & 1u, st1 = (stop >> p) & 1u, z1 = (m_zero >> p) & 1u;
            // a candidate inside a planned lattice is passed over (its value is about to change);
            // one whose lattice meets a planned lattice ends the walk; one without any positive
            // metric is taken only alone, and only while nothing is pending (its pixel is read
            // from the image, which must be up to date)
            const unsigned zero_ok = (p == 0 && pendm == 0) ? 1u : 0u;
            const unsigned take = c1 & ~sk1 & ~st1 & (~z1 | zero_ok) & 1u;
            picked |= take << p;
            M += (int) take;
            skip |= m_ins[p] & (0u - take);
            stop |= m_ovl[p] & (0u - take);
            zs |= take & z1;
            const unsigned halt = (c1 & ~sk1 & (st1 | z1)) | (take & ((unsigned) M >= room ? 1u : 0u));
            alive &= c1 & ~halt & 1u;
        }
        zero_special = zs != 0;
        skipped = skip;
        if (!(m_real & 1u) && pendm == 0 && rest_floor == 0)
            done_now = true;            // no tiles at all
    }
    if (role == ROLE_NEW)
        MC_STAMP(11);
    if (stop_after_first) {
        picked = 0;
        M = 0;
        done_now = true;
    }
    // ---- how many steps each planned lattice takes -------------------------------------------------
    // The plan is the merge of the planned lattices' sequences sq[0] > sq[1] > ... (the value at the
    // peak before each of its steps) above a level below which nothing is proven: the bound of the
    // walk; the first candidate that is neither planned nor inside a planned lattice; for every
    // planned lattice the value its peak has when it may not be stepped again (the last allowed step,
    // the threshold, a value that does not come down).
    mkey_t sq[STEPS + 1];
    float psq[STEPS][PMAX];
    int rsteps = 0;                     // lane p < 8: steps of the candidate at sorted position p
    int next_steps_max = 0;
#pragma unroll
    for (int st_ = 0; st_ < STEPS; st_++) {
        sq[st_ + 1] = 0;
#pragma unroll
        for (int p = 0; p < PMAX; p++)
            psq[st_][p] = spix[p];
    }
    sq[0] = sk;
    int opportunity = 0;                // (single-step kernel, keeper: the repeated steps that could have been planned)
    if (STEPS == 1) {
        // (single steps, as the walk planned them; a lattice a higher peak's next value would beat
        // is found out by the next verification)
        rsteps = (!repair && lane < MC_MAX && (picked >> lane & 1u)) ? 1 : 0;
        next_steps_max = (!repair && M > 0) ? 1 : 0;
        if (role == ROLE_KEEPER && M > 0 && !mispredicted && !zero_special && cool == 0) {
            // What the host goes by when it chooses between the two kernels: how many more steps the
            // planned peaks could take before they are down to what is not planned (each on its
            // own; the second-best records left aside: the other kernel mostly lifts them).
            mkey_t level = kmax(list_floor, overflow >= MC_REAL ? overflow | 0x3ffu : 0);
            const unsigned u = m_real & ~picked & ~skipped;
            if (u)
                level = kmax(level, lane_key(sk, __builtin_ctz(u)) | 0x3ffu);
            float v[PMAX];
#pragma unroll
            for (int p = 0; p < PMAX; p++)
                v[p] = spix[p];
            mkey_t before = sk;
            bool going = lane < MC_MAX && (picked >> lane & 1u);
            for (int st_ = 1; st_ < (PMAX == 1 ? MC_STEPS : MC_STEPS / 2); st_++) {
                const float m = mc_peak_step<MODE, PMAX>(v, centre, g.loop_gain, g.P);
                const mkey_t key = ((mkey_t) __float_as_uint(m) << 32) | (unsigned) sk;
                going = going && key > level && key < before && !(m < threshold);
                before = key;
                const unsigned long long more = __builtin_amdgcn_ballot_w64(going);
                if (!more)
                    break;
                opportunity += __builtin_popcountll(more);
            }
        }
    } else if (!repair) {
        const bool mine_picked = lane < MC_MAX && (picked >> lane & 1u);
        const int rmax = (mispredicted || zero_special || first_alone || cool > 0 || limit - count < g.mmax * g.rmax)
                             ? 1 : min(g.rmax, STEPS);
        mkey_t level = bound2;
        {
            const unsigned u = m_real & ~picked & ~skipped;
            if (u)
                level = kmax(level, lane_key(sk, __builtin_ctz(u)) | 0x3ffu);
        }
        float v[PMAX];
#pragma unroll
        for (int p = 0; p < PMAX; p++)
            v[p] = spix[p];
        const float m1 = mc_peak_step<MODE, PMAX>(v, centre, g.loop_gain, g.P);
        sq[1] = ((mkey_t) __float_as_uint(m1) << 32) | (unsigned) sk;
        // Nearly always no planned peak is still above the level after one step: one step each, as
        // the walk planned them (what the rest of this block works out then comes to just that).
        if (!__builtin_amdgcn_ballot_w64(mine_picked && (sq[1] > level || sq[1] >= sq[0]))) {
            rsteps = mine_picked ? 1 : 0;
            next_steps_max = M > 0 ? 1 : 0;
        } else {
            int end = 0;                // first step this lattice may not take (0: not known yet)
            mkey_t lev = 0;
            int depth = 1;
            float m = m1;
#pragma unroll
            for (int st_ = 1; st_ <= STEPS; st_++) {
                if (depth == st_) {
                    if (st_ > 1) {
                        m = mc_peak_step<MODE, PMAX>(v, centre, g.loop_gain, g.P);
                        sq[st_] = ((mkey_t) __float_as_uint(m) << 32) | (unsigned) sk;
                    }
                    if (st_ < STEPS) {
#pragma unroll
                        for (int p = 0; p < PMAX; p++)
                            psq[st_ < STEPS ? st_ : 0][p] = v[p];
                    }
                    if (end == 0 && (st_ == rmax || m < threshold || sq[st_] >= sq[st_ - 1])) {
                        end = st_;
                        lev = sq[st_] >= sq[st_ - 1] ? sq[st_ - 1] - 1 : sq[st_];
                    }
                    // deeper only while some planned lattice is still above the level there
                    if (st_ < rmax && __builtin_amdgcn_ballot_w64(mine_picked && end == 0 && sq[st_] > level))
                        depth = st_ + 1;
                }
            }
            if (end == 0) {
                // (not above the level at this depth: bounds nothing new)
                end = depth;
                lev = 0;
            }
            {
                const mkey_t r = row_max(mine_picked ? lev : 0);
                level = kmax(level, lane_key(r, 0));
            }
#pragma unroll
            for (int st_ = 0; st_ < STEPS; st_++)
                if (st_ < depth)
                    rsteps += (mine_picked && st_ < end && sq[st_] > level) ? 1 : 0;
            // (the walk's first pick stands where nothing else can be said: it is the largest tile)
            if ((mispredicted || zero_special) && picked && lane == __builtin_ctz(picked))
                rsteps = 1;
            picked = (unsigned) __builtin_amdgcn_ballot_w64(lane < MC_MAX && rsteps > 0) & 0xffu;
            M = __builtin_popcount(picked);
#pragma unroll
            for (int st_ = 1; st_ <= STEPS; st_++)
                if (st_ <= depth && __builtin_amdgcn_ballot_w64(lane < MC_MAX && rsteps >= st_))
                    next_steps_max = st_;
        }
    }
    MC_STAMP(4);
    MC_COUNT(17, M);
    MC_COUNT(18, committed);
    // the sorted position of planned component m: the m-th set bit of `picked`
    auto position = [&](int m) {
        unsigned x = picked;
        for (int i = 0; i < m; i++)
            x &= x - 1u;
        return __builtin_ctz(x | 0x100u);
    };
    auto lane_int = [&](int v, int from) { return __builtin_amdgcn_readlane(v, from); };
    auto lane_float = [&](float v, int from) {
        return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), from));
    };
    const unsigned no_bits[STEPS] = {};

    if (role == ROLE_NEW) {
        if (comp >= M)
            return;
        const int p = position(comp);
        const int ny = lane_int(c.y, p), nx = lane_int(c.x, p);
        const int tx = lane_int(cbx, p) + (int) blockIdx.x, ty = lane_int(cby, p) + (int) blockIdx.y;
        const int nsteps = lane_int(rsteps, p);
        float ns[STEPS][PMAX];
        unsigned peak_bits[STEPS];
#pragma unroll
        for (int st_ = 0; st_ < STEPS; st_++) {
            peak_bits[st_] = 0;
#pragma unroll
            for (int i = 0; i < PMAX; i++)
                ns[st_][i] = 0.0f;
            if (st_ < nsteps) {
                if (STEPS > 1)
                    peak_bits[st_] = (unsigned) lane_int((int) (unsigned) (sq[st_] >> 32), p);
#pragma unroll
                for (int i = 0; i < PMAX; i++)
                    ns[st_][i] = g.loop_gain * lane_float(psq[st_][i], p);     // clean.py:1044
            }
        }
        if (zero_special) {
            // a tile without any positive metric won: its record holds the (x0, y0) start position
            // of clean.py:950, whose pixel is read now (nothing is pending: the image is up to date)
            const bool ok = ny >= 0 && ny < g.height && nx >= 0 && nx < g.width;
#pragma unroll
            for (int i = 0; i < PMAX; i++)
                ns[0][i] = g.loop_gain
                           * (ok && i < g.P ? dirty[i * g.pol_stride + (int64_t) ny * g.row_stride + nx] : 0.0f);
        }
        // the pending steps of a committed lattice that holds this block, if any
        const unsigned hit = (unsigned) __builtin_amdgcn_ballot_w64(
            lane < MC_MAX && (pendm >> lane & 1u) && (unsigned) (tx - plx) < (unsigned) g.lat_x
            && (unsigned) (ty - ply) < (unsigned) g.lat_y);
        const int pend = hit ? __builtin_ctz(hit) : 0;
        const int pend_n = hit ? lane_int(held, pend) : 0;
        float ps[STEPS][PMAX];
#pragma unroll
        for (int st_ = 0; st_ < STEPS; st_++) {
#pragma unroll
            for (int i = 0; i < PMAX; i++)
                ps[st_][i] = 0.0f;
            if (st_ < pend_n) {
#pragma unroll
                for (int i = 0; i < PMAX; i++)
                    ps[st_][i] = g.loop_gain * lane_float(pq[st_][i], pend);
            }
        }
        // (the peak's tile and pixel, for the blocks' question whether it is still the first pixel)
        const unsigned plo = STEPS > 1 ? (unsigned) lane_int((int) (unsigned) sk, p) : 0u;
        const unsigned ptile = ~(plo >> 10) & 0x3FFFFFu;
        MC_STAMP(5);
        mc_block<MODE, PMAX, STEPS>(dirty, psf, g, tx, ty, pend_n, lane_int(pc.y, pend),
                                    lane_int(pc.x, pend), ps, true, nsteps, ny, nx, ns, peak_bits,
                                    (int) (ptile >> 11) * g.tiles_x + (int) (ptile & 2047u), (int) (plo & 1023u),
                                    &scratch->deltas[parity ^ 1][comp * g.seg + (int) blockIdx.y * g.lat_x
                                                                 + (int) blockIdx.x],
                                    s, dbg_v, dbg_t0);
        MC_FLUSH();
        return;
    }
    if (role == ROLE_COMMIT) {
        const int tx = lane_int(plx, comp) + (int) blockIdx.x, ty = lane_int(ply, comp) + (int) blockIdx.y;
        const unsigned covered = (unsigned) __builtin_amdgcn_ballot_w64(
            lane < MC_MAX && (picked >> lane & 1u) && (unsigned) (tx - cbx) < (unsigned) g.lat_x
            && (unsigned) (ty - cby) < (unsigned) g.lat_y);
        if (covered)
            return;                     // a workgroup of the new lattice writes this block
        const int pend_n = lane_int(held, comp);
        float ps[STEPS][PMAX];
#pragma unroll
        for (int st_ = 0; st_ < STEPS; st_++) {
#pragma unroll
            for (int i = 0; i < PMAX; i++)
                ps[st_][i] = 0.0f;
            if (st_ < pend_n) {
#pragma unroll
                for (int i = 0; i < PMAX; i++)
                    ps[st_][i] = g.loop_gain * lane_float(pq[st_][i], comp);
            }
        }
        mc_block<MODE, PMAX, STEPS>(dirty, psf, g, tx, ty, pend_n, lane_int(pc.y, comp),
                                    lane_int(pc.x, comp), ps, false, 0, 0, 0, ps, no_bits, 0, 0, nullptr, s,
                                    dbg_v, dbg_t0);
        return;
    }

    // ---- keeper: the committed steps, the planned components, the next state ------------------------
    // The committed steps in the order of their keys: lane i of wave 0 holds lattice i of the
    // previous plan; its steps go through LDS to one thread each, which finds its place in the log.
    if (STEPS == 1 || steps_max <= 1) {
        // (single steps: the plan is in the order of its keys)
        if (tid < MC_MAX && held > 0) {
            float *entry = log + (int64_t) (count0 + __builtin_popcount(pendm & ((1u << tid) - 1u))) * (3 + g.P);
            entry[0] = pc.value;
            entry[1] = __int_as_float(pc.y);
            entry[2] = __int_as_float(pc.x);
            for (int p = 0; p < g.P; p++) {
                const float sc = g.loop_gain * ppix[p];
                float *mp = model + p * g.pol_stride + (int64_t) pc.y * g.row_stride + pc.x;
                entry[3 + p] = sc;
                *mp += sc;                                          // clean.py:1047
            }
        }
    } else {
        if (tid < MC_MAX) {
#pragma unroll
            for (int st_ = 0; st_ < STEPS; st_++) {
                s.seqk[tid * MC_STEPS + st_] = st_ < held ? kq[st_] : 0;
#pragma unroll
                for (int p = 0; p < PMAX; p++)
                    s.seqp[tid * MC_STEPS + st_][p] = pq[st_][p];
            }
#pragma unroll
            for (int st_ = STEPS; st_ < MC_STEPS; st_++)
                s.seqk[tid * MC_STEPS + st_] = 0;
            if (held > 0) {
                for (int p = 0; p < g.P && p < PMAX; p++) {
                    float *mp = model + p * g.pol_stride + (int64_t) pc.y * g.row_stride + pc.x;
                    float m = *mp;
#pragma unroll
                    for (int st_ = 0; st_ < STEPS; st_++)
                        if (st_ < held)
                            m += g.loop_gain * pq[st_][p];          // clean.py:1047
                    *mp = m;
                }
            }
        }
        __syncthreads();
        if (tid < MC_MAX * MC_STEPS) {
            const mkey_t own = s.seqk[tid];
            if (own >= MC_REAL) {
                int place = 0;
                const ulonglong2 *pairs = reinterpret_cast<const ulonglong2 *>(s.seqk);
#pragma unroll 4
                for (int i = 0; i < MC_MAX * MC_STEPS; i += 2) {
                    const ulonglong2 two = pairs[i >> 1];
                    place += two.x > own ? 1 : 0;
                    place += two.y > own ? 1 : 0;
                }
                // (the position: that of the lattice's first step, whose key is never a made-up one)
                const mc_cand oc = mc_decode(s.seqk[tid & ~(MC_STEPS - 1)], g.border);
                float *entry = log + (int64_t) (count0 + place) * (3 + g.P);
                entry[0] = __uint_as_float((unsigned) (own >> 32));
                entry[1] = __int_as_float(oc.y);
                entry[2] = __int_as_float(oc.x);
                for (int p = 0; p < g.P && p < PMAX; p++)
                    entry[3 + p] = g.loop_gain * s.seqp[tid][p];
            }
        }
    }
    if (tid < MC_MAX && (picked >> tid & 1u)) {
        // (lane p of wave 0 holds the candidate at sorted position p)
        mc_record r;
        r.key = sk;
        r.pad[0] = rsteps;
        r.pad[1] = 0;
        const bool ok = c.y >= 0 && c.y < g.height && c.x >= 0 && c.x < g.width;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            r.pix[i] = spix[i];
            if (zero_special)
                r.pix[i] = ok && i < g.P ? dirty[i * g.pol_stride + (int64_t) c.y * g.row_stride + c.x] : 0.0f;
        }
        next->plan[__builtin_popcount(picked & ((1u << tid) - 1u))] = r;
    }
    if (tid == 0) {
        // repeated steps that went wrong cost a launch that commits nothing: single steps for a
        // while after that, for longer each time in a row
        int cool_next = cool > 0 && !repair ? cool - 1 : cool, penalty_next = penalty ? penalty : MC_COOL;
        if (repair) {
            penalty_next = min(MC_COOL_MAX, 2 * penalty_next);
            cool_next = penalty_next;
        } else if (steps_max > 1 && !mispredicted) {
            penalty_next = MC_COOL;
        }
        *reinterpret_cast<int4 *>(next) = make_int4(count, done_now ? 1 : 0, limit, first_alone ? st.w : threshold_bits);
        next->planned = M;
        next->launches = st2.w + 1;
        next->gen = gen;
        next->aux = (repair ? 0 : next_steps_max) | cool_next << 8 | penalty_next << 16;
        // (the repeated-steps kernel counts the steps committed beyond the first of their lattice and
        // launch; the single-step kernel those it could have planned)
        const int repeated = repeated0 + (STEPS == 1 ? opportunity : committed - __builtin_popcount(pendm));
        next->repeated = repeated;
        next->pad2 = 0;
        // (what the host reads goes out last: the word in host memory is a long way off)
        *reinterpret_cast<int4 *>(scratch->head) = make_int4(count, done_now ? 1 : 0, limit, first_alone ? st.w : threshold_bits);
        if (progress) {
            // (the second word first, and only when it moves: whoever sees the launch counted finds
            // its other figures there)
            if (repeated != repeated0)
                __hip_atomic_store(progress + 1, ((unsigned long long) (unsigned) gen << 32) | (unsigned) repeated,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(progress, progress_word(st2.w + 1, gen, done_now, count),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    MC_STAMP(6);
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
                                                            float threshold, int gen, int mode,
                                                            int relative, double noise_threshold, double left,
                                                            unsigned long long *progress)
{
    __shared__ rest_lds sr;
    __shared__ int s_tau;
    __shared__ int s_hist[1024];
    __shared__ int s_above[MC_THREADS];
    const int tid = threadIdx.x;
    mc_state *st = &scratch->st[0];
    const float *tile_pix = reinterpret_cast<const float *>(scratch + 1);
    // Where to cut the tile maxima for the first list: the value above which the 128 best tiles lie,
    // to a quarter of an octave, from one pass (a histogram over the top eleven bits of the float
    // patterns, which order like the values) -- stepping down from the maximum by a factor 0.7 per
    // pass took a dozen passes on a field whose first source stands far above the rest.
    for (int i = tid; i < 1024; i += MC_THREADS)
        s_hist[i] = 0;
    if (tid == 0)
        s_tau = -1;
    __syncthreads();
    for (int t = tid; t < g.tiles_x * g.tiles_y; t += MC_THREADS) {
        const int bits = __float_as_int(tile_max[t]);
        atomicAdd(&s_hist[min(max(bits, 0) >> 21, 1023)], 1);
    }
    __syncthreads();
    s_above[tid] = s_hist[4 * tid] + s_hist[4 * tid + 1] + s_hist[4 * tid + 2] + s_hist[4 * tid + 3];
    __syncthreads();
    for (int d = 1; d < MC_THREADS; d <<= 1) {          // tiles in this thread's bins and above
        const int v = tid + d < MC_THREADS ? s_above[tid + d] : 0;
        __syncthreads();
        s_above[tid] += v;
        __syncthreads();
    }
    {
        constexpr int WANT = 128;
        const int higher = tid + 1 < MC_THREADS ? s_above[tid + 1] : 0;
        if (s_above[tid] >= WANT && higher < WANT) {
            int acc = higher, bin = 4 * tid;
            for (int k = 3; k >= 0; k--) {
                acc += s_hist[4 * tid + k];
                if (acc >= WANT) {
                    bin = 4 * tid + k;
                    break;
                }
            }
            s_tau = (bin << 21) - 1;    // (bin 0: -1, every tile)
        }
    }
    __syncthreads();
    if (tid == 0) {
        *reinterpret_cast<int4 *>(st) = make_int4(0, 0, limit, __float_as_int(threshold));
        st->planned = 0;
        st->launches = 0;
        st->gen = gen;
        *reinterpret_cast<int4 *>(scratch->head) = make_int4(0, 0, limit, __float_as_int(threshold));
        scratch->pad[0] = 0x4d554c54;   // "MULT": which form the buffer holds (Clean.last_launches)
        scratch->pad[1] = relative;
        *reinterpret_cast<double *>(&scratch->pad[2]) = noise_threshold;
        *reinterpret_cast<double *>(&scratch->pad[4]) = left;
    }
#ifdef KIMG_MC_STAMPS
    long long dbg_v[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#else
    long long *dbg_v = nullptr;
#endif
    mc_build_rest(tile_max, tile_pos, tile_pix, nullptr, g, s_tau, 1, INT_MIN / 2,
                  INT_MIN / 2, 0, 0, make_float4(0.0f, 0.0f, 0.0f, 0.0f), false, make_int4(0, 0, 0, 0), make_int4(0, 0, 0, 0), 0, 0, st, sr, dbg_v, 0);
    // What the host starts from (the word of launch 0 carries it in place of a count): how many steps
    // the eight best tiles could take, each at its own peak, before they are down to the ninth -- a
    // field with a few sources far above the rest is worth the repeated-steps kernel from the start.
    __syncthreads();
    if (tid < 64 && progress) {
        // (lane i < 8: the i-th best tile; all of a wave's lanes take part in the sum)
        const int n = st->top_n;
        const float ref = n > MC_MAX ? __uint_as_float((unsigned) (st->top[MC_MAX].key >> 32))
                                     : __uint_as_float((unsigned) (st->top_floor >> 32));
        // (a step takes the value at the peak down by the loop gain; the metric is the value, or its square)
        const float per_step = -logf(fmaxf(1.0f - g.loop_gain, 1e-3f)) * (mode == KIMG_CLEAN_I ? 1.0f : 2.0f);
        float steps = 0.0f;
        if (tid < n && tid < MC_MAX) {
            const float v = __uint_as_float((unsigned) (st->top[tid].key >> 32));
            if (ref > 0.0f && v > ref && per_step > 0.0f)
                steps = fminf(logf(v / ref) / per_step, 1000.0f);
        }
        for (int off = 4; off > 0; off >>= 1)
            steps += __shfl_down(steps, off, WAVE);
        if (tid != 0)
            return;
        __hip_atomic_store(progress + 1, (unsigned long long) (unsigned) gen << 32, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
        // (the first component's metric comes later: until then a tag no call has)
        __hip_atomic_store(progress + 2, (unsigned long long) ((unsigned) gen ^ 0x8000u) << 32, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(progress, progress_word(0, gen, false, (int) steps), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    }
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
    int repeats;            // the kernel built for repeated steps (else: single steps)
};

// One launch.  `repeats`: with the kernel that plans up to `rmax` steps per lattice (rmax = 1: that
// kernel, which reads any plan, leaving a plan of single steps behind, which either kernel reads).
int enqueue_launch(const multi_args &a, hipStream_t s, int parity, bool repeats, int rmax)
{
    const dim3 grid(a.g.lat_x, a.g.lat_y, 1 + 2 * a.g.mmax);
    mc_geom g = a.g;
    g.rmax = repeats ? rmax : 1;
#define LAUNCH(MODE, PMAX, STEPS) cycle_multi_kernel<MODE, PMAX, STEPS><<<grid, MC_THREADS, 0, s>>>( \
        a.dirty, a.model, a.psf, a.tile_max, a.tile_pos, g, a.scratch, parity, a.log, a.progress)
#define LAUNCH2(MODE, PMAX) do { if (repeats) LAUNCH(MODE, PMAX, (PMAX == 1 ? MC_STEPS : MC_STEPS / 2)); \
                                 else LAUNCH(MODE, PMAX, 1); } while (0)
    if (a.mode == KIMG_CLEAN_I && a.g.P == 1)
        LAUNCH2(KIMG_CLEAN_I, 1);
    else if (a.mode == KIMG_CLEAN_I)
        LAUNCH2(KIMG_CLEAN_I, 4);
    else if (a.g.P == 1)
        LAUNCH2(KIMG_CLEAN_SUMSQ, 1);
    else
        LAUNCH2(KIMG_CLEAN_SUMSQ, 4);
#undef LAUNCH2
#undef LAUNCH
    return kimg_launch_status();
}

// hipGraphs of MULTI_GRAPH launches, cached per argument set (as cycles_graph in clean.hip)
constexpr int MULTI_GRAPH = 16;
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
    // (captured on the library's own stream of this thread, launched on the caller's: see
    // kimg_capture_stream)
    hipStream_t cs = kimg_capture_stream();
    if (cs == nullptr || hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal) != hipSuccess)
        return nullptr;
    int rc = 0;
    // (the last launch of a graph of the repeated-steps kernel plans single steps: whatever comes
    // next, of either kind, can read what it leaves behind)
    for (int i = 0; i < MULTI_GRAPH && rc == 0; i++)
        rc = enqueue_launch(a, cs, i & 1, a.repeats != 0, i == MULTI_GRAPH - 1 ? 1 : a.g.rmax);
    const hipError_t ended = hipStreamEndCapture(cs, &graph);
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
    if (nb > MC_THREADS || tiles_x > MC_MAX_TILES || tiles_y > MC_MAX_TILES
        || (int64_t) tiles_x * tiles_y < 4)         // (the list's scan reads the tile maxima four at a time)
        return 0;
    int seg = 16;
    while (seg < nb)
        seg *= 2;
    const int m = MC_THREADS / seg;
    return m < MC_MAX ? m : MC_MAX;
}

size_t kimg_clean_multi_state_bytes(int tiles_x, int tiles_y)
{
    return sizeof(mc_scratch) + (size_t) tiles_x * tiles_y * 4 * sizeof(float) + 1024;     // (+ stamps of a test build: 3 rows of 32 words)
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
                         int tiles_x, int tiles_y, int max_cycles, int components, int repeats,
                         bool relative, double noise_threshold, double left_for_next,
                         void *state, float *log, hipStream_t s, int *cycles_done, float *first_peak)
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
    // (a peak is stepped up to MC_STEPS times per launch, half that with several polarizations)
    a.g.rmax = num_polarizations == 1 ? MC_STEPS : MC_STEPS / 2;
    if ((repeats & 0xf) > 0 && (repeats & 0xf) < a.g.rmax)
        a.g.rmax = repeats & 0xf;
    const bool always_repeat = (repeats & 0x10) != 0 && a.g.rmax > 1;
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
                                                threshold, (int) gen, mode, relative ? 1 : 0, noise_threshold,
                                                left_for_next, a.progress);
        rc = kimg_launch_status();
    }
    // Pace: keep the device a graph or two ahead of what has been seen to complete, and stop when
    // the keeper says done.  Near the end the number of launches still needed is estimated from the
    // components per launch so far, so that few launches run after the loop has ended (each costs a
    // kernel boundary).
    // Which kernel: single steps (the first graph goes out at once) unless the start-up kernel's
    // estimate says the field is a dominated one, or until the keeper reports that the repeated steps
    // it could have planned are 0.4 of what a window committed; then the repeated-steps kernel
    // for as long as repeated steps are a fifth of what gets committed, judged over windows of
    // launches that ran wholly under the choice; a try that did not help is not repeated for a while
    // (twice as long each time in a row).
    multi_graph *graphs[2] = {nullptr, nullptr};
    multi_args variants[2] = {a, a};
    variants[1].repeats = 1;
    const bool can_repeat = a.g.rmax > 1 && !always_repeat;
    bool repeating = always_repeat, hinted = false;
    int choice_from = 0;                // launches enqueued before the current choice
    int wait_until = 0, backoff = 1;    // no new try before so many launches are enqueued
    int streak = 0;                     // windows in a row that showed steps to repeat
    int win_l = 0, win_c = 0, win_r = 0;
    int enqueued = 0;
    const double t_start = now_s();
    double t_progress = t_start;
    unsigned long long last = ~0ull;
    while (rc == 0) {
        const unsigned long long word = seen[0];
        if (word != last) {
            last = word;
            t_progress = now_s();
        }
        const bool started = ((word >> 32) & 0xffu) == gen;     // (else: a word of an earlier call)
        const int launches = started ? (int) (word >> 40) : 0;
        // (the word of launch 0 carries the start-up kernel's estimate of the steps there are to repeat)
        const int count = started && launches > 0 ? (int) (word & 0x7fffffffu) : 0;
        if (started && (word & 0x80000000u))
            break;
        if (can_repeat && !hinted && started) {
            // (the estimate arrives with the word of launch 0; if the first launch has overtaken
            // it, the windows below decide)
            hinted = true;
            if (launches == 0 && (int) (word & 0x7fffffffu) >= 2 * MULTI_GRAPH) {
                repeating = true;
                choice_from = enqueued;
            }
        }
        if (can_repeat && started && launches > 0) {
            const unsigned long long word1 = seen[1];           // (written before the word read above)
            const int repeated = (word1 >> 32) == gen ? (int) (unsigned) word1 : win_r;
            if (win_l < choice_from) {
                if (launches >= choice_from) {
                    win_l = launches;
                    win_c = count;
                    win_r = repeated;
                }
            } else if (launches - win_l >= 8) {
                // (under the single-step kernel: the repeated steps that could have been planned, as a
                // share of what was committed; under the other: those that were committed.  A launch
                // of the repeated-steps kernel is a sixth longer.)
                const double share = count > win_c ? (double) (repeated - win_r) / (count - win_c) : 0.0;
                if (!repeating) {
                    // (two windows in a row: what is enqueued now runs sixteen to thirty-two launches
                    // from now, and a dominated phase that is over by then makes the change a loss)
                    streak = share >= 0.4 ? streak + 1 : 0;
                    if (streak >= 2 && enqueued >= wait_until) {
                        repeating = true;
                        choice_from = enqueued;
                        streak = 0;
                    }
                } else if (share < 0.2) {
                    repeating = false;
                    choice_from = enqueued;
                    wait_until = enqueued + MULTI_GRAPH * backoff;
                    backoff = backoff < 32 ? 2 * backoff : 32;
                } else {
                    backoff = 1;
                }
                win_l = launches;
                win_c = count;
                win_r = repeated;
            }
        }
        const int in_flight = enqueued - launches;
        // components per launch so far (at least 1, optimistic before anything is known)
        const double per = launches > 0 && count > 0 ? (double) count / launches : (double) mmax;
        int need = (int) ((max_cycles - count) / per) + 2 - in_flight;
        // (two graphs ahead; one while the start-up kernel's estimate is not in: what is enqueued
        // before it runs with single steps)
        const int ahead = can_repeat && !hinted ? MULTI_GRAPH : 2 * MULTI_GRAPH;
        if (need > ahead - in_flight)
            need = ahead - in_flight;
        if (in_flight == 0 && need < 2)
            need = 2;
        const int v = repeating ? 1 : 0;
        if (enqueued == 0 && need >= MULTI_GRAPH) {
            // (a graph takes the host tens of microseconds to launch, the start-up kernels are over
            // sooner: four plain launches bridge the gap)
            for (int i = 0; i < 4 && rc == 0; i++)
                rc = enqueue_launch(a, s, i & 1, repeating, i == 3 ? 1 : a.g.rmax);
            enqueued += 4;
            continue;
        }
#ifndef KIMG_MC_NO_GRAPH
        if (need >= MULTI_GRAPH && !graphs[v])
            graphs[v] = multi_graph_for(variants[v], s);
#endif
        if (need >= MULTI_GRAPH && graphs[v]) {
            he = hipGraphLaunch(graphs[v]->exec, s);
            if (he != hipSuccess)
                rc = -(int) he;
            enqueued += MULTI_GRAPH;
            continue;
        }
        if (need >= 2 || (need > 0 && in_flight == 0)) {
            // (short of a graph: two launches, or four with repeated steps, the last of which plans
            // single steps again)
            const int n = repeating && need >= 4 ? 4 : 2;
            for (int i = 0; i < n && rc == 0; i++)
                rc = enqueue_launch(a, s, i & 1, repeating, i == n - 1 ? 1 : a.g.rmax);
            enqueued += n;
            continue;
        }
        // nothing to enqueue: wait for the device
        if (now_s() - t_progress > 2e-3) {
            const hipError_t busy = hipStreamQuery(s);
            if (busy != hipSuccess && busy != hipErrorNotReady) {
                rc = -(int) busy;       // (a launch failed: the word will never move)
                break;
            }
            if (busy == hipSuccess) {
                // the stream is idle but the word has not moved: read the state itself
                mc_state st0;
                he = hipMemcpy(&st0, &a.scratch->st[enqueued & 1], 32, hipMemcpyDeviceToHost);
                if (he != hipSuccess) {
                    rc = -(int) he;
                    break;
                }
                seen[0] = progress_word(st0.launches, (int) gen, st0.done != 0, st0.count);
                if (st0.launches < enqueued && now_s() - t_progress > 10.0) {
                    rc = KIMG_ETIMEOUT; // (everything enqueued has run, and the state says it has not)
                    break;
                }
                continue;
            }
            if (now_s() - t_progress > 120.0) {
                rc = KIMG_ETIMEOUT;     // (the stream is busy with something that does not end)
                break;
            }
        }
        sched_yield();
    }
    if (rc == 0 && (cycles_done || first_peak)) {
        // what the caller can have without reading the device back: the count of the word that said
        // done, and the first component's metric (the keeper's word of the launch that committed it)
        const unsigned long long word = seen[0], word2 = seen[2];
        int count = (int) (word & 0x7fffffffu);
        float peak = 0.0f;
        bool have_peak = (word2 >> 32) == gen;
        if (have_peak) {
            const unsigned bits = (unsigned) word2;
            memcpy(&peak, &bits, sizeof(peak));
        }
        if (first_peak && !have_peak && count > 0) {
            // (the word did not arrive -- the fall-back path above ended the loop: read the log)
            he = hipMemcpyAsync(&peak, log, sizeof(float), hipMemcpyDeviceToHost, s);
            if (he == hipSuccess)
                he = hipStreamSynchronize(s);
            if (he != hipSuccess)
                rc = -(int) he;
        }
        if (cycles_done)
            *cycles_done = count;
        if (first_peak)
            *first_peak = peak;
    }
    for (int v = 0; v < 2; v++)
        if (graphs[v])
            multi_graph_release(graphs[v], s);
    progress_release(slot);
    return rc;
}

// (kimg_preload, api.hip)
KIMG_PRELOAD_THIS_UNIT(mc_tile_pix_kernel)
