/* CPU restatement of the katsdpimager host loops -- TEST INFRASTRUCTURE ONLY.
 *
 * Parity oracle and timed single-thread CPU baseline ("port") for the HIP
 * path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library; the product (katsdpimager_amd, libkimg.so) never does.
 *
 * Each function restates one numba-jitted loop of the reference (file:line
 * relative to ska-sa/katsdpimager @ 2024_10_08), with the same loop order and
 * the same float32/complex64 arithmetic.  Built with -ffp-contract=off: numba
 * does not contract a*b+c into FMA.  Pinned against golden vectors produced by
 * the imported reference (tests/test_oracle_golden.py).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

int oracle_version(void) { return 1; }

typedef struct { float re, im; } c64;
typedef struct { double re, im; } c128;

static inline c64 c64_mul(c64 a, c64 b)
{
    c64 r;
    r.re = a.re * b.re - a.im * b.im;
    r.im = a.re * b.im + a.im * b.re;
    return r;
}

/* _grid, grid.py:1032-1052, complex64 grid.
 * kernel [W][OV][K] c64; grid [P][G][G]; weights_grid [P][G][G] f32;
 * uv, sub_uv [N][2] i16; w_plane [N] i16; vis [N][P] c64. */
void oracle_grid_c64(const c64 *kernel, int oversample, int ksize,
                     c64 *grid, int P, int G, const float *weights_grid,
                     const int16_t *uv, const int16_t *sub_uv, const int16_t *w_plane,
                     const c64 *vis, long N)
{
    const int uv_bias = (ksize - 1) / 2 - G / 2;               /* :1038 */
    const long plane = (long) G * G;
    c64 sample[16];
    for (long row = 0; row < N; row++) {
        const int u0 = uv[2 * row] - uv_bias;                  /* :1040 */
        const int v0 = uv[2 * row + 1] - uv_bias;
        const int sub_u = sub_uv[2 * row], sub_v = sub_uv[2 * row + 1];
        const int wu = uv[2 * row] + G / 2;                    /* :1043 */
        const int wv = uv[2 * row + 1] + G / 2;
        for (int i = 0; i < P; i++) {                          /* :1045-1046 */
            float wgt = weights_grid[i * plane + (long) wv * G + wu];
            sample[i].re = vis[row * P + i].re * wgt;
            sample[i].im = vis[row * P + i].im * wgt;
        }
        const c64 *kv = kernel + ((long) w_plane[row] * oversample + sub_v) * ksize;
        const c64 *ku = kernel + ((long) w_plane[row] * oversample + sub_u) * ksize;
        for (int j = 0; j < ksize; j++)
            for (int k = 0; k < ksize; k++) {
                c64 ks = c64_mul(kv[j], ku[k]);                /* :1049 */
                ks.im = -ks.im;                                /* :1050 conj */
                for (int pol = 0; pol < P; pol++) {
                    c64 upd = c64_mul(sample[pol], ks);
                    c64 *g = &grid[pol * plane + (long) (v0 + j) * G + (u0 + k)];
                    g->re += upd.re;                           /* :1052 */
                    g->im += upd.im;
                }
            }
    }
}

/* _grid with a complex128 grid (the reference's own unit test runs float64,
 * test_grid.py:33-42).  sample[] has the grid dtype but is filled from a
 * complex64 product; kernel_sample stays complex64. */
void oracle_grid_c128(const c64 *kernel, int oversample, int ksize,
                      c128 *grid, int P, int G, const float *weights_grid,
                      const int16_t *uv, const int16_t *sub_uv, const int16_t *w_plane,
                      const c64 *vis, long N)
{
    const int uv_bias = (ksize - 1) / 2 - G / 2;
    const long plane = (long) G * G;
    c128 sample[16];
    for (long row = 0; row < N; row++) {
        const int u0 = uv[2 * row] - uv_bias;
        const int v0 = uv[2 * row + 1] - uv_bias;
        const int sub_u = sub_uv[2 * row], sub_v = sub_uv[2 * row + 1];
        const int wu = uv[2 * row] + G / 2;
        const int wv = uv[2 * row + 1] + G / 2;
        for (int i = 0; i < P; i++) {
            float wgt = weights_grid[i * plane + (long) wv * G + wu];
            sample[i].re = (float) (vis[row * P + i].re * wgt);
            sample[i].im = (float) (vis[row * P + i].im * wgt);
        }
        const c64 *kv = kernel + ((long) w_plane[row] * oversample + sub_v) * ksize;
        const c64 *ku = kernel + ((long) w_plane[row] * oversample + sub_u) * ksize;
        for (int j = 0; j < ksize; j++)
            for (int k = 0; k < ksize; k++) {
                c64 ks = c64_mul(kv[j], ku[k]);
                double wr = ks.re, wi = -ks.im;
                for (int pol = 0; pol < P; pol++) {
                    c128 *g = &grid[pol * plane + (long) (v0 + j) * G + (u0 + k)];
                    g->re += sample[pol].re * wr - sample[pol].im * wi;
                    g->im += sample[pol].re * wi + sample[pol].im * wr;
                }
            }
    }
}

/* _degrid, grid.py:1138-1154, complex64 grid.  vis modified in place. */
void oracle_degrid_c64(const c64 *kernel, int oversample, int ksize,
                       const c64 *values, int P, int G,
                       const int16_t *uv, const int16_t *sub_uv, const int16_t *w_plane,
                       const float *weights, c64 *vis, long N)
{
    const int uv_bias = (ksize - 1) / 2 - G / 2;               /* :1141 */
    const long plane = (long) G * G;
    c64 sample[16];
    for (long row = 0; row < N; row++) {
        const int u0 = uv[2 * row] - uv_bias;
        const int v0 = uv[2 * row + 1] - uv_bias;
        const int sub_u = sub_uv[2 * row], sub_v = sub_uv[2 * row + 1];
        for (int i = 0; i < P; i++)
            sample[i].re = sample[i].im = 0.0f;
        const c64 *kv = kernel + ((long) w_plane[row] * oversample + sub_v) * ksize;
        const c64 *ku = kernel + ((long) w_plane[row] * oversample + sub_u) * ksize;
        for (int j = 0; j < ksize; j++)
            for (int k = 0; k < ksize; k++) {
                c64 wgt = c64_mul(kv[j], ku[k]);               /* :1150 */
                for (int pol = 0; pol < P; pol++) {
                    c64 t = c64_mul(wgt, values[pol * plane + (long) (v0 + j) * G + (u0 + k)]);
                    sample[pol].re += t.re;                    /* :1152 */
                    sample[pol].im += t.im;
                }
            }
        for (int i = 0; i < P; i++) {                          /* :1154 */
            float w = weights[row * P + i];
            vis[row * P + i].re -= w * sample[i].re;
            vis[row * P + i].im -= w * sample[i].im;
        }
    }
}

/* _degrid with complex128 grid values (test_grid.py:114-135 runs float64). */
void oracle_degrid_c128(const c64 *kernel, int oversample, int ksize,
                        const c128 *values, int P, int G,
                        const int16_t *uv, const int16_t *sub_uv, const int16_t *w_plane,
                        const float *weights, c64 *vis, long N)
{
    const int uv_bias = (ksize - 1) / 2 - G / 2;
    const long plane = (long) G * G;
    c128 sample[16];
    for (long row = 0; row < N; row++) {
        const int u0 = uv[2 * row] - uv_bias;
        const int v0 = uv[2 * row + 1] - uv_bias;
        const int sub_u = sub_uv[2 * row], sub_v = sub_uv[2 * row + 1];
        for (int i = 0; i < P; i++)
            sample[i].re = sample[i].im = 0.0;
        const c64 *kv = kernel + ((long) w_plane[row] * oversample + sub_v) * ksize;
        const c64 *ku = kernel + ((long) w_plane[row] * oversample + sub_u) * ksize;
        for (int j = 0; j < ksize; j++)
            for (int k = 0; k < ksize; k++) {
                c64 wgt = c64_mul(kv[j], ku[k]);
                double wr = wgt.re, wi = wgt.im;
                for (int pol = 0; pol < P; pol++) {
                    c128 g = values[pol * plane + (long) (v0 + j) * G + (u0 + k)];
                    sample[pol].re += wr * g.re - wi * g.im;
                    sample[pol].im += wr * g.im + wi * g.re;
                }
            }
        for (int i = 0; i < P; i++) {
            double w = weights[row * P + i];
            /* vis is complex64: the subtraction result is rounded to float */
            vis[row * P + i].re = (float) (vis[row * P + i].re - w * sample[i].re);
            vis[row * P + i].im = (float) (vis[row * P + i].im - w * sample[i].im);
        }
    }
}

/* _predict_host, predict.py:419-438.  lmn [S][3] f32 (l, m, n-1); flux [S][P]. */
void oracle_predict(c64 *vis, const int16_t *uv, const int16_t *sub_uv, const int16_t *w_plane,
                    const float *weights, const float *lmn, const float *flux,
                    long N, int S, int P,
                    float oversample, float uv_scale, float w_scale, float w_bias)
{
    const float m2pi = (float) (-2.0 * M_PI);                  /* complex64(-2j*pi).imag */
    c64 accum[16];
    for (long i = 0; i < N; i++) {
        float u = (uv[2 * i] * oversample + sub_uv[2 * i] + 0.5f) * uv_scale;         /* :428 */
        float v = (uv[2 * i + 1] * oversample + sub_uv[2 * i + 1] + 0.5f) * uv_scale; /* :429 */
        float w = w_plane[i] * w_scale + w_bias;                                      /* :430 */
        for (int p = 0; p < P; p++)
            accum[p].re = accum[p].im = 0.0f;
        for (int j = 0; j < S; j++) {
            float phase = lmn[3 * j] * u + lmn[3 * j + 1] * v + lmn[3 * j + 2] * w;   /* :433 */
            float arg = m2pi * phase;
            c64 rot;
            rot.re = (float) cos((double) arg);                                       /* :434 */
            rot.im = (float) sin((double) arg);
            for (int p = 0; p < P; p++) {
                accum[p].re += rot.re * flux[j * P + p];                              /* :436 */
                accum[p].im += rot.im * flux[j * P + p];
            }
        }
        for (int p = 0; p < P; p++) {
            float wgt = weights[i * P + p];
            vis[i * P + p].re -= accum[p].re * wgt;                                   /* :437-438 */
            vis[i * P + p].im -= accum[p].im * wgt;
        }
    }
}

/* CleanHost._update_tile + _tile_peak, clean.py:946-968, 1003-1012, for the
 * tile range [ty0,ty1) x [tx0,tx1).  image [P][H][W] f32; tile_pos [ty][tx][2]. */
void oracle_update_tiles(const float *image, int P, int H, int W, int border, int tile_size,
                         int mode, float *tile_max, int32_t *tile_pos, int tiles_x,
                         int ty0, int tx0, int ty1, int tx1)
{
    const long plane = (long) H * W;
    for (int ty = ty0; ty < ty1; ty++)
        for (int tx = tx0; tx < tx1; tx++) {
            int x0 = tx * tile_size + border;
            int y0 = ty * tile_size + border;
            int x1 = x0 + tile_size < W - border ? x0 + tile_size : W - border;
            int y1 = y0 + tile_size < H - border ? y0 + tile_size : H - border;
            int best0 = x0, best1 = y0;        /* clean.py:950: best_pos = (x0, y0) */
            float best = 0.0f;
            for (int y = y0; y < y1; y++)
                for (int x = x0; x < x1; x++) {
                    float value;
                    if (mode == 0)
                        value = fabsf(image[(long) y * W + x]);
                    else {
                        value = 0.0f;
                        for (int pol = 0; pol < P; pol++) {
                            float pix = image[pol * plane + (long) y * W + x];
                            value += pix * pix;
                        }
                    }
                    if (value > best) {
                        best = value;
                        best0 = y;
                        best1 = x;
                    }
                }
            tile_max[ty * tiles_x + tx] = best;
            tile_pos[2 * (ty * tiles_x + tx)] = best0;
            tile_pos[2 * (ty * tiles_x + tx) + 1] = best1;
        }
}

/* ------------------------------------------------------------------------
 * Visibility preprocessing, preprocess.cpp:313-372 and :390-513.
 *
 * The reference's _preprocess extension needs Eigen3 and cannot be built in
 * this image; this restates its arithmetic.  Pinned by the known-answer
 * vectors of test_preprocess.py:76-136 (tests/test_oracle_known_answers.py).
 * One deliberate choice where the reference leaves the order to Eigen's
 * unroller: the Mueller products are summed in index order.
 * ---------------------------------------------------------------------- */
typedef struct {            /* channel_config, preprocess.cpp:54-61 */
    float max_w;
    int32_t w_slices, w_planes, oversample;
    float cell_size;
} pp_config;

static inline c64 mulz_c(c64 a, c64 b)        /* MulZ<complex<float>>::operator*, mulz.h:37-40 */
{
    c64 z = {0.0f, 0.0f};
    int a_nz = (a.re != 0.0f) || (a.im != 0.0f);
    int b_nz = (b.re != 0.0f) || (b.im != 0.0f);
    return (a_nz && b_nz) ? c64_mul(a, b) : z;
}

static inline float mulz_f(float a, float b)
{
    return (a != 0.0f && b != 0.0f) ? a * b : 0.0f;
}

static void pp_subpixel(float x, int32_t oversample, int16_t *pixel, int16_t *sub)   /* :313-323 */
{
    int32_t xs = (int32_t) floorf(x * (float) oversample);
    int32_t p = xs / oversample, s = xs % oversample;
    if (s < 0) { p--; s += oversample; }
    *pixel = (int16_t) p;
    *sub = (int16_t) s;
}

/* Per-visibility conversion of add_impl2 (:435-507) for one channel.
 * uvw [n][3]; weights, vis [n][Q]; feed angles [n] or NULL (then `stokes` is the P x Q matrix
 * used directly, :198-205); otherwise stokes is P x 4 and circular 4 x Q (:216-258).
 * Outputs: key [n][6] = (u, v, sub_u, sub_v, w_plane, w_slice); out_w [n][P]; out_vis [n][P].
 * Flagged inputs give an all-zero record (:446-454). */
void oracle_pp_convert(int P, int Q, long n, const float *uvw, const float *weights, const c64 *vis,
                       const float *fa1, const float *fa2, const c64 *stokes, const c64 *circular,
                       const pp_config *conf, int16_t *key, float *out_w, c64 *out_vis)
{
    const float uv_scale = 1.0f / conf->cell_size;                                       /* :428 */
    const float w_scale = (conf->w_slices - 0.5f) * conf->w_planes / conf->max_w;        /* :429 */
    const int max_slice_plane = conf->w_slices * conf->w_planes - 1;                     /* :430 */
    for (long i = 0; i < n; i++) {
        const float *wi = weights + i * Q;
        const c64 *vi = vis + i * Q;
        int flagged = 0;
        for (int q = 0; q < Q; q++)
            if (wi[q] == 0.0f) flagged = 1;
        if (flagged) {
            memset(key + 6 * i, 0, 6 * sizeof(int16_t));
            memset(out_w + i * P, 0, P * sizeof(float));
            memset(out_vis + i * P, 0, P * sizeof(c64));
            continue;
        }
        c64 M[4][4];
        if (!fa1) {
            for (int p = 0; p < P; p++)
                for (int q = 0; q < Q; q++) M[p][q] = stokes[p * Q + q];
        } else {                                                                         /* :244-258 */
            c64 r1 = {cosf(fa1[i]), sinf(fa1[i])}, r2 = {cosf(fa2[i]), sinf(fa2[i])};
            c64 r2c = {r2.re, -r2.im};
            c64 rr = c64_mul(r1, r2c), rl = c64_mul(r1, r2);
            c64 scale[4] = {rr, rl, {rl.re, -rl.im}, {rr.re, -rr.im}};
            c64 mu[4][4];
            for (int k = 0; k < 4; k++)
                for (int q = 0; q < Q; q++) mu[k][q] = c64_mul(circular[k * Q + q], scale[k]);
            for (int p = 0; p < P; p++)
                for (int q = 0; q < Q; q++) {
                    c64 acc = {0.0f, 0.0f};
                    for (int k = 0; k < 4; k++) {
                        c64 t = c64_mul(stokes[p * 4 + k], mu[k][q]);
                        acc.re += t.re;
                        acc.im += t.im;
                    }
                    M[p][q] = acc;
                }
        }
        c64 xvis[4];
        float xw[4];
        for (int p = 0; p < P; p++) {
            c64 acc = {0.0f, 0.0f};
            float var = 0.0f;
            for (int q = 0; q < Q; q++) {
                c64 t = mulz_c(M[p][q], vi[q]);                                          /* :456 */
                acc.re += t.re;
                acc.im += t.im;
                float m2 = M[p][q].re * M[p][q].re + M[p][q].im * M[p][q].im;            /* cwiseAbs2 */
                var += mulz_f(m2, 1.0f / fabsf(wi[q]));                                  /* :468-471 */
            }
            xvis[p] = acc;
            xw[p] = 1.0f / var;
        }
        float u = uvw[3 * i], v = uvw[3 * i + 1], w = uvw[3 * i + 2];
        if (w < 0.0f) {                                                                  /* :476-482 */
            u = -u; v = -v; w = -w;
            for (int p = 0; p < P; p++) xvis[p].im = -xvis[p].im;
        }
        for (int p = 0; p < P; p++) {                                                    /* :483-496 */
            float weight = xw[p];
            c64 s = {xvis[p].re * weight, xvis[p].im * weight};
            if (!isfinite(s.re) || !isfinite(s.im)) {
                s.re = s.im = 0.0f;
                weight = 0.0f;
            }
            out_vis[i * P + p] = s;
            out_w[i * P + p] = weight;
        }
        u = u * uv_scale;
        v = v * uv_scale;
        w = truncf(w * w_scale + conf->w_planes * 0.5f);                                 /* :501 */
        int wsp = (int) w;
        if (wsp > max_slice_plane) wsp = max_slice_plane;
        int16_t *k = key + 6 * i;
        pp_subpixel(u, conf->oversample, &k[0], &k[2]);
        pp_subpixel(v, conf->oversample, &k[1], &k[3]);
        k[4] = (int16_t) (wsp % conf->w_planes);
        k[5] = (int16_t) (wsp / conf->w_planes);
    }
}

/* compress (:334-372): drop flagged (weights[0] == 0), merge adjacent equal keys with sequential
 * float32 sums, stable bucket sort by w_slice.  counts[w_slices] receives the run length per slice;
 * returns the number of output records (written to out_* in slice order). */
long oracle_pp_compress(int P, long n, int w_slices, const int16_t *key, const float *w, const c64 *vis,
                        int16_t *out_key, float *out_w, c64 *out_vis, long *counts,
                        int16_t *tmp_key, float *tmp_w, c64 *tmp_vis)
{
    long out_pos = 0, i = 0;
    for (int s = 0; s < w_slices; s++) counts[s] = 0;
    while (i < n && w[i * P] == 0.0f) i++;
    if (i == n) return 0;
    int16_t last_key[6];
    float last_w[4];
    c64 last_vis[4];
    memcpy(last_key, key + 6 * i, sizeof(last_key));
    memcpy(last_w, w + i * P, P * sizeof(float));
    memcpy(last_vis, vis + i * P, P * sizeof(c64));
    for (i++; i <= n; i++) {
        int end = (i == n);
        if (!end && w[i * P] == 0.0f) continue;
        if (!end && memcmp(key + 6 * i, last_key, sizeof(last_key)) == 0) {
            for (int p = 0; p < P; p++) {
                last_vis[p].re += vis[i * P + p].re;
                last_vis[p].im += vis[i * P + p].im;
            }
            for (int p = 0; p < P; p++) last_w[p] += w[i * P + p];
        } else {
            counts[last_key[5]]++;
            memcpy(tmp_key + 6 * out_pos, last_key, sizeof(last_key));
            memcpy(tmp_w + out_pos * P, last_w, P * sizeof(float));
            memcpy(tmp_vis + out_pos * P, last_vis, P * sizeof(c64));
            out_pos++;
            if (!end) {
                memcpy(last_key, key + 6 * i, sizeof(last_key));
                memcpy(last_w, w + i * P, P * sizeof(float));
                memcpy(last_vis, vis + i * P, P * sizeof(c64));
            }
        }
    }
    long start[1024];
    long sum = 0;
    for (int s = 0; s < w_slices; s++) { start[s] = sum; sum += counts[s]; }
    for (long j = 0; j < out_pos; j++) {
        long d = start[tmp_key[6 * j + 5]]++;
        memcpy(out_key + 6 * d, tmp_key + 6 * j, 6 * sizeof(int16_t));
        memcpy(out_w + d * P, tmp_w + j * P, P * sizeof(float));
        memcpy(out_vis + d * P, tmp_vis + j * P, P * sizeof(c64));
    }
    return out_pos;
}
