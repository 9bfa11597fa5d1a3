/* libkimg -- C ABI of the MI355X (gfx950) imaging hot path.
 *
 * Drop-in boundary for the per-channel imaging loop of ska-sa/katsdpimager:
 * each entry point replaces one device-kernel launch site of the reference's
 * operator classes (file:line relative to the reference checkout).  The
 * reference reaches its kernels through katsdpsigproc's
 * `command_queue.enqueue_kernel(...)`; a maintainer binds these functions with
 * ctypes/cffi instead.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller unless the name
 *    ends in `_host`; nothing is allocated or freed here except FFT plans;
 *  - arrays come with explicit element strides (row, polarization) exactly
 *    like the reference kernels' arguments;
 *  - `stream` is a hipStream_t (NULL = default stream); all calls are
 *    asynchronous on it and may be captured into a hipGraph unless noted;
 *  - return value: 0 on success, a negated hipError_t on a HIP failure, or a
 *    KIMG_E* code below for argument errors.  Kernels themselves never report
 *    errors (same as the reference).
 *  - complex numbers are interleaved float (re, im) = numpy complex64.
 */
#ifndef KIMG_H
#define KIMG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KIMG_VERSION 4

#define KIMG_EINVAL (-10001)      /* bad argument (null pointer, negative size ...) */
#define KIMG_EUNSUPPORTED (-10002) /* parameter combination not supported by this build */
#define KIMG_EWORKSPACE (-10003)   /* workspace too small */
#define KIMG_ETIMEOUT (-10004)     /* workgroups of a persistent kernel did not see each other in time
                                    * (reported after the stream was synchronised by the caller: see
                                    * kimg_clean_cycles) */

/* Arithmetic of the gridder / degridder matrix instructions (argument `arith`):
 *   KIMG_ARITH_FP32        v_mfma_f32_32x32x2_f32 -- every product and sum in float32, bit-identical
 *                          to the fmaf chain of the reference's kernels (grid.py:1049-1052).  Default.
 *   KIMG_ARITH_SPLIT_FP16  operands carried as fp16 (hi, lo) pairs (22 significant bits, lo*lo
 *                          dropped), two visibilities per v_mfma_f32_32x32x16_f16, float32
 *                          accumulation.  Faster, narrower than the reference's arithmetic: opt-in.
 * The generic (non-MFMA) kernels always compute in float32 and ignore it. */
#define KIMG_ARITH_FP32 0
#define KIMG_ARITH_SPLIT_FP16 1

/* Kernel choice of kimg_grid / kimg_degrid (argument `variant`) */
#define KIMG_VARIANT_AUTO 0     /* MFMA window kernel when the parameters allow it */
#define KIMG_VARIANT_GENERIC 1  /* one wave per visibility, any kernel width */
#define KIMG_VARIANT_MFMA 2     /* MFMA window kernel or KIMG_EUNSUPPORTED */
#define KIMG_VARIANT_BINNED 3   /* sort the visibilities by grid tile on the device
                                 * (bins of window-slack cells, stable radix sort, gather), then the
                                 * MFMA window kernel over the sorted copies -- for streams without
                                 * locality (time order, shuffled); needs the scratch of
                                 * kimg_grid_binned_workspace_bytes / kimg_degrid_binned_workspace_bytes
                                 * (the degridder also sorts the weights and scatters its results
                                 * back into the caller's order); KIMG_EUNSUPPORTED where MFMA is */

/* Form of the device-resident CLEAN loop (argument `form` of kimg_clean_cycles) */
#define KIMG_CLEAN_FORM_AUTO 0      /* one launch per cycle when the patch's lattice blocks fit the CUs */
#define KIMG_CLEAN_FORM_TWO_LAUNCH 1
#define KIMG_CLEAN_FORM_ONE_LAUNCH 2    /* falls back to two launches when the patch is too large */
#define KIMG_CLEAN_FORM_PERSISTENT 3    /* the whole loop in one launch of resident workgroups (small
                                         * patches: at most 64 lattice blocks, tile maxima in LDS);
                                         * falls back to ONE_LAUNCH otherwise.  Measured slower than
                                         * ONE_LAUNCH (agent-scope hand-offs cost what the kernel
                                         * boundary costs): AUTO does not take it. */
#define KIMG_CLEAN_FORM_ONE_WORKGROUP 4 /* the whole loop in ONE workgroup: tile records and the PSF
                                         * patch in its LDS, no kernel boundary and no hand-off per
                                         * cycle.  One polarization, at most 8 x 8 lattice blocks,
                                         * 6 bytes of LDS per tile + the patch; falls back to AUTO's
                                         * choice otherwise.  Measured slower than ONE_LAUNCH (7.0 vs
                                         * 6.3 us per cycle: one CU's memory pipe): AUTO does not
                                         * take it. */

#define KIMG_CLEAN_FORM_MULTI 5     /* SEVERAL components per launch: a launch verifies the components
                                     * the last one evaluated speculatively, commits the verified
                                     * prefix and evaluates the next up to 8 (csrc/clean_multi.hip);
                                     * results bit-identical to the other forms.  At most 256 lattice
                                     * blocks per patch (8 components per launch up to 32 blocks, 4
                                     * up to 64, 2 up to 128) and 2047 tiles per axis; falls back to
                                     * AUTO's other choices otherwise.  A planned component may be
                                     * stepped several times in a launch (its peak's value follows a
                                     * scalar recursion every workgroup evaluates; up to 8 steps, 4
                                     * with several polarizations).  `form | n << 8` caps the
                                     * components per launch at n (1..8), `| r << 16` the steps per
                                     * component at r (1..8; the loop goes over to repeated steps
                                     * only while it sees few components per launch: `| 1 << 20`
                                     * makes it take them from the first launch on).  AUTO takes this form when
                                     * at least 2 components fit and at least 4 cycles are asked for.
                                     * HOST-PACED: the number of launches depends on the data, so the
                                     * call watches a progress word the device writes and returns
                                     * once the loop has ended -- it blocks the calling thread for
                                     * about as long as the loop runs, and cannot be captured. */

#define KIMG_CLEAN_I 0      /* clean.py:29 */
#define KIMG_CLEAN_SUMSQ 1  /* clean.py:31 */

int kimg_version(void);

/* Load every code object of the library on the current device now, in the calling thread.  The runtime
 * otherwise loads a code object at the first launch of one of its kernels, and first launches that
 * several host threads make at the same moment (channels imaged concurrently) are not safe against
 * that load: call this once per device before any thread uses the library.  Idempotent. */
int kimg_preload(void);
/* Static string describing a return code of this library. */
const char *kimg_error_string(int code);
/* How many of the device's CUs the window kernels (kimg_grid / kimg_degrid, MFMA variants) fill with
 * their resident workgroups: 256 (all; the default, also set by 0) down to 1.  Process-wide, takes
 * effect with the next launch.  A process that keeps several channels in flight on one GPU
 * (frontend.process_channel_stream) sets 192: the window kernels' workgroups stay for a whole launch
 * (0.7 ms on a stored W-slice) and leave no room for another channel's CLEAN workgroups (1024
 * threads, 64 KB of LDS), whose latency-bound chain then stands still; with 64 CUs left free a
 * 12-channel stream with four in flight takes 6.1 instead of 7.2 ms per channel, one channel alone
 * 16.1 instead of 15.6.  No counterpart in the reference (its kernels are not resident). */
int kimg_set_window_cus(int cus);
int kimg_get_window_cus(void);
/* The same PER CALL, which is what callers should use (the process-wide setting above is kept as
 * the default for calls that bring none, and is deprecated: two imagers of one process with
 * different needs would race on it): `variant | KIMG_WINDOW_CUS(n)` in the `variant` argument of
 * kimg_grid / kimg_degrid, n = 1 .. 256; 0 = the default. */
#define KIMG_WINDOW_CUS(n) ((n) << 8)

/* ---- convolution kernel table: grid.py:235-334 antialias_w_kernel, as called for every W plane
 * by ConvolutionKernel.__init__ (grid.py:358-389), evaluated on the device in float64.
 *   table  complex64 [w_planes][oversample][kernel_width] (device, written)
 *   ws     float64 [w_planes] (device): the w of each plane in wavelengths, grid.py:382-383
 *   beta   Kaiser-Bessel shape parameter, grid.py:374-378
 * table[w][s][t] = sample t*oversample + (oversample-1-s) of the central oversample*kernel_width
 * samples of step * FFT(ifftshift(aa(l) exp(2 pi i (-w (-l^2/2 - 5 l^4/24) + half_subcell l)))),
 * l on an image_oversample-times finer grid of step 1/(kernel_width cell_wavelengths
 * image_oversample).  oversample*kernel_width must be even (KIMG_EINVAL, grid.py:268);
 * KIMG_EUNSUPPORTED when oversample*kernel_width*image_oversample > 5120 (the plane's samples and
 * the roots of unity are kept in LDS). */
int kimg_kernel_table(void *table, const double *ws, int w_planes, int kernel_width,
                      int oversample, int image_oversample, double cell_wavelengths,
                      double antialias_width, double beta, void *stream);

/* ---- gridding: grid.py:786-867 Gridder.static_run/_run + imager_kernels/grid.mako:63-197
 * grid[p][v0+j][u0+k] += vis[r][p] * weights_grid[p][v+Gg/2][u+Gg/2]
 *                        * conj(kern[w][sub_v][j] * kern[w][sub_u][k]),
 * u0 = u - ((K-1)/2 - Gg/2)  (grid.py:1038-1041).
 *   grid            complex64 [P][grid_size][grid_size] (strides in complex elements)
 *   weights_grid    float32   [P][grid_size][grid_size] (strides in elements)
 *   uv              int16 [N][4] = (u, v, sub_u, sub_v)       grid.py:661-664
 *   w_plane         int16 [N]
 *   vis             complex64 [N][P]
 *   convolve_kernel complex64 [w_planes][oversample][kernel_width], unpadded
 *   workspace       device scratch, at least kimg_grid_workspace_bytes(max N, P, w_planes,
 *                   oversample, kernel_width) bytes: 256 bytes (the chunk counter from which the
 *                   waves of a long launch draw their work; without it -- NULL is accepted when the
 *                   table fits LDS -- they take their chunks in a fixed order, a few per cent slower)
 *                   plus, when the kernel table is too large for LDS -- more than 512 rows
 *                   w_planes*oversample for widths <= 32, 256 for 33..64 -- a zero-padded copy of it,
 *                   built there on every call.  One call at a time per workspace.
 *   variant         KIMG_VARIANT_*: automatic = MFMA window kernel when supported (kernel_width
 *                   <= 64), else the generic scatter kernel
 *   arith           KIMG_ARITH_* (above); anything else is KIMG_EINVAL
 *   Out-of-range coordinates (footprint outside the grid, sub_uv >= oversample, w_plane >=
 *   w_planes) contribute nothing instead of faulting.
 */
size_t kimg_grid_workspace_bytes(int64_t max_vis, int num_polarizations, int w_planes,
                                 int oversample, int kernel_width);
/* Scratch of kimg_grid with KIMG_VARIANT_BINNED (includes the above): sort keys and indices, the
 * sorted copies of uv / w_plane / vis (18 + 8 P bytes per visibility) and the sort's own scratch. */
size_t kimg_grid_binned_workspace_bytes(int64_t max_vis, int num_polarizations, int w_planes,
                                        int oversample, int kernel_width);
/* How well a visibility stream suits the window kernel: *count (device uint32, zeroed by the call)
 * receives the number of records whose (u, v) cell differs from their predecessor's by more than
 * the window slack (32 - kernel_width; 32 - (kernel_width + 1) / 2 for widths 33..64) along
 * either axis: each of them costs a whole-window flush in the direct window kernel.  A caller may
 * choose KIMG_VARIANT_BINNED when count / num_vis is more than a few percent. */
int kimg_grid_jumps(const int16_t *uv, int64_t num_vis, int kernel_width, uint32_t *count,
                    void *stream);
int kimg_grid(void *grid, int64_t grid_row_stride, int64_t grid_pol_stride, int grid_size,
              int num_polarizations,
              const float *weights_grid, int64_t wg_row_stride, int64_t wg_pol_stride,
              const int16_t *uv, const int16_t *w_plane, const void *vis, int64_t num_vis,
              const void *convolve_kernel, int w_planes, int oversample, int kernel_width,
              void *workspace, size_t workspace_bytes, int variant, int arith, void *stream);

/* ---- degridding: grid.py:985-1029 Degridder.static_run/_run + degrid.mako:77-199
 * vis[r][p] -= weights[r][p] * sum_{j,k} kern[w][sub_v][j]*kern[w][sub_u][k]*grid[p][v0+j][u0+k]
 *   weights  float32 [N][P] statistical weights
 *   variant, arith as for kimg_grid
 */
int kimg_degrid(const void *grid, int64_t grid_row_stride, int64_t grid_pol_stride, int grid_size,
                int num_polarizations,
                const int16_t *uv, const int16_t *w_plane, const float *weights, void *vis,
                int64_t num_vis,
                const void *convolve_kernel, int w_planes, int oversample, int kernel_width,
                void *workspace, size_t workspace_bytes, int variant, int arith, void *stream);
/* Device scratch kimg_degrid needs, as for kimg_grid: 256 bytes (the chunk counter of long launches;
 * optional when the table fits LDS) plus, when the kernel table is too large for LDS, a padded copy
 * of it, built there on every call.  One call at a time per workspace. */
size_t kimg_degrid_workspace_bytes(int num_polarizations, int w_planes, int oversample,
                                   int kernel_width);
/* Scratch of kimg_degrid with KIMG_VARIANT_BINNED (includes the above). */
size_t kimg_degrid_binned_workspace_bytes(int64_t max_vis, int num_polarizations, int w_planes,
                                          int oversample, int kernel_width);

/* ---- direct prediction: predict.py:386-416 Predict._run + predict.mako:10-87
 * vis[r][p] -= weights[r][p] * sum_s flux[s][p] * exp(-2 pi i (l u + m v + (n-1) w)),
 * u = (uv.x*oversample + uv.z + 0.5)*uv_scale, w = w_plane*w_scale + w_bias.
 *   lmn float32 [S][3] (l, m, n-1);  flux float32 [S][P]
 */
int kimg_predict(void *vis, const int16_t *uv, const int16_t *w_plane, const float *weights,
                 const float *lmn, const float *flux, int64_t num_vis, int num_sources,
                 int num_polarizations, int oversample, float uv_scale, float w_scale,
                 float w_bias, void *stream);

/* ---- imaging weights: weight.py:155-176 / 261-284 / 357-376 + grid_weights.mako,
 * density_weights.mako, mean_weight.mako; fill = katsdpsigproc fill.FillTemplate (weight.py:403).
 *   kimg_grid_weights:    grid[p][v+H/2][u+W/2] += weights[r][p]   (uv = first 2 of 4 int16)
 *   kimg_mean_weight:     sums[0] = sum w, sums[1] = sum w^2 over polarization 0
 *   kimg_density_weights: in place d = w != 0 ? 1/(a*w+b) : 0;
 *                         sums[0..2] = sum w, sum d*w, sum d*d*w over polarization 0
 *   `sums` are device float64 and are zeroed by the call.
 */
int kimg_grid_weights(float *grid, int64_t row_stride, int64_t pol_stride, int width, int height,
                      int num_polarizations, const int16_t *uv, const float *weights,
                      int64_t num_vis, void *stream);
int kimg_mean_weight(double *sums, const float *grid, int64_t row_stride, int width, int height,
                     void *stream);
int kimg_density_weights(double *sums, float *grid, int64_t row_stride, int64_t pol_stride,
                         int width, int height, int num_polarizations, float a, float b,
                         void *stream);
/* Robust weighting without the host between the two kernels (weight.py:525-531): a = robust /
 * (mean_sums[1] / mean_sums[0]) on the device, in the host's arithmetic (doubles, rounded to float32),
 * mean_sums = what kimg_mean_weight left on the DEVICE; robust = (5 * 10^-robustness)^2. */
int kimg_density_weights_robust(double *sums, float *grid, int64_t row_stride, int64_t pol_stride,
                                int width, int height, int num_polarizations,
                                const double *mean_sums, double robust, float b, void *stream);
int kimg_fill(float *data, int64_t count, float value, void *stream);

/* ---- visibility preprocessing: preprocess.cpp:390-513 (visibility_collector<P>::add_impl2) and
 * :334-372 (compress); the reference runs these on host cores (OpenMP) behind
 * preprocess.VisibilityCollector.add (preprocess.py:117-150).
 *   kimg_preprocess_convert: one channel, one buffer of num_vis inputs.  Per visibility: drop if any
 *       input weight is 0; xvis = M vis, xweights = 1/(|M|^2 (1/|w|)) with MulZ products; M =
 *       mueller_stokes [P][Q] when the feed angles are NULL, else mueller_stokes [P][4] x
 *       diag(RR, RL, conj RL, conj RR) x mueller_circular [4][Q] (:244-258); w<0 flip + conjugate;
 *       vis *= weight; non-finite -> 0; quantise (subpixel_coord :313-323, w :501-506).
 *       The two matrices are HOST pointers (interleaved re, im float32); everything else is device.
 *       Outputs: key int16 [N][6] = (u, v, sub_u, sub_v, w_plane, w_slice), out_weights float32 [N][P],
 *       out_vis complex64 [N][P]; dropped inputs give an all-zero record.
 *   kimg_preprocess_compress: skip records with weights[0] == 0, sum runs of adjacent equal keys
 *       left to right in float32, then order the results by w_slice keeping arrival order inside a
 *       slice.  Outputs are in the gridder's layout: out_uv int16 [M][4] = (u, v, sub_u, sub_v),
 *       out_w_plane int16 [M], out_weights [M][P], out_vis [M][P], and counts uint64 [w_slices]
 *       (device) = run length per slice, M = sum(counts).  Output arrays need room for N records.
 *       merge_window > 0: runs never cross a multiple of merge_window records -- the call then
 *       compresses num_vis / merge_window of the reference's buffers (preprocess.cpp:431-509: every
 *       buffer is compressed on its own) in ONE pass over the device, with the results the
 *       per-buffer calls would have given, concatenated per slice.  0 = one buffer.
 *       workspace: kimg_preprocess_workspace_bytes(N, P) bytes of device memory.
 *   kimg_real_to_complex: dst[i] = (src[i], 0): feeds weights as visibilities for the PSF pass
 *       (frontend.py:511) from a device-resident store.
 */
int kimg_preprocess_convert(int num_polarizations, int num_input_polarizations, int64_t num_vis,
                            const float *uvw, const float *weights, const void *vis,
                            const float *feed_angle1, const float *feed_angle2,
                            const float *mueller_stokes_host, const float *mueller_circular_host,
                            float max_w, int w_slices, int w_planes, int oversample, float cell_size,
                            int16_t *key, float *out_weights, void *out_vis, void *stream);
size_t kimg_preprocess_workspace_bytes(int64_t num_vis, int num_polarizations);
int kimg_preprocess_compress(int num_polarizations, int64_t num_vis, int w_slices,
                             const int16_t *key, const float *weights, const void *vis,
                             int16_t *out_uv, int16_t *out_w_plane, float *out_weights, void *out_vis,
                             uint64_t *counts, int64_t merge_window, void *workspace,
                             size_t workspace_bytes, void *stream);
int kimg_real_to_complex(void *dst, const float *src, int64_t count, void *stream);

/* ---- once-per-channel re-ordering of a stored W-slice (csrc/store.hip).  No launch site of the
 * reference corresponds to it: the reference's preprocessor leaves a slice in arrival order
 * (baseline-sorted load blocks, adjacent-merged: loader_ms.py:465-468, preprocess.cpp:334-397) and
 * its gridder bins per pass inside the kernel (grid.mako; grid.py:436-463 get_bin_size).  The
 * resident store (preprocess.VisibilityCollectorDevice) calls this once when it is closed; every
 * later pass of the channel runs the window kernels on the result.
 *   order: strips of (32 - taps + 1) grid columns (taps = kernel_width, or (kernel_width + 1) / 2
 *       for widths above 32), each strip sorted by v, odd strips backwards; stable.
 *   merge != 0: the sort also covers (u in strip, sub_v, sub_u, w_plane), and every run of records
 *       with equal (u, v, sub_u, sub_v, w_plane) becomes ONE record whose weights and visibilities
 *       are the run's sums, added left to right in float32 in arrival order -- compress() of
 *       preprocess.cpp:334-372 applied to whole-slice runs instead of arrival runs.
 *   Outputs need room for num_vis records; *out_count (device uint64) = records written.
 *   workspace: kimg_store_reorder_workspace_bytes(num_vis) bytes (28 per record + sort scratch). */
size_t kimg_store_reorder_workspace_bytes(int64_t num_vis);
int kimg_store_reorder(int num_polarizations, int64_t num_vis, int kernel_width, int oversample,
                       int w_planes, int merge, const int16_t *uv, const int16_t *w_plane,
                       const float *weights, const void *vis, int16_t *out_uv, int16_t *out_w_plane,
                       float *out_weights, void *out_vis, uint64_t *out_count, void *workspace,
                       size_t workspace_bytes, void *stream);

/* ---- grid <-> image: image.py:649-673 GridToImage._run, :716-740 ImageToGrid._run,
 * :153-180 _LayerImage._run + layer_to_image.mako / image_to_layer.mako.
 *   kimg_grid_to_layer: zero the GxG layer and copy the centred Gg x Gg grid of one
 *       polarization into its corners (DC at [0][0]) -- the reference's zero + 4 copy_region.
 *   kimg_layer_to_grid: the inverse copy (corners -> centred grid).
 *   kimg_layer_to_image: image[pol] += Re(layer * e^{2 pi i w (n-1)}) * n / (k1d[y] k1d[x]),
 *       with fftshift; l = x*lm_scale + lm_bias.  layer is row-contiguous GxG.
 *   kimg_image_to_layer: layer = image[pol] / (k1d[y] k1d[x] n) * e^{-2 pi i w (n-1)}.
 */
int kimg_grid_to_layer(void *layer, int layer_size, const void *grid, int64_t grid_row_stride,
                       int grid_size, void *stream);
int kimg_layer_to_grid(void *grid, int64_t grid_row_stride, int grid_size, const void *layer,
                       int layer_size, void *stream);
int kimg_layer_to_image(float *image, int64_t image_row_stride, const void *layer, int size,
                        const float *kernel1d, float lm_scale, float lm_bias, float w,
                        void *stream);
int kimg_image_to_layer(void *layer, const float *image, int64_t image_row_stride, int size,
                        const float *kernel1d, float lm_scale, float lm_bias, float w,
                        void *stream);
/* The same pair for a layer whose transform is only wanted for its real part, i.e. w = 0 (the
 * phase factor of layer_to_image is then exactly 1): the reference notes at image.py:561-566 that a
 * complex-to-real transform would do; here it does.
 *   kimg_grid_to_half_layer: half_layer[ly][lx], lx = 0 .. G/2 (rows of G/2 + 1 complex values),
 *       = (g(k) + conj g(-k)) / 2 of the zero-padded, corner-DC layer g that kimg_grid_to_layer
 *       would have made: the Hermitian part of g, whose inverse transform is Re F^-1[g].
 *   kimg_real_layer_to_image: image[pol] += layer * n / (k1d[y] k1d[x]) with fftshift, from the REAL
 *       output of kimg_rfft_exec(direction +1); layer_row_stride in floats (G + 2 when the
 *       transform ran in place on the half layer). */
int kimg_grid_to_half_layer(void *half_layer, int layer_size, const void *grid,
                            int64_t grid_row_stride, int grid_size, void *stream);
int kimg_real_layer_to_image(float *image, int64_t image_row_stride, const float *layer,
                             int64_t layer_row_stride, int size, const float *kernel1d,
                             float lm_scale, float lm_bias, void *stream);
/* ... and the other way (ImageToGrid, image.py:676-740) at w = 0, where the layer is real:
 *   kimg_image_to_real_layer: layer = image[pol] / (k1d[y] k1d[x] n) with fftshift, rows of
 *       layer_row_stride floats (G + 2 for the in-place real-to-complex kimg_rfft_exec, direction -1);
 *   kimg_half_layer_to_grid: the centred grid from the half spectrum [G][G/2 + 1] that transform
 *       leaves, F(-k) = conj F(k) for the columns it does not hold. */
int kimg_image_to_real_layer(float *layer, int64_t layer_row_stride, const float *image,
                             int64_t image_row_stride, int size, const float *kernel1d,
                             float lm_scale, float lm_bias, void *stream);
int kimg_half_layer_to_grid(void *grid, int64_t grid_row_stride, int grid_size,
                            const void *half_layer, int layer_size, void *stream);
/* The whole of GridToImage.__call__ / ImageToGrid.__call__ (image.py:609-673, :676-740) for one
 * polarization at w = 0 in two launches, with transforms of the library's own (even layer sizes
 * 16 .. 8192 with no prime factor above 7 -- every size parameters.py:17-25 of the reference picks
 * in that range; mixed radix 4 / 2 / 3 / 5 / 7 in LDS): only the Gg/2 + 1 columns of the half layer the grid reaches
 * are transformed, the fold / padding and the image correction are the prologue and epilogue of
 * the transform kernels, and what passes between the two launches is (Gg/2 + 1) x G cells in
 * `workspace` (16-byte aligned, kimg_grid_image_real_workspace_bytes; the layer buffer will do).
 * Results equal the route above up to the rounding of the transform.  accumulate = 0: the image
 * is written, not added to (what the reference gets by zeroing it first, imaging.py:258-261:
 * saves the fill and the read).
 *   kimg_grid_image_real_supported: 1 when the two functions take these sizes, else 0. */
int kimg_grid_image_real_supported(int layer_size, int grid_size);
size_t kimg_grid_image_real_workspace_bytes(int layer_size, int grid_size);
int kimg_grid_to_image_real(float *image, int64_t image_row_stride, int layer_size,
                            const void *grid, int64_t grid_row_stride, int grid_size,
                            const float *kernel1d, float lm_scale, float lm_bias, int accumulate,
                            void *workspace, size_t workspace_bytes, void *stream);
int kimg_image_to_grid_real(void *grid, int64_t grid_row_stride, int grid_size,
                            const float *image, int64_t image_row_stride, int layer_size,
                            const float *kernel1d, float lm_scale, float lm_bias,
                            void *workspace, size_t workspace_bytes, void *stream);
/* ... and for any w (a slice of the W stack away from w = 0: the layer is complex, image.py:781-799
 * and :836-848 with the phase e^{+-2 pi i w (n - 1)}): the same two launches over the Gg columns the
 * grid reaches and over pairs of rows, each row with a complex transform of its own.  Same sizes as
 * the w = 0 pair (kimg_grid_image_real_supported); workspace Gg x G cells. */
size_t kimg_grid_image_w_workspace_bytes(int layer_size, int grid_size);
int kimg_grid_to_image_w(float *image, int64_t image_row_stride, int layer_size,
                         const void *grid, int64_t grid_row_stride, int grid_size,
                         const float *kernel1d, float lm_scale, float lm_bias, float w,
                         int accumulate, void *workspace, size_t workspace_bytes, void *stream);
int kimg_image_to_grid_w(void *grid, int64_t grid_row_stride, int grid_size,
                         const float *image, int64_t image_row_stride, int layer_size,
                         const float *kernel1d, float lm_scale, float lm_bias, float w,
                         void *workspace, size_t workspace_bytes, void *stream);

/* ConvolveBeam.__call__ (beam.py:351-398) for a square image of a size the functions above take
 * (kimg_grid_image_real_supported(size, size)), in three launches on the library's own transforms:
 * rows (two real rows per complex transform) -> per column of the half spectrum: forward transform,
 * times amplitude * exp((a v + b u) v + c u^2) (kimg_fourier_beam's factor; 1 / (H W) folded into
 * the amplitude as there), inverse transform -> rows back.  In place on `image`; workspace =
 * (size / 2 + 1) x size cells, 16-byte aligned (the operator's `fourier` buffer). */
int kimg_convolve_beam(float *image, int64_t row_stride, int size, float amplitude, float a,
                       float b, float c, void *workspace, size_t workspace_bytes, void *stream);

/* 2-D complex-to-complex FFT plans (katsdpsigproc.fft.FftTemplate, image.py:585-600,629,698)
 * on rocFFT (called directly); unnormalised, in place.  direction: -1 forward, +1 inverse. */
int kimg_fft_plan_create(void **plan, int size_y, int size_x);
int kimg_fft_exec(void *plan, void *layer, int direction, void *stream);
int kimg_fft_plan_destroy(void *plan);

/* Restoring-beam convolution (beam.py:204-398: FourierBeam._run :283-311 + fourier_beam.mako,
 * ConvolveBeam :351-398 = R2C FFT, multiply, C2R FFT), one polarization plane at a time.
 *   kimg_rfft_*: out-of-place real <-> half-complex 2-D plans; image float32 [H][W] dense,
 *       fourier complex64 [H][W/2+1] dense; direction -1 = R2C forward, +1 = C2R inverse
 *       (unnormalised; the C2R transform overwrites `fourier`).
 *   kimg_fourier_beam: data[y][x] *= amplitude * exp((a v + b u) v + c u u), u = x,
 *       v = y < H/2 ? y : y - H  (the caller folds 1/(H W) and the axis scaling into
 *       amplitude, a, b, c exactly as beam.py:287-299).  width = W/2+1 columns. */
int kimg_rfft_plan_create(void **plan, int height, int width);
int kimg_rfft_exec(void *plan, float *image, void *fourier, int direction, void *stream);
int kimg_rfft_plan_destroy(void *plan);
int kimg_fourier_beam(void *data, int64_t row_stride, int width, int height, float amplitude,
                      float a, float b, float c, void *stream);

/* ---- image-plane streams: image.py:351-367, :439-458, :539-558 (+ scale.mako,
 * add_image.mako, apply_primary_beam.mako).  scale_host: P floats on the HOST. */
int kimg_scale(float *image, int64_t row_stride, int64_t pol_stride, int width, int height,
               int num_polarizations, const float *scale_host, void *stream);
/* The scaling of frontend.py:541-545 (dirty and PSF by 1 / the PSF's central pixel) without the host in
 * between: out[p] = 1 / image[p][y][x] on the device (np.reciprocal of a float32), and kimg_scale with
 * its factors read from device memory.  scale, out: P floats on the DEVICE. */
int kimg_pixel_reciprocal(const float *image, int64_t row_stride, int64_t pol_stride, int width,
                          int height, int num_polarizations, int x, int y, float *out, void *stream);
int kimg_scale_device(float *image, int64_t row_stride, int64_t pol_stride, int width, int height,
                      int num_polarizations, const float *scale, void *stream);
int kimg_add_image(float *dest, int64_t dest_row_stride, int64_t dest_pol_stride,
                   const float *src, int64_t src_row_stride, int64_t src_pol_stride,
                   int width, int height, int num_polarizations, void *stream);
int kimg_apply_primary_beam(float *image, int64_t row_stride, int64_t pol_stride,
                            const float *beam_power, int64_t beam_row_stride,
                            int width, int height, int num_polarizations,
                            float threshold, float replacement, void *stream);

/* Output statistics of the restore step (frontend.py:171-209, host loops in the reference):
 *   kimg_image_peak:   find_peak -- max |image| over all polarizations and pixels with
 *       |image| * pbeam[y][x] > 7.5 * noise (pbeam float32 [H][W], may be NULL = 1; NaNs never
 *       pass).  *peak (device float32) receives the maximum, 0 when no pixel qualifies (the
 *       reference returns NaN then; the host wrapper does the same).
 *   kimg_image_nansum: get_totals -- sums[p] (device float64 [P], zeroed by the call) = sum of
 *       the non-NaN pixels of polarization p. */
int kimg_image_peak(const float *image, int64_t row_stride, int64_t pol_stride, const float *pbeam,
                    int64_t beam_row_stride, int width, int height, int num_polarizations,
                    float noise, float *peak, void *stream);
int kimg_image_nansum(const float *image, int64_t row_stride, int64_t pol_stride, int width,
                      int height, int num_polarizations, double *sums, void *stream);

/* ---- CLEAN support: clean.py:123-163 PsfPatch.__call__ + psf_patch.mako
 * bound (device int32[2], zeroed by the call) receives max |x-mid_x|, max |y-mid_y| over
 * pixels in [min_x,max_x]x[min_y,max_y] where any polarization has |psf| >= threshold. */
int kimg_psf_patch(const float *psf, int64_t row_stride, int64_t pol_stride, int num_polarizations,
                   int min_x, int min_y, int max_x, int max_y, int mid_x, int mid_y,
                   float threshold, int32_t *bound, void *stream);

/* Noise estimate support (clean.py:295-353 NoiseEst.__call__ + rank.mako, host semantics of
 * clean.py:938-943): one radix-select pass.  hist (device uint32[256], zeroed by the call)
 * receives the histogram of byte `pass` (3 = most significant) of the bit pattern of |x| over
 * the region inside `border`, restricted to values whose higher bytes equal `prefix`. */
int kimg_abs_histogram(const float *image, int64_t row_stride, int64_t pol_stride,
                       int width, int height, int num_polarizations, int border,
                       int pass, uint32_t prefix, uint32_t *hist, void *stream);
/* out[0] = count of |x| <= value, out[1] = bit pattern of min |x| > value (0xFFFFFFFF if none);
 * out is device uint32[2], initialised by the call. */
int kimg_abs_count_le(const float *image, int64_t row_stride, int64_t pol_stride,
                      int width, int height, int num_polarizations, int border,
                      float value, uint32_t *out, void *stream);

/* ---- CLEAN minor cycle: clean.py:451-480 _UpdateTiles.__call__, :566-587 _FindPeak._run,
 * :683-726 _SubtractPsf.__call__ (+ update_tiles.mako, find_peak.mako, subtract_psf.mako).
 * Tie-breaks follow the reference HOST path bit-exactly: first strict maximum in row-major
 * order within a tile (clean.py:953-958), first maximum tile in row-major order
 * (np.argmax, clean.py:1062).
 *   dirty/model/psf float32 [P][height][width]; tile_max float32 [tiles_y][tiles_x];
 *   tile_pos int32 [tiles_y][tiles_x][2] = (y, x); tiles are 32x32 starting at `border`.
 */
int kimg_update_tiles(const float *dirty, int64_t row_stride, int64_t pol_stride,
                      int width, int height, int num_polarizations, int border, int mode,
                      float *tile_max, int32_t *tile_pos, int tiles_x, int tiles_y,
                      int tile_x0, int tile_y0, int tile_x1, int tile_y1, void *stream);
int kimg_find_peak(const float *dirty, int64_t row_stride, int64_t pol_stride,
                   int num_polarizations, const float *tile_max, const int32_t *tile_pos,
                   int tiles_x, int tiles_y,
                   float *peak_value, int32_t *peak_pos, float *peak_pixel, void *stream);
int kimg_subtract_psf(float *dirty, float *model, int64_t row_stride, int64_t pol_stride,
                      int width, int height, int num_polarizations,
                      const float *psf, int64_t psf_row_stride, int64_t psf_pol_stride,
                      int psf_width, int psf_height, int patch_width, int patch_height,
                      const float *peak_pixel, int pos_x, int pos_y, float loop_gain,
                      void *stream);

/* The whole noise estimate of NoiseEst.__call__ / noise_est_host (clean.py:305-353, :938-943) without
 * host round trips: four radix-select passes with the byte chosen on the device, the counting pass
 * when the number of samples is even, then out[0] = median(|x| inside the border) * median_to_rms
 * (float32 arithmetic as numpy's: (lo + hi) / 2 * scale).  scratch: kimg_noise_est_scratch_bytes()
 * bytes of device memory, initialised by the call; out: device float32[1]. */
size_t kimg_noise_est_scratch_bytes(void);
int kimg_noise_est(const float *image, int64_t row_stride, int64_t pol_stride,
                   int width, int height, int num_polarizations, int border,
                   float median_to_rms, void *scratch, float *out, void *stream);

/* Device-resident minor-cycle loop (replaces the per-cycle host round trip of
 * clean.py:848-891): runs up to `max_cycles` cycles of find-peak -> threshold test ->
 * subtract -> tile update without host synchronisation.
 *   state  device scratch, kimg_clean_state_bytes(P, tiles_x, tiles_y) bytes, initialised by the
 *          call (loop state, and for the one-launch-per-cycle form used with small PSF patches
 *          the per-tile peak pixel values and the tile records in flight between cycles)
 *   log    device float32 [max_cycles][3 + P]: (metric, y, x as float bits, loop_gain*pixel[p])
 *   form   KIMG_CLEAN_FORM_*
 *   After the stream is synchronised, ((int32*)state)[0] holds the number of cycles done
 *   (stops early when the peak metric < threshold, clean.py:879-880); ((int32*)state)[1] is 2 if
 *   the persistent form gave up waiting (KIMG_ETIMEOUT for the caller to raise; the images are
 *   then undefined).
 */
size_t kimg_clean_state_bytes(int num_polarizations, int tiles_x, int tiles_y);
int kimg_clean_cycles(float *dirty, float *model, int64_t row_stride, int64_t pol_stride,
                      int width, int height, int num_polarizations,
                      const float *psf, int64_t psf_row_stride, int64_t psf_pol_stride,
                      int psf_width, int psf_height, int patch_width, int patch_height,
                      int border, int mode, float loop_gain, float threshold,
                      float *tile_max, int32_t *tile_pos, int tiles_x, int tiles_y,
                      int max_cycles, int form, void *state, float *log, void *stream);

/* The minor cycles of ONE MAJOR CYCLE in one call (the loop of frontend.py:560-585): the first cycle
 * runs without a threshold (clean.py:879: its component is always taken); the rest stop below
 *     max(noise_threshold, left_for_next * power of the first peak)      (frontend.py:568-575)
 * -- or do not run at all if the first peak itself is not above that -- worked out on the device in
 * the host's arithmetic (doubles; a metric as a flux and back as clean.py:166-184 has it), so that the
 * host round trip between the first cycle and the others is gone.  Arguments as kimg_clean_cycles;
 * noise_threshold = noise estimate x the clean threshold (in sigma), left_for_next = 1 - major gain;
 * max_cycles counts the first cycle.  The log's first row is the first cycle's.  Runs where the
 * multi-component form does (form: KIMG_CLEAN_FORM_AUTO or _MULTI with its caps); KIMG_EUNSUPPORTED
 * otherwise, and the caller takes the two steps of the reference.
 * cycles_done, first_peak (host pointers, either may be null): the number of cycles done and the
 * metric of the first one, which the call has from the words the device writes for it -- what a
 * driver needs to decide on the next major cycle without reading the device back (state and log can
 * then be fetched while the next stage runs). */
int kimg_clean_major_cycles(float *dirty, float *model, int64_t row_stride, int64_t pol_stride,
                            int width, int height, int num_polarizations,
                            const float *psf, int64_t psf_row_stride, int64_t psf_pol_stride,
                            int psf_width, int psf_height, int patch_width, int patch_height,
                            int border, int mode, float loop_gain, double noise_threshold,
                            double left_for_next, float *tile_max, int32_t *tile_pos, int tiles_x,
                            int tiles_y, int max_cycles, int form, void *state, float *log, void *stream,
                            int *cycles_done, float *first_peak);

/* The same loop for several channels of a band at once: cycle i of every channel runs in ONE
 * launch (the reference loops over channels serially, frontend.py:749-767, and within a channel
 * over cycles with a host round trip each, clean.py:848-891).  A minor cycle is a latency chain
 * that occupies a fraction of the device; the channels' images have the same shape, so their
 * cycles share the kernel boundary.  Each channel keeps its own images, PSF and patch size, tile
 * arrays, state / log buffers (as for kimg_clean_cycles), threshold and cycle limit, and stops on
 * its own; the results per channel are bit-identical to kimg_clean_cycles on that channel alone.
 *   channels_host  HOST array of num_channels (<= KIMG_CLEAN_BATCH_MAX) descriptors; the device
 *                  pointers in them follow the conventions of kimg_clean_cycles
 *   Every channel's patch must allow the one-launch-per-cycle form (at most 32 x 32 lattice
 *   blocks and (patch_width / 32 + 2) * (patch_height / 32 + 3) <= 256): KIMG_EUNSUPPORTED
 *   otherwise, and the caller runs the channels one by one. */
#define KIMG_CLEAN_BATCH_MAX 8
typedef struct kimg_clean_channel {
    float *dirty, *model;
    const float *psf;
    float *tile_max;
    int32_t *tile_pos;
    void *state;
    float *log;
    int32_t patch_width, patch_height;
    float threshold;
    int32_t max_cycles;
} kimg_clean_channel;
int kimg_clean_cycles_batch(const kimg_clean_channel *channels_host, int num_channels,
                            int64_t row_stride, int64_t pol_stride, int width, int height,
                            int num_polarizations, int64_t psf_row_stride, int64_t psf_pol_stride,
                            int psf_width, int psf_height, int border, int mode, float loop_gain,
                            int tiles_x, int tiles_y, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* KIMG_H */
