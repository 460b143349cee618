/* watroo_hip.h - C ABI of libwatroo_hip.so, the MI355X (gfx950) a-trous wavelet engine.
 *
 * The reference (frederic-auchere/wavelets, "watroo" 0.0.4) is pure Python and has no
 * FFI/plugin interface of its own: its boundary to native code is the handful of calls it
 * makes into OpenCV / numexpr / scipy / numpy.  Each entry point below replaces one of those
 * call sites (cited as file:line relative to /root/reference) for 2-D float32 images, plus
 * the plan/buffer plumbing a device-resident engine needs.  Host Python
 * (wavelets_amd/_lib.py) binds these with ctypes; INTEGRATION.md shows the stub a watroo
 * maintainer would add.
 *
 * Conventions
 *  - every function returns 0 on success, non-zero on failure; wt_last_error() returns a
 *    thread-local message (argument errors, HIP errors, RCCL errors).
 *  - plain pointers and sizes only; host pointers are borrowed for the duration of a call.
 *  - one HIP stream per context; calls on a context are serialised on that stream and are
 *    asynchronous unless they return a host value (upload/download/median/reduce sync).
 *  - images are row-major float32.  A plan describes ONE row strip [row0,row0+nrows) of a
 *    global H x W image (the whole image when nranks == 1).  Planes live in HBM with a row
 *    pitch of round_up(W,4) floats and `halo` margin rows above and below the strip that are
 *    filled by the RCCL halo exchange (nranks > 1) - borders of the GLOBAL image are always
 *    handled by symmetric reflection (cv2.BORDER_REFLECT == np.pad 'symmetric').
 *  - plane ids: 0..max_level are the coefficient planes (plane s = detail w_s, plane
 *    `level` = final smooth); WT_PLANE_INPUT holds the image handed to the transform;
 *    WT_PLANE_OUT the reconstruction; WT_PLANE_SCRATCH(i), i in [0,WT_NUM_SCRATCH), are
 *    general purpose (allocated on first use).
 */
#ifndef WATROO_HIP_H
#define WATROO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WT_ABI_VERSION 8

typedef struct wt_ctx wt_ctx;   /* device + stream (+ RCCL communicator) */
typedef struct wt_plan wt_plan; /* geometry + device planes of one image strip */

enum { WT_TRIANGLE = 0, WT_B3SPLINE = 1 };      /* watroo/wavelets.py:232-287 */

#define WT_PLANE_INPUT (-1)
#define WT_PLANE_OUT (-2)
#define WT_NUM_SCRATCH 32
#define WT_PLANE_SCRATCH(i) (-3 - (i))
#define WT_PLANE_NONE (-1000)

/* ---- library / device ------------------------------------------------------------- */
int wt_abi_version(void);
const char *wt_last_error(void);
int wt_device_count(int *count);
/* process-wide tuning / A-B switches (every setting produces the same bits).
 * "row_kernel" (default 1): single-scale operators use the LDS row kernel where the dilation
 *   allows, 0 forces the chain-march kernel;  "lattice_kernel" (1): lattice kernel for d >= 64;
 *   "bilateral_paired" (1): the float32 bilateral march fetches an operand pair with one 8-byte load (0: two 4-byte loads:
 *                  the generic path of polyphase borders; same bits).
 * "overlap" (1): multi-GPU strips run the halo exchange of the next pass beside the interior
 *   rows of the current one (second stream), 0 = every exchange between the passes;
 *   "overlap_reserve" (16): compute units the interior launch leaves to the RCCL kernels.
 * "fused_fast" (1): fused passes use the single-bounce / whole-group addressing where the image
 *   allows (halo <= image; any width since round 6); 0 forces the generic (multi-bounce, gather) addressing.
 * "scatter" (4; env WT_SCATTER): planes >= 8 MiB of single-GPU plans created from now on are
 *   mapped over shuffled 2-MiB physical chunks created in groups worth this many planes
 *   (DESIGN.md section 2); 0 = one hipMalloc per plane (see wt_plane_ptr).
 * "split_dry" (0): measurement aid - launch the passes of a strip plan split into edge and
 *   interior rows as "overlap" does, without exchanging (FLAG_NO_EXCHANGE runs).
 * "stencil64" (1): float64 images with a built-in family run the tuned per-scale kernels (chain / lattice /
 *   row kernels for double, the float64 bilateral march); 0 = the generic float64 kernels (identical bits for
 *   the filters, 1e-12 for the bilateral operator).  "fused64", "f64_pairs", "select64_list", "hist_window",
 *   "tri4", "host_pipeline", "scatter_strips": further A/B switches of the same kind.
 * "wow_overlap" (1): behind wt_decompose_bilateral / wt64_decompose_bilateral the per-scale wow updates and
 *   the MAD median run on a side stream beside the bilateral scales still queued; 0 = serial order (same bits).
 * "axis_filter" (1): wt_axis_filter / wt64_axis_filter use the tiled kernels; 0 = the tap-list operator
 *   (same bits).
 * (the 2-MiB chunks of "scatter" are 8 MiB since ABI version 7; env WT_SCATTER_CHUNK_KB) */
int wt_set_option(const char *name, int value);

/* ---- context ---------------------------------------------------------------------- */
int wt_ctx_create(int device, wt_ctx **out);
int wt_ctx_destroy(wt_ctx *ctx);
int wt_ctx_sync(wt_ctx *ctx);
/* hipMemGetInfo of the context's device: out[0] = free bytes, out[1] = total bytes (leak checks
 * around plan create / destroy cycles; sizing of strips for 288 GB of HBM). */
int wt_device_memory(wt_ctx *ctx, int64_t out[2]);
/* hipEvent stopwatch on the context's stream (the stream every kernel is launched on). */
int wt_timer_start(wt_ctx *ctx);
int wt_timer_stop(wt_ctx *ctx, float *elapsed_ms);
/* Per-kernel HIP-event profile: when enabled every kernel launch is bracketed by events.
 * wt_profile_entry copies the i-th kernel name (<= 63 chars) and its call count and total
 * device milliseconds.  Used by bench.py for the live roofline figure. */
int wt_profile_enable(wt_ctx *ctx, int on);
int wt_profile_reset(wt_ctx *ctx);
int wt_profile_count(wt_ctx *ctx, int *n);
int wt_profile_entry(wt_ctx *ctx, int i, char *name64, int64_t *calls, double *total_ms);

/* The translation units the library's device code is built as ("core", "transform", "fused_f32_k5_acc0", ...):
 * count and names, in the order a context's warm-up threads know them (wt_ctx_create loads the runtime's copy
 * machinery and the host-side units in the background, a plan's creation the fused passes of its family;
 * wt_ctx_sync waits for both; WATROO_HIP_NO_WARMUP=1 disables).  Build / test bookkeeping - the reference,
 * interpreted Python, has nothing to load (watroo/wavelets.py:1-11). */
int wt_unit_count(void);
const char *wt_unit_name(int i);

/* ---- multi-GPU (one process per GPU; RCCL over xGMI) --------------------------------- */
/* 128-byte ncclUniqueId; rank 0 creates it, the launcher broadcasts it out of band. */
int wt_comm_unique_id(void *id128);
int wt_ctx_comm_init(wt_ctx *ctx, int rank, int nranks, const void *id128);
/* rank / size as the RCCL communicator itself reports them (ncclCommUserRank / ncclCommCount);
 * 0 / 1 without a communicator.  bench.py prints the size as "rccl_ranks". */
int wt_ctx_comm_info(wt_ctx *ctx, int *rank, int *nranks);
/* ncclGetVersion of the loaded RCCL (0: unknown), and one line describing the context's GPU ("device=<hip
 * ordinal> pci=<bus id> cus=<n> name=<...>", NUL-terminated, truncated to cap): what bench.py lists per rank
 * in the multi-GPU line (the reference has no counterpart: it runs on one host, watroo/utils.py:83-219). */
int wt_comm_version(int *version);
int wt_ctx_device_info(wt_ctx *ctx, char *buf, int cap);
/* test hook: nranks==1 periodic self exchange through RCCL send/recv (plumbing check). */
int wt_comm_selftest(wt_ctx *ctx, int64_t nfloats, int *ok);

/* ---- plan --------------------------------------------------------------------------- */
/* Whole image on one GPU. */
int wt_plan_create(wt_ctx *ctx, int64_t H, int64_t W, int family, int max_level,
                   wt_plan **out);
/* Row strip [row0,row0+nrows) of a global H x W image.  halo_rows = margin rows to allocate
 * above/below (>= the largest halo any requested op needs; 0 lets the library size it from
 * max_level).  rank/nranks give the strip's position: neighbours are rank-1 / rank+1. */
int wt_plan_create_strip(wt_ctx *ctx, int64_t H, int64_t W, int family, int max_level,
                         int64_t row0, int64_t nrows, int64_t halo_rows, int rank,
                         int nranks, wt_plan **out);
/* wt_plan_create with the placement of THIS plan's planes given instead of taken from the process-wide "scatter"
 * option: scatter = 0 keeps every plane on plain hipMalloc, n > 0 maps planes >= 8 MiB over shuffled physical chunks
 * created in groups worth n planes.  What the Python API's plan pool uses for its numpy-to-numpy calls
 * (watroo/utils.py:83-102 crosses PCIe both ways: the mapping costs more than it earns there; no counterpart in the
 * reference). */
int wt_plan_create_placed(wt_ctx *ctx, int64_t H, int64_t W, int family, int max_level, int scatter,
                          wt_plan **out);
int wt_plan_destroy(wt_plan *plan);
/* geometry query: out[0..7] = H, W, pitch, row0, nrows, halo, max_level, family */
int wt_plan_info(wt_plan *plan, int64_t out[8]);
/* Device memory held by the plan right now: out[0] = all bytes (hipMalloc'ed planes, the bounce
 * plane of host transfers, mapped chunks, idle chunks), out[1] = bytes of planes mapped over
 * scattered 2-MiB chunks, out[2] = bytes of idle chunks (created in groups, not yet mapped:
 * wt_plan_trim gives them back), out[3] = 1 when this context has stopped scattering planes
 * because the virtual-memory API failed on it (wt_ctx_scatter_status names the failing call; the
 * fused passes run ~20 % slower on physically contiguous planes).  Leaves
 * wt_last_error alone. */
int wt_plan_memory(wt_plan *plan, int64_t out[4]);
/* *disabled = 1 when the context has fallen back to plain hipMalloc per plane; `reason` (up to
 * `cap` bytes, may be NULL) then names the HIP call that failed when it happened.  The fallback
 * is also announced once on stderr.  (No reference counterpart: device-memory management.) */
int wt_ctx_scatter_status(wt_ctx *ctx, int *disabled, char *reason, int cap);
/* Release the plan's idle physical chunks (a plan keeps up to three planes' worth after its last
 * plane was allocated).  The Python plan pool calls this when a plan is handed back. */
int wt_plan_trim(wt_plan *plan);
/* Decomposition schedule (host logic, needs no GPU): for `level` scales of `family`, writes
 * up to `cap` passes as triples {first_scale, n_scales, halo_rows_of_input}.  The same
 * schedule drives the kernels and the halo exchange; tests use it for the gloo CPU model. */
int wt_schedule(int family, int level, int fused, int32_t *triples, int cap, int *n_passes);
/* Border rule of the single-scale operators on this plan: 0 (default) symmetric reflection =
 * cv2.BORDER_REFLECT; 1 = symmetric reflection inside each polyphase component of the
 * operator's dilation - the rule atrous_recursive applies to its sub-arrays
 * (watroo/wavelets.py:354-390); 2 = scipy 'mirror' (reflection without edge duplication), the
 * border of the 1-D branch (watroo/wavelets.py:66-69; a 1-D signal is a 1 x N image); 3 = 'mirror'
 * inside each polyphase component (atrous_recursive on a 1-D signal).
 * Modes 1-3: per-scale kernels only (the bilateral and 3-D operators accept 0 and 1), single GPU. */
int wt_plan_set_border(wt_plan *plan, int border);
/* User-defined scaling function (a subclass of AbstractScalingFunction with its own
 * coefficients_1d, watroo/wavelets.py:152-229): `ntaps` odd 1-D taps (<= 15) replace the plan's
 * built-in family for wt_decompose / wt_decompose_pass (one scale) / wt_atrous_scale /
 * wt_smooth / wt_local_variance / wt_bilateral_conv / wt_decompose_bilateral and the 3-D
 * operators, which then run generic kernels (no fused passes; scratch planes 12, 13 and 15 are
 * used internally); the fused wow update fails.  ntaps = 0 restores the family.  Single GPU.
 * Orientation: taps are applied like cv2.filter2D applies its kernel (correlation, tap j at
 * offset (j - ntaps/2) * 2^s; watroo/wavelets.py:39-45).  The range-weighted operator is a true
 * convolution in the reference (kernel index i at offset (ntaps/2 - i) * 2^s, :87-91) and
 * applies the same taps accordingly.  A caller that stored the taps reversed - to get scipy's
 * convolution for a 1-D signal (:65-69) - passes flag bit3 to the bilateral entry points. */
int wt_plan_set_taps(wt_plan *plan, const float *taps, int ntaps);
/* dst plane <- window of a (larger) source plan's plane starting at (y0, x0); device copy.
 * (atrous_recursive pads by hw*2^(level-1) and crops at the end, watroo/wavelets.py:394-406) */
int wt_crop_plane(wt_plan *src, int src_plane, wt_plan *dst, int dst_plane, int64_t y0,
                  int64_t x0);
/* the reverse: dst[y0:y0+src.nrows, x0:x0+src.W] = src (both windows in LOCAL rows; same
 * device).  With wt_crop_plane this scatters / gathers row strips of a resident image without
 * a host round trip (tools/check_large.py builds and checks a 32768^2 image this way). */
int wt_paste_plane(wt_plan *src, int src_plane, wt_plan *dst, int dst_plane, int64_t y0,
                   int64_t x0);
/* general form: dst[dy:dy+rows, dx:dx+cols] = src[sy:sy+rows, sx:sx+cols] (local rows, same
 * device) - the crop of a padded CUBE back to its (Z, Y, X) block is one window per z slice
 * (atrous_recursive on 3-D data, watroo/wavelets.py:394-406). */
int wt_copy_window(wt_plan *src, int src_plane, wt_plan *dst, int dst_plane, int64_t sy,
                   int64_t sx, int64_t dy, int64_t dx, int64_t rows, int64_t cols);
/* Device pointer of a plane's local row 0 (zero-copy interop / virtual-strip tests).
 * RESTRICTION: on a single-GPU plan a plane of 8 MiB or more is one contiguous VIRTUAL range
 * mapped over many 2-MiB physical allocations (hipMemMap).  Kernels read and write it like any
 * pointer; hipMemcpy / hipMemcpy2D, IPC handles and RCCL transports may refuse a range that
 * spans several mappings (the library's own transfers bounce through a hipMalloc'ed plane for
 * that reason).  Consumers that need plain hipMalloc memory create their plans after
 * wt_set_option("scatter", 0) (or with WT_SCATTER=0 in the environment); strip plans
 * (nranks > 1) are always plain hipMalloc. */
int wt_plane_ptr(wt_plan *plan, int plane, void **dev_ptr);

/* ---- host <-> device ---------------------------------------------------------------- */
/* Page-locked host memory for results handed back to the caller.  A fresh pageable array costs
 * a first-touch page fault per 4 KiB during the download (8192^2: 17-31 ms instead of 4.7 ms at
 * 57 GB/s); the Python host side keeps a pool of these blocks behind the ndarrays it returns
 * (wavelets_amd/_lib.py host_empty) - the reference returns fresh numpy arrays at
 * watroo/utils.py:98,205,219 and watroo/wavelets.py:426. */
int wt_host_alloc(wt_ctx *ctx, size_t bytes, void **host_ptr);
int wt_host_free(void *host_ptr);
/* host image = this strip's rows, `host_stride` floats between rows (>= W). */
int wt_upload(wt_plan *plan, int plane, const float *host, int64_t host_stride);
/* plane <- (float) of an image of another element type, widened (byte-swapped) on the device: what the
 * reference does not recast to float64 (watroo/wavelets.py:297) - uint8 pictures, raw big-endian FITS
 * integers - is served in float32 here, without a host astype.  Type codes: WT_INT8 .. WT_FLOAT64,
 * | WT_BYTESWAPPED (defined with wt64_upload_int below). */
int wt_upload_int(wt_plan *plan, int plane, const void *host, int64_t host_pitch_bytes, int dtype);
int wt_download(wt_plan *plan, int plane, float *host, int64_t host_stride);
int wt_copy_plane(wt_plan *plan, int src, int dst);
int wt_fill_plane(wt_plan *plan, int plane, float value);
/* In-process stand-in for the RCCL halo exchange between two plans on the SAME device
 * ("virtual strips", tests/test_gpu_strips.py): `upper` owns the rows just above `lower`;
 * upper's last `rows` rows go to lower's top margin and lower's first `rows` rows to upper's
 * bottom margin - exactly the rows and margins the RCCL exchange moves. */
int wt_halo_exchange_local(wt_plan *upper, wt_plan *lower, int plane, int64_t rows);
/* RCCL halo exchange of `rows` margin rows of `plane` with the strip neighbours. */
int wt_halo_exchange(wt_plan *plan, int plane, int64_t rows);

/* ---- the hot path --------------------------------------------------------------------- */
/* AtrousTransform.atrous_standard, bilateral=None  (watroo/wavelets.py:408-444):
 * planes[0..level-1] <- detail, planes[level] <- smooth, from plane `src` (left intact).
 * flags: bit0 = allow fused multi-scale passes (default path), bit1 = skip halo exchange
 * (caller did it / virtual strips), bit2 (bilateral only) = materialise the variance plane
 * with a separate kernel instead of forming it inside the bilateral kernel, bit3 (bilateral with
 * user-defined taps only) = the plan's taps are stored reversed (see wt_plan_set_taps), bit4
 * (wt_decompose / wt_decompose_pass whose first pass is a fused pass from scale 0) = that pass
 * also histograms the first radix level of |w_0| as it produces the plane
 * (np.median(np.abs(data[0])), watroo/wavelets.py:127): a wt_abs_median(plan, 0, .) that follows
 * before anything else touches plane 0 reads the plane twice instead of three times.  Same
 * result; costs nothing measurable. */
int wt_decompose(wt_plan *plan, int src, int level, int flags);
/* one pass of the schedule (wt_schedule): scales [s0,s0+ns) from plane `cur` (c_{s0}) into
 * detail planes s0..s0+ns-1 and plane `nxt` (c_{s0+ns}); exchanges the pass halo first.
 * flags bit4: as for wt_decompose. */
int wt_decompose_pass(wt_plan *plan, int cur, int nxt, int s0, int ns, int flags);
/* wt_decompose followed by wt_plane_sum(0, level+1, dst) - the transform and its synthesis
 * np.sum(coefficients, axis=0) (watroo/wavelets.py:408-444 then watroo/utils.py:98,205) - in the
 * SAME passes: all level+1 planes are written as usual and the plane-order sum rides along
 * (each pass reads the running sum and writes it back), so the planes are not re-read:
 * 4*(L+2) + 8*passes - 4 B/pixel of traffic instead of 8*(L+2).  Bit-identical to the two-call
 * form.  Schedules with single-scale passes other than the one that ends 4 or 7 scales (9 scales
 * and more, user-defined taps, non-symmetric borders) run as the two calls. */
int wt_decompose_sum(wt_plan *plan, int src, int level, int dst, int flags);
/* Host-to-host form of wt_decompose_sum: host_in (H x W floats, row stride in_stride) -> planes
 * 0..level and the reconstruction in plane dst on the device AND in host_out.  Equivalent to
 * wt_upload(WT_PLANE_INPUT) + wt_decompose_sum + wt_download(dst), with identical bits, but the
 * three legs are PIPELINED over blocks of rows (block_rows, 0 = H/16): the passes run on a block as
 * soon as its rows and the pass's halo rows have arrived, finished rows of the reconstruction go
 * down while later blocks are still coming up (PCIe is full duplex): about one transfer leg
 * instead of two (8192^2: 10.2 -> ~6 ms).  Single-GPU plans with a fully fused schedule;
 * anything else (and wt_set_option("host_pipeline", 0)) runs the three legs in turn.  Host
 * buffers are used as they are (pageable or page-locked; the runtime locks pages per copy).
 * Reference flow: watroo/utils.py:83-102 (numpy in, numpy out). */
int wt_decompose_sum_host(wt_plan *plan, const float *host_in, int64_t in_stride, int level,
                          int dst, float *host_out, int64_t out_stride, int block_rows);
/* utils.denoise(data, weights, noise=<given>) host to host (watroo/utils.py:83-102): as
 * wt_decompose_sum_host with Coefficients.denoise placed between the passes - the first k_passes
 * passes of the fused schedule (wt_schedule) run plain, wt_denoise_sum over their n_den planes
 * (thresholds tau[k], weights wgt[k]; tau <= 0: weight only) starts the plane sum on the rows they
 * have finished, the remaining passes carry it, finished rows of dst go down while later blocks of
 * the image are still coming up.  The planes are left unthresholded (denoise() does not return
 * them).  Bit-identical to the serial sequence.  Requires 0 < k_passes < number of passes, n_den =
 * the scales of those passes, an all-fused schedule and an image worth pipelining (else an error:
 * run the serial sequence). */
int wt_denoise_sum_host(wt_plan *plan, const float *host_in, int64_t in_stride, int level,
                        int k_passes, int n_den, const double *tau, const double *wgt, int soft,
                        int dst, float *host_out, int64_t out_stride, int block_rows);
/* *ok = 1 when wt_decompose_sum(plan, ., level, ., bit0) runs as accumulate passes (else it is the
 * two-call form).  The host uses it to interleave Coefficients.denoise with the passes:
 * wt_decompose_pass for the passes that produce the thresholded planes, wt_abs_median,
 * wt_denoise_sum over those planes into the sum plane, wt_decompose_pass_sum (first = 0) for the
 * rest - the same bits as transform, denoise, sum (watroo/utils.py:95-98) with the planes read
 * once less. */
int wt_plan_fused_ok(wt_plan *plan, int level, int *ok);
/* one pass of that (fused passes only): `first` = the sum starts with this pass's first detail
 * plane, `last` = the smooth plane `nxt` is added and the sum is complete. */
int wt_decompose_pass_sum(wt_plan *plan, int cur, int nxt, int s0, int ns, int flags,
                          int sum_plane, int first, int last);
/* one scale of the above on explicit planes (per-scale operator; virtual-strip tests):
 * dst_c <- h_s (*) src ; dst_w <- src - dst_c (dst_w may be WT_PLANE_NONE). */
int wt_atrous_scale(wt_plan *plan, int src, int dst_c, int dst_w, int s, int flags);
/* convolution(arr, scaling_function, s)  (watroo/wavelets.py:35-45; cv2.filter2D with the
 * zero-stuffed kernel of :191-197, BORDER_REFLECT).  square_input: smooth src*src
 * (utils.py:177,194). */
int wt_smooth(wt_plan *plan, int src, int dst, int s, int square_input, int flags);
/* sdev_loc(image, sf, s, variance)  (watroo/wavelets.py:24-32), times f1 then f2
 * (the sigma_bilateral**2 and (s+1) factors of :434-436). */
int wt_local_variance(wt_plan *plan, int src, int dst, int s, float f1, float f2,
                      int take_sqrt, int flags);
/* atrous_convolution(image, kernel, bilateral_variance, s, 'symmetric')
 * (watroo/wavelets.py:74-105; numexpr expression :97).  flags: bit1, bit3 as for wt_decompose. */
int wt_bilateral_conv(wt_plan *plan, int src, int var, int dst, int s, int flags);
/* atrous_standard with bilateral (watroo/wavelets.py:421-442): sigma_b[level] */
int wt_decompose_bilateral(wt_plan *plan, int src, int level, const double *sigma_b,
                           int bilateral_scaling, int flags);
/* np.sum(coefficients, axis=0) (watroo/utils.py:98,205): dst <- sum planes[first..first+n) */
int wt_plane_sum(wt_plan *plan, int first, int count, int dst);
/* The same sum in two parts, for wow() behind a bilateral transform (watroo/utils.py:174-205: the planes of the
 * first scales are final long before the transform's last scales have run).  wt_plane_sum_early: dst <- planes
 * [0, count), queued on the side stream behind the per-scale updates already there (*done = 1), or nothing at all
 * when the plan is not in that overlapped state (*done = 0: sum in one piece with wt_plane_sum).
 * wt_plane_sum_resume: dst <- dst + planes [first, first + count).  Additions in plane order: the bits of
 * wt_plane_sum(plan, 0, first + count, dst).  dst: WT_PLANE_OUT or a scratch plane. */
int wt_plane_sum_early(wt_plan *plan, int count, int dst, int *done);
int wt_plane_sum_resume(wt_plan *plan, int first, int count, int dst);
/* np.median(np.abs(data[0])) (watroo/wavelets.py:127): exact radix select, fp32 result */
int wt_abs_median(wt_plan *plan, int plane, float *median);
/* Coefficients.significance (watroo/wavelets.py:129-143): dst <- erf(|c|/tau) (soft) or
 * |c| > tau (hard, 1.0/0.0).  tau = sigma*noise*sigma_e[scale]; with noise_plane !=
 * WT_PLANE_NONE tau is multiplied per pixel by that plane (ndarray noise map, :133). */
int wt_significance(wt_plan *plan, int plane, int dst, double tau, int soft, int noise_plane);
/* Coefficients.denoise body (watroo/wavelets.py:149): plane *= wgt * significance */
int wt_denoise(wt_plan *plan, int plane, double tau, double wgt, int soft, int noise_plane);
/* Coefficients.denoise (watroo/wavelets.py:145-149) of the first n_den planes fused with the
 * plane sum (watroo/utils.py:98): dst <- sum_k plane[first+k] * (wgt[k]*significance_k) for
 * k < n_den, plain planes after.  tau[k] <= 0: significance one.  write_back != 0 also stores
 * the thresholded planes (exactly denoise-then-sum); 0 leaves the planes untouched
 * (utils.denoise discards them).  Bit-identical to wt_denoise + wt_plane_sum. */
int wt_denoise_sum(wt_plan *plan, int first, int count, int dst, int n_den, const double *tau,
                   const double *wgt, int soft, int noise_plane, int write_back);
/* wow per-scale update (watroo/utils.py:193-203) fused:
 *   c <- c * significance(tau)            (skipped when tau <= 0)
 *   gamma_plane += c                      (skipped when gamma_plane == WT_PLANE_NONE)
 *   c <- c * (factor / sqrt(clip(P)))     P = power_plane (conv of c^2), clip <=0 -> 1e-15;
 *                                         power_plane == WT_PLANE_NONE: c <- c * factor */
int wt_wow_update(wt_plan *plan, int plane, int power_plane, double tau, int soft,
                  int noise_plane, float factor, int gamma_plane);
/* The whitening form of the above (power = conv_s(c^2), watroo/utils.py:193-196) in ONE kernel:
 * the local power is formed in registers, never written; the updated plane replaces plane
 * `plane` by a pointer swap with WT_PLANE_SCRATCH(3).  Bit-identical to
 * wt_smooth(square_input=1) + wt_wow_update. */
int wt_wow_scale(wt_plan *plan, int plane, int s, double tau, int soft, int noise_plane,
                 float factor, int gamma_plane, int flags);
/* global reductions for wow (watroo/utils.py:180-187,209-211): out = {sum, sumsq, min, max} */
int wt_reduce(wt_plan *plan, int plane, double out[4]);
/* gamma blend (watroo/utils.py:212-217):
 *   g <- clip((g-gmin)/(gmax-gmin),0,1)**(1/gamma); recon <- (1-h)*recon + h*g */
int wt_gamma_blend(wt_plan *plan, int recon, int gamma_plane, float gmin, float gmax,
                   float inv_gamma, float h);
/* ---- 3-D cubes (watroo/wavelets.py:46-64) ---------------------------- */
/* A (Z, Y, X) cube lives on a plan as a (Z*Y) x X image (depth = Z).  convolution() 3-D branch:
 * per-slice 2-D filter, then the same K-tap dilated filter along axis 0 (BORDER_REFLECT). */
int wt_smooth3d(wt_plan *plan, int src, int dst, int s, int depth);
/* atrous_standard on the cube: planes[0..level-1] <- detail, planes[level] <- smooth. */
int wt_decompose3d(wt_plan *plan, int src, int level, int depth);
/* bilateral transform of cubes, one scale at a time (watroo/wavelets.py:433-440 on 3-D input):
 * wt_local_variance3d = sdev_loc(cube, variance=True) * f1 * f2 (3-D smoothing of I and I^2,
 * scratch planes 13-15); wt_bilateral3d_conv = atrous_convolution with the K^3 kernel and that
 * variance (watroo/wavelets.py:74-105).  The detail plane is wt_binary(SUB). */
int wt_local_variance3d(wt_plan *plan, int src, int dst, int s, int depth, float f1, float f2);
int wt_bilateral3d_conv(wt_plan *plan, int src, int var, int dst, int s, int depth);

/* ---- Richardson-Lucy support (watroo/utils.py:222-290) -------------- */
/* cv2.filter2D(src, -1, kernel, dst, (-1,-1), 0, BORDER_REFLECT) with a small arbitrary kernel
 * (watroo/utils.py:257,286): correlation, anchor = kernel centre.  `kernel` is a host pointer
 * to kh*kw floats.  Up to 4096 taps run as one LDS-tiled launch; larger PSFs (the reference has
 * no size limit; up to 2^22 taps here) are applied in bands of rows / columns that accumulate
 * (O(kh*kw) per pixel: the direct form, not an FFT). */
int wt_filter2d(wt_plan *plan, int src, int dst, const float *kernel, int kh, int kw, int flags);
/* General form: explicit anchor (ay, ax) and border WT_BORDER_SYMMETRIC or WT_BORDER_PERIODIC.
 * The periodic border with anchor k/2 (correlation) or k-1-k/2 (flipped kernel = convolution)
 * is the circular product the reference forms with rfft2/irfft2 when fft=True
 * (watroo/utils.py:245-254, 284); whole-image plans only. */
#define WT_BORDER_SYMMETRIC 0
#define WT_BORDER_PERIODIC 3
int wt_filter2d_ex(wt_plan *plan, int src, int dst, const float *kernel, int kh, int kw, int ay,
                   int ax, int border, int flags);
/* elementwise dst = a OP b: 0 a-b, 1 a+b, 2 a*b, 3 a/b, 4 (a+b)/b  (watroo/utils.py:259,280-281,288) */
int wt_binary(wt_plan *plan, int op, int a, int b, int dst);
/* Generic tap-list operator: atrous_convolution(image, kernel, bilateral_variance, s, mode) for
 * ANY kernel and np.pad mode (watroo/wavelets.py:74-105) - what the tuned operators above do not
 * take (non-separable kernels, even / large tap counts, borders other than 'symmetric').
 *   var == WT_PLANE_NONE:  dst = [center_weight * I] + sum_t weights[t] * I_t      (tap order kept)
 *   else (range weights):  dst = ([cw * I] + sum_t e_t I_t) / ([cw] + sum_t e_t),
 *                          e_t = weights[t] * exp(-(I - I_t)^2 / var / 2)            (ref:97)
 * I_t = the sample at offsets[3t .. 3t+2] = (dz, dy, dx); the border rule pad_mode (WT_PAD_*:
 * np.pad's 'symmetric', 'reflect', 'edge', 'wrap', 'constant' with fill_value) applies per axis.
 * depth = 0: an image (1 x N: a signal), Z > 0: a (Z, Y, X) cube stored as a (Z*Y) x X image.
 * One sample per thread, taps read from a device list (<= 65536): a correctness path, not a
 * tuned one.  The Python layer builds the list in the reference's tap order. */
#define WT_PAD_SYMMETRIC 0
#define WT_PAD_REFLECT 1
#define WT_PAD_EDGE 2
#define WT_PAD_WRAP 3
#define WT_PAD_CONSTANT 4
int wt_taps_conv(wt_plan *plan, int src, int var, int dst, const int32_t *offsets,
                 const float *weights, int ntaps, float center_weight, int has_center,
                 int depth, int pad_mode, float fill_value);
/* The same with the two border rules of the RECURSIVE algorithm (watroo/wavelets.py:330-406: every
 * polyphase sub-array of stride `dilation` is filtered on its own with the base operator, i.e. an
 * out-of-range index is extended inside its own residue class modulo `dilation`): symmetric
 * (cv2.BORDER_REFLECT per sub-array, images and cubes) or scipy's 'mirror' (signals).  `dilation`
 * is ignored by the np.pad modes above. */
#define WT_PAD_POLY_SYMMETRIC 5
#define WT_PAD_POLY_MIRROR 6
int wt_taps_conv_ex(wt_plan *plan, int src, int var, int dst, const int32_t *offsets,
                    const float *weights, int ntaps, float center_weight, int has_center,
                    int depth, int pad_mode, float fill_value, int dilation);
/* One axis of the separable kernel of a scaling function (watroo/wavelets.py:170-187: kernel = outer product of
 * coefficients_1d; conv_s = convolution() with the zero-stuffed kernel, :35-69, :191-197): dst[i] = sum_j
 * weights[j] * src[pad(i + offsets[j])] along axis 2 (x), 1 (y, inside every slice) or 0 (z, across the depth
 * slices of a cube); pad_mode / fill_value / dilation as wt_taps_conv_ex.  Up to 33 taps run on tiled kernels
 * (LDS row segments along x, an LDS ring of rows down the polyphase chains along y / z); anything else on the
 * tap-list operator, with identical bits. */
int wt_axis_filter(wt_plan *plan, int src, int dst, int axis, const int32_t *offsets,
                   const float *weights, int ntaps, int depth, int pad_mode, float fill_value,
                   int dilation);
/* sdev_loc's last step (watroo/wavelets.py:27-32) from the two smoothed moments: dst = (meansq -
 * mean^2, values <= 0 -> 1e-20, optionally sqrt) * f1 * f2 - for callers that form conv(I) and
 * conv(I^2) themselves (scaling functions the tuned wt_local_variance does not take) */
int wt_variance_from_moments(wt_plan *plan, int mean, int meansq, int dst, float f1, float f2,
                             int take_sqrt);
/* Circular products of richardson_lucy(fft=True) (watroo/utils.py:245-254, 284) through a
 * hand-written FFT, for whole-image plans whose height and width are products of 2s, 3s and 5s
 * (2 .. 8192; powers of two: radix 2, other lengths: mixed radix 2 / 3 / 5):
 *   wt_fft_spectrum(plan, src)        kernel spectrum of the plan <- rfft2-equivalent of plane src
 *                                     (the PSF as the caller placed it periodically, :246-250)
 *   wt_fft_apply(plan, src, dst, conj) dst = irfft2(rfft2(src) * K)  (conj: * conj(K), :254 / :284)
 * wt_fft_supported: host logic, *ok = 1 when an H x W image qualifies.  Other sizes (and small PSFs,
 * where it is faster) keep the direct periodic form of wt_filter2d_ex. */
int wt_fft_supported(int64_t H, int64_t W, int *ok);
int wt_fft_spectrum(wt_plan *plan, int src);
int wt_fft_apply(wt_plan *plan, int src, int dst, int conj);
/* multiresolution-support update of a residual plane (watroo/utils.py:263-276):
 * sig = significance(plane, tau); hard: mrs = persistent ? max(mrs,sig) : sig, plane *= mrs;
 * soft: mrs = persistent ? mrs*sig : sig, plane *= mrs**inv_pow */
int wt_mrs_update(wt_plan *plan, int plane, int mrs_plane, double tau, int soft, int noise_plane,
                  int persistent, float inv_pow);
/* generalized_anscombe (watroo/wavelets.py:14-21) */
int wt_anscombe(wt_plan *plan, int src, int dst, float alpha, float g, float sigma,
                int inverse);

/* ---- float64 engine ------------------------------------------------------------------------
 * The reference computes float64 inputs in float64 and promotes int / big-endian inputs to
 * float64 (watroo/wavelets.py:297,319-320).  A wt_plan64 holds double planes (same plane ids as a
 * wt_plan: 0..max_level, WT_PLANE_INPUT, WT_PLANE_OUT, 32 scratch planes; wt64_decompose uses
 * scratch 0 and 1) and runs the decomposition (standard and, through border modes 1 / 3, recursive;
 * plain or bilateral), the Coefficients operators and wow's per-scale loop in double arithmetic
 * on generic kernels with the given 1-D taps (cv2.filter2D's correlation order).  Single GPU, whole images.  `depth`: 0 = an H x W image (a
 * 1 x N image is a signal: no column pass; its 'mirror' border is border mode 2), Z > 0 = a
 * (Z, Y, X) cube stored as a (Z*Y) x X image (watroo/wavelets.py:46-63). */
typedef struct wt_plan64 wt_plan64;
int wt64_plan_create(wt_ctx *ctx, int64_t H, int64_t W, int max_level, const double *taps,
                     int ntaps, wt_plan64 **plan);
int wt64_plan_destroy(wt_plan64 *plan);
int wt64_plan_set_border(wt_plan64 *plan, int border);
int wt64_upload(wt_plan64 *plan, int plane, const double *host, int64_t host_pitch);
/* plane <- (double) of an image of another element type, widened on the device: the reference recasts
 * integer and big-endian input to float64 on the host before anything else (watroo/wavelets.py:297,
 * 319-320: int16 .. int64, '>f4', '>f8' - what a FITS file holds).  host: nrows rows of W elements of type
 * `dtype`, host_pitch_bytes apart; dtype | WT_BYTESWAPPED: the elements are in the other byte order.
 * Integers are exact up to 2^53 and round to nearest even beyond (numpy's astype). */
#define WT_INT8 1
#define WT_UINT8 2
#define WT_INT16 3
#define WT_UINT16 4
#define WT_INT32 5
#define WT_UINT32 6
#define WT_INT64 7
#define WT_UINT64 8
#define WT_FLOAT32 9
#define WT_FLOAT64 10
#define WT_BYTESWAPPED 16
int wt64_upload_int(wt_plan64 *plan, int plane, const void *host, int64_t host_pitch_bytes, int dtype);
int wt64_download(wt_plan64 *plan, int plane, double *host, int64_t host_pitch);
/* AtrousTransform.atrous_standard (watroo/wavelets.py:408-444).  Images (depth 0) under the
 * symmetric border with the taps of a built-in family run the FUSED multi-scale passes of the
 * float32 engine instantiated for double (two pixels per lane; wavelets_amd/csrc/wt_fused.h),
 * any width and height (H >= 2; scales beyond the fused passes: one generic kernel each);
 * signals, cubes, user-defined taps and non-default borders run one generic kernel per scale.
 * wt_set_option("fused64", 0) forces the generic kernels (A/B; results agree to rounding: the
 * fused passes filter columns first, the generic kernels rows first). */
int wt64_decompose(wt_plan64 *plan, int src, int level, int depth);
/* the same with flags: bit4 (16) = where the first pass is a fused one it also histograms the first
 * radix level of |w_0| (as flag bit4 of wt_decompose): a wt64_abs_median(plan, 0) that follows -
 * Coefficients.get_noise, watroo/wavelets.py:126-127 - then reads the plane once less */
int wt64_decompose_ex(wt_plan64 *plan, int src, int level, int depth, int flags);
/* Pass-level entry points of the fused float64 schedule (wt_schedule; as wt_decompose_pass /
 * wt_decompose_pass_sum / wt_plan_fused_ok): they let the caller place Coefficients.denoise between
 * the passes (watroo/utils.py:95-98: transform, denoise, sum) */
int wt64_plan_fused_ok(wt_plan64 *plan, int level, int *ok);
int wt64_decompose_pass(wt_plan64 *plan, int cur, int nxt, int s0, int ns, int flags);
int wt64_decompose_pass_sum(wt_plan64 *plan, int cur, int nxt, int s0, int ns, int sum_plane,
                            int first, int last);
/* wt_decompose_sum in float64: planes 0..level and np.sum(planes, axis=0) (plane order) -> dst,
 * the sum carried through the fused passes where they apply (*fused = 1), else transform +
 * wt64_plane_sum (*fused = 0; fused may be NULL). */
int wt64_decompose_sum(wt_plan64 *plan, int src, int level, int dst, int *fused);
/* convolution(arr, scaling_function, s) (watroo/wavelets.py:35-69); square_input: of src*src */
int wt64_smooth(wt_plan64 *plan, int src, int dst, int s, int square_input, int depth);
/* sdev_loc (watroo/wavelets.py:24-32) times f1 then f2 */
int wt64_local_variance(wt_plan64 *plan, int src, int dst, int s, double f1, double f2,
                        int take_sqrt, int depth);
/* atrous_convolution(image, kernel, bilateral_variance, s) (watroo/wavelets.py:74-105);
 * taps_reversed as flag bit3 of wt_bilateral_conv */
int wt64_bilateral_conv(wt_plan64 *plan, int src, int var, int dst, int s, int depth,
                        int taps_reversed);
/* AtrousTransform(bilateral=...)(image, level) in float64 (watroo/wavelets.py:408-444, bilateral branch
 * :433-440): as wt_decompose_bilateral - planes 0..level from plane src (not 0..level, not scratch 0 / 1),
 * variance = sdev_loc(c_s) * sigma_b[s]**2 (* (s + 1) under bilateral_scaling).  Built-in taps: one marching
 * kernel per scale (variance formed in its register window, both output planes written). */
int wt64_decompose_bilateral(wt_plan64 *plan, int src, int level, const double *sigma_b,
                             int bilateral_scaling);
/* wt_axis_filter in float64 */
int wt64_axis_filter(wt_plan64 *plan, int src, int dst, int axis, const int32_t *offsets,
                     const double *weights, int ntaps, int depth, int pad_mode, double fill_value,
                     int dilation);
/* wt_taps_conv / wt_taps_conv_ex / wt_variance_from_moments in float64 */
int wt64_taps_conv(wt_plan64 *plan, int src, int var, int dst, const int32_t *offsets,
                   const double *weights, int ntaps, double center_weight, int has_center,
                   int depth, int pad_mode, double fill_value);
int wt64_taps_conv_ex(wt_plan64 *plan, int src, int var, int dst, const int32_t *offsets,
                      const double *weights, int ntaps, double center_weight, int has_center,
                      int depth, int pad_mode, double fill_value, int dilation);
int wt64_variance_from_moments(wt_plan64 *plan, int mean, int meansq, int dst, double f1, double f2,
                               int take_sqrt);
/* device copy of a window between planes of two plans (crop of atrous_recursive, :405-406) */
int wt64_copy_window(wt_plan64 *src, int src_plane, wt_plan64 *dst, int dst_plane, int64_t sy,
                     int64_t sx, int64_t dy, int64_t dx, int64_t rows, int64_t cols);
/* np.median(np.abs(plane)) (watroo/wavelets.py:127): exact select on the 63-bit keys - two radix
 * levels (the first may ride on the transform, wt64_decompose_ex), then the keys of the selected bin
 * are gathered and one workgroup finishes on that list; radix passes to the end where the bin does
 * not fit the list (ties) */
int wt64_abs_median(wt_plan64 *plan, int plane, double *median);
/* mode 0: dst = Coefficients.significance (watroo/wavelets.py:129-143); mode 1: dst = src * (wgt *
 * significance) (Coefficients.denoise with dst = src, :145-149).  tau <= 0: significance one;
 * noise_plane: per-pixel noise map multiplying tau, or WT_PLANE_NONE */
int wt64_significance(wt_plan64 *plan, int src, int dst, double tau, double wgt, int soft,
                      int noise_plane, int mode);
/* np.sum(planes[first..first+count), axis=0) in plane order (watroo/utils.py:98) */
int wt64_plane_sum(wt_plan64 *plan, int first, int count, int dst);
/* wt_plane_sum_early / wt_plane_sum_resume on a float64 plan (watroo/utils.py:174-205 with float64 data) */
int wt64_plane_sum_early(wt_plan64 *plan, int count, int dst, int *done);
int wt64_plane_sum_resume(wt_plan64 *plan, int first, int count, int dst);
/* wt_denoise_sum in float64: Coefficients.denoise over the first n_den planes (tau[k] <= 0: weight
 * only) fused with the plane sum of planes [first, first + count) -> dst (watroo/wavelets.py:145-149,
 * utils.py:98); write_back stores the thresholded planes */
int wt64_denoise_sum(wt_plan64 *plan, int first, int count, int dst, int n_den, const double *tau,
                     const double *wgt, int soft, int noise_plane, int write_back);
/* dst = a OP b: 0 add, 1 sub, 2 mul, 3 div, 4 (a+b)/b */
int wt64_binary(wt_plan64 *plan, int op, int a, int b, int dst);
/* the operators of utils.wow without bilateral filtering (watroo/utils.py:157-217) in float64:
 * per-scale update (as wt_wow_update), gamma blend, fill, {sum, sumsq, min, max} */
int wt64_wow_update(wt_plan64 *plan, int plane, int power_plane, double tau, int soft,
                    int noise_plane, double factor, int gamma_plane);
/* one scale of the wow loop on a float64 image (watroo/utils.py:193-203): local power conv_s(c^2) and the
 * update in a row pass + a column pass with the update as its epilogue, in place (no power plane);
 * identical bits to wt64_smooth(square) + wt64_wow_update. */
int wt64_wow_scale(wt_plan64 *plan, int plane, int s, double tau, int soft, int noise_plane, double factor,
                   int gamma_plane);
int wt64_gamma_blend(wt_plan64 *plan, int recon, int gamma_plane, double gmin, double gmax,
                     double inv_gamma, double h);
int wt64_fill_plane(wt_plan64 *plan, int plane, double value);
int wt64_reduce(wt_plan64 *plan, int plane, double out[4]);
/* Richardson-Lucy support in float64 (watroo/utils.py:222-290), as wt_filter2d_ex / wt_mrs_update;
 * `kernel` is a host pointer to kh*kw doubles */
int wt64_filter2d(wt_plan64 *plan, int src, int dst, const double *kernel, int kh, int kw, int ay,
                  int ax, int border);
int wt64_mrs_update(wt_plan64 *plan, int plane, int mrs_plane, double tau, int soft,
                    int noise_plane, int persistent, double inv_pow);
/* wt_fft_spectrum / wt_fft_apply in float64 */
int wt64_fft_spectrum(wt_plan64 *plan, int src);
int wt64_fft_apply(wt_plan64 *plan, int src, int dst, int conj);
/* generalized_anscombe (watroo/wavelets.py:14-21) */
int wt64_anscombe(wt_plan64 *plan, int src, int dst, double alpha, double g, double sigma,
                  int inverse);

#ifdef __cplusplus
}
#endif
#endif /* WATROO_HIP_H */
