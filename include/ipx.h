/* ipx.h -- C ABI of libipx.so, the MI355X (gfx950) kernel library behind the reference's
 * /api/process image hot path (duyavaliable/image-processing, server.py).
 *
 * The reference has no FFI of its own: its "operator interface" for this path is the set of
 * whole-array cv2 / numpy call sites inside api_process().  Each entry point below replaces one of
 * those call sites (cited as server.py:LINE) and is what a ctypes binding in server.py would bind
 * (see INTEGRATION.md).
 *
 * Conventions
 *  - Plain C, no exceptions cross the boundary.  Return 0 on success, a negative IPX_ERR_* code on
 *    failure; ipx_last_error() returns the calling thread's last message.
 *  - All image pointers are DEVICE pointers (hipMalloc / PyTorch-ROCm tensor.data_ptr()) on the
 *    calling thread's current HIP device.  The caller owns every buffer; the library never frees
 *    caller memory.  src and dst must not overlap unless an entry point says in-place is allowed.
 *    The neighbourhood operators (3x3 convolutions, separable filters, Lee, median, morphology and their fused forms) CHECK
 *    this and return IPX_ERR_INVALID: they read pixels around the one they write, so an overlapping destination would be
 *    read after it was overwritten (cv2 copies in that case; the reference never asks for it).  Two column ranges of one
 *    pitched buffer (equal strides, e.g. the left and right half of a frame) share no byte and are accepted.
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream).  All work is enqueued on
 *    it; no entry point synchronises the device except ipx_pairwise_sum_lut_f32, whose result is a
 *    host scalar (it says so).  The library keeps no mutable global state, so
 *    concurrent calls from different host threads (Flask workers, one thread per GPU) are safe as
 *    long as they do not share output buffers.
 *  - Images are uint8, C-contiguous inside a row, channels interleaved (BGR order where it matters),
 *    as in the reference (H x W or H x W x 3 numpy arrays).  A batch is `n` images of identical
 *    geometry; image i starts at base + i*img_stride bytes and row y at + y*row_stride bytes.
 *    row_stride >= w*c; strides are in bytes.
 *  - Convolution borders are OpenCV's BORDER_DEFAULT (BORDER_REFLECT_101), as in the reference; the
 *    rank / morphological / Lee filters follow the border rule of the call they replace (stated there).
 *    Convolutions accept optional halo rows (`top_halo` / `bottom_halo`, `r` rows each with the
 *    source row stride, NULL = reflect at the image edge) so that a row-partitioned giant image can be
 *    filtered shard by shard after a halo exchange.
 */
#ifndef IPX_H
#define IPX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IPX_OK 0
#define IPX_ERR_INVALID (-1)     /* bad argument (null pointer, non-positive size, stride < row bytes ...) */
#define IPX_ERR_HIP (-2)         /* a HIP runtime call failed; message holds hipGetErrorString */
#define IPX_ERR_UNSUPPORTED (-3) /* valid request this build does not implement (e.g. channels > 4) */

#define IPX_MAX_GAUSS_K 255      /* largest separable kernel size accepted by ipx_sep_filter_q8_u8 */

/* library version (major*10000 + minor*100 + patch) */
int ipx_version(void);
/* thread-local message of the last failing call ("" if none) */
const char* ipx_last_error(void);
/* 0 if the current HIP device is a gfx950 (MI355X), IPX_ERR_UNSUPPORTED otherwise */
int ipx_check_device(void);

/* ---- pixel-wise --------------------------------------------------------------------------- */

/* cv2.cvtColor(img, COLOR_BGR2GRAY) on uint8 -- server.py:517 (and :511 BGRA2BGR when src_channels==4:
 * the alpha byte is dropped, not blended).  gray = (3735*B + 19235*G + 9798*R + 16384) >> 15.
 * src: n x h x w x src_channels (3 or 4), dst: n x h x w. */
int ipx_bgr2gray_u8(void* stream, const uint8_t* src, uint8_t* dst, int n, int h, int w, int src_channels,
                    int64_t src_row_stride, int64_t src_img_stride,
                    int64_t dst_row_stride, int64_t dst_img_stride);

/* The 16-bit branch of ensure_gray_uint8_from_bytes -- server.py:516-523: cv2.cvtColor(img, COLOR_BGR2GRAY) on uint16 (a
 * 4th sample, alpha, is dropped first, :509-513), then (img >> 8).astype(np.uint8).  OpenCV converts 16-bit samples with the
 * same Q15 coefficients as bytes: gray16 = (3735*B + 19235*G + 9798*R + 16384) >> 15, dst = gray16 >> 8 -- NOT the 8-bit
 * formula on the samples' high bytes (the two can differ by one level).  src: n x h x w x src_channels 16-bit samples (given as
 * bytes: 2 * src_channels per pixel, row_stride >= 2*w*src_channels); src_channels 1 (gray: dst = sample >> 8), 2 (gray + alpha:
 * the same on the first sample -- OpenCV decodes such a file to B = G = R = gray, whose gray value is gray), 3 or 4.
 * flags: IPX_U16_BIG_ENDIAN = samples as a PNG file stores them (most significant byte first; the scan lines
 * ipx_png_unfilter_u8 reconstructs go straight in), else little-endian (a numpy uint16 array); IPX_U16_RGB_ORDER = R first
 * (file order), else B first (OpenCV's memory order).  dst: n x h x w uint8. */
#define IPX_U16_BIG_ENDIAN 1
#define IPX_U16_RGB_ORDER 2
int ipx_gray_from_u16_u8(void* stream, const uint8_t* src, uint8_t* dst, int n, int h, int w, int src_channels, int flags,
                         int64_t src_row_stride, int64_t src_img_stride,
                         int64_t dst_row_stride, int64_t dst_img_stride);

/* dst = lut[src] with one 256-entry table given in HOST memory (copied into the launch, the pointer
 * may be reused as soon as the call returns).  Replaces the numpy pixel-wise expressions, each of
 * which is a function of the pixel value alone: negative server.py:376, gamma :370-372, log :385-388
 * (table depends on the image maximum), piecewise :459-467, and cv2.threshold :380.
 * `c` is the interleaved channel count (the table is applied to every byte).  In-place (dst==src,
 * same strides) is allowed. */
int ipx_lut_u8(void* stream, const uint8_t* src, uint8_t* dst, int n, int h, int w, int c,
               int64_t src_row_stride, int64_t src_img_stride,
               int64_t dst_row_stride, int64_t dst_img_stride, const uint8_t* lut_host);

/* Same with per-image tables in DEVICE memory: image i uses lut_dev + i*lut_stride (lut_stride 0 =
 * one shared table).  Second half of cv2.equalizeHist, server.py:246.  In-place allowed. */
int ipx_lut_dev_u8(void* stream, const uint8_t* src, uint8_t* dst, int n, int h, int w, int c,
                   int64_t src_row_stride, int64_t src_img_stride,
                   int64_t dst_row_stride, int64_t dst_img_stride,
                   const uint8_t* lut_dev, int64_t lut_stride);

/* k tables applied to ONE read of the source: dsts[j] = lut_j[src] for j < k <= IPX_MAX_MULTI_LUT.  The pixel-wise modes of
 * the reference all start from the same gray image (negative server.py:376, threshold :380, gamma :370-372, log :385-388,
 * piecewise :459-467); a batch pipeline that wants several of them (BASELINE.json configs[2]) pays 1 + k bytes per pixel
 * instead of 2 k.  `dsts` is a HOST array of k distinct device pointers (all with the dst strides; none may alias src);
 * image i, output j uses the 256-byte table at luts_dev + i*lut_img_stride + 256*j (DEVICE memory; lut_img_stride 0 = one
 * shared set, else >= 256 k). */
#define IPX_MAX_MULTI_LUT 8
int ipx_lut_multi_u8(void* stream, const uint8_t* src, uint8_t* const* dsts, int k, int n, int h, int w, int c,
                     int64_t src_row_stride, int64_t src_img_stride,
                     int64_t dst_row_stride, int64_t dst_img_stride,
                     const uint8_t* luts_dev, int64_t lut_img_stride);

/* Fused cvtColor(BGR2GRAY) + table: dst = lut[gray(src)] (server.py:360 followed by any of :362-388,
 * :443-467 in one pass, 4 B/px instead of 4 + 2). */
int ipx_bgr2gray_lut_u8(void* stream, const uint8_t* src, uint8_t* dst, int n, int h, int w, int src_channels,
                        int64_t src_row_stride, int64_t src_img_stride,
                        int64_t dst_row_stride, int64_t dst_img_stride, const uint8_t* lut_host);

/* ---- histogram ---------------------------------------------------------------------------- */

/* Scratch bytes ipx_hist_u8 / ipx_equalize_hist_u8 / ipx_bgr2gray_equalize_u8 need for a batch of this geometry (device
 * memory, 256-byte aligned).  Pass such a buffer as `workspace`, or NULL to let the call take it from the
 * stream-ordered allocator (hipMallocAsync/hipFreeAsync on `stream`).
 * A workspace the CALLER passes must have been prepared ONCE, after it was allocated, with ipx_hist_workspace_init
 * (below), on a stream ordered before its first use.  Its first bytes hold the tickets with which the last block of an
 * image to finish counting recognises itself (it then sums the partial counts and builds the table in the same launch);
 * they are zero between calls -- the block that wins a ticket puts it back -- so the same buffer serves any number of
 * later calls of any geometry it is large enough for, with no memset per call.  Everything behind the tickets is scratch whose
 * contents don't matter.  A workspace that was never initialised gives undefined histograms and tables.  Calls that share
 * one workspace must not overlap in time (one stream, or ordered by events). */
size_t ipx_hist_workspace_bytes(int n, int h, int w);

/* Zeroes the tickets at the head of `workspace` (one hipMemsetAsync of 16.5 KiB on `stream`).  workspace_bytes: the size
 * of the buffer, at least ipx_hist_workspace_bytes of the smallest batch it will serve.  No counterpart in the reference
 * (cv2.equalizeHist owns its scratch). */
int ipx_hist_workspace_init(void* stream, void* workspace, size_t workspace_bytes);

/* 256-bin histogram of each image: hist_dev[i*256 + v] = #pixels of image i equal to v (uint32).
 * First half of cv2.equalizeHist (server.py:246); also the source of s.max() for the log transform
 * (:387) and of the denoise_manual statistics (:193-200). */
int ipx_hist_u8(void* stream, const uint8_t* src, int n, int h, int w,
                int64_t src_row_stride, int64_t src_img_stride,
                uint32_t* hist_dev, void* workspace, size_t workspace_bytes);

/* LUT construction of cv2.equalizeHist from n histograms (device -> device): first non-zero bin i0,
 * scale = 255.f/(total - hist[i0]) in float32, lut[i] = saturate_u8(rint(float(cumsum)*scale));
 * a single-valued image maps to itself.  Separate so that a row-partitioned image can all-reduce
 * its histogram between ipx_hist_u8 and this call. */
int ipx_equalize_lut_u8(void* stream, const uint32_t* hist_dev, int n, uint8_t* lut_dev);

/* The table of mode=log from n histograms (device -> device) -- server.py:385-388: r = img/255 (float32), s = np.log1p(c*r),
 * s /= (s.max() + 1e-12), out = (s*255.0).astype(np.uint8), a function of the pixel value given the image's maximum.
 * curve_dev: the 256 float32 values log1p(c * level/255) (DEVICE memory; the host evaluates them with numpy so that they are
 * numpy's bits); s.max() is the curve at the highest level whose histogram count is non-zero.  Image i's table goes to
 * lut_dev + i*lut_stride (lut_stride >= 256; tables of a batch can be written straight into the per-image table sets of
 * ipx_lut_multi_u8).  Entries above the highest level present are never looked up and read 255.  For batch pipelines
 * (BASELINE.json configs[2]) that must not leave the stream for a host round trip; a single request builds the table on the
 * host (ipx/luts.py) from ipx_hist_u8's counts. */
int ipx_log_lut_u8(void* stream, const uint32_t* hist_dev, int n, const float* curve_dev, uint8_t* lut_dev, int64_t lut_stride);

/* cv2.equalizeHist(img) -- server.py:246.  hist_out_dev (n*256 uint32) and lut_out_dev (n*256 uint8)
 * may be NULL; when given they receive the histogram and table that were used.  dst may be NULL: the table pass is then
 * skipped (histograms and tables only; hist_out_dev or lut_out_dev must be given) for callers that apply the tables
 * themselves, e.g. together with Sobel in ipx_sobel_lut_u8. */
int ipx_equalize_hist_u8(void* stream, const uint8_t* src, uint8_t* dst, int n, int h, int w,
                         int64_t src_row_stride, int64_t src_img_stride,
                         int64_t dst_row_stride, int64_t dst_img_stride,
                         uint32_t* hist_out_dev, uint8_t* lut_out_dev,
                         void* workspace, size_t workspace_bytes);

/* ipx_equalize_hist_u8 for callers that CANNOT ALLOCATE (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  It has no counterpart in the reference.  Same arguments and results as
 * ipx_equalize_hist_u8 minus the workspace: the scratch (per-block partial counts, histograms, tables) is a
 * file-scope __device__ array of IPX_NOALLOC_SCRATCH_BYTES inside the library, one copy per GPU, held until the
 * process ends; the call makes no hipMalloc / hipFree / allocating library call.  Limit: batches whose
 * ipx_hist_workspace_bytes(n, h, w) exceeds IPX_NOALLOC_SCRATCH_BYTES are rejected with IPX_ERR_UNSUPPORTED
 * (any batch of up to 256 images fits, whatever the image size).  All calls on one GPU share that GPU's array: calls that
 * overlap in time on one GPU (two streams, two threads) race -- serialise them.  Only the tickets at the array's head have
 * to start at zero: they do when the code object is loaded, and the kernel puts each one back to zero itself (every other
 * word that is read is written first by the same call), so no call clears anything.  The array used is that of the GPU `stream`
 * belongs to (the current device for the null stream).  The FIRST call per GPU looks the array's address up
 * (hipGetSymbolAddress, which may load the code object): make one call outside a graph capture first; every later
 * call can be captured (tests/test_zz_no_alloc.py captures and replays one). */
#define IPX_NOALLOC_SCRATCH_BYTES (1536u * 1024u)
int ipx_equalize_hist_noalloc_u8(void* stream, const uint8_t* src, uint8_t* dst, int n, int h, int w,
                                 int64_t src_row_stride, int64_t src_img_stride,
                                 int64_t dst_row_stride, int64_t dst_img_stride,
                                 uint32_t* hist_out_dev, uint8_t* lut_out_dev);

/* Fused server.py:245-246 for colour input: gray = cv2.cvtColor(img, BGR2GRAY) (kept: `gray` is an output, n x h x w)
 * and dst = cv2.equalizeHist(gray).  The histogram is taken while the gray image is written, so the source is read
 * once (3 + 1 + 1 + 1 bytes per pixel instead of 3 + 1 and 1 + 1 + 1).  dst may alias gray (in-place table pass).
 * hist_out_dev / lut_out_dev / workspace as in ipx_equalize_hist_u8.  dst may be NULL: the table pass is then skipped (gray,
 * histograms and tables only; hist_out_dev or lut_out_dev must be given) for callers that apply the table themselves,
 * e.g. with others in one ipx_lut_multi_u8 pass. */
int ipx_bgr2gray_equalize_u8(void* stream, const uint8_t* src, uint8_t* gray, uint8_t* dst, int n, int h, int w,
                             int src_channels, int64_t src_row_stride, int64_t src_img_stride,
                             int64_t gray_row_stride, int64_t gray_img_stride,
                             int64_t dst_row_stride, int64_t dst_img_stride,
                             uint32_t* hist_out_dev, uint8_t* lut_out_dev,
                             void* workspace, size_t workspace_bytes);

/* ---- 3x3 convolutions --------------------------------------------------------------------- */

/* clip(f64(ch) - cv2.Laplacian(ch, CV_64F), 0, 255).astype(u8) per channel -- server.py:266-276,
 * i.e. clip(5c - n - s - e - w, 0, 255) with REFLECT_101.  c = 1..4 interleaved channels. */
int ipx_laplacian_sharpen_u8(void* stream, const uint8_t* src, uint8_t* dst, int n, int h, int w, int c,
                             int64_t src_row_stride, int64_t src_img_stride,
                             int64_t dst_row_stride, int64_t dst_img_stride,
                             const uint8_t* top_halo, const uint8_t* bottom_halo);

/* 3x3 normalised box blur, round-half-even(sum/9) (cv2.blur convention).  Named by BASELINE.json
 * configs[0]; NOT present in the reference -- definition is this build's. */
int ipx_box3_u8(void* stream, const uint8_t* src, uint8_t* dst, int n, int h, int w, int c,
                int64_t src_row_stride, int64_t src_img_stride,
                int64_t dst_row_stride, int64_t dst_img_stride,
                const uint8_t* top_halo, const uint8_t* bottom_halo);

/* Sobel magnitude min(255, |gx| + |gy|), gx = [-1 0 1; -2 0 2; -1 0 1], gy = gx^T.  Named by
 * BASELINE.json configs[1]; NOT present in the reference -- definition is this build's. */
int ipx_sobel_u8(void* stream, const uint8_t* src, uint8_t* dst, int n, int h, int w, int c,
                 int64_t src_row_stride, int64_t src_img_stride,
                 int64_t dst_row_stride, int64_t dst_img_stride,
                 const uint8_t* top_halo, const uint8_t* bottom_halo);

/* ipx_sobel_u8 AND ipx_lut_dev_u8 of the same gray batch in one pass: dst_sobel = Sobel magnitude of src (as above),
 * dst_lut = table_i[src] with image i's 256-byte table at lut_dev + i*lut_stride (DEVICE memory; lut_stride 0 = one shared
 * table) -- e.g. the second half of cv2.equalizeHist, server.py:246, with the tables ipx_bgr2gray_equalize_u8 (dst = NULL)
 * produced.  The source is read once: 1 + 2 bytes per pixel instead of 2 + 2 (BASELINE.json configs[1] runs both on the gray
 * image).  Gray images (one channel); both destinations have the dst strides and must not overlap src or each other. */
int ipx_sobel_lut_u8(void* stream, const uint8_t* src, uint8_t* dst_sobel, uint8_t* dst_lut, int n, int h, int w,
                     int64_t src_row_stride, int64_t src_img_stride,
                     int64_t dst_row_stride, int64_t dst_img_stride,
                     const uint8_t* lut_dev, int64_t lut_stride);

/* The same for any of the three 3x3 operators (op: IPX_CONV3_*; dst_conv = ipx_laplacian_sharpen_u8 / ipx_box3_u8 / ipx_sobel_u8
 * of src, dst_lut = table_i[src]) and with the optional halo rows of a row-partitioned image (n == 1; SURVEY.md section 8(e):
 * BASELINE.json configs[4] runs hist-eq and a 3x3 filter on every row block -- after the histogram all-reduce the table pass and
 * the filter share ONE read of the block, 4 bytes per pixel and step instead of 5). */
#define IPX_CONV3_SHARPEN 0
#define IPX_CONV3_BOX 1
#define IPX_CONV3_SOBEL 2
int ipx_conv3_lut_u8(void* stream, int op, const uint8_t* src, uint8_t* dst_conv, uint8_t* dst_lut, int n, int h, int w,
                     int64_t src_row_stride, int64_t src_img_stride,
                     int64_t dst_row_stride, int64_t dst_img_stride,
                     const uint8_t* lut_dev, int64_t lut_stride,
                     const uint8_t* top_halo, const uint8_t* bottom_halo);

/* ---- separable Gaussian ------------------------------------------------------------------- */

/* Host helper: the Q0.8 kernel cv2.GaussianBlur(img,(k,k),sigma) uses on uint8 input (fixed tables
 * for sigma<=0 and k<=9, else sigma = 0.15k+0.35; error-diffused rounding, taps sum to 256).
 * out holds k uint16.  k odd, 1 <= k <= IPX_MAX_GAUSS_K. */
int ipx_gauss_kernel_q8(int k, double sigma, uint16_t* out_host);

/* Separable fixed-point filter, the arithmetic of cv2.GaussianBlur on uint8 -- server.py:428:
 * row pass u8 x Q0.8 -> Q8.8 (exact), column pass -> Q8.16, dst = (acc + 2^15) >> 16, REFLECT_101,
 * each interleaved channel on its own.  kx_q8 / ky_q8 are HOST arrays of k odd taps each summing to
 * 256 (every tap <= 256).  Halo rows (r = k/2 each) as described above. */
int ipx_sep_filter_q8_u8(void* stream, const uint8_t* src, uint8_t* dst, int n, int h, int w, int c,
                         int64_t src_row_stride, int64_t src_img_stride,
                         int64_t dst_row_stride, int64_t dst_img_stride,
                         const uint16_t* kx_q8, const uint16_t* ky_q8, int k,
                         const uint8_t* top_halo, const uint8_t* bottom_halo);

/* cv2.GaussianBlur(img, (k,k), 0) -- server.py:428: ipx_gauss_kernel_q8 + ipx_sep_filter_q8_u8. */
int ipx_gaussian_blur_u8(void* stream, const uint8_t* src, uint8_t* dst, int n, int h, int w, int c,
                         int64_t src_row_stride, int64_t src_img_stride,
                         int64_t dst_row_stride, int64_t dst_img_stride, int k,
                         const uint8_t* top_halo, const uint8_t* bottom_halo);

/* cv2.GaussianBlur(img, (k,k), 0) -- server.py:428 -- AND cv2.cvtColor(img, COLOR_BGR2GRAY) -- server.py:517 -- of the same
 * BGR batch in one pass: dst = the blurred colour image (n x h x w x 3), gray = the gray image of the UNBLURRED source
 * (n x h x w), exactly as the two entry points produce them.  For pipelines that need both from one colour batch
 * (BASELINE.json configs[1]: 5x5 Gaussian of the frame; Sobel and equalizeHist of its gray image): the colour source is
 * read once.  One launch for k = 3 and 5 on rows of whole 16-byte lanes that are at least 4 KiB long (4K / 8K frames);
 * every other case runs the two entry points one after the other.  No halo rows. */
/* 1 if ipx_gaussian_blur_gray_u8 serves images of this size and kernel size with ONE launch (one read of the colour batch), else 0
 * (it then runs the two entry points): lets a pipeline choose between it and ipx_bgr2gray_equalize_u8's fused gray + histogram
 * for the gray half.  Host-only, touches no device.  No counterpart in the reference. */
int ipx_gaussian_blur_gray_one_pass(int h, int w, int k);
int ipx_gaussian_blur_gray_u8(void* stream, const uint8_t* src, uint8_t* dst, uint8_t* gray, int n, int h, int w,
                              int64_t src_row_stride, int64_t src_img_stride,
                              int64_t dst_row_stride, int64_t dst_img_stride,
                              int64_t gray_row_stride, int64_t gray_img_stride, int k);

/* ipx_gaussian_blur_gray_u8 AND ipx_laplacian_sharpen_u8 of the same BGR batch in one pass -- server.py:428, :270-276 and :517:
 * dst_gauss = the blurred colour batch, dst_sharp = clip(5c - n - s - e - w) per channel (both n x h x w x 3, sharing the dst
 * strides), gray = the gray image of the source, each exactly as its own entry point produces it.  For batch pipelines that run all
 * three on every colour frame (BASELINE.json configs[2]): the colour batch is read once, 10 bytes per pixel instead of 6 + 6 + 4.
 * One launch for k = 5 on the shapes ipx_gaussian_blur_gray_one_pass accepts; otherwise the entry points one after the other.
 * No halo rows; the destinations must not overlap the source or each other. */
int ipx_gaussian_sharpen_gray_u8(void* stream, const uint8_t* src, uint8_t* dst_gauss, uint8_t* dst_sharp, uint8_t* gray,
                                 int n, int h, int w,
                                 int64_t src_row_stride, int64_t src_img_stride,
                                 int64_t dst_row_stride, int64_t dst_img_stride,
                                 int64_t gray_row_stride, int64_t gray_img_stride, int k);

/* ---- speckle filter ---------------------------------------- */

/* lee_filter(img, window_size) -- server.py:711-732, reached through mode=denoise_manual with
 * method=lee (server.py:224-226).  Gray images only, as in the reference.  k = window_size / 2; for
 * every interior pixel (k <= y < h-k, k <= x < w-k) the float32 mean and variance of its
 * (2k+1) x (2k+1) window are formed exactly as numpy forms them (pairwise float32 summation) and
 * out = mean + var / (var + mean^2 / (mean + 1e-12)) * (pixel - mean), or mean where var == 0;
 * clip to [0,255] and truncate.  Border pixels keep their input value.  window_size in 1..31. */
int ipx_lee_filter_u8(void* stream, const uint8_t* src, uint8_t* dst, int n, int h, int w,
                      int64_t src_row_stride, int64_t src_img_stride,
                      int64_t dst_row_stride, int64_t dst_img_stride, int window_size);

/* np.add.reduce(table[src].astype(np.float32)) over a CONTIGUOUS uint8 array of `count` elements
 * on the device, bit for bit: numpy's float32 reduction order (8192-element buffers added in
 * sequence, pairwise summation inside each).  With table = the levels it is the sum behind
 * np.mean(flat), with table = fl(fl(v - mean)^2) the one behind np.var / np.std of
 * mode=denoise_manual -- server.py:197-200.  table_host: 256 floats (host).  BLOCKING: synchronises
 * `stream` and writes the float32 sum to *result_host. */
int ipx_pairwise_sum_lut_f32(void* stream, const uint8_t* src, int64_t count,
                             const float* table_host, float* result_host);

/* ---- rank / morphological filters ------------------------- */

/* cv2.medianBlur(img, k) -- server.py:215 (denoise_manual, method=median) and :430 (mode=median).
 * Gray uint8, odd k in 1..31, BORDER_REPLICATE; the exact median of the k x k window.  src != dst. */
int ipx_median_blur_u8(void* stream, const uint8_t* src, uint8_t* dst, int n, int h, int w,
                       int64_t src_row_stride, int64_t src_img_stride,
                       int64_t dst_row_stride, int64_t dst_img_stride, int k);

/* cv2.erode (op = 0) / cv2.dilate (op = 1) with a structuring element -- the two halves of
 * cv2.morphologyEx(img, MORPH_OPEN / MORPH_CLOSE, se), server.py:433-434.  se_host: HOST array of
 * se_rows x se_cols bytes (non-zero = member), odd sides in 1..31, anchor at the centre; pixels
 * outside the image do not take part (OpenCV's default morphology border).  src != dst. */
int ipx_morph_u8(void* stream, const uint8_t* src, uint8_t* dst, int n, int h, int w,
                 int64_t src_row_stride, int64_t src_img_stride,
                 int64_t dst_row_stride, int64_t dst_img_stride,
                 int op, const uint8_t* se_host, int se_rows, int se_cols);

/* ---- decode staging ----------------------------------------- */

/* The reconstruction ("un-filtering") stage of PNG decoding, PNG specification section 9 -- inside the reference it is
 * part of cv2.imdecode (server.py:499 IMREAD_UNCHANGED, :252 IMREAD_COLOR; libpng's row filters).  The host inflates the
 * IDAT stream (zlib) and hands the filtered scan lines over as they are: `raw` holds h rows of [filter type byte]
 * [w * bytes_per_pixel filtered bytes], raw_row_stride bytes apart (>= 1 + w * bytes_per_pixel); dst receives the h rows of
 * reconstructed bytes.  bytes_per_pixel: 1, 2, 3, 4, 6 or 8 (gray, gray + alpha, RGB, RGBA at 8 or 16 bits per sample;
 * samples stay in file order, 16-bit ones big-endian).  Non-interlaced images only; palette and sub-byte depths are the
 * host's business.  workspace: ipx_png_unfilter_workspace_bytes(h, w, bytes_per_pixel) bytes of device memory, contents
 * don't matter (the call clears what it needs).  *status_dev (device uint32) is set to 0, then bit 0 if a row carried a
 * filter type above 4 (not a PNG: that row is copied unfiltered), bit 1 if a wave gave up waiting for the rows above it
 * (never seen; the bound exists so that no wave can spin for ever): the pixels are valid only if it reads 0 afterwards. */
size_t ipx_png_unfilter_workspace_bytes(int h, int w, int bytes_per_pixel);
int ipx_png_unfilter_u8(void* stream, const uint8_t* raw, int64_t raw_row_stride, uint8_t* dst, int64_t dst_row_stride,
                        int h, int w, int bytes_per_pixel, void* workspace, size_t workspace_bytes, uint32_t* status_dev);

/* base64.b64encode(buf) of the encoded PNG -- server.py:536 (RFC 4648 standard alphabet, '=' padding).  src: n bytes
 * (device, 4-byte aligned); dst: 4 * ceil(n / 3) characters (device, 16-byte aligned). */
int ipx_base64_encode_u8(void* stream, const uint8_t* src, int64_t n, uint8_t* dst);

#ifdef __cplusplus
}
#endif
#endif /* IPX_H */
