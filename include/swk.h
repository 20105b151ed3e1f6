/*
 * swk.h -- C ABI of libswk.so, the MI355X (gfx950) implementation of
 * swiftwatcher's per-frame segmentation hot path.
 *
 * The reference (joshuacwnewton/swiftwatcher) is pure Python and has no FFI of its
 * own; the boundary it offers is its Python call surface
 * (swiftwatcher/data_structures.py:116-217 FrameQueue.preprocess_queue/segment_queue,
 * swiftwatcher/image_filtering.py:188-369 free functions).  Each entry point below
 * names the reference function it replaces.  The ctypes binding a maintainer would add
 * is shown in INTEGRATION.md and shipped in swiftwatcher_amd/_lib.py.
 *
 * Conventions
 *   - every function returns int32 status: 0 = SWK_OK, negative = error;
 *     swk_last_error(ctx) gives the text of the last failure on that context.
 *   - plain pointers and sizes only.  Buffers are caller-allocated and C-contiguous;
 *     the library never frees or retains a caller pointer after the call returns.
 *   - "mem" fields say where a caller buffer lives: SWK_MEM_HOST (pageable or pinned
 *     host memory; the library copies) or SWK_MEM_DEVICE (a HIP device pointer on the
 *     context's GPU, e.g. torch.Tensor.data_ptr()).
 *   - one swk_ctx per process per GPU; a context is not thread-safe (serialise the calls on it: the Python layer holds one
 *     lock per context); calls are synchronous (work runs on the context's own non-blocking HIP stream and is waited for;
 *     the library issues nothing on the null stream, so it can run beside a thread that captures a HIP graph in
 *     thread-local capture mode).
 *   - there is NO CPU fallback: without a usable gfx950 device swk_ctx_create fails
 *     with SWK_ERR_NOGPU and nothing else can be called.
 */
#ifndef SWK_H
#define SWK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SWK_ABI_VERSION 2

enum {
    SWK_OK = 0,
    SWK_ERR_ARG = -1,       /* bad argument / unsupported parameter value   */
    SWK_ERR_HIP = -2,       /* a HIP runtime call failed                    */
    SWK_ERR_NOGPU = -3,     /* no usable gfx950 device                      */
    SWK_ERR_CAPACITY = -4,  /* request exceeds what the context was sized for */
    SWK_ERR_NOMEM = -5,
    SWK_ERR_STALE = -6      /* the batch a call refers to is no longer held by the context */
};

enum { SWK_MEM_HOST = 0, SWK_MEM_DEVICE = 1 };

/* component numbering of cc_labeling (see swk_ccl_u8) */
enum { SWK_ORDER_RASTER = 0, SWK_ORDER_BLOCK2X2 = 1 };

/* BGR->gray fixed-point weights (see swk_bgr2gray) */
enum { SWK_GRAY_Q14 = 0, SWK_GRAY_Q15 = 1 };

typedef struct swk_ctx swk_ctx;

/* Algorithm constants.  The reference hard-codes all of them; swk_params_default()
 * fills in exactly those literals (file:line next to each field). */
typedef struct swk_params {
    double  lmbda;            /* IALM lambda = 0.01          image_filtering.py:256 */
    double  tol;              /* IALM tolerance = 0.001      image_filtering.py:256 */
    int32_t maxiter;          /* IALM max iterations = 100   image_filtering.py:257 */
    int32_t bil_d;            /* bilateral diameter = 7      data_structures.py:194 */
    double  bil_sigma_color;  /* = 15                        data_structures.py:194 */
    double  bil_sigma_space;  /* = 1                         data_structures.py:194 */
    int32_t bil_fma;          /* 0: sum += v*w (mul, add); 1: fused multiply-add.
                                 OpenCV builds differ; unpinned, default 0          */
    int32_t thresh;           /* THRESH_TOZERO level = 15    data_structures.py:198 */
    int32_t open_kh, open_kw; /* grey opening window = 3,3   data_structures.py:202 */
    int32_t connectivity;     /* 8: what cv2.connectedComponents(frame, 4) really runs
                                 (the 4 lands in the `labels` slot, image_filtering.py:327);
                                 4 also supported                                    */
    int32_t label_order;      /* SWK_ORDER_BLOCK2X2 (OpenCV 8-way default, BBDT) or
                                 SWK_ORDER_RASTER (SAUF; always used for 4-way)      */
    int32_t gray_mode;        /* SWK_GRAY_Q14 = OpenCV 4.1.0 (requirements.txt:8)    */
    int32_t reserved_;
} swk_params;

/* One region of one frame: what get_segment_properties()/regionprops yields that
 * the rest of swiftwatcher reads (image_filtering.py:332-335, data_structures.py:16-30).
 * bbox = (r0, c0, r1, c1) half-open; centroid = (sum_r/area, sum_c/area) -- the caller
 * divides in float64, which is bit-identical to skimage's coords.mean(axis=0). */
typedef struct swk_segment {
    int32_t label;
    int32_t r0, c0, r1, c1;
    int32_t reserved_;
    int64_t area;
    int64_t sum_r, sum_c;
} swk_segment;

/* A batch of RPCA windows taken from a stream of frames.
 * Window w, queue position j (0 = newest, data_structures.py:134) is the frame at
 *   frames + (w*n + j)*frame_stride, pixel (r, c) of its ROI at
 *   + (y0 + r)*row_stride + (x0 + c)*channels.
 * Passing whole 1080p frames with (x0, y0, Hc, Wc) = crop_region reproduces
 * crop_frame() (image_filtering.py:199-203); passing pre-cropped ROIs uses x0=y0=0.
 * frame_stride may be negative: `frames` is then the LAST frame in memory (a window that lies in the order it was read,
 * oldest first, is handed over without reversing it: queue position 0 = the newest = the last one read). */
typedef struct swk_input {
    const uint8_t *frames;
    int32_t mem;            /* SWK_MEM_HOST / SWK_MEM_DEVICE */
    int32_t channels;       /* 3 = BGR (convert_grayscale runs), 1 = already gray (passes
                               through, image_filtering.py:193-194) */
    int32_t nwin;           /* windows in this batch */
    int32_t n;              /* frames per window = FrameQueue queue_size (21 default); up to 128 (65 .. 128 run on plain
                               float64 kernels: correct, not fast) */
    int32_t Hc, Wc;         /* ROI rows, cols */
    int32_t x0, y0;         /* ROI origin inside each frame */
    int64_t frame_stride;   /* bytes */
    int64_t row_stride;     /* bytes */
} swk_input;

/* Every pointer is optional (NULL = not wanted).  Planes are u8 [nwin*n][Hc][Wc] in the
 * same frame order as the input; they are the per-stage images FrameQueue stores under
 * "grayscale", "RPCA", "bilateral", "thresh_15", "opened", "cc_labeling"
 * (data_structures.py:183-208). */
typedef struct swk_output {
    int32_t mem;            /* where ALL non-NULL buffers below live */
    int32_t seg_cap;        /* capacity of segs per frame (<= 255; labels are u8) */
    uint8_t *gray, *rpca, *bilateral, *thresh, *opened, *labels;
    double  *A, *E;         /* [nwin][Hc*Wc][n] float64, the reference's (pixels, frames)
                               layout (image_filtering.py:235-237); E costs an extra
                               8 B/element/iteration of HBM traffic when requested */
    int32_t *iters;         /* [nwin] IALM iterations executed */
    int32_t *nseg;          /* [nwin*n] regions found per frame (may exceed seg_cap) */
    swk_segment *segs;      /* [nwin*n][seg_cap], ascending label */
    int32_t planes_on_device; /* nonzero: the six u8 planes above are DEVICE pointers even when mem = SWK_MEM_HOST
                               (FrameQueue keeps the stage images on the GPU and copies one only when somebody reads
                               it, data_structures.py:183-208 stores them but the counting loop never looks) */
    int32_t reserved_;
} swk_output;

/* ---- lifecycle ---------------------------------------------------------------- */
int32_t swk_abi_version(void);
void    swk_params_default(swk_params *p);
/* Sizes device workspaces for batches up to max_windows x max_n frames of max_Hc x max_Wc. */
int32_t swk_ctx_create(int32_t device, int32_t max_windows, int32_t max_n,
                       int32_t max_Hc, int32_t max_Wc, swk_ctx **out);
void    swk_ctx_destroy(swk_ctx *ctx);
const char *swk_last_error(const swk_ctx *ctx);   /* ctx may be NULL: last create error */
/* Bytes of device memory the context holds (for sizing against 288 GB HBM). */
int64_t swk_ctx_device_bytes(const swk_ctx *ctx);

/* Page-locked host memory for staging buffers the caller fills and then passes as swk_input.frames (SWK_MEM_HOST): the
 * copy to the device is then a single DMA.  Allocated on `device` (the GPU the buffer will be copied to; a thread that never
 * chose a device would otherwise create a context on GPU 0), usable from any; no swk_ctx needed; free with swk_pinned_free. */
int32_t swk_pinned_alloc(int32_t device, int64_t bytes, void **out);
int32_t swk_pinned_free(void *p);
/* Host-side staging (no GPU, no context): rows [y0, y0 + rows) x bytes [x_bytes, x_bytes + row_bytes) of each of `count` frames
 * (frames[f] = address of frame f's first byte, row_stride bytes per row) copied densely into dst [count][rows][row_bytes] --
 * the stack FrameQueue.segment_queue hands to swk_batch_run, i.e. crop_frame (image_filtering.py:199-203) for a whole window.
 * threads > 1: the frames are split over a small persistent pool (video frames are cold in the caches; one core copies a
 * 21-frame 1080p window's ROI in about half a millisecond). */
int32_t swk_stage_frames(const uint8_t *const *frames, int32_t count, int64_t row_stride, int32_t y0, int32_t rows, int64_t x_bytes,
                         int64_t row_bytes, uint8_t *dst, int32_t threads);
/* Host side, no context: `count` boxes cut out of a window's frames in one call -- box i = rows [boxes[4i], boxes[4i+1]) x columns
 * [boxes[4i+2], boxes[4i+3]) of frame frame_of[i] (frames[f] = first byte of frame f, row_stride bytes per row, pixel_bytes per
 * pixel), copied densely to out + offsets[i].  The segment images of extract_segment_images (image_filtering.py:338-369) for
 * frames whose memory is reused (the ROI-stream reader's blocks): the segments then hold their crops, not the frames. */
int32_t swk_cut_boxes(const uint8_t *const *frames, int32_t nframes, int64_t row_stride, int32_t pixel_bytes, int32_t count,
                      const int32_t *frame_of, const int32_t *boxes, const int64_t *offsets, uint8_t *out);
/* Device memory on the context's GPU for outputs the caller wants to keep there (swk_output with planes_on_device, or
 * mem = SWK_MEM_DEVICE), and a synchronous copy of a piece of it to host memory.  The library never frees such a buffer
 * by itself; swk_device_free waits for the context's stream first. */
int32_t swk_device_alloc(swk_ctx *ctx, int64_t bytes, void **out);
int32_t swk_device_free(swk_ctx *ctx, void *p);
int32_t swk_device_read(swk_ctx *ctx, const void *src_device, void *dst_host, int64_t bytes);

/* ---- the hot path --------------------------------------------------------------
 * Replaces, for a batch of windows, FrameQueue.preprocess_queue + segment_queue
 * (data_structures.py:171-217): crop -> gray -> RPCA/IALM -> bilateral -> to-zero
 * threshold -> 3x3 grey opening -> connected components -> region properties.
 * Windows are independent (no state is carried between them). */
int32_t swk_batch_run(swk_ctx *ctx, const swk_input *in, const swk_params *p, swk_output *out);

/* ---- stage-level entry points (host buffers; used by the parity tests and by the
 *      image_filtering.* drop-in functions) ------------------------------------- */
/* convert_grayscale (image_filtering.py:188-196): [count][H][W][3] -> [count][H][W] */
int32_t swk_bgr2gray(swk_ctx *ctx, const uint8_t *bgr, int32_t count, int32_t H, int32_t W,
                     int32_t gray_mode, uint8_t *gray);
/* inexact_augmented_lagrange_multiplier (image_filtering.py:256-301) on one window.
 * planes: u8 [n][P] (frame j = column j of the reference's X); A, E: float64 [P][n]. */
int32_t swk_ialm(swk_ctx *ctx, const uint8_t *planes, int32_t n, int32_t P,
                 double lmbda, double tol, int32_t maxiter,
                 double *A, double *E, int32_t *iters);
/* rpca() tail (image_filtering.py:244-245): S = clip(-E, 0, 255).astype(uint8) */
int32_t swk_rpca_epilogue(swk_ctx *ctx, const double *E, int64_t count, uint8_t *S);
/* bilateral_blur (image_filtering.py:304-307): [count][H][W] u8 */
int32_t swk_bilateral_u8(swk_ctx *ctx, const uint8_t *src, int32_t count, int32_t H, int32_t W,
                         int32_t d, double sigma_color, double sigma_space, int32_t use_fma,
                         uint8_t *dst);
/* thresh_to_zero (image_filtering.py:310-316) */
int32_t swk_thresh_tozero_u8(swk_ctx *ctx, const uint8_t *src, int64_t count, int32_t thresh,
                             uint8_t *dst);
/* grayscale_opening with SE (3,3) (image_filtering.py:319-322): [count][H][W] u8 */
int32_t swk_grey_open3x3_u8(swk_ctx *ctx, const uint8_t *src, int32_t count, int32_t H, int32_t W,
                            uint8_t *dst);
/* grayscale_opening with any SE (kh, kw) (image_filtering.py:319-322: scipy.ndimage.grey_opening(size=SE), border mode 'reflect',
 * scipy's placement of even windows).  The reference's loop only asks for (3, 3) (data_structures.py:202: swk_batch_run's fused
 * filter kernel and swk_grey_open3x3_u8); this is the stage function's general case. */
int32_t swk_grey_open_u8(swk_ctx *ctx, const uint8_t *src, int32_t count, int32_t H, int32_t W, int32_t kh, int32_t kw, uint8_t *dst);
/* resize_frame (image_filtering.py:206-212): cv2.resize(frame, (dW, dH)) = INTER_LINEAR on 8-bit pixels, [count][H][W][channels] ->
 * [count][dH][dW][channels].  PARITY UNPINNED (OpenCV 4.1.0's generic 8u arithmetic restated); dead code in the reference (both call
 * sites are commented out, data_structures.py:179-181), kept for the completeness of the module's surface. */
int32_t swk_resize_linear_u8(swk_ctx *ctx, const uint8_t *src, int32_t count, int32_t H, int32_t W, int32_t channels, int32_t dH, int32_t dW,
                             uint8_t *dst);
/* cc_labeling (image_filtering.py:325-329) before the uint8 cast: int32 labels, and the
 * component count per plane. */
int32_t swk_ccl_u8(swk_ctx *ctx, const uint8_t *src, int32_t count, int32_t H, int32_t W,
                   int32_t connectivity, int32_t label_order, int32_t *labels, int32_t *ncomp);
/* get_segment_properties (image_filtering.py:332-335) on u8 label planes. */
int32_t swk_regionprops_u8(swk_ctx *ctx, const uint8_t *labels, int32_t count, int32_t H, int32_t W,
                           int32_t seg_cap, swk_segment *segs, int32_t *nseg);

/* ---- classifier input (segment_classification.py:18-24) -------------------------------------------------
 * ToPILImage -> Resize((24,24)) (Pillow's antialiased bilinear resampling, 8-bit fixed point, restated
 * exactly) -> Pad(100) -> ToTensor -> Normalize(mean, std) for nseg segment crops.
 *   crops    packed uint8 H x W x 3 images (host), crop i at byte offsets[i] with hw[2i] rows, hw[2i+1] columns
 *            (each side 1..512)
 *   patches  optional uint8 [nseg][24][24][3] (host): the resized crops
 *   net      optional float32 [nseg][3][224][224]; net_mem says whether it is a host or a device pointer
 *            (a torch tensor's data_ptr() feeds the network without another copy) */
int32_t swk_classifier_input(swk_ctx *ctx, const uint8_t *crops, int64_t crops_bytes, const int64_t *offsets,
                             const int32_t *hw, int32_t nseg, const float mean[3], const float std_[3],
                             uint8_t *patches, float *net, int32_t net_mem);
/* Same, writing only the centred (24 + 2 pad)^2 window of each 224x224 input: net is [nseg][3][24+2pad][24+2pad].
 * pad = 100 is swk_classifier_input; pad = 8 (rows/cols 92..131) is all the receptive-field cropped network of
 * swiftwatcher_amd/segment_classification.py reads. */
int32_t swk_classifier_input_window(swk_ctx *ctx, const uint8_t *crops, int64_t crops_bytes, const int64_t *offsets,
                                    const int32_t *hw, int32_t nseg, const float mean[3], const float std_[3],
                                    int32_t pad, int32_t channels_last, uint8_t *patches, float *net, int32_t net_mem);
/* channels_last: memory order of the float32 network input: 0 = planes [3][side][side] per segment (an NCHW tensor),
 * 1 = [side][side][3] (what the library's convolution kernels read; a torch tensor of shape (n, 3, side, side) in
 * torch.channels_last memory format).  swk_classifier_input writes planes. */

/* Classifier inputs cut on the device from the frames and region records of a swk_batch_run with device buffers
 * (in->mem = SWK_MEM_DEVICE, BGR; segs / nseg / net / seg_frame are device pointers): segment k of the batch is
 * region i of frame f in frame order.  Its crop box is extract_segment_images' (image_filtering.py:338-369): bbox
 * grown to at least min_h x min_w (floor / ceil split), translated by (x0, y0), sliced out of the full frame_h x
 * frame_w frame -- intersected with the frame where the reference's unchecked slice would leave it.  Segments
 * [first, first + net_cap) get their (24 + 2 pad)^2 network inputs written to net; seg_frame (optional) receives
 * the frame index of each.  *total = segments in the batch (regions beyond seg_cap per frame do not count);
 * *skipped = boxes that were empty or larger than 512 pixels (their input is the blank image). */
int32_t swk_segment_inputs(swk_ctx *ctx, const swk_input *in, int32_t frame_h, int32_t frame_w,
                           const swk_segment *segs, const int32_t *nseg, int32_t seg_cap, int32_t min_h, int32_t min_w,
                           const float mean[3], const float std_[3], int32_t pad, int32_t channels_last, int32_t first,
                           int32_t net_cap, float *net, int32_t *seg_frame, int32_t *total, int32_t *skipped);
/* The same for the LAST swk_batch_run of the context, from what that call left on the device: its frames (the copy the
 * library made of a SWK_MEM_HOST input, or the caller's device frames, which must still be alive) and its region records.
 * A host input that carries a margin around the ROI (x0, y0 > 0 in a densely packed buffer: the library then uploads the
 * whole buffer) lets boxes grow into that margin like extract_segment_images grows them into the full frame
 * (image_filtering.py:338-369): with a margin of min_seg_size / 2 (or up to the frame's edge) every crop equals the
 * reference's.  This is how SegmentClassifier scores all segments of a FrameQueue window in one batch at the window's
 * first classifier call (__main__.py:84-85).  SWK_ERR_STALE when another call has reused the buffers since.
 * *total is read as well: >= 0 = the caller's own count of the batch's segments (sum of min(nseg, seg_cap) of that batch_run's
 * host output; checked, and it saves the round trip that fetches the count from the device), -1 = not known. */
int32_t swk_segment_inputs_last(swk_ctx *ctx, int32_t min_h, int32_t min_w, const float mean[3], const float std_[3],
                                int32_t pad, int32_t channels_last, int32_t first, int32_t net_cap, float *net,
                                int32_t *seg_frame, int32_t *total, int32_t *skipped);

/* Glue of the receptive-field cropped classifier, launched on the CALLER's HIP stream (PyTorch's current stream;
 * no context): channels-last dense float32 tensors, (n, c, h, w) = memory [n][h][w][c], c multiple of 4.
 * bias_relu_place: dst[n][off_y+y][off_x+x][c_off+ch] = max(src[n][crop_y+y][crop_x+x][ch] + bias[ch], 0), the three
 * passes PyTorch makes between two convolutions (bias add, ReLU, copy into the next layer's tile) as one.
 * maxpool3s2: nn.MaxPool2d(3, 2) without padding, output [n][(h-3)/2+1][(w-3)/2+1][c]. */
int32_t swk_nhwc_bias_relu_place(void *stream, const float *src, int32_t n, int32_t sh, int32_t sw, int32_t c, int32_t crop_y,
                                 int32_t crop_x, int32_t h, int32_t w, const float *bias, float *dst, int32_t dH, int32_t dW,
                                 int32_t dC, int32_t off_y, int32_t off_x, int32_t c_off);
int32_t swk_nhwc_maxpool3s2(void *stream, const float *src, int32_t n, int32_t h, int32_t w, int32_t c, float *dst);
/* head2_relu_mean: the classifier's two-class head (Conv2d(c, 2, 1), ReLU, AdaptiveAvgPool2d(1); the reference re-heads torchvision's
 * SqueezeNet this way, segment_classification.py:47-67) over the px live positions of the last Fire's output, x [n][px][c]:
 *   out[n][k] = (sum_p max(sum_ch x[n][p][ch] w[k][ch] + bias[k], 0) + ring[k]) / n_pos
 * ring[k] = the head's sum over the positions that do not depend on the segment, n_pos = all positions.  c in {256, 512, 768, 1024}.
 * Fixed summation order: a segment's scores do not depend on the batch it is in. */
int32_t swk_nhwc_head2_relu_mean(void *stream, const float *x, int32_t n, int32_t px, int32_t c, const float *w, const float *bias,
                                 const float *ring, float n_pos, float *out);
/* conv7x7s2_bias_relu: the network's first convolution (Conv2d(3, 96, 7, stride 2), no padding) with bias and ReLU on the f32 matrix
 * cores, for the m x m outputs starting at output (lo, lo) of a side x side channels-last input:
 *   dst[n][y][x][co] = max(sum src[n][2 (lo + y) + dy][2 (lo + x) + dx][c] * weight[co][c][dy][dx] + bias[co], 0)
 * src [n][side][side][3] (side even, 2 (lo + m - 1) + 8 <= side: a zero-weighted eighth patch row is read), weight the Conv2d weight
 * [96][3][7][7] in plain (contiguous) layout, dst [n][m][m][96]. */
int32_t swk_nhwc_conv7x7s2_bias_relu(void *stream, const float *src, int32_t n, int32_t side, int32_t lo, int32_t m, const float *weight,
                                     const float *bias, int32_t cout, float *dst);
/* conv1x1_bias_relu_place: a 1 x 1 convolution fused with everything up to the next layer's tile (the squeeze and expand1x1
 * convolutions of a Fire module), on the f32 matrix cores:
 *   dst[n][off_y+y][off_x+x][c_off+co] = max(sum_ci src[n][crop_y+y][crop_x+x][ci] * weight[co][ci] + bias[co], 0)
 * src [n][sh][sw][cin] (cin a multiple of 16), weight [cout][cin] (a Conv2d weight with a 1 x 1 kernel), cout <= 256,
 * dst [n][dH][dW][dC]; cout, dC and c_off multiples of 4 (the kernel stores four channels at a time), src and dst 16-byte
 * aligned. */
int32_t swk_nhwc_conv1x1_bias_relu_place(void *stream, const float *src, int32_t n, int32_t sh, int32_t sw, int32_t cin, int32_t crop_y,
                                         int32_t crop_x, int32_t h, int32_t w, const float *weight, const float *bias, int32_t cout,
                                         float *dst, int32_t dH, int32_t dW, int32_t dC, int32_t off_y, int32_t off_x, int32_t c_off);

/* maxpool3s2 + conv1x1_bias_relu_place as ONE kernel (a MaxPool2d(3, 2) followed by a Fire module's squeeze: the pooled tensor never
 * goes to memory): src [n][t][t][cin] (cin a multiple of 32), pooled size p = (t - 3) / 2 + 1 (p * p <= 96), weight [cout][cin] with
 * cout <= 64 a multiple of 4;  dst[n][off_y + y][off_x + x][co] = max(sum_ci weight[co][ci] max_{3x3, stride 2} src + bias[co], 0).
 * ring (or NULL): ONE tile [t][t][cin] whose pixels outside the square [live_lo, live_lo + live_n)^2 equal those of every tile of src
 * (the receptive-field cropped network's tiles carry a ring of segment-independent values): those pixels are then read from it, so
 * that the ring of n tiles is not fetched n times. */
int32_t swk_nhwc_maxpool3s2_conv1x1_bias_relu_place(void *stream, const float *src, int32_t n, int32_t t, int32_t cin, const float *weight,
                                                    const float *bias, int32_t cout, float *dst, int32_t dH, int32_t dW, int32_t dC,
                                                    int32_t off_y, int32_t off_x, const float *ring, int32_t live_lo, int32_t live_n);


/* conv3x3_bias_relu_place: the 3 x 3 expand convolution of a Fire module as a VALID convolution over the t x t squeeze tile
 * (the tile carries the halo), fused with bias, ReLU and the placement behind the expand1x1 channels:
 *   dst[n][off_y+y][off_x+x][c_off+co] = max(sum src[n][y+dy][x+dx][ci] * W[co][ci][dy][dx] + bias[co], 0),  y, x < t - 2
 * src [n][t][t][cin] (cin a multiple of 16); weight_t = the Conv2d weight re-laid as [dy][dx][ci][co]; cout <= 256. */
int32_t swk_nhwc_conv3x3_bias_relu_place(void *stream, const float *src, int32_t n, int32_t t, int32_t cin, const float *weight_t,
                                         const float *bias, int32_t cout, float *dst, int32_t dH, int32_t dW, int32_t dC, int32_t off_y,
                                         int32_t off_x, int32_t c_off);

/* The same convolution by Winograd's F(2x2, 3x3) on the f32 matrix cores (2.25 x fewer multiplies; csrc/cnn_wino3x3.hip) for
 * the Fire shapes cin = cout / 4 in {32, 48, 64}; other shapes are refused (SWK_ERR_ARG) and take the direct kernel above.
 * weight_w = the filter transform G g G^T in the kernel's operand layout, made once per layer by
 * swk_winograd_f2x2_3x3_weights (host code, float64 then rounded) from the Conv2d weight [cout][cin][3][3] into
 * out[16 * cin * 32 * ceil(cout / 32)].  Results differ from the direct kernel by float32 rounding of the transform
 * (<= 1e-5 of the output scale, tests/test_classifier.py).  cout, dC, c_off multiples of 4; src, dst, weight_w 16-byte aligned. */
int32_t swk_winograd_f2x2_3x3_weights(const float *weight, int32_t cout, int32_t cin, float *out);
int32_t swk_nhwc_conv3x3_winograd_bias_relu_place(void *stream, const float *src, int32_t n, int32_t t, int32_t cin, const float *weight_w,
                                                  const float *bias, int32_t cout, float *dst, int32_t dH, int32_t dW, int32_t dC,
                                                  int32_t off_y, int32_t off_x, int32_t c_off);

/* ---- host-side tracker kernels (no GPU, no context): SURVEY section 8f rank 1 -----------------------
 * Cost matrix of SegmentTracker.formulate_cost_matrix (segment_tracking.py:46-102, 179-254): square, size
 * n_prev + n_curr, row-major.  Centroids are (row, col) float64 pairs; prev_hist0 = centroid of the first
 * segment in each previous segment's history (ignored where prev_has_hist is 0). */
int32_t swk_track_costs(const double *prev_c, const double *prev_hist0, const uint8_t *prev_has_hist,
                        const double *curr_c, int32_t n_prev, int32_t n_curr, double *cost);
/* apply_hungarian_algorithm (segment_tracking.py:257-263): scipy.optimize.linear_sum_assignment's algorithm
 * with its tie rule; col4row[i] = column assigned to row i (n_rows <= n_cols). */
int32_t swk_lsap(const double *cost, int32_t n_rows, int32_t n_cols, int32_t *col4row);

/* ---- ROI mask of a video (host side, no GPU, no context): SURVEY section 8f rank 3 -----------------------
 * image_filtering.py:99-180, run once per video on its first frame.  PARITY UNPINNED (OpenCV 4.1.0 semantics restated:
 * exact integer rules, see csrc/roi_mask.cpp).  Stage-level pieces first: */
/* cv2.medianBlur (image_filtering.py:125-131): ksize x ksize per channel, BORDER_REPLICATE; src/dst [H][W][channels] */
int32_t swk_median_blur_u8(const uint8_t *src, int32_t H, int32_t W, int32_t channels, int32_t ksize, uint8_t *dst);
/* cv2.threshold(image, 0, 255, THRESH_BINARY + THRESH_OTSU) (image_filtering.py:143-151): dst (optional) = src > t ? 255 : 0,
 * *thresh (optional) = Otsu's threshold t */
int32_t swk_otsu_threshold_u8(const uint8_t *src, int64_t count, uint8_t *dst, int32_t *thresh);
/* cv2.Canny(image, low, high) (image_filtering.py:154-159), aperture 3, L1 gradient */
int32_t swk_canny_u8(const uint8_t *src, int32_t H, int32_t W, int32_t low, int32_t high, uint8_t *dst);
/* cv2.dilate(image, ones((N, 1)), anchor=(0, 0)) (image_filtering.py:162-170): edges grow upwards by N - 1 rows */
int32_t swk_dilate_up_u8(const uint8_t *src, int32_t H, int32_t W, int32_t N, uint8_t *dst);
/* generate_regions (image_filtering.py:20-28): frame = first BGR frame [H][W][3] with row_stride bytes per row,
 * corners = {x1, y1, x2, y2} of the chimney's top edge.  Writes crop_region = {x0, y0, x1, y1} (generate_crop_region)
 * and the ROI mask, uint8 0 / 255, [y1 - y0][x1 - x0] (mask_capacity bytes available).  SWK_ERR_ARG when a region
 * leaves the frame. */
int32_t swk_roi_mask(const uint8_t *frame, int32_t H, int32_t W, int64_t row_stride, const int32_t corners[4],
                     int32_t crop_region[4], uint8_t *mask, int64_t mask_capacity);

/* A/B switches, profiling hooks and diagnostics of the library are declared in swk_debug.h: nothing a caller of the boundary above needs. */

#ifdef __cplusplus
}
#endif
#endif /* SWK_H */
