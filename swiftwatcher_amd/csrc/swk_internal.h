// Internal declarations shared by the HIP translation units of libswk.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "swk.h"
#include "swk_debug.h"

namespace swk {

constexpr int kMaxN = 64;          // frames per window supported by the matrix-core IALM kernels
constexpr int kMaxNWide = 128;     // ... and by the plain f64 kernels that take over above that (k_ialm_pass_wide, k_ialm_small_wide)

// Kernels that take more than 64 KB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize, and the attribute
// belongs to the (kernel, device) pair: `mask` (one static per kernel instantiation) keeps a bit per device it has been
// set on, so a second context on another device of the same process gets it too.  A failure is parked in
// g_launch_error (one atomic for the process: launchers do not know a context) and surfaces at the next sync() of a context.
extern std::atomic<int> g_launch_error;
inline bool ensure_dyn_lds(const void *fn, size_t bytes, unsigned long long &mask)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess && dev < 64 && ((mask >> dev) & 1ull)) return true;
    if (e == hipSuccess) e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) { g_launch_error = (int)e; return false; }
    if (dev < 64) mask |= 1ull << dev;
    return true;
}
inline void note_launch() { hipError_t e = hipGetLastError(); if (e != hipSuccess) g_launch_error = (int)e; }

// Division of a 31-bit dividend by a launch-invariant divisor as one multiply-high and a shift (a 64-bit integer division is some
// 150 vector instructions on gfx950, a 32-bit one 30 -- and in the matrix kernels every vector instruction is paid in MFMA time).
// For n < 2^31 and d >= 1: l = ceil(log2 d), mul = floor(2^(31+l) / d) + 1, n / d = umulhi(n, mul) >> (l - 1); d = 1 is the identity.
struct FastDiv {
    unsigned mul, shift, d;
    FastDiv() : mul(0), shift(0), d(1) {}
    explicit FastDiv(unsigned div) : mul(0), shift(0), d(div)
    {
        if (div > 1) {
            unsigned l = 0;
            while ((1ull << l) < div) ++l;
            mul = (unsigned)(((1ull << (31 + l)) / div) + 1);
            shift = l - 1;
        }
    }
    __host__ __device__ unsigned div(unsigned n) const
    {
#if defined(__HIP_DEVICE_COMPILE__)
        return d == 1 ? n : __umulhi(n, mul) >> shift;
#else
        return n / d;
#endif
    }
};

// Per-window scalar state of the IALM loop (device resident).
struct IalmScal { double mu, inv_mu, thr; };
struct IalmWin {
    IalmScal cur;                  // mu of the iteration being finished (image_filtering.py:282-294)
    IalmScal nxt;                  // mu of the iteration being started  (:295)
    double dual_norm;              // :271
    double dnorm;                  // :275
    unsigned long long sumsq;      // exact sum of squares of the u8 window
    unsigned int maxv;
    int iter;                      // iterations completed
    int done;
    int sweeps;                    // Jacobi sweeps used by the last eigen solve (diagnostic)
    int ws, ws_prev;               // M-state pass: did / does the pass of this (the previous) iteration write the sparse image
    int redo;                      // the window has to be run again: bit 0 = the last iteration's sparse image was not
                                   // written, bit 1 = a partial norm could not rule out that an iteration was the last,
                                   // bit 2 = the float32 stopping norm fell inside the guard band around the tolerance: this
                                   // window alone is run again by the A/Y-state pass (float64 norm, like the reference)
    int ru, wu;                    // M-state pass: this pass reads / writes ALL of U (else frames 0..3 only)
    double last_ratio;             // last full ||Z||_F / ||X||_F that was formed
    int int_gram;                  // the Gram matrix of the first iteration came from k_gram_u8 (unscaled X^T X)
    unsigned long long pass_b16;   // algorithmic bytes per element moved by the passes 1..iter of this window, in 1/16 B
    double zf2;                    // ||(G/s)^(-1/2)||_F^2 of the last Newton-Schulz solve (0: none yet): where the next solve's bound starts
    double norm_err;               // M-state pass: bound on the relative error of the float32 / binary16 stopping norm at the last test
    double cond_sum;               // ||G_1||_F sum_i 1 / lambda_i(G_1) of the first solve (>= cond(G_1)): the ill-conditioning estimate
    int refine;                    // accurate first iteration (ialm_refine.hip): 0 not needed, 1 asked for by k_ialm_small (k = 0),
                                   // 2 done, 3 given up (rank deficient: the standard route's defined result stands),
                                   // 4 wanted but not done (a window without an integer start that is too large for one workgroup's
                                   // double-double Gram matrix)
};

struct IalmBuffers {
    const uint8_t *X;              // [nwin][n][P]
    double *A, *Y;                 // [nwin][n][P]
    uint8_t *S;                    // [nwin][n][P]
    uint8_t *Salt;                 // M-state pass: second sparse-image buffer (iteration parity), else null
    uint16_t *U;                   // M-state pass: Y/mu as binary16 (x 1/128) for the stopping norm, planes like A; b.A holds M
    double *E;                     // optional [nwin][n][P]
    double *Bm;                    // [nwin][n][n]   I - W/mu
    double *Vprev;                 // [nwin][n][n]   eigenvectors of the previous solve (warm start)
    double *gpart;                 // [nwin][nblk][n][n]
    double *zzpart;                // [nwin][nblk] partial sums of z^2; the M-state pass appends [nwin][nblk] largest |U_{k-1}| per block
    IalmWin *win;                  // [nwin]
    int *active;                   // device counter of windows not yet converged
    int nwin, n, P, nblk;
    double spec;                   // M-state pass: sparse image written only once ||Z|| < spec * tol * ||X|| (<= 0: always)
    int use_gram8;                 // k_gram_u8 ran: k_ialm_init decides per window whether its result stands
    double nspec;                  // M-state pass: ||Z|| formed every other iteration while above nspec * tol * ||X|| (<= 0: always)
    double guard;                  // M-state pass: relative half-width of the band around tol in which the float32 norm does
                                   // not decide (<= 0: off; the A/Y-state pass forms the norm in float64 and needs none)
    double refine;                 // a window whose estimated first-iteration error eps * cond_sum / mu_0 exceeds this gets the accurate
                                   // first iteration of ialm_refine.hip (<= 0: never)
    int nred;                      // Gram slabs the small-matrix kernel still has to sum (1 after k_gram_reduce)
    int fpad;                      // planes allocated per window in A, Y, E: n rounded up to 16
    int64_t pstride;               // plane pitch (elements) of A, Y, E: P rounded up to 16 -> 128-B aligned rows
};

// ialm.hip
void launch_ialm_stats(hipStream_t s, const IalmBuffers &b);
void launch_ialm_init(hipStream_t s, const IalmBuffers &b, double lmbda);
// k = iteration number of the pass (0 = Gram-only start pass); variants 4 / 5 = M-state pass (no A/E outputs)
void launch_ialm_pass(hipStream_t s, const IalmBuffers &b, int mode, int variant, int k, int tune = 0);
// ialm_mstate.hip: the M-state pass instantiated per k-step count
void launch_ialm_pass_m(hipStream_t s, const IalmBuffers &b, int mode, int k, int tune, bool pipe);
int  ialm_mstate_fpad(int n);
void launch_select_sparse(hipStream_t s, const IalmBuffers &b);
// method: 0 = Newton-Schulz on the f64 matrix cores (Jacobi only as fallback), 1 = cyclic Jacobi
void launch_ialm_small(hipStream_t s, const IalmBuffers &b, int k, double lmbda, double tol, int maxiter, int method);
// windows of 65 .. 128 frames: cyclic Jacobi with its matrices in global memory (work: 3 (n + 2)^2 doubles per window)
void launch_ialm_small_wide(hipStream_t s, const IalmBuffers &b, int k, double lmbda, double tol, int maxiter, double *work);
size_t ialm_small_wide_doubles(int n);
// ialm_refine.hip: accurate first iteration (B_1) of the windows k_ialm_small flagged at k = 0; one launch after that step
void launch_ialm_refine_start(hipStream_t s, const IalmBuffers &b);
// sums the nblk Gram partial slabs of every live window into slab 0, in fixed order, chip-wide
void launch_gram_reduce(hipStream_t s, const IalmBuffers &b);
// ialm_gram8.hip: exact X^T X, sum of squares and max of every window on the i8 matrix cores
bool gram_u8_supported(const IalmBuffers &b);
void launch_gram_u8(hipStream_t s, const IalmBuffers &b);
void launch_planes_to_pn(hipStream_t s, const double *planes, double *out, int nwin, int n, int P, int64_t pstride, int fpad);
void launch_rpca_epilogue(hipStream_t s, const double *E, int64_t count, uint8_t *S);
int  ialm_pass_nblk(int variant, int n, int P, int nwin);

// filters.hip
struct BilateralTables {           // device copies of the weight tables
    float *color_w;                // [256]
    float *space_w;                // [maxk]
    int8_t *tap_dr, *tap_dc;       // [maxk]
    int maxk, radius;
};
void launch_gray(hipStream_t s, const uint8_t *frames, int channels, int64_t frame_stride, int64_t row_stride,
                 int x0, int y0, int F, int H, int W, int gray_mode, uint8_t *out);
void launch_bilateral(hipStream_t s, const uint8_t *src, int F, int H, int W, const BilateralTables &t,
                      int use_fma, uint8_t *dst);
void launch_thresh(hipStream_t s, const uint8_t *src, int64_t count, int thresh, uint8_t *dst);
void launch_open3x3(hipStream_t s, const uint8_t *src, int F, int H, int W, uint8_t *dst);
// any window (scipy's placement of even ones); tmp: F x H x W bytes for the eroded image
void launch_grey_open(hipStream_t s, const uint8_t *src, int F, int H, int W, int kh, int kw, uint8_t *tmp, uint8_t *dst);
// cv2.resize INTER_LINEAR on 8-bit pixels; per-axis index / weight tables made by the host
void launch_resize_linear(hipStream_t s, const uint8_t *src, int F, int H, int W, int ch, int dH, int dW, const int *xi, const short *xw,
                          const int *yi, const short *yw, uint8_t *dst);
// fused bilateral (radius 3) + to-zero threshold + 3x3 opening; optional intermediates
void launch_filter_fused(hipStream_t s, const uint8_t *src, int F, int H, int W, const BilateralTables &t,
                         int use_fma, int thresh, uint8_t *bil_out, uint8_t *thr_out, uint8_t *open_out);

// classify_input.hip
void launch_classifier_input(hipStream_t s, const uint8_t *crops, const int64_t *offsets, const int32_t *hw, int nseg,
                             uint8_t *patches, float *net, int pad, bool nhwc, const float *mean, const float *sd);

void launch_segment_prefix(hipStream_t s, const int32_t *nseg, int F, int seg_cap, int32_t *offsets);
void launch_segment_inputs(hipStream_t s, const uint8_t *frames, int64_t frame_stride, int64_t row_stride, int frame_h, int frame_w,
                           int x0, int y0, const swk_segment *segs, const int32_t *offsets, int F, int seg_cap, int min_h, int min_w,
                           int first, int count, float *net, int32_t *seg_frame, int pad, bool nhwc, const float *mean, const float *sd,
                           int32_t *oversize);

// ccl.hip
struct CclBuffers {
    int32_t *parent;               // [F][Pp] in the id space chosen by label_order
    uint32_t *rootbits;            // [F][words]
    int32_t *wordprefix;           // [F][words]
    int32_t *ncomp;                // [F]
    int32_t *table;                // [F][256][8] region accumulators (ints)
    unsigned long long *sums;      // [F][256][2] sum_r, sum_c
    int words;                     // per frame
    int Pp;                        // padded id-space size per frame
};
void launch_ccl(hipStream_t s, const uint8_t *src, int F, int H, int W, int connectivity, int order,
                const CclBuffers &b, int32_t *labels32 /*optional*/, uint8_t *labels8 /*optional*/);
void launch_regionprops(hipStream_t s, const uint8_t *labels8, int F, int H, int W, const CclBuffers &b,
                        int seg_cap, swk_segment *segs, int32_t *nseg);
// one-workgroup-per-frame fused labelling (+ region properties); needs the frame's bitmap in LDS
bool ccl_frame_supported(int H, int W);
void launch_ccl_frame(hipStream_t s, const uint8_t *src, int F, int H, int W, int connectivity, int order, const CclBuffers &b,
                      int32_t *labels32, uint8_t *labels8, bool props, int seg_cap, swk_segment *segs, int32_t *nseg);
size_t ccl_words(int H, int W);
size_t ccl_padded(int H, int W);


// expand1x1 of the Fire shapes with float32 products formed from split bf16 operands (cnn_expand_bf16.hip); SWK_ERR_ARG for other shapes
int launch_expand1x1_split_bf16(hipStream_t s, const float *src, int64_t rows, int sh, int sw, int cin, int crop_y, int crop_x, int h, int w,
                                const float *wgt, const float *bias, int cout, float *dst, int dH, int dW, int dC, int off_y, int off_x, int c_off);
extern int g_expand_split_bf16;          // A/B switch (swk_set_cnn_tuning knob 1): 1 = the split-bf16 kernel for the expand1x1 shapes
}  // namespace swk
