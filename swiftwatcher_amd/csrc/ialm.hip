// RPCA by inexact ALM on gfx950 -- replaces image_filtering.py:220-301 of the reference.
//
// Formulation.  The reference's singular-value step keeps ALL singular values
// (`svp = (S > 1/mu).shape[0]`, image_filtering.py:285, is the vector's length), so
//     A = U diag(S - 1/mu) V^T = M - (1/mu) * polar(M),   polar(M) = M (M^T M)^(-1/2).
// No SVD of the (pixels x frames) matrix is needed: one streaming pass accumulates the
// n x n Gram matrix G = M^T M, a one-workgroup Jacobi eigen-solve turns it into
// B = I - G^(-1/2)/mu, and the next streaming pass applies A = M B while already
// accumulating the Gram matrix of the following iteration.  Per iteration every element
// is read once (X u8, A f64, Y f64) and written once (A, Y, + the u8 sparse image):
// 34 bytes.  E and M are recomputed, never stored.
//
// Data layout in HBM: frame-major planes.  X u8 [win][n][P], A/Y f64 [win][n][P]; a
// wave reads 64 consecutive pixels of one frame per instruction (512 B for f64).
//
// All-zero frames (the null frames io_video.py:40-44 pads the last window with) give G a
// zero eigenvalue; the reference leaves that direction to LAPACK (arbitrary), here it gets
// weight 0, i.e. null frames are excluded from the decomposition.
#include "swk_internal.h"

namespace swk {

// ---------------------------------------------------------------------------------
// statistics: exact integer sum of squares and max of each window (image_filtering.py:269-275)
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ialm_stats(const uint8_t *__restrict__ X, IalmWin *win, int64_t per_win)
{
    const int w = blockIdx.y;
    const uint8_t *x = X + (int64_t)w * per_win;
    unsigned long long ss = 0;
    unsigned int mx = 0;
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, gsize = (int64_t)gridDim.x * blockDim.x;
    if ((((uintptr_t)x) & 3) == 0) {                         // dword path: 4 pixels per load
        const uint32_t *x4 = (const uint32_t *)x;
        const int64_t nq = per_win >> 2;
        for (int64_t i = gtid; i < nq; i += gsize) {
            const uint32_t v = x4[i];
            const unsigned int a = v & 255u, b2 = (v >> 8) & 255u, c = (v >> 16) & 255u, d = v >> 24;
            ss += a * a + b2 * b2 + c * c + d * d;
            const unsigned int m01 = a > b2 ? a : b2, m23 = c > d ? c : d;
            const unsigned int m = m01 > m23 ? m01 : m23;
            mx = m > mx ? m : mx;
        }
        for (int64_t i = (nq << 2) + gtid; i < per_win; i += gsize) { const unsigned int v = x[i]; ss += v * v; mx = v > mx ? v : mx; }
    } else {
        for (int64_t i = gtid; i < per_win; i += gsize) {
            unsigned int v = x[i];
            ss += v * v;
            mx = v > mx ? v : mx;
        }
    }
    for (int off = 32; off; off >>= 1) {
        ss += __shfl_down(ss, off);
        unsigned int o = __shfl_down(mx, off);
        mx = o > mx ? o : mx;
    }
    if ((threadIdx.x & 63) == 0) {          // integer atomics: exact, order independent
        atomicAdd(&win[w].sumsq, ss);
        atomicMax(&win[w].maxv, mx);
    }
}

__global__ void k_ialm_init(IalmWin *win, int *active, int nwin, double lmbda, int use_gram8)
{
    int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nwin) return;
    IalmWin &s = win[w];
    double norm_two = sqrt((double)s.sumsq);            // :269 (= Frobenius norm of the window)
    double norm_inf = (double)s.maxv / lmbda;           // :270
    s.dual_norm = norm_two > norm_inf ? norm_two : norm_inf;   // :271
    s.dnorm = norm_two;                                  // :275
    s.nxt.mu = 1.25 / norm_two;                          // :276
    s.nxt.inv_mu = 1.0 / s.nxt.mu;
    s.nxt.thr = lmbda / s.nxt.mu;
    s.cur = s.nxt;
    s.iter = 0;
    s.sweeps = 0;
    s.ws = 1; s.ws_prev = 1; s.redo = 0;
    s.ru = 1; s.wu = 1; s.last_ratio = 1e300; s.pass_b16 = 0;
    // an all-zero window has nothing to decompose (the reference would divide by zero)
    s.done = s.sumsq == 0 ? 1 : 0;
    if (!s.done) atomicAdd(active, 1);
    // does the integer Gram matrix stand?  Only if the first shrinkage removes nothing: the largest entry of
    // X + Y_0/mu_0 (:282; Y_0 = X/dual, :272) stays within the threshold lmbda/mu_0 (:283)
    const double xmax = (double)s.maxv;
    const double raw_max = xmax + s.nxt.inv_mu * (xmax / s.dual_norm);
    s.int_gram = (use_gram8 && !s.done && raw_max <= s.nxt.thr) ? 1 : 0;
}

__device__ __forceinline__ double shrink(double raw, double thr)
{
    // np.maximum(raw - thr, 0) + np.minimum(raw + thr, 0)          :283
    return fmax(raw - thr, 0.0) + fmin(raw + thr, 0.0);
}

__device__ __forceinline__ uint8_t sparse_u8(double e)
{
    double v = -e;                                       // :244
    v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);        // :245 clip
    return (uint8_t)v;                                   // astype(uint8) truncates
}

// ---------------------------------------------------------------------------------
// Variant 1: one wave per 64-pixel tile, frames looped at run time, products on the f64
// VALU with the tile staged in LDS.  Works for every n <= 64; it is the fallback and the
// on-device cross-check of the MFMA kernel.
//   MODE 0: implicit start state (A=0, Y=X/dual_norm); only accumulates Gram(M_1)
//   MODE 1: first full iteration, previous state implicit (no A/Y reads)
//   MODE 2: steady state
// ---------------------------------------------------------------------------------
constexpr int kLdsRow = 65;

template <int MODE, bool WRITE_E>
__global__ __launch_bounds__(64) void k_ialm_pass_v1(IalmBuffers b)
{
    extern __shared__ double lds[];
    const int n = b.n, P = b.P;
    const int n8 = (n + 7) & ~7;
    double *sm = lds;                    // [n8][65]  M of the iteration being finished
    double *se = lds + n8 * kLdsRow;     // [n8][65]  E of it, then M of the next iteration
    const int w = blockIdx.y;
    const IalmWin &st = b.win[w];
    if (st.done) return;
    if (MODE == 0 && st.int_gram) return;        // the start pass's only product already came from k_gram_u8 (exact X^T X)
    const int t = threadIdx.x;
    const double inv_mu = st.cur.inv_mu, thr = st.cur.thr, mu = st.cur.mu;
    const double inv_mu2 = st.nxt.inv_mu, thr2 = st.nxt.thr;
    const double dual = st.dual_norm;
    const int64_t wbase = (int64_t)w * n * P, ps = b.pstride;
    const uint8_t *X = b.X + wbase;
    double *A = b.A + (int64_t)w * b.fpad * ps, *Y = b.Y + (int64_t)w * b.fpad * ps;
    uint8_t *S = b.S + wbase;
    double *Eo = WRITE_E ? b.E + (int64_t)w * b.fpad * ps : nullptr;
    const double *Bm = b.Bm + (int64_t)w * n * n;

    for (int j = n; j < n8; ++j) { sm[j * kLdsRow + t] = 0.0; se[j * kLdsRow + t] = 0.0; }

    const int ti = t >> 3, tj = t & 7;
    double g[8][8];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int c = 0; c < 8; ++c) g[a][c] = 0.0;
    double zz = 0.0;

    const int ntiles = (P + 63) / 64;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int p = tile * 64 + t;
        const bool valid = p < P;
        const int pc = valid ? p : P - 1;
        if (MODE != 0) {
            for (int j = 0; j < n; ++j) {
                const int64_t idx = (int64_t)j * ps + pc;
                const double x = (double)X[(int64_t)j * P + pc];
                double a, y;
                if (MODE == 1) { a = 0.0; y = x / dual; }                 // :272-273
                else { a = A[idx]; y = Y[idx]; }
                const double raw = (x - a) + inv_mu * y;                  // :282
                const double e = shrink(raw, thr);                        // :283
                const double m = (x - e) + inv_mu * y;                    // :284 (SVD input)
                sm[j * kLdsRow + t] = m;
                se[j * kLdsRow + t] = e;
            }
        }
        for (int i = 0; i < n; ++i) {
            const int64_t idx = (int64_t)i * ps + pc;
            const double x = (double)X[(int64_t)i * P + pc];
            double a_new, y;
            if (MODE == 0) {
                a_new = 0.0;
                y = x / dual;
            } else {
                double acc = 0.0;
                for (int j = 0; j < n; ++j) acc += sm[j * kLdsRow + t] * Bm[j * n + i];   // :290
                a_new = acc;
                const double e = se[i * kLdsRow + t];
                const double z = (x - a_new) - e;                         // :293
                const double y_prev = MODE == 1 ? x / dual : Y[idx];
                y = y_prev + mu * z;                                      // :294
                if (valid) {
                    zz += z * z;
                    A[idx] = a_new;
                    Y[idx] = y;
                    S[(int64_t)i * P + pc] = sparse_u8(e);
                    if (WRITE_E) Eo[idx] = e;
                }
            }
            const double raw2 = (x - a_new) + inv_mu2 * y;
            const double e2 = shrink(raw2, thr2);
            const double m2 = (x - e2) + inv_mu2 * y;
            se[i * kLdsRow + t] = valid ? m2 : 0.0;
        }
        __syncthreads();
        // Gram of the 64-pixel tile: thread (ti, tj) owns rows 8ti.., cols 8tj..
        if (8 * ti < n && 8 * tj < n) {
            for (int q = 0; q < 64; ++q) {
                double ra[8], rb[8];
#pragma unroll
                for (int a = 0; a < 8; ++a) ra[a] = se[(8 * ti + a) * kLdsRow + q];
#pragma unroll
                for (int c = 0; c < 8; ++c) rb[c] = se[(8 * tj + c) * kLdsRow + q];
#pragma unroll
                for (int a = 0; a < 8; ++a)
#pragma unroll
                    for (int c = 0; c < 8; ++c) g[a][c] += ra[a] * rb[c];
            }
        }
        __syncthreads();
    }
    double *gp = b.gpart + ((int64_t)w * b.nblk + blockIdx.x) * n * n;
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int i = 8 * ti + a, j = 8 * tj + c;
            if (i < n && j < n) gp[i * n + j] = g[a][c];
        }
    if (MODE != 0) {
        for (int off = 32; off; off >>= 1) zz += __shfl_down(zz, off);
        if (t == 0) b.zzpart[(int64_t)w * b.nblk + blockIdx.x] = zz;
    }
}

// ---------------------------------------------------------------------------------
// Windows of 65 .. 128 frames: variant 1 with four waves per 64-pixel tile.  The per-pixel loops over the frames are dealt to the
// waves (frame j or i = wave, wave + 4, ...: every sum keeps variant 1's order, so for n <= 64 the two kernels agree bit for bit --
// the GPU test runs both), the tile's Gram is formed by 16 x 16 threads owning 8 x 8 entries each.  LDS: 2 x n8 x 65 doubles
// (133 KB at 128 frames).  A correctness path for queues no BASELINE configuration uses (data_structures.py:120: queue_size is free).
// ---------------------------------------------------------------------------------
template <int MODE, bool WRITE_E>
__global__ __launch_bounds__(256) void k_ialm_pass_wide(IalmBuffers b)
{
    extern __shared__ double lds[];
    const int n = b.n, P = b.P;
    const int n8 = (n + 7) & ~7;
    double *sm = lds;                    // [n8][65]  M of the iteration being finished
    double *se = lds + n8 * kLdsRow;     // [n8][65]  E of it, then M of the next iteration
    const int w = blockIdx.y;
    const IalmWin &st = b.win[w];
    if (st.done) return;
    const int tid = threadIdx.x, t = tid & 63, wave = tid >> 6;
    const double inv_mu = st.cur.inv_mu, thr = st.cur.thr, mu = st.cur.mu;
    const double inv_mu2 = st.nxt.inv_mu, thr2 = st.nxt.thr;
    const double dual = st.dual_norm;
    const int64_t wbase = (int64_t)w * n * P, ps = b.pstride;
    const uint8_t *X = b.X + wbase;
    double *A = b.A + (int64_t)w * b.fpad * ps, *Y = b.Y + (int64_t)w * b.fpad * ps;
    uint8_t *S = b.S + wbase;
    double *Eo = WRITE_E ? b.E + (int64_t)w * b.fpad * ps : nullptr;
    const double *Bm = b.Bm + (int64_t)w * n * n;

    for (int j = n + wave; j < n8; j += 4) { sm[j * kLdsRow + t] = 0.0; se[j * kLdsRow + t] = 0.0; }

    const int ti = tid >> 4, tj = tid & 15;          // this thread's 8 x 8 block of the Gram matrix
    double g[8][8];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int c = 0; c < 8; ++c) g[a][c] = 0.0;
    double zz = 0.0;

    const int ntiles = (P + 63) / 64;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int p = tile * 64 + t;
        const bool valid = p < P;
        const int pc = valid ? p : P - 1;
        if (MODE != 0) {
            for (int j = wave; j < n; j += 4) {
                const int64_t idx = (int64_t)j * ps + pc;
                const double x = (double)X[(int64_t)j * P + pc];
                double a, y;
                if (MODE == 1) { a = 0.0; y = x / dual; }                 // :272-273
                else { a = A[idx]; y = Y[idx]; }
                const double raw = (x - a) + inv_mu * y;                  // :282
                const double e = shrink(raw, thr);                        // :283
                const double m = (x - e) + inv_mu * y;                    // :284 (SVD input)
                sm[j * kLdsRow + t] = m;
                se[j * kLdsRow + t] = e;
            }
            __syncthreads();
        }
        for (int i = wave; i < n; i += 4) {
            const int64_t idx = (int64_t)i * ps + pc;
            const double x = (double)X[(int64_t)i * P + pc];
            double a_new, y;
            if (MODE == 0) {
                a_new = 0.0;
                y = x / dual;
            } else {
                double acc = 0.0;
                for (int j = 0; j < n; ++j) acc += sm[j * kLdsRow + t] * Bm[j * n + i];   // :290
                a_new = acc;
                const double e = se[i * kLdsRow + t];
                const double z = (x - a_new) - e;                         // :293
                const double y_prev = MODE == 1 ? x / dual : Y[idx];
                y = y_prev + mu * z;                                      // :294
                if (valid) {
                    zz += z * z;
                    A[idx] = a_new;
                    Y[idx] = y;
                    S[(int64_t)i * P + pc] = sparse_u8(e);
                    if (WRITE_E) Eo[idx] = e;
                }
            }
            const double raw2 = (x - a_new) + inv_mu2 * y;
            const double e2 = shrink(raw2, thr2);
            const double m2 = (x - e2) + inv_mu2 * y;
            se[i * kLdsRow + t] = valid ? m2 : 0.0;
        }
        __syncthreads();
        if (8 * ti < n && 8 * tj < n) {
            for (int q = 0; q < 64; ++q) {
                double ra[8], rb[8];
#pragma unroll
                for (int a = 0; a < 8; ++a) ra[a] = se[(8 * ti + a) * kLdsRow + q];
#pragma unroll
                for (int c = 0; c < 8; ++c) rb[c] = se[(8 * tj + c) * kLdsRow + q];
#pragma unroll
                for (int a = 0; a < 8; ++a)
#pragma unroll
                    for (int c = 0; c < 8; ++c) g[a][c] += ra[a] * rb[c];
            }
        }
        __syncthreads();
    }
    double *gp = b.gpart + ((int64_t)w * b.nblk + blockIdx.x) * n * n;
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int i = 8 * ti + a, j = 8 * tj + c;
            if (i < n && j < n) gp[i * n + j] = g[a][c];
        }
    if (MODE != 0) {
        // variant 1 sums a pixel's z^2 over the frames in order; here a pixel's frames are dealt to four waves: per-wave sums in frame
        // order, then the waves in order (float64: the stopping ratio moves in its 14th digit against variant 1)
        for (int off = 32; off; off >>= 1) zz += __shfl_down(zz, off);
        __shared__ double zw[4];
        if (t == 0) zw[wave] = zz;
        __syncthreads();
        if (tid == 0) b.zzpart[(int64_t)w * b.nblk + blockIdx.x] = ((zw[0] + zw[1]) + zw[2]) + zw[3];
    }
}

template <int MODE, bool WE>
static void launch_wide(hipStream_t s, const IalmBuffers &b)
{
    const int n8 = (b.n + 7) & ~7;
    const size_t lds = (size_t)2 * n8 * kLdsRow * sizeof(double);
    static unsigned long long attr_mask = 0;
    if (!ensure_dyn_lds((const void *)k_ialm_pass_wide<MODE, WE>, 2 * kMaxNWide * kLdsRow * 8, attr_mask)) return;
    hipLaunchKernelGGL((k_ialm_pass_wide<MODE, WE>), dim3(b.nblk, b.nwin), dim3(256), lds, s, b);
    note_launch();
}

// planes [nwin][n][P] -> reference layout [nwin][P][n]
__global__ void k_planes_to_pn(const double *__restrict__ planes, double *__restrict__ out, int n, int P, int64_t ps, int fpad)
{
    const int w = blockIdx.y;
    const int64_t total = (int64_t)n * P;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int p = (int)(i / n), j = (int)(i % n);
    out[(int64_t)w * total + i] = planes[(int64_t)w * fpad * ps + (int64_t)j * ps + p];
}

__global__ void k_rpca_epilogue(const double *__restrict__ E, int64_t count, uint8_t *__restrict__ S)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
        S[i] = sparse_u8(E[i]);
}

// ---------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------
int ialm_pass_nblk(int variant, int n, int P, int nwin)
{
    (void)n;
    if (variant >= 2 && variant != 6) {
        // 256-thread blocks, two resident per CU (LDS and registers).
        // Several rounds of blocks per CU keep the tail short when a few CUs are busy with another
        // group's eigen-solve; the cap bounds the Gram partial slabs (nblk x n^2 doubles per window).
        const int ntiles = (P + 15) / 16;
        int per_win = (256 * 2 * 3 + nwin - 1) / nwin;   // two blocks resident per CU, three rounds
        const int cap = (ntiles + 7) / 8;          // at least two tiles per wave
        if (per_win > cap) per_win = cap;
        // the Gram partial slabs (nblk x n^2 doubles per window) are summed by k_gram_reduce, four waves per 64
        // entries: 512 slabs are 128 loads per lane.  A lone window (the unchanged CLI's call pattern) needs that
        // many blocks to put two on every CU.
        if (per_win > 512) per_win = 512;
        if (per_win < 1) per_win = 1;
        return per_win;
    }
    const int ntiles = (P + 63) / 64;
    // enough blocks to fill 256 CUs a few times over, few enough that the Gram partial
    // slabs stay small next to the 34 B/element stream
    int per_win = (256 * 8 + nwin - 1) / nwin;
    if (per_win < 4) per_win = 4;
    if (per_win > 128) per_win = 128;
    if (per_win > ntiles) per_win = ntiles;
    return per_win;
}

void launch_ialm_stats(hipStream_t s, const IalmBuffers &b)
{
    const int64_t per_win = (int64_t)b.n * b.P;
    int bx = (int)((per_win + 256 * 16 - 1) / (256 * 16));
    if (bx > 256) bx = 256;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(k_ialm_stats, dim3(bx, b.nwin), dim3(256), 0, s, b.X, b.win, per_win);
}

void launch_ialm_init(hipStream_t s, const IalmBuffers &b, double lmbda)
{
    hipLaunchKernelGGL(k_ialm_init, dim3((b.nwin + 63) / 64), dim3(64), 0, s, b.win, b.active, b.nwin, lmbda, b.use_gram8);
}

template <int MODE, bool WE>
static void launch_v1(hipStream_t s, const IalmBuffers &b)
{
    const int n8 = (b.n + 7) & ~7;
    const size_t lds = (size_t)2 * n8 * kLdsRow * sizeof(double);
    static unsigned long long attr_mask = 0;
    if (!ensure_dyn_lds((const void *)k_ialm_pass_v1<MODE, WE>, 2 * kMaxN * kLdsRow * 8, attr_mask)) return;
    hipLaunchKernelGGL((k_ialm_pass_v1<MODE, WE>), dim3(b.nblk, b.nwin), dim3(64), lds, s, b);
    note_launch();
}

void launch_ialm_pass_v2(hipStream_t s, const IalmBuffers &b, int mode);   // ialm_mfma.hip
bool ialm_v2_supported(int n);


void launch_ialm_pass(hipStream_t s, const IalmBuffers &b, int mode, int variant, int k, int tune)
{
    if (variant == 6) {          // 65 .. 128 frames (also accepted below that: the tests compare it with variant 1)
        const bool we6 = b.E != nullptr;
        if (mode == 0) launch_wide<0, false>(s, b);
        else if (mode == 1) { if (we6) launch_wide<1, true>(s, b); else launch_wide<1, false>(s, b); }
        else { if (we6) launch_wide<2, true>(s, b); else launch_wide<2, false>(s, b); }
        return;
    }
    if (variant >= 4) { launch_ialm_pass_m(s, b, mode, k, tune, variant == 4); return; }
    if (variant == 2) { launch_ialm_pass_v2(s, b, mode); return; }
    const bool we = b.E != nullptr;
    if (mode == 0) launch_v1<0, false>(s, b);
    else if (mode == 1) { if (we) launch_v1<1, true>(s, b); else launch_v1<1, false>(s, b); }
    else { if (we) launch_v1<2, true>(s, b); else launch_v1<2, false>(s, b); }
}

void launch_planes_to_pn(hipStream_t s, const double *planes, double *out, int nwin, int n, int P, int64_t pstride, int fpad)
{
    const int64_t total = (int64_t)n * P;
    hipLaunchKernelGGL(k_planes_to_pn, dim3((unsigned)((total + 255) / 256), nwin), dim3(256), 0, s, planes, out, n, P, pstride, fpad);
}

void launch_rpca_epilogue(hipStream_t s, const double *E, int64_t count, uint8_t *S)
{
    int64_t blocks = (count + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_rpca_epilogue, dim3((unsigned)blocks), dim3(256), 0, s, E, count, S);
}

}  // namespace swk
