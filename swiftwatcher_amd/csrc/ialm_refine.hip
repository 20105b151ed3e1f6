// The first IALM iteration of an ill-conditioned window, without squaring its condition number.
//
// Every iteration computes A = M - (1/mu) polar(M) (image_filtering.py:284-290 with the always-full `svp` of :285), and the library
// forms polar(M) = M G^(-1/2) from the n x n Gram matrix G = M^T M: an eigenvalue of G is known to eps * lambda_max, so the directions
// of the small singular values come out with a relative error of eps * cond(M)^2 / 2 -- and 1/mu is largest in iteration 1, where
// M_1 = c X is the (scaled) window itself: a rank-one sky plus sensor noise.  Measured on reference-made windows of few pixels and
// many frames (tests/golden/ialm_47x94x64*.npz; the numpy experiments are described in DESIGN.md section 2): the error of A, E at the END
// of the iteration is the error made in iteration 1 (an SVD there and the Gram route everywhere else reproduces the all-SVD result),
// because such windows run 23-25 iterations whose smallest singular value shrinks along with 1/mu, so nothing damps an early error.
//
// For a window that starts on the integer matrix cores (ialm_gram8.hip: M_1 = c X, G_1 = c^2 X^T X with K = X^T X an EXACT integer
// matrix) the accurate polar factor needs no second pass over the pixels:
//     K = L L^T            Cholesky in double-double arithmetic (106 bits: the eps * cond^2 of the squaring becomes 1e-32 * cond^2);
//     R = L^T rounded      X = Q R with Q orthonormal: R carries the singular values of X to eps * cond(X), like an SVD of X would
//     U = polar(R)         coupled Newton-Schulz on the f64 matrix cores: Y_0 = R / s, Z_0 = Y_0^T;  T = a I - c Z Y;  Y <- Y T, Z <- T Z
//     W = R^-1 U           = (R^T R)^(-1/2) = K^(-1/2), used as the PRODUCT (not symmetrised): M W = (M R^-1) U = Q U, so the errors of
//                          R^-1 (eps * cond relative) meet the orthonormal Q instead of sigma_max
//     B_1 = I - W / (c mu_0)
// A window whose first shrinkage clips (it starts with the f64 pass: about 1e5 elements or fewer for a daylight sky) has no integer K;
// its K = M_1^T M_1 is accumulated in double-double from the pixels by the same workgroup (M_1 is a function of the 8-bit value).
// One workgroup per flagged window, once per window (after the small-matrix step of k = 0); k_ialm_small flags a window when its own
// estimate of the first iteration's error, eps * ||G||_F sum_i 1/lambda_i / mu_0, exceeds IalmBuffers::refine.  Rank-deficient windows
// (null frames, duplicated frames: a pivot vanishes) keep the standard route's defined result.
#include "swk_internal.h"
#include "ialm_small_dev.h"

namespace swk {

// ---- double-double arithmetic (error-free transformations; the library is built with -ffp-contract=off) ----
struct dd { double hi, lo; };
__device__ __forceinline__ dd quick_two_sum(double a, double b) { const double s = a + b; return dd{s, b - (s - a)}; }
__device__ __forceinline__ dd two_sum(double a, double b)
{
    const double s = a + b, bb = s - a;
    return dd{s, (a - (s - bb)) + (b - bb)};
}
__device__ __forceinline__ dd two_prod(double a, double b) { const double p = a * b; return dd{p, __builtin_fma(a, b, -p)}; }
__device__ __forceinline__ dd dd_add(dd a, dd b)
{
    dd s = two_sum(a.hi, b.hi);
    const dd t = two_sum(a.lo, b.lo);
    s.lo += t.hi;
    s = quick_two_sum(s.hi, s.lo);
    s.lo += t.lo;
    return quick_two_sum(s.hi, s.lo);
}
__device__ __forceinline__ dd dd_neg(dd a) { return dd{-a.hi, -a.lo}; }
__device__ __forceinline__ dd dd_mul(dd a, dd b)
{
    dd p = two_prod(a.hi, b.hi);
    p.lo += a.hi * b.lo + a.lo * b.hi;
    return quick_two_sum(p.hi, p.lo);
}
__device__ __forceinline__ dd dd_mul_d(dd a, double b)
{
    dd p = two_prod(a.hi, b);
    p.lo += a.lo * b;
    return quick_two_sum(p.hi, p.lo);
}
__device__ __forceinline__ dd dd_div(dd a, dd b)
{
    const double q1 = a.hi / b.hi;
    dd r = dd_add(a, dd_neg(dd_mul_d(b, q1)));
    const double q2 = r.hi / b.hi;
    r = dd_add(r, dd_neg(dd_mul_d(b, q2)));
    const double q3 = r.hi / b.hi;
    return dd_add(quick_two_sum(q1, q2), dd{q3, 0.0});
}
__device__ __forceinline__ dd dd_sqrt(dd a)          // a > 0 (Karp / Markstein: one Newton step on the f64 root)
{
    const double x = 1.0 / sqrt(a.hi), ax = a.hi * x;
    const dd r = dd_add(a, dd_neg(two_prod(ax, ax)));
    return quick_two_sum(ax, r.hi * (x * 0.5));
}

template <int NB> struct RefCfg {
    static constexpr int NPAD = 16 * NB;
    static constexpr int PITCH = NPAD + 2;
    static constexpr int NT = NB * NB;
    static constexpr size_t mat = (size_t)NPAD * PITCH;
    static constexpr size_t lds_bytes = (4 * mat + 64) * sizeof(double) + 16 * sizeof(int);
};

template <int NB>
__global__ __launch_bounds__(kSmallThreads) void k_ialm_refine_start(IalmBuffers b)
{
    using C = RefCfg<NB>;
    constexpr int NPAD = C::NPAD, PITCH = C::PITCH, NT = C::NT, KS = 4 * NB;
    extern __shared__ double sm[];
    double *Z = sm, *Y = sm + C::mat, *T = sm + 2 * C::mat, *Ri = sm + 3 * C::mat;       // Z / Y hold K's high / low words first
    double *red = sm + 4 * C::mat;                                                        // [64]
    int *flags = (int *)(red + 64);
    const int w = blockIdx.x, tid = threadIdx.x, n = b.n, nthreads = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6;
    IalmWin &st = b.win[w];
    if (st.done || st.refine != 1) return;          // k_ialm_small (k = 0) asked for this window
    const double inv_mu = st.cur.inv_mu;            // 1 / mu_0 (the prologue of k = 0 has run)
    const bool from_int = st.int_gram != 0;
    const double c1 = from_int ? 1.0 + inv_mu / st.dual_norm : 1.0;          // integer start: M_1 = c1 X and K = X^T X; else K = M_1^T M_1
    double *Khi = Z, *Klo = Y;
    for (int idx = tid; idx < NPAD * PITCH; idx += nthreads) { Khi[idx] = 0.0; Klo[idx] = 0.0; T[idx] = 0.0; Ri[idx] = 0.0; }
    if (tid == 0) flags[0] = 0;
    __syncthreads();
    if (from_int) {
        // ---- K = X^T X, exact: the integer kernel's slabs (block pairs ib <= jb; sums of integers below 2^53 are exact in any order) ----
        const int nblk = b.nblk, nred = b.nred;
        const double *gp = b.gpart + (int64_t)w * nblk * n * n;
        for (int idx = tid; idx < n * n; idx += nthreads) {
            const int i = idx / n, j = idx - i * n;
            const int src = (i >> 4) <= (j >> 4) ? idx : j * n + i;
            double acc = 0.0;
            for (int bk = 0; bk < nred; ++bk) acc += gp[(int64_t)bk * n * n + src];
            Khi[i * PITCH + j] = acc;
            if (i == j && !(acc > 0.0)) flags[0] = 1;          // a null frame: rank deficient
        }
    } else {
        // ---- the window's first shrinkage clips (it did not start on the integer cores): K = M_1^T M_1 accumulated in double-double
        //      from the pixels.  M_1 takes one of 256 values (the start pass's expressions, image_filtering.py:272, 282-284, on the 8-bit
        //      value); 16 NB pixels at a time are tabulated into LDS, a thread owns the pairs (i >= j) tid, tid + 1024, ... ----
        const double dual = st.dual_norm, thr = st.cur.thr;
        const uint8_t *X = b.X + (int64_t)w * n * b.P;
        const int npairs = n * (n + 1) / 2;
        // (one workgroup does it: bounded to windows where that stays in the milliseconds -- such windows are small by construction,
        //  0.008 ||X||_F < 1.8 max(X) means about 1e5 elements for a daylight sky; larger ones keep the standard route, counted)
        if ((int64_t)b.P * ((npairs + kSmallThreads - 1) / kSmallThreads) > 400000) { if (tid == 0) st.refine = 4; return; }
        constexpr int MAXP = (NPAD * (NPAD + 1) / 2 + kSmallThreads - 1) / kSmallThreads;
        dd acc[MAXP];
        int pi[MAXP], pj[MAXP];
#pragma unroll
        for (int q = 0; q < MAXP; ++q) {
            acc[q] = dd{0.0, 0.0};
            const int pr = tid + q * kSmallThreads;
            // pair index -> (i, j), i >= j: row i starts at i (i + 1) / 2
            int i = (int)((sqrt(8.0 * (double)pr + 1.0) - 1.0) * 0.5);
            while (i * (i + 1) / 2 > pr) --i;
            while ((i + 1) * (i + 2) / 2 <= pr) ++i;
            pi[q] = pr < npairs ? i : -1;
            pj[q] = pr - i * (i + 1) / 2;
        }
        // (CH pixels at a time: a row of the staging tile holds PITCH = 16 NB + 2 values)
        constexpr int CH = NPAD;
        for (int p0 = 0; p0 < b.P; p0 += CH) {
            __syncthreads();
            for (int idx = tid; idx < n * CH; idx += nthreads) {
                const int f = idx / CH, p = p0 + (idx - f * CH);
                double m = 0.0;
                if (p < b.P) {
                    const double x = (double)X[(int64_t)f * b.P + p];
                    const double y = x / dual;                              // :272
                    const double raw = (x - 0.0) + inv_mu * y;              // :282 (A_0 = 0)
                    const double e = fmax(raw - thr, 0.0) + fmin(raw + thr, 0.0);   // :283
                    m = (x - e) + inv_mu * y;                               // :284
                }
                T[f * PITCH + (idx - f * CH)] = m;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < MAXP; ++q) {
                if (pi[q] < 0) continue;
                const double *ri = T + pi[q] * PITCH, *rj = T + pj[q] * PITCH;
                dd a = acc[q];
                for (int p = 0; p < CH; ++p) a = dd_add(a, two_prod(ri[p], rj[p]));
                acc[q] = a;
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < MAXP; ++q) {
            if (pi[q] < 0) continue;
            Khi[pi[q] * PITCH + pj[q]] = acc[q].hi; Klo[pi[q] * PITCH + pj[q]] = acc[q].lo;
            if (pi[q] == pj[q] && !(acc[q].hi > 0.0)) flags[0] = 1;
        }
        __syncthreads();
        for (int idx = tid; idx < NPAD * PITCH; idx += nthreads) T[idx] = 0.0;
    }
    __syncthreads();
    double trace = 0.0;
    for (int i = 0; i < n; ++i) trace += Khi[i * PITCH + i];          // ||R||_F^2 = trace K (every thread, same order)
    // ---- double-double Cholesky, right-looking, lower triangle in place ----
    bool ok = flags[0] == 0;
    for (int j = 0; j < n && ok; ++j) {
        const dd piv = dd{Khi[j * PITCH + j], Klo[j * PITCH + j]};
        // a pivot that cancels to nothing: duplicated frames (io_video.py:51-53) or a numerically singular window
        if (!(piv.hi > 1e-13 * trace)) { ok = false; break; }
        const dd d = dd_sqrt(piv);
        __syncthreads();                                              // every thread has read the pivot
        for (int i = j + tid; i < n; i += nthreads) {
            const dd v = i == j ? d : dd_div(dd{Khi[i * PITCH + j], Klo[i * PITCH + j]}, d);
            Khi[i * PITCH + j] = v.hi; Klo[i * PITCH + j] = v.lo;
        }
        __syncthreads();
        const int m = n - j - 1;                                      // trailing block: rows / columns j+1 .. n-1, lower triangle
        for (int idx = tid; idx < m * m; idx += nthreads) {
            const int a = idx / m, c = idx - a * m;
            if (c > a) continue;
            const int i = j + 1 + a, k2 = j + 1 + c;
            const dd li = dd{Khi[i * PITCH + j], Klo[i * PITCH + j]}, lk = dd{Khi[k2 * PITCH + j], Klo[k2 * PITCH + j]};
            const dd v = dd_add(dd{Khi[i * PITCH + k2], Klo[i * PITCH + k2]}, dd_neg(dd_mul(li, lk)));
            Khi[i * PITCH + k2] = v.hi; Klo[i * PITCH + k2] = v.lo;
        }
        __syncthreads();
    }
    if (!ok) { if (tid == 0) st.refine = 3; return; }
    // ---- L rounded to f64 into T (lower triangle), R^-1 = (L^-1)^T row by row: thread c forms column c of L^-1 by forward substitution ----
    for (int idx = tid; idx < n * n; idx += nthreads) {
        const int i = idx / n, j = idx - i * n;
        T[i * PITCH + j] = j <= i ? Khi[i * PITCH + j] + Klo[i * PITCH + j] : 0.0;
    }
    __syncthreads();
    if (tid < n) {
        const int c = tid;
        double *row = Ri + c * PITCH;                                 // Ri[c][r] = (L^-1)[r][c], r >= c
        row[c] = 1.0 / T[c * PITCH + c];
        for (int r = c + 1; r < n; ++r) {
            double acc = 0.0;
            for (int k2 = c; k2 < r; ++k2) acc += T[r * PITCH + k2] * row[k2];
            row[r] = -acc / T[r * PITCH + r];
        }
    }
    __syncthreads();
    double acc = 0.0;
    for (int idx = tid; idx < n * n; idx += nthreads) { const double v = Ri[(idx / n) * PITCH + idx % n]; acc += v * v; }
    const double ri_f2 = block_sum(acc, red);                         // ||R^-1||_F^2 >= 1 / sigma_min(R)^2
    // ---- Z = L / s, Y = Z^T = R / s (s = ||R||_F: singular values in (0, 1]); padded directions are decoupled unit values ----
    const double s = sqrt(trace) * (1.0 + 1e-12), inv_s = 1.0 / s;
    for (int idx = tid; idx < NPAD * NPAD; idx += nthreads) {
        const int i = idx / NPAD, j = idx - i * NPAD;
        const bool live = i < n && j < n;
        const double lij = live ? (j <= i ? T[i * PITCH + j] * inv_s : 0.0) : (i == j ? 1.0 : 0.0);
        const double lji = live ? (i <= j ? T[j * PITCH + i] * inv_s : 0.0) : (i == j ? 1.0 : 0.0);
        Z[i * PITCH + j] = lij;
        Y[i * PITCH + j] = lji;
    }
    __syncthreads();
    // ---- coupled Newton-Schulz for the polar factor of Y_0; the scaled step of ialm_small.hip with its bound tracked in place ----
    double lo = 0.999 / (sqrt(ri_f2) * s);                            // sigma_min(Y_0) >= 1 / (||R^-1||_F s)
    const int pt = wave < NT ? wave : 0, pti = pt / NB, ptj = pt - pti * NB;
    bool final_step = false, converged = false, plain = false;
    double prev_res2 = 1e300;
    for (int it = 0; it < 100; ++it) {
        if (prev_res2 < 0.25) plain = true;
        const bool unit = plain || !(lo < 0.9999);
        double ta = 1.5, tc = 0.5;
        if (!unit) {
            const double alpha = sqrt(3.0 / (1.0 + lo + lo * lo));
            ta = 1.5 * alpha; tc = 0.5 * alpha * alpha * alpha;
            lo = lo * (ta - tc * lo * lo);
        }
        double r2 = 0.0;
        if (wave < NT) {
            d4 p = mm_tile<PITCH, KS>(Z, Y, pti, ptj, lane);
            d4 tt;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * pti + (lane >> 4) + 4 * r, col = 16 * ptj + (lane & 15);
                const double id = row == col ? 1.0 : 0.0;
                const double e = id - p[r];
                r2 += e * e;
                tt[r] = unit ? id + 0.5 * e : ta * id - tc * p[r];
            }
            store_tile<PITCH>(T, tt, pti, ptj, lane);
        }
        r2 = wave_sum(r2);
        if (lane == 0) red[wave] = r2;
        __syncthreads();
        double res2 = 0.0;
        for (int i = 0; i < nthreads / 64; ++i) res2 += red[i];
        d4 o0 = {0.0, 0.0, 0.0, 0.0}, o1 = {0.0, 0.0, 0.0, 0.0};
        const int j0 = wave, j1 = wave + nthreads / 64;
        const int t0 = j0 % NT, t1 = j1 % NT;
        const int ti0 = t0 / NB, tj0 = t0 - ti0 * NB, ti1 = t1 / NB, tj1 = t1 - ti1 * NB;
        if (j0 < 2 * NT) o0 = j0 < NT ? mm_tile<PITCH, KS>(Y, T, ti0, tj0, lane) : mm_tile<PITCH, KS>(T, Z, ti0, tj0, lane);
        if (j1 < 2 * NT) o1 = j1 < NT ? mm_tile<PITCH, KS>(Y, T, ti1, tj1, lane) : mm_tile<PITCH, KS>(T, Z, ti1, tj1, lane);
        __syncthreads();
        if (j0 < 2 * NT) store_tile<PITCH>(j0 < NT ? Y : Z, o0, ti0, tj0, lane);
        if (j1 < 2 * NT) store_tile<PITCH>(j1 < NT ? Y : Z, o1, ti1, tj1, lane);
        __syncthreads();
        if (!(res2 == res2)) break;
        prev_res2 = res2;
        if (final_step) { converged = true; break; }
        if (res2 < 1e-8 && unit) final_step = true;
    }
    if (!converged) { if (tid == 0) st.refine = 3; return; }
    // ---- W = R^-1 U, B_1 = I - W / (c1 mu_0) ----
    if (wave < NT) {
        const d4 wv = mm_tile<PITCH, KS>(Ri, Y, pti, ptj, lane);
        double *Bm = b.Bm + (int64_t)w * n * n;
        const double scale = inv_mu / c1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * pti + (lane >> 4) + 4 * r, col = 16 * ptj + (lane & 15);
            if (row < n && col < n) Bm[row * n + col] = (row == col ? 1.0 : 0.0) - scale * wv[r];
        }
    }
    if (tid == 0) st.refine = 2;
}

template <int NB>
static void launch_refine_nb(hipStream_t s, const IalmBuffers &b)
{
    static unsigned long long attr_mask = 0;
    if (!ensure_dyn_lds((const void *)k_ialm_refine_start<NB>, RefCfg<NB>::lds_bytes, attr_mask)) return;
    hipLaunchKernelGGL((k_ialm_refine_start<NB>), dim3(b.nwin), dim3(kSmallThreads), RefCfg<NB>::lds_bytes, s, b);
    note_launch();
}

void launch_ialm_refine_start(hipStream_t s, const IalmBuffers &b)
{
    switch ((b.n + 15) / 16) {
    case 1: launch_refine_nb<1>(s, b); break;
    case 2: launch_refine_nb<2>(s, b); break;
    case 3: launch_refine_nb<3>(s, b); break;
    default: launch_refine_nb<4>(s, b); break;
    }
}

}  // namespace swk
