// The small-matrix step of every IALM iteration, one workgroup per window (256 / 512 / 1024 threads for up to 16 / 32 / 64 frames):
// convergence test (image_filtering.py:297), mu <- 1.5 mu (:295), reduction of the per-block Gram
// partials, W = G^(-1/2) of the n x n Gram matrix, B = I - W/mu for the next streaming pass.
//
// G^(-1/2) by the coupled Newton-Schulz iteration
//     Y0 = G/s, Z0 = I;   T = (3I - Z Y)/2;   Y <- Y T,  Z <- T Z;      Z -> (G/s)^(-1/2)
// (s = ||G||_F >= lambda_max, so every eigenvalue of Y0 lies in (0, 1] and the iteration converges,
// x2.25 per step for the small ones -- x6.8 with the scaled steps described in the kernel -- then quadratically).  It is nothing
// but 64x64x64 matrix products (the k-steps that hold live frames: template parameter KS):
// v_mfma_f64_16x16x4_f64 on operands in LDS.  The iterates are polynomials in G (symmetric, commuting)
// only in exact arithmetic, and the iteration is stable only as long as the rounding errors stay of the
// form a true product leaves: mirroring the upper triangle, or reading the left operand transposed
// (both tried: cheaper, and both diverge once cond(G) is large), is not allowed -- every product is the
// full NPAD x NPAD matrix product in the written order.  8-12 steps of 3 products (tools/small_stamp.py times their parts).
// Zero rows of G (all-zero "null" frames, io_video.py:40-44) are decoupled by construction, iterate
// as a unit diagonal, and get weight 0 at the end (null frames are excluded).
// If the iteration has not reached the quadratic regime after 60 steps (numerically singular G) the
// cyclic Jacobi eigen-solver below takes over; it also serves as the reference implementation
// (swk_set_eig_method).
#include "swk_internal.h"
#include "ialm_small_dev.h"

namespace swk {


// Diagnostic build (tools/small_stamp.sh, -DSWK_SMALL_STAMP): window 0's thread 0 leaves the 100 MHz wall clock at the phase
// boundaries of every launch (row = iteration k); swk_small_stamp_read copies the table out.  Not part of libswk.so.
#ifdef SWK_SMALL_STAMP
__device__ long long g_small_stamp[64][10];          // 0..6 wall clock at the phase boundaries, 8 / 9 shader clock around the solver loop
#define SWK_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && k < 64) g_small_stamp[k][i] = wall_clock64(); } while (0)
#define SWK_STAMP_CYC(i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && k < 64) g_small_stamp[k][i] = clock64(); } while (0)
__device__ long long g_ns_stamp[64][8];          // shader clock inside solver step 3 of every launch
#define SWK_STAMP_NS(i) do { if (it == 3 && threadIdx.x == 0 && blockIdx.x == 0 && k < 64) g_ns_stamp[k][i] = clock64(); } while (0)
#else
#define SWK_STAMP_NS(i) do { } while (0)
#define SWK_STAMP_CYC(i) do { } while (0)
#define SWK_STAMP(i) do { } while (0)
#endif

// Coefficients of the scaled Newton-Schulz steps (see the kernel): the sequence depends on nothing but the starting bound lo = 1e-3, so
// the host forms it once (IEEE sqrt and division, as the kernel did per step and thread: 430 cycles of a step) and passes it by value.
// A window's Gram matrix moves little from one IALM iteration to the next, and the last solve leaves ||Z||_F^2 = sum_i 1 / x_i^2 behind:
// sqrt(n) / ||Z||_F is the scale of the small singular values, half of it the next solve's starting bound -- the table whose start lies
// at or below that is used (kNsStarts of them; 12 solver steps from 1e-3 become 8 to 11).
constexpr int kNsScaled = 12, kNsStarts = 6;
struct NsSteps {
    double start[kNsStarts];                                   // the starting bounds, ascending; start[0] = 1e-3 serves a first solve
    double ta[kNsStarts][kNsScaled], tc[kNsStarts][kNsScaled];
    int count[kNsStarts];                                      // steps until the tracked bound reaches 1
};

template <int NB> struct NsCfg {
    static constexpr int NPAD = 16 * NB;
    static constexpr int PITCH = NPAD + 2;
    static constexpr int NT = NB * NB;
    static constexpr size_t mat = (size_t)NPAD * PITCH;
    static constexpr size_t jac = (size_t)3 * kMaxN * kJac;                    // G, V, W of the fallback
    static constexpr size_t doubles = (3 * mat > jac ? 3 * mat : jac) + kSmallThreads + 64 + kMaxN;
    static constexpr size_t lds_bytes = doubles * sizeof(double) + 128 * sizeof(int);   // pq[32], flag, dead[64]
};

// First solve of a window (k = 0): remember how ill-conditioned G_1 is and ask for the accurate first iteration (ialm_refine.hip)
// when the Gram route's error in it, about eps * cond(G_1) / mu_0 in A_1, is not negligible.  cond_sum = ||G_1||_F sum_i 1 / lambda_i.
__device__ __forceinline__ void note_conditioning(const IalmBuffers &b, IalmWin &st, double cond_sum, double inv_mu)
{
    st.cond_sum = cond_sum;
    if (b.refine > 0.0 && 1.1e-16 * cond_sum * inv_mu > b.refine) st.refine = 1;
}

template <int NB, int KS>
__global__ __launch_bounds__(kSmallThreads) void k_ialm_small(IalmBuffers b, int k, double lmbda, double tol, int maxiter, int method, NsSteps steps)
{
    using C = NsCfg<NB>;
    constexpr int NPAD = C::NPAD, PITCH = C::PITCH, NT = C::NT;
    extern __shared__ double sm[];
    double *Y = sm, *Z = sm + C::mat, *T = sm + 2 * C::mat;
    // up to 48 frames a second pair of matrices fits (inside the Jacobi fallback's space): a solver step writes its Y T, T Z there and
    // the pairs change roles -- one barrier per step less than overwriting in place
    constexpr bool DB = 5 * C::mat <= C::jac;
    double *Yn = DB ? sm + 3 * C::mat : Y, *Zn = DB ? sm + 4 * C::mat : Z;
    double *red = sm + (3 * C::mat > C::jac ? 3 * C::mat : C::jac);        // [1024]
    double2 *cs = (double2 *)(red + kSmallThreads);                           // [32]
    double *wgt = red + kSmallThreads + 64;                                   // [64]
    int *ints = (int *)(wgt + kMaxN);                                         // pq[32], flag, sweeps, dead mask...
    const int w = blockIdx.x, tid = threadIdx.x, n = b.n, nthreads = blockDim.x;      // 256 threads for n <= 16, 512 for n <= 32, else 1024
    const int lane = tid & 63, wave = tid >> 6;
    IalmScal cur;
    SWK_STAMP(0);
    if (!small_prologue(b, w, k, lmbda, tol, maxiter, red, cur)) return;
    SWK_STAMP(1);
    double *Bm = b.Bm + (int64_t)w * n * n;
    IalmWin &st = b.win[w];

    bool need_jacobi = method == 1;
    int ns_iters = 0;
    if (!need_jacobi) {
        // ---- Y = G (padded with zeros), Z = I ----
        for (int idx = tid; idx < NPAD * PITCH; idx += nthreads) { Y[idx] = 0.0; Z[idx] = 0.0; }
        __syncthreads();
        gram_reduce(b, w, Y, PITCH, k);
        __syncthreads();
        SWK_STAMP(2);
        // s = ||G||_F (fixed-order reduction)
        double acc = 0.0;
        for (int idx = tid; idx < n * n; idx += nthreads) { const double v = Y[(idx / n) * PITCH + idx % n]; acc += v * v; }
        const double sc = sqrt(block_sum(acc, red)) * (1.0 + 1e-12);
        const double inv_sc = 1.0 / sc;
        if (tid < NPAD) ints[40 + tid] = (tid >= n || Y[tid * PITCH + tid] == 0.0) ? 1 : 0;      // dead directions
        __syncthreads();
        for (int idx = tid; idx < NPAD * NPAD; idx += nthreads) {
            const int i = idx / NPAD, j = idx - i * NPAD;
            const bool dead = ints[40 + i] || ints[40 + j];
            Y[i * PITCH + j] = dead ? (i == j ? 1.0 : 0.0) : Y[i * PITCH + j] * inv_sc;
            Z[i * PITCH + j] = i == j ? 1.0 : 0.0;
        }
        __syncthreads();
        bool final_step = false, converged = false;
        SWK_STAMP(3);
        SWK_STAMP_CYC(8);
        // Scaled Newton-Schulz: with the singular values x = sqrt(eig(ZY)) known to lie in [lo, 1], the step
        // x <- x (3 alpha / 2 - alpha^3 x^2 / 2), alpha = sqrt(3 / (1 + lo + lo^2)), maps [lo, 1] onto [p(lo), 1] with
        // p(lo) = p(1): small values grow 2.6x per step instead of 1.5x.  Any x in (0, 1] stays in (0, 1] for any
        // alpha in [1, sqrt 3], so a wrong guess for lo only costs speed; as lo -> 1 the step is the plain one.
        // lo starts at 1e-3 for a window's first solve, i.e. cond(G) up to 1e6 relative to ||G||_F, later where the last solve's
        // ||Z||_F puts it (NsSteps).
        // Once ||I - ZY||_F < 1/2 (every x above 0.7) the plain step takes over: it converges quadratically from there
        // and the stopping rule below counts plain steps.
        // (The bound is tracked by the recurrence lo <- lo (3 alpha / 2 - alpha^3 lo^2 / 2): it depends on nothing but its start, so the
        // host hands the coefficients of its steps over as tables, one per starting bound: NsSteps.)
        double prev_res2 = 1e300;
        bool plain = n == 1;
        // the table this solve runs on (uniform), then lane i keeps the coefficients of its scaled step i (read back with a lane
        // index: no memory access inside the loop)
        int tab = 0;
        {
            const double zf2 = st.zf2;
            const double guess = zf2 > 0.0 ? 0.5 * sqrt((double)n / zf2) : 0.0;
#pragma unroll
            for (int i = 1; i < kNsStarts; ++i) tab = guess >= steps.start[i] ? i : tab;
        }
        const int nscaled = steps.count[tab];
        const double my_ta = lane < kNsScaled ? steps.ta[tab][lane] : 1.5, my_tc = lane < kNsScaled ? steps.tc[tab][lane] : 0.5;
        // this wave's tile of Z Y (phase 1) never changes: which of the lane's four elements sit in a dead row or column
        const int pt = wave < NT ? wave : 0, pti = pt / NB, ptj = pt - pti * NB;
        int deadm = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (ints[40 + 16 * pti + (lane >> 4) + 4 * r] | ints[40 + 16 * ptj + (lane & 15)]) deadm |= 1 << r;
        for (int it = 0; it < 60; ++it) {
            SWK_STAMP_NS(0);
            if (prev_res2 < 0.25) plain = true;
            const bool unit = plain || it >= nscaled;          // alpha == 1: the plain step
            const double ta = unit ? 1.5 : lane_value(my_ta, it), tc = unit ? 0.5 : lane_value(my_tc, it);
            // phase 1: P = Z Y;  T = ta I - tc P  (= (3I - P)/2 once alpha = 1);  residual ||I - P||_F^2
            double r2 = 0.0;
            SWK_STAMP_NS(1);
            if (wave < NT) {
                d4 p = mm_tile<PITCH, KS>(Z, Y, pti, ptj, lane);
                d4 tt;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * pti + (lane >> 4) + 4 * r, col = 16 * ptj + (lane & 15);
                    const double id = row == col ? 1.0 : 0.0;
                    const double e = id - p[r];
                    // dead directions (zero rows/columns of G) are decoupled 1x1 blocks held at 1 (up to 4 KS; beyond, the
                    // skipped k-steps leave them 0: they never enter a live element, and they stay out of the residual)
                    const bool dead = (deadm >> r) & 1;
                    r2 += dead ? 0.0 : e * e;
                    tt[r] = dead ? id : (unit ? id + 0.5 * e : ta * id - tc * p[r]);   // (3I - P)/2 = I + (I - P)/2
                }
                store_tile<PITCH>(T, tt, pti, ptj, lane);
            }
            SWK_STAMP_NS(2);
            r2 = wave_sum(r2);
            if (lane == 0) red[wave] = r2;
            SWK_STAMP_NS(3);
            __syncthreads();
            double res2 = 0.0;
            for (int i = 0; i < nthreads / 64; ++i) res2 += red[i];
            SWK_STAMP_NS(4);
            // phase 2: Y T and T Z into registers (both read the old Y, Z), then write back
            d4 o0 = {0.0, 0.0, 0.0, 0.0}, o1 = {0.0, 0.0, 0.0, 0.0};
            const int j0 = wave, j1 = wave + nthreads / 64;
            const int t0 = j0 % NT, t1 = j1 % NT;
            const int ti0 = t0 / NB, tj0 = t0 - ti0 * NB, ti1 = t1 / NB, tj1 = t1 - ti1 * NB;
            if (j0 < 2 * NT) o0 = j0 < NT ? mm_tile<PITCH, KS>(Y, T, ti0, tj0, lane) : mm_tile<PITCH, KS>(T, Z, ti0, tj0, lane);
            if (j1 < 2 * NT) o1 = j1 < NT ? mm_tile<PITCH, KS>(Y, T, ti1, tj1, lane) : mm_tile<PITCH, KS>(T, Z, ti1, tj1, lane);
            SWK_STAMP_NS(5);
            if (!DB) __syncthreads();
            SWK_STAMP_NS(6);
            if (j0 < 2 * NT) store_tile<PITCH>(j0 < NT ? Yn : Zn, o0, ti0, tj0, lane);
            if (j1 < 2 * NT) store_tile<PITCH>(j1 < NT ? Yn : Zn, o1, ti1, tj1, lane);
            __syncthreads();
            if (DB) { double *t_ = Y; Y = Yn; Yn = t_; t_ = Z; Z = Zn; Zn = t_; }
            SWK_STAMP_NS(7);
            ns_iters = it + 1;
            if (!(res2 == res2)) break;                                      // NaN: give up, Jacobi decides
            prev_res2 = res2;
            if (final_step) { converged = true; break; }
            if (res2 < 1e-8 && unit) final_step = true;              // ||I - ZY|| < 1e-4: two more plain steps reach 1e-16
        }
        SWK_STAMP_CYC(9);
        SWK_STAMP(4);
        double solved_zf2 = 0.0;
        if (converged) {
            // A (numerically) zero eigenvalue that is not a zero row of G -- the duplicated last frame of every video
            // (io_video.py:51-53) makes two columns of M equal -- converges here to a weight of 1e6 and more on a
            // direction made of rounding noise.  ||Z||_F^2 = sum s / lambda_i gives it away: such a matrix goes to the
            // Jacobi solver, which gives eigenvalues below 1e-13 lambda_max the weight 0 (the project's defined
            // behaviour for zero singular directions).
            double zacc = 0.0;
            for (int idx = tid; idx < n * n; idx += nthreads) { const double v = Z[(idx / n) * PITCH + idx % n]; zacc += v * v; }
            const double zf2 = block_sum(zacc, red);
            solved_zf2 = zf2;
            if (!(zf2 < 1e11)) converged = false;
            else if (tid == 0) st.zf2 = zf2;
        }
        SWK_STAMP(5);
        if (converged) {
            const double wscale = 1.0 / sqrt(sc);
            for (int idx = tid; idx < n * n; idx += nthreads) {
                const int i = idx / n, j = idx - i * n;
                const bool dead = ints[40 + i] || ints[40 + j];
                const double wv = dead ? 0.0 : 0.5 * (Z[i * PITCH + j] + Z[j * PITCH + i]) * wscale;
                Bm[idx] = (i == j ? 1.0 : 0.0) - cur.inv_mu * wv;
            }
            if (tid == 0) {
                st.sweeps = ns_iters;
                if (k == 0) note_conditioning(b, st, solved_zf2, cur.inv_mu);
            }
            SWK_STAMP(6);
            return;
        }
        need_jacobi = true;
        __syncthreads();
    }
    // ---- Jacobi (reference method / fallback) ----
    double *G = sm, *V = sm + kMaxN * kJac, *Wm = sm + 2 * kMaxN * kJac;
    gram_reduce(b, w, G, kJac, k);
    __syncthreads();
    int sweeps = 0;
    jacobi_invsqrt(G, V, Wm, n, cs, ints, wgt, ints + 36, &sweeps);
    for (int idx = tid; idx < n * n; idx += nthreads) {
        const int i = idx / n, j = idx - i * n;
        Bm[idx] = (i == j ? 1.0 : 0.0) - cur.inv_mu * Wm[i * kJac + j];
    }
    if (tid == 0) {
        st.sweeps = 100 + sweeps;
        if (k == 0) {
            // the same estimate from the eigenvalues (wgt = lambda^-1/2, 0 for a dead direction): ||G||_F sum 1 / lambda_i
            double f2 = 0.0, inv = 0.0;
            for (int i = 0; i < n; ++i) {
                const double wi = wgt[i];
                if (wi > 0.0) { const double lam = 1.0 / (wi * wi); f2 += lam * lam; inv += wi * wi; }
            }
            note_conditioning(b, st, sqrt(f2) * inv, cur.inv_mu);
        }
    }
}

// With many slabs per window (small batches use many blocks per window) the sum is done by the whole chip first: a
// block of 16 waves takes 64 matrix entries (lane = entry: coalesced rows of a slab), each wave a sixteenth of the slabs
// with four loads in flight, and the partial sums are combined in a fixed order -- reproducible results.  (One thread per
// entry walking all slabs, the first version, was a chain of up to 512 dependent cache misses: 32 us for a lone window.)
__global__ __launch_bounds__(1024) void k_gram_reduce(IalmBuffers b)
{
    const int w = blockIdx.y, n = b.n;
    if (b.win[w].done) return;
    __shared__ double part[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + lane;
    // the passes only write frame-block pairs ib <= jb
    const bool live = idx < n * n && ((idx / n) >> 4) <= ((idx % n) >> 4);
    double *gp = b.gpart + (int64_t)w * b.nblk * n * n;
    const int per = (b.nblk + 15) / 16, b0 = wave * per, b1 = min(b0 + per, b.nblk);
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (live) {
        const int64_t nn = (int64_t)n * n;
        int bk = b0;
        for (; bk + 3 < b1; bk += 4) {
            a0 += gp[(bk + 0) * nn + idx]; a1 += gp[(bk + 1) * nn + idx];
            a2 += gp[(bk + 2) * nn + idx]; a3 += gp[(bk + 3) * nn + idx];
        }
        for (; bk < b1; ++bk) a0 += gp[bk * nn + idx];
    }
    part[wave][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (wave == 0 && live) {
        double acc = 0.0;
        for (int g = 0; g < 16; ++g) acc += part[g][lane];
        gp[idx] = acc;
    }
}

// ---------------------------------------------------------------------------------
// Windows of 65 .. 128 frames (FrameQueue(queue_size) is unbounded in the reference, data_structures.py:120; no BASELINE configuration
// uses such a queue).  The three n x n matrices of the eigen-solve do not fit the LDS next to each other, so they live in global memory
// (131 KB each at n = 128: L2 resident) and the solver is the cyclic Jacobi iteration alone -- the same code, another pitch.  A
// correctness path: milliseconds per IALM iteration, one workgroup per window.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(kSmallThreads) void k_ialm_small_wide(IalmBuffers b, int k, double lmbda, double tol, int maxiter, double *work)
{
    __shared__ double red[kSmallThreads];
    __shared__ double2 cs[kMaxNWide / 2];
    __shared__ double wgt[kMaxNWide];
    __shared__ int pq[kMaxNWide / 2];
    __shared__ int flags[2];
    const int w = blockIdx.x, tid = threadIdx.x, n = b.n, nthreads = blockDim.x, pitch = n + 2;
    IalmScal cur;
    if (!small_prologue(b, w, k, lmbda, tol, maxiter, red, cur)) return;
    double *G = work + (size_t)w * 3 * pitch * pitch, *V = G + (size_t)pitch * pitch, *Wm = V + (size_t)pitch * pitch;
    gram_reduce(b, w, G, pitch, k);
    __syncthreads();
    int sweeps = 0;
    jacobi_invsqrt(G, V, Wm, n, cs, pq, wgt, flags, &sweeps, pitch);
    double *Bm = b.Bm + (int64_t)w * n * n;
    for (int idx = tid; idx < n * n; idx += nthreads) {
        const int i = idx / n, j = idx - i * n;
        Bm[idx] = (i == j ? 1.0 : 0.0) - cur.inv_mu * Wm[i * pitch + j];
    }
    if (tid == 0) b.win[w].sweeps = 100 + sweeps;
}

size_t ialm_small_wide_doubles(int n) { return (size_t)3 * (n + 2) * (n + 2); }

void launch_ialm_small_wide(hipStream_t s, const IalmBuffers &b, int k, double lmbda, double tol, int maxiter, double *work)
{
    hipLaunchKernelGGL(k_ialm_small_wide, dim3(b.nwin), dim3(kSmallThreads), 0, s, b, k, lmbda, tol, maxiter, work);
    note_launch();
}

void launch_gram_reduce(hipStream_t s, const IalmBuffers &b)
{
    hipLaunchKernelGGL(k_gram_reduce, dim3((b.n * b.n + 63) / 64, b.nwin), dim3(1024), 0, s, b);
}

static NsSteps ns_steps()
{
    NsSteps t{};
    const double starts[kNsStarts] = {1e-3, 2.5e-3, 6e-3, 1.5e-2, 4e-2, 1e-1};
    for (int j = 0; j < kNsStarts; ++j) {
        double lo = t.start[j] = starts[j];
        t.count[j] = 0;
        for (int i = 0; i < kNsScaled; ++i) {
            double alpha = 1.0;
            if (lo < 0.9999) {
                alpha = sqrt(3.0 / (1.0 + lo + lo * lo));
                t.count[j] = i + 1;
            }
            t.ta[j][i] = 1.5 * alpha;
            t.tc[j][i] = 0.5 * alpha * alpha * alpha;
            lo = lo < 0.9999 ? lo * (t.ta[j][i] - t.tc[j][i] * lo * lo) : 1.0;
        }
    }
    return t;
}

template <int NB, int KS>
static void launch_small_ks(hipStream_t s, const IalmBuffers &b, int k, double lmbda, double tol, int maxiter, int method)
{
    static unsigned long long attr_mask = 0;
    static const NsSteps steps = ns_steps();
    if (!ensure_dyn_lds((const void *)k_ialm_small<NB, KS>, NsCfg<NB>::lds_bytes, attr_mask)) return;
    // up to 32 frames the matrices are 2 x 2 tiles: eight waves take the eight products Y T, T Z of a step one each (two per SIMD, so
    // that one's matrix instructions run under the other's operand reads) without idling eight more at every barrier; one tile: four
    hipLaunchKernelGGL((k_ialm_small<NB, KS>), dim3(b.nwin), dim3(NB == 1 ? 256 : NB == 2 ? 512 : kSmallThreads), NsCfg<NB>::lds_bytes, s, b, k, lmbda,
                       tol, maxiter, method, steps);
    note_launch();
}

template <int NB>
static void launch_small_nb(hipStream_t s, const IalmBuffers &b, int k, double lmbda, double tol, int maxiter, int method)
{
    // k-steps that hold live frames, rounded up to even: 4 NB, or two fewer when the last eight rows of the padded matrix are empty
    if (b.n <= 16 * NB - 8) launch_small_ks<NB, 4 * NB - 2>(s, b, k, lmbda, tol, maxiter, method);
    else launch_small_ks<NB, 4 * NB>(s, b, k, lmbda, tol, maxiter, method);
}

void launch_ialm_small(hipStream_t s, const IalmBuffers &b, int k, double lmbda, double tol, int maxiter, int method)
{
    switch ((b.n + 15) / 16) {
    case 1: launch_small_nb<1>(s, b, k, lmbda, tol, maxiter, method); break;
    case 2: launch_small_nb<2>(s, b, k, lmbda, tol, maxiter, method); break;
    case 3: launch_small_nb<3>(s, b, k, lmbda, tol, maxiter, method); break;
    default: launch_small_nb<4>(s, b, k, lmbda, tol, maxiter, method); break;
    }
}

}  // namespace swk

#ifdef SWK_SMALL_STAMP
#pragma GCC visibility push(default)
extern "C" int32_t swk_small_stamp_read(long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(swk::g_small_stamp), sizeof(swk::g_small_stamp)) == hipSuccess ? 0 : -1;
}
extern "C" int32_t swk_ns_stamp_read(long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(swk::g_ns_stamp), sizeof(swk::g_ns_stamp)) == hipSuccess ? 0 : -1;
}
#pragma GCC visibility pop
#endif
