// Device helpers shared by the small-matrix kernels of the IALM (ialm_small.hip: the per-iteration step; ialm_refine.hip: the
// accurate first iteration of ill-conditioned windows): wave / workgroup sums, the convergence prologue, the Gram slab reduction,
// the cyclic Jacobi eigen-solver and 16 x 16 tile products on the f64 matrix cores over matrices in LDS.
#pragma once
#include "swk_internal.h"

namespace swk {

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int kSmallThreads = 1024;
constexpr int kJac = 65;              // LDS row pitch of the Jacobi matrices

__device__ __forceinline__ void round_robin_pair(int m, int r, int k, int &p, int &q)
{
    if (k == 0) { p = m - 1; q = r; }
    else { p = (r + k) % (m - 1); q = (r - k + (m - 1)) % (m - 1); }
    if (p > q) { int tmp = p; p = q; q = tmp; }
}

// Sum over the 64 lanes of a wave by DPP moves (quad permutes, row shifts, row broadcasts: register-file operations) instead of six
// dependent trips through the LDS crossbar (__shfl: 625 cycles of a solver step, tools/small_stamp.py); every lane gets the total.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v)
{
    const long long bits = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)bits, CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(bits >> 32), CTRL, ROW_MASK, 0xf, false);
    // lanes without a source (row shifts at a row's start, rows masked out) read +0.0
    return v + __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double wave_sum(double v)
{
    v = dpp_add<0xb1, 0xf>(v);          // quad_perm:[1,0,3,2]
    v = dpp_add<0x4e, 0xf>(v);          // quad_perm:[2,3,0,1]: every lane holds its quad's sum
    v = dpp_add<0x114, 0xf>(v);         // row_shr:4
    v = dpp_add<0x118, 0xf>(v);         // row_shr:8: lanes 12..15 of a row hold the row's sum
    v = dpp_add<0x142, 0xa>(v);         // row_bcast:15 into rows 1, 3
    v = dpp_add<0x143, 0xc>(v);         // row_bcast:31 into rows 2, 3: lane 63 holds the wave's sum
    const long long bits = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int)bits, 63), hi = __builtin_amdgcn_readlane((int)(bits >> 32), 63);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}

// v of lane `index` (uniform), for every lane
__device__ __forceinline__ double lane_value(double v, int index)
{
    const long long bits = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int)bits, index), hi = __builtin_amdgcn_readlane((int)(bits >> 32), index);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}

// Sum over the workgroup, the waves' sums added in wave order (reproducible); every thread gets it.  red: one double per wave.
__device__ __forceinline__ double block_sum(double v, double *red)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    v = wave_sum(v);
    __syncthreads();                                             // an earlier sum may still be being read
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double total = 0.0;
    for (int i = 0; i < nwaves; ++i) total += red[i];
    return total;
}

// Maximum over the workgroup; every thread gets it.  red: one double per wave.
__device__ __forceinline__ double block_max(double v, double *red)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    for (int off = 32; off; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double m = 0.0;
    for (int i = 0; i < nwaves; ++i) m = fmax(m, red[i]);
    return m;
}

// Convergence test and scalar update.  Returns false when the window is (now) finished.
__device__ __forceinline__ bool small_prologue(const IalmBuffers &b, int w, int k, double lmbda, double tol, int maxiter,
                                               double *red, IalmScal &cur)
{
    const int tid = threadIdx.x, nblk = b.nblk, nthreads = blockDim.x;
    IalmWin &st = b.win[w];
    if (st.done) return false;
    // every thread takes its copy of the window state BEFORE the reduction's barriers: thread 0 rewrites ru / wu
    // further down, and all waves must take the same branches around the barriers that follow
    const bool full = st.ru != 0;
    const double dnorm = st.dnorm;
    const double inv_mu_prev = st.cur.inv_mu;                    // 1 / mu_{k-1}: the scale of U_{k-1} = Y_{k-1} / mu_{k-1}
    if (k >= 1) {
        double acc = 0.0;
        for (int i = tid; i < nblk; i += nthreads) acc += b.zzpart[(int64_t)w * nblk + i];
        const double zz = block_sum(acc, red);
        // M-state pass (b.guard > 0): how wrong can the stopping norm be?  z = P_k - U_{k-1} is formed in float32 from the binary16 copy
        // of U_{k-1} (relative rounding eps_h = 2^-11; below 2^-14 x 128 absolute 2^-25 x 128) and summed in float32:
        //   ||z^||^2 - ||z||^2 = -2 z.delta + ||delta||^2,   |delta_i| <= eps_h |U_i|
        //   * ||delta||^2 <= eps_h^2 ||U_{k-1}||_F^2 = eps_h^2 n / mu_{k-1}^2 EXACTLY known: Y_{k-1} = polar(M_{k-1}) has n unit singular
        //     values (image_filtering.py:290-294 with the always-full svp of :285), so ||Y||_F = sqrt(n);
        //   * z.delta is a sum of independent bounded terms (the rounding errors): |z.delta| <= kappa eps_h max|U| ||z|| except with
        //     probability 2 exp(-kappa^2 / 2) (Hoeffding; kappa = 8: 2.5e-14) -- max|U| is MEASURED by the passes near the decision.
        //     (The deterministic bound ||delta|| / ||z|| <= eps_h sqrt(n) / (mu_{k-1} tol ||X||_F) is 1 % for a 64-frame window that stops
        //     after 15 iterations: 300 x what is observed and no basis for a band)
        //   * float32 accumulation: a lane adds nlane terms, relative error <= nlane 2^-24 in the worst case; lanes are added in float64.
        // The band in which the comparison with tol is not trusted is the larger of the configured one and 4 x this bound.
        double band = b.guard > 0.0 ? b.guard : 0.0;
        if (b.U != nullptr && full) {          // the M-state pass (the bound is kept as a diagnostic also when the band is switched off)
            double um = 0.0;
            for (int i = tid; i < nblk; i += nthreads) um = fmax(um, b.zzpart[(int64_t)(b.nwin + w) * nblk + i]);
            um = block_max(um, red);
            const double T = tol * dnorm, eps_h = 4.8828125e-4;
            const int ntiles = (b.P + 15) >> 4, groups = (ntiles + 7) >> 3;
            const double nlane = 2.0 * (double)((groups + nblk - 1) / nblk) * (double)((b.n + 3) >> 2);
            double e = 0.5 * (nlane + 2.0) * 5.9604644775390625e-8;
            if (k >= 2) {
                const double rho = eps_h * sqrt((double)b.n) * inv_mu_prev / T;
                e += 8.0 * eps_h * um / T + 0.5 * rho * rho + sqrt((double)b.P * (double)b.n) * 3.814697265625e-6 / T;
            }
            if (tid == 0) st.norm_err = e;
            if (band > 0.0) band = fmax(band, 4.0 * e);
        }
        // a pass that read all of U_{k-1} delivers ||Z_k||^2; one that read only frames 0..3 of it (IalmWin::ru == 0)
        // delivers the sum over those frames: a LOWER bound
        const double ratio = sqrt(zz) / dnorm;                   // :297
        if (tid == 0) {
            // bookkeeping for the roofline: what pass k had to move per element (M-state pass; 1/16-byte units):
            // X 1 + M 8 written (+ 8 read after the first pass) + U 2 or 2/16 each way + the sparse image if stored
            unsigned u = 16 + 128 + (st.wu ? 32 : 2) + (st.ws ? 16 : 0);
            if (k >= 2) u += 128 + (st.ru ? 32 : 2);
            st.pass_b16 += u;
        }
        // M-state pass: the norm is a float32 sum over a binary16 copy of Y/mu (relative error about 1e-6).  Inside the guard
        // band the comparison with tol is not trusted: the window stops here and the host runs it again with the float64 norm.
        if (full && band > 0.0 && k < maxiter && fabs(ratio / tol - 1.0) < band) {
            if (tid == 0) { st.iter = k; st.done = 1; st.redo |= 4; atomicSub(b.active, 1); }
            return false;
        }
        if (!full && ratio < tol * (1.0 + band) && k < maxiter) {
            // the bound cannot rule out that this iteration is the last: give the window up, the host runs the
            // batch again with every norm formed
            if (tid == 0) { st.iter = k; st.done = 1; st.redo |= 2; atomicSub(b.active, 1); }
            return false;
        }
        if (tid == 0 && full) st.last_ratio = ratio;
        if ((full && ratio < tol) || k >= maxiter) {
            // the answer's sparse image is the one pass k-1 wrote (ialm_mfma.hip, M-state pass): if that pass ran
            // with its stores switched off, the speculation below failed and the host runs the batch again
            if (tid == 0) { st.iter = k; st.done = 1; if (!st.ws_prev) st.redo |= 1; atomicSub(b.active, 1); }
            return false;
        }
        if (tid == 0) {
            if (full) st.last_ratio = ratio;
            const double known = st.last_ratio;
            // far from the stopping threshold the next iteration cannot be the last but one: its pass skips the
            // sparse-image stores (a u8 plane written in 16-byte row pieces costs 2.5x its share of the bytes) ...
            st.ws_prev = st.ws; st.ws = (b.spec <= 0.0 || known < b.spec * tol) ? 1 : 0;
            // ... and further out the full norm is formed every other iteration only: a pass that writes just
            // frames 0..3 of U is followed by one that reads just those (3.75 of 21 B per element saved per pair);
            // the partial norm still proves that the skipped iteration did not converge
            const bool far = b.nspec > 0.0 && known >= b.nspec * tol;
            const int wrote = st.wu;
            st.ru = wrote;                                       // pass k+1 can read all of U_k only if pass k wrote it
            st.wu = far ? (wrote ? 0 : 1) : 1;
        }
    } else if (tid == 0) {
        st.ws_prev = st.ws; st.ws = 1;
        st.ru = 1;                                               // pass 1 forms ||Z_1|| from U_0 = X / (dual mu_0)
        st.wu = b.nspec > 0.0 ? 0 : 1;
    }
    cur = st.nxt;
    IalmScal nxt;
    nxt.mu = cur.mu * 1.5;                                       // :295 (min(mu*rho, mu*1e7) == mu*rho)
    nxt.inv_mu = 1.0 / nxt.mu;
    nxt.thr = lmbda / nxt.mu;
    __syncthreads();
    if (tid == 0) { st.cur = cur; st.nxt = nxt; st.iter = k; }
    return true;
}

// Deterministic reduction of the per-block Gram partials into an LDS matrix of the given pitch.
// The MFMA pass only fills frame-block pairs ib <= jb (G is symmetric): the rest is mirrored.
__device__ __forceinline__ void gram_reduce(const IalmBuffers &b, int w, double *G, int pitch, int k)
{
    const int n = b.n, nblk = b.nblk, nred = b.nred, nthreads = blockDim.x;
    // first iteration from the integer kernel: the slabs hold X^T X and M_1 = (1 + 1/(mu_0 dual)) X (ialm_gram8.hip);
    // st.cur is mu_0 at this point (small_prologue has run)
    const IalmWin &st = b.win[w];
    double scale = 1.0;
    if (k == 0 && st.int_gram) { const double s1 = 1.0 + st.cur.inv_mu / st.dual_norm; scale = s1 * s1; }
    const double *gp = b.gpart + (int64_t)w * nblk * n * n;
    for (int idx = threadIdx.x; idx < n * n; idx += nthreads) {
        const int i = idx / n, j = idx - i * n;
        const int src = (i >> 4) <= (j >> 4) ? idx : j * n + i;
        double acc = 0.0;
        for (int bk = 0; bk < nred; ++bk) acc += gp[(int64_t)bk * n * n + src];
        G[i * pitch + j] = acc * scale;
    }
}

// ---------------------------------------------------------------------------------
// Cyclic Jacobi, round-robin ordering: every round applies m/2 disjoint rotations, G <- J^T G J per 2x2
// block in place, V <- V J alongside.  Leaves W = V diag(lambda^-1/2) V^T in Wout (pitch kJac);
// eigenvalues below 1e-13 lambda_max get weight 0.
// ---------------------------------------------------------------------------------
static __device__ void jacobi_invsqrt(double *G, double *V, double *Wout, int n, double2 *cs, int *pq, double *wgt, int *flag, int *sweeps_out,
                               int pitch = kJac)
{
    const int tid = threadIdx.x, nthreads = blockDim.x;
    for (int idx = tid; idx < n * n; idx += nthreads) { const int i = idx / n, j = idx - i * n; V[i * pitch + j] = i == j ? 1.0 : 0.0; }
    if ((n & 1) && tid <= n) {                 // zero row/column at the dummy index of an odd n
        G[n * pitch + tid] = 0.0;
        G[tid * pitch + n] = 0.0;
    }
    __syncthreads();
    const int m = n + (n & 1), half = m / 2;
    int sweeps = 0;
    for (int sweep = 0; sweep < 40; ++sweep) {
        if (tid == 0) *flag = 0;
        __syncthreads();
        for (int r = 0; r < m - 1; ++r) {
            if (tid < half) {
                int p, q;
                round_robin_pair(m, r, tid, p, q);
                double c = 1.0, s = 0.0;
                if (q < n) {
                    const double gpq = G[p * pitch + q], gpp = G[p * pitch + p], gqq = G[q * pitch + q];
                    const double g2 = gpq * gpq, dd = fabs(gpp * gqq);
                    if (gpq != 0.0 && g2 > 1e-30 * dd) {
                        if (g2 > 1e-16 * dd) *flag = 1;
                        // t = sgn(tau) / (|tau| + sqrt(1 + tau^2)), tau = (gqq - gpp) / (2 gpq), without forming tau
                        const double d = gqq - gpp, b2 = 2.0 * gpq;
                        const double tt = (d >= 0.0 ? b2 : -b2) / (fabs(d) + sqrt(d * d + b2 * b2));
                        c = rsqrt(1.0 + tt * tt);
                        s = tt * c;
                    }
                }
                cs[tid] = make_double2(c, s);
                pq[tid] = p | (q << 8);
            }
            __syncthreads();
            for (int idx = tid; idx < half * half; idx += nthreads) {
                const int ka = idx / half, kb = idx - ka * half;
                const double2 ra = cs[ka], rb = cs[kb];
                if (ra.y == 0.0 && rb.y == 0.0) continue;
                const int pqa = pq[ka], pqb = pq[kb];
                const int pa = pqa & 255, qa = pqa >> 8, pb = pqb & 255, qb = pqb >> 8;
                const double g00 = G[pa * pitch + pb], g01 = G[pa * pitch + qb];
                const double g10 = G[qa * pitch + pb], g11 = G[qa * pitch + qb];
                const double r00 = ra.x * g00 - ra.y * g10, r01 = ra.x * g01 - ra.y * g11;
                const double r10 = ra.y * g00 + ra.x * g10, r11 = ra.y * g01 + ra.x * g11;
                G[pa * pitch + pb] = rb.x * r00 - rb.y * r01;
                G[pa * pitch + qb] = rb.y * r00 + rb.x * r01;
                G[qa * pitch + pb] = rb.x * r10 - rb.y * r11;
                G[qa * pitch + qb] = rb.y * r10 + rb.x * r11;
            }
            for (int idx = tid; idx < n * half; idx += nthreads) {
                const int i = idx / half, kk = idx - i * half;
                const double2 rk = cs[kk];
                if (rk.y == 0.0) continue;
                const int pqk = pq[kk];
                const int p = pqk & 255, q = pqk >> 8;
                const double vp = V[i * pitch + p], vq = V[i * pitch + q];
                V[i * pitch + p] = rk.x * vp - rk.y * vq;
                V[i * pitch + q] = rk.y * vp + rk.x * vq;
            }
            __syncthreads();
        }
        ++sweeps;
        const int big = *flag;
        __syncthreads();
        // quadratic convergence: every off-diagonal this sweep met was below 1e-8 (relative) before it was
        // rotated, so the sweep leaves them near 1e-16
        if (!big) break;
    }
    if (tid < 64) {
        // (n <= 64: one value per lane; the wide kernel's 65 .. 128 frames: two)
        double lam = tid < n ? G[tid * pitch + tid] : 0.0;
        double lam2 = tid + 64 < n ? G[(tid + 64) * pitch + tid + 64] : 0.0;
        double lmax = fmax(lam, lam2);
        for (int off = 32; off; off >>= 1) lmax = fmax(lmax, __shfl_xor(lmax, off));
        if (tid < n) wgt[tid] = lam > 1e-13 * lmax ? 1.0 / sqrt(lam) : 0.0;
        if (tid + 64 < n) wgt[tid + 64] = lam2 > 1e-13 * lmax ? 1.0 / sqrt(lam2) : 0.0;
    }
    __syncthreads();
    for (int idx = tid; idx < n * n; idx += nthreads) {
        const int i = idx / n, j = idx - i * n;
        double acc = 0.0;
        for (int kk = 0; kk < n; ++kk) acc += V[i * pitch + kk] * wgt[kk] * V[j * pitch + kk];
        Wout[i * pitch + j] = acc;
    }
    __syncthreads();
    *sweeps_out = sweeps;
}

// ---------------------------------------------------------------------------------
// 16x16 output tile (ti, tj) of L*R, L and R stored [NPAD][PITCH] row-major in LDS.
// A operand: lane l holds L[16ti + (l&15)][4kk + (l>>4)]; B operand: R[4kk + (l>>4)][16tj + (l&15)].
// PITCH = NPAD + 2 makes the A reads conflict-free (16 rows land 4 banks apart) and the B reads 2-way.
// Result: lane l, component r <-> element (16ti + (l>>4) + 4r, 16tj + (l&15)).
// ---------------------------------------------------------------------------------
// KS = k-steps that hold live frames (ceil(n / 4) rounded up to even, a template parameter of the kernel): rows and columns from 4 KS
// on are dead directions whose only entries are the unit diagonal, so for a live output element the steps beyond contribute exact
// zeros and are left out (the CLI's queue of 21 frames: 6 of 8).
template <int PITCH, int KS>
__device__ __forceinline__ d4 mm_tile(const double *L, const double *R, int ti, int tj, int lane)
{
    // two accumulator chains (even / odd k-steps): a dependent f64 MFMA waits out the previous one's latency
    d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    const int lo = lane & 15, hi = lane >> 4;
    const double *lp = L + (16 * ti + lo) * PITCH + hi;
    const double *rp = R + hi * PITCH + 16 * tj + lo;
    double a[KS], bb[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) { a[kk] = lp[4 * kk]; bb[kk] = rp[4 * kk * PITCH]; }
#pragma unroll
    for (int kk = 0; kk < KS; kk += 2) {
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], bb[kk], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk + 1], bb[kk + 1], acc1, 0, 0, 0);
    }
    return acc0 + acc1;
}

template <int PITCH>
__device__ __forceinline__ void store_tile(double *M, d4 v, int ti, int tj, int lane)
{
    const int col = 16 * tj + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) M[(16 * ti + (lane >> 4) + 4 * r) * PITCH + col] = v[r];
}

}  // namespace swk
