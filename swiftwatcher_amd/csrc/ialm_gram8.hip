// Start of the IALM on the integer matrix cores.
//
// The first streaming pass only has to produce the Gram matrix of M_1 = X - E_1 + Y_0/mu_0 (image_filtering.py:272-284).
// With lmbda = 0.01 the first threshold lmbda/mu_0 = 0.008 ||X||_F dwarfs every entry of X + Y_0/mu_0 for all but
// toy windows, so E_1 = 0, M_1 = (1 + 1/(mu_0 dual)) X, and
//     G_1 = (1 + 1/(mu_0 dual))^2 X^T X
// with X^T X an INTEGER matrix: v_mfma_i32_16x16x64_i8 forms it exactly, reading each u8 pixel once, and its
// diagonal is the sum of squares ||X||_F^2 (:269) that the statistics sweep would otherwise need a second read for.
// u8 does not fit the signed operands, so the kernel multiplies x' = x - 128 (one XOR per dword) and corrects:
//     sum x_i x_j = sum x'_i x'_j + 128 (sum x_i + sum x_j) - 16384 * pixels.
// Both MFMA operands are the SAME registers (lane = frame within the block, 16 consecutive pixels per lane), so
// whatever order the instruction gives the 64 k-values inside a lane group, A and B agree on it.
// k_ialm_init checks E_1 = 0 per window (IalmWin::int_gram); a window that fails gets the f64 start pass.
#include "swk_internal.h"

namespace swk {

typedef int v4i __attribute__((ext_vector_type(4)));

template <int NB, bool ALIGNED>
__global__ __launch_bounds__(256) void k_gram_u8(IalmBuffers b)
{
    constexpr int NPAD = 16 * NB, NPAIR = NB * (NB + 1) / 2;
    __shared__ long long sG[NPAD * NPAD];
    __shared__ unsigned long long sS[NPAD];
    __shared__ unsigned int sMax;
    __shared__ long long sCnt;
    const int w = blockIdx.y, n = b.n, P = b.P;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int fi = lane & 15, g = lane >> 4;
    // ALIGNED: one descriptor per window (its range check zeroes what lies past the window).  Otherwise the dword
    // loads are relative to the 4-byte aligned start of the whole batch and carry the window offset themselves.
    const unsigned wbase = ALIGNED ? 0u : (unsigned)w * (unsigned)n * (unsigned)P;
    const __amdgpu_buffer_rsrc_t rX = ALIGNED
        ? __builtin_amdgcn_make_buffer_rsrc((void *)(b.X + (int64_t)w * n * P), 0, n * P, 0x00020000)
        : __builtin_amdgcn_make_buffer_rsrc((void *)b.X, 0, (int)(((int64_t)b.nwin * n * P + 3) & ~3ll), 0x00020000);   // whole dwords:
    // the range check drops a dword that is only partly inside; X lives in the library's own, padded allocation
    v4i acc[NPAIR];
#pragma unroll
    for (int i = 0; i < NPAIR; ++i) acc[i] = v4i{0, 0, 0, 0};
    unsigned int sum[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) sum[i] = 0u;
    unsigned int mx = 0u;
    long long cnt = 0;
    for (int i = tid; i < NPAD * NPAD; i += 256) sG[i] = 0;
    if (tid < NPAD) sS[tid] = 0ull;
    if (tid == 0) { sMax = 0u; sCnt = 0; }
    const int nchunks = (P + 63) >> 6;
    for (int c = blockIdx.x * 4 + wave; c < nchunks; c += gridDim.x * 4) {
        const int p0 = c * 64 + 16 * g;
        v4i a[NB];
#pragma unroll
        for (int fb = 0; fb < NB; ++fb) {
            const int frame = fb * 16 + fi;
            // past the last frame or the last pixel of the window the range check returns zeros; a row's last 16
            // pixels may run into the next row, which the tail mask below removes
            const bool ok = frame < n && p0 < P;
            const unsigned off = wbase + (unsigned)frame * (unsigned)P + (unsigned)p0;
            if (ALIGNED) {
                a[fb] = __builtin_amdgcn_raw_buffer_load_b128(rX, ok ? off : 0x80000000u, 0, 0);
            } else {
                // rows that do not start on a 16-byte boundary (e.g. 214 x 107, 850 x 425): five aligned dwords and
                // a byte funnel shift per lane
                const unsigned base = ok ? (off & ~3u) : 0x80000000u, sh = off & 3u;
                unsigned d[5];
#pragma unroll
                for (int q = 0; q < 5; ++q) d[q] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rX, base, 4 * q, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) a[fb][q] = (int)__builtin_amdgcn_alignbyte(d[q + 1], d[q], sh);
            }
        }
        const int left = P - p0;                       // valid pixels among this lane's 16
        cnt += (P - c * 64 < 64 ? P - c * 64 : 64);
#pragma unroll
        for (int fb = 0; fb < NB; ++fb) {
            const bool fvalid = fb * 16 + fi < n;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                unsigned v = (unsigned)a[fb][d];
                const int lv = left - 4 * d;           // valid bytes of this dword
                unsigned keep = lv >= 4 ? 0xffffffffu : (lv <= 0 ? 0u : ((1u << (8 * lv)) - 1u));
                if (!fvalid) keep = 0u;
                v &= keep;
                sum[fb] = __builtin_amdgcn_udot4(v, 0x01010101u, sum[fb], false);
                const unsigned m01 = (v & 0xffu) > ((v >> 8) & 0xffu) ? (v & 0xffu) : ((v >> 8) & 0xffu);
                const unsigned m23 = ((v >> 16) & 0xffu) > (v >> 24) ? ((v >> 16) & 0xffu) : (v >> 24);
                const unsigned m = m01 > m23 ? m01 : m23;
                mx = m > mx ? m : mx;
                a[fb][d] = (int)((v ^ 0x80808080u) & keep);   // x - 128 as i8; 0 where there is no pixel
            }
        }
        int pair = 0;
#pragma unroll
        for (int ib = 0; ib < NB; ++ib)
#pragma unroll
            for (int jb = ib; jb < NB; ++jb) {
                acc[pair] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[ib], a[jb], acc[pair], 0, 0, 0);
                ++pair;
            }
    }
    __syncthreads();
    // ---- block-level sums (integers: exact, order independent) ----
    {
        int pair = 0;
#pragma unroll
        for (int ib = 0; ib < NB; ++ib)
#pragma unroll
            for (int jb = ib; jb < NB; ++jb) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * ib + 4 * g + r, j = 16 * jb + fi;       // D: lane holds rows 4 (l >> 4) + r, column l & 15
                    atomicAdd((unsigned long long *)&sG[i * NPAD + j], (unsigned long long)(long long)acc[pair][r]);
                }
                ++pair;
            }
#pragma unroll
        for (int fb = 0; fb < NB; ++fb) atomicAdd(&sS[fb * 16 + fi], (unsigned long long)sum[fb]);
        for (int off = 32; off; off >>= 1) { const unsigned o = __shfl_down(mx, off); mx = o > mx ? o : mx; }
        if (lane == 0) { atomicMax(&sMax, mx); atomicAdd((unsigned long long *)&sCnt, (unsigned long long)cnt); }
    }
    __syncthreads();
    const long long pixels = sCnt;
    double *gp = b.gpart + ((int64_t)w * b.nblk + blockIdx.x) * n * n;
    unsigned long long diag = 0ull;
    for (int idx = tid; idx < n * n; idx += 256) {
        const int i = idx / n, j = idx % n;
        if ((i >> 4) > (j >> 4)) continue;
        const long long v = sG[i * NPAD + j] + 128ll * (long long)(sS[i] + sS[j]) - 16384ll * pixels;
        gp[idx] = (double)v;
        if (i == j) diag += (unsigned long long)v;
    }
    for (int off = 32; off; off >>= 1) diag += __shfl_down(diag, off);
    if (lane == 0 && diag) atomicAdd(&b.win[w].sumsq, diag);
    if (tid == 0) atomicMax(&b.win[w].maxv, sMax);
}

bool gram_u8_supported(const IalmBuffers &b)
{
    // i32 accumulators: at most 2^16 pixels per wave; dword loads: the window base must be 4-byte aligned
    if ((((uintptr_t)b.X) & 3) || (int64_t)b.nwin * b.n * b.P >= (1ll << 31)) return false;
    const long long per_wave = ((long long)b.P + 4ll * b.nblk - 1) / (4ll * b.nblk) + 64;
    return per_wave <= 65536;
}

template <int NB>
static void launch_gram_nb(hipStream_t s, const IalmBuffers &b)
{
    const dim3 grid(b.nblk, b.nwin), block(256);
    // 16-byte loads need every frame row of every window on a 16-byte boundary
    const bool aligned = (b.P & 15) == 0 && (((uintptr_t)b.X) & 15) == 0;
    if (aligned) hipLaunchKernelGGL((k_gram_u8<NB, true>), grid, block, 0, s, b);
    else hipLaunchKernelGGL((k_gram_u8<NB, false>), grid, block, 0, s, b);
}

void launch_gram_u8(hipStream_t s, const IalmBuffers &b)
{
    switch ((b.n + 15) / 16) {
    case 1: launch_gram_nb<1>(s, b); break;
    case 2: launch_gram_nb<2>(s, b); break;
    case 3: launch_gram_nb<3>(s, b); break;
    default: launch_gram_nb<4>(s, b); break;
    }
}

}  // namespace swk
