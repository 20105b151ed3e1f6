// Connected-component labelling and region properties on gfx950 -- replaces
// cc_labeling (image_filtering.py:325-329, cv2.connectedComponents) and
// get_segment_properties (image_filtering.py:332-335, skimage regionprops) of the reference.
//
// Label equivalence by union-find over a per-frame id space chosen so that the smallest id
// of a component is also its rank key under OpenCV's numbering:
//   SWK_ORDER_BLOCK2X2 (8-way, BBDT): id = 4*((r>>1)*ceil(W/2) + (c>>1)) + 2*(r&1) + (c&1)
//                                     -> components ordered by their first 2x2 block
//   SWK_ORDER_RASTER   (SAUF)       : id = r*W + c -> ordered by their first pixel
// Roots are the minimum id (atomicMin union), so the final label of a component is
// 1 + (number of roots with a smaller id): a bitmap of roots + per-word prefix counts +
// popcount gives the rank with no sort.  Results are independent of scheduling: bit-exact.
#include "swk_internal.h"
#include <limits.h>

namespace swk {

size_t ccl_padded(int H, int W) { return (size_t)((H + 1) / 2) * ((W + 1) / 2) * 4; }
size_t ccl_words(int H, int W) { return (ccl_padded(H, W) + 31) / 32; }

__device__ __forceinline__ int pix_id(int r, int c, int W, int Wb, int order)
{
    return order == SWK_ORDER_BLOCK2X2 ? ((((r >> 1) * Wb + (c >> 1)) << 2) | ((r & 1) << 1) | (c & 1)) : r * W + c;
}

__device__ __forceinline__ int uf_find(int *parent, int i)
{
    for (;;) {
        const int p = __hip_atomic_load(&parent[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p == i) return i;
        i = p;
    }
}

__device__ __forceinline__ void uf_union(int *parent, int a, int b)
{
    for (;;) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }      // a > b: hang a under b
        const int old = atomicMin(&parent[a], b);
        if (old == a) return;
        a = old;                                            // somebody re-parented a meanwhile: retry from there
    }
}

__global__ __launch_bounds__(256) void k_ccl_init(const uint8_t *__restrict__ src, int H, int W, int order,
                                                  int *__restrict__ parent, int Pp)
{
    const int f = blockIdx.z, r = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= W) return;
    if (src[((int64_t)f * H + r) * W + c]) {
        const int id = pix_id(r, c, W, (W + 1) / 2, order);
        parent[(int64_t)f * Pp + id] = id;
    }
}

__global__ __launch_bounds__(256) void k_ccl_union(const uint8_t *__restrict__ src, int H, int W, int order, int conn8,
                                                   int *__restrict__ parent, int Pp)
{
    const int f = blockIdx.z, r = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= W) return;
    const uint8_t *img = src + (int64_t)f * H * W;
    if (!img[r * W + c]) return;
    int *par = parent + (int64_t)f * Pp;
    const int Wb = (W + 1) / 2;
    const int id = pix_id(r, c, W, Wb, order);
    if (c > 0 && img[r * W + c - 1]) uf_union(par, id, pix_id(r, c - 1, W, Wb, order));
    if (r > 0) {
        const uint8_t *up = img + (r - 1) * W;
        if (up[c]) uf_union(par, id, pix_id(r - 1, c, W, Wb, order));
        else if (conn8) {
            // with the pixel above set, NW and NE are already joined to it through their own W/E links
            if (c > 0 && up[c - 1]) uf_union(par, id, pix_id(r - 1, c - 1, W, Wb, order));
            if (c + 1 < W && up[c + 1]) uf_union(par, id, pix_id(r - 1, c + 1, W, Wb, order));
        }
    }
}

__global__ __launch_bounds__(256) void k_ccl_flatten(const uint8_t *__restrict__ src, int H, int W, int order,
                                                     int *__restrict__ parent, int Pp, uint32_t *__restrict__ rootbits, int words)
{
    const int f = blockIdx.z, r = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= W) return;
    if (!src[((int64_t)f * H + r) * W + c]) return;
    int *par = parent + (int64_t)f * Pp;
    const int id = pix_id(r, c, W, (W + 1) / 2, order);
    const int root = uf_find(par, id);
    if (root == id) atomicOr(&rootbits[(int64_t)f * words + (id >> 5)], 1u << (id & 31));
    else __hip_atomic_store(&par[id], root, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// exclusive prefix of popcounts over the root bitmap words of one frame
__global__ __launch_bounds__(256) void k_ccl_wordprefix(const uint32_t *__restrict__ rootbits, int words,
                                                        int *__restrict__ wordprefix, int *__restrict__ ncomp)
{
    __shared__ int s_sum[256];
    const int f = blockIdx.x, t = threadIdx.x;
    const uint32_t *bits = rootbits + (int64_t)f * words;
    int *pre = wordprefix + (int64_t)f * words;
    const int per = (words + 255) / 256;
    const int w0 = t * per, w1 = w0 + per < words ? w0 + per : words;
    int cnt = 0;
    for (int w = w0; w < w1; ++w) cnt += __popc(bits[w]);
    s_sum[t] = cnt;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {           // Hillis-Steele inclusive scan
        const int v = t >= off ? s_sum[t - off] : 0;
        __syncthreads();
        s_sum[t] += v;
        __syncthreads();
    }
    int run = s_sum[t] - cnt;
    for (int w = w0; w < w1; ++w) { pre[w] = run; run += __popc(bits[w]); }
    if (t == 255) ncomp[f] = s_sum[255];
}

__global__ __launch_bounds__(256) void k_ccl_label(const uint8_t *__restrict__ src, int H, int W, int order,
                                                   const int *__restrict__ parent, int Pp,
                                                   const uint32_t *__restrict__ rootbits, const int *__restrict__ wordprefix,
                                                   int words, int32_t *__restrict__ labels32, uint8_t *__restrict__ labels8)
{
    const int f = blockIdx.z, r = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= W) return;
    const int64_t o = ((int64_t)f * H + r) * W + c;
    int label = 0;
    if (src[o]) {
        const int id = pix_id(r, c, W, (W + 1) / 2, order);
        int root = parent[(int64_t)f * Pp + id];
        // one more hop covers a node flattened before its parent was (parent of a non-root is a root
        // after k_ccl_flatten only if the chain was already compressed; walk to be safe)
        for (;;) { const int p = parent[(int64_t)f * Pp + root]; if (p == root) break; root = p; }
        const uint32_t wbits = rootbits[(int64_t)f * words + (root >> 5)];
        label = wordprefix[(int64_t)f * words + (root >> 5)] + __popc(wbits & ((1u << (root & 31)) - 1u)) + 1;
    }
    if (labels32) labels32[o] = label;
    if (labels8) labels8[o] = (uint8_t)(label & 0xff);          // labeled_frame.astype(np.uint8), :329
}

void launch_ccl(hipStream_t s, const uint8_t *src, int F, int H, int W, int connectivity, int order,
                const CclBuffers &b, int32_t *labels32, uint8_t *labels8)
{
    if (connectivity == 4) order = SWK_ORDER_RASTER;       // OpenCV's 4-way algorithm numbers in raster order
    const int conn8 = connectivity == 8;
    hipError_t me = hipMemsetAsync(b.parent, 0xFF, (size_t)F * b.Pp * sizeof(int32_t), s);
    if (me == hipSuccess) me = hipMemsetAsync(b.rootbits, 0, (size_t)F * b.words * sizeof(uint32_t), s);
    if (me != hipSuccess) { g_launch_error = (int)me; return; }          // surfaces at the context's next sync()
    for (int f0 = 0; f0 < F; f0 += 32768) {
        const int fc = F - f0 < 32768 ? F - f0 : 32768;
        const dim3 grid((W + 255) / 256, H, fc), blk(256);
        const uint8_t *sp = src + (int64_t)f0 * H * W;
        int *par = b.parent + (int64_t)f0 * b.Pp;
        uint32_t *rb = b.rootbits + (int64_t)f0 * b.words;
        int *wp = b.wordprefix + (int64_t)f0 * b.words;
        hipLaunchKernelGGL(k_ccl_init, grid, blk, 0, s, sp, H, W, order, par, b.Pp);
        hipLaunchKernelGGL(k_ccl_union, grid, blk, 0, s, sp, H, W, order, conn8, par, b.Pp);
        hipLaunchKernelGGL(k_ccl_flatten, grid, blk, 0, s, sp, H, W, order, par, b.Pp, rb, b.words);
        hipLaunchKernelGGL(k_ccl_wordprefix, dim3(fc), blk, 0, s, rb, b.words, wp, b.ncomp + f0);
        hipLaunchKernelGGL(k_ccl_label, grid, blk, 0, s, sp, H, W, order, par, b.Pp, rb, wp, b.words,
                           labels32 ? labels32 + (int64_t)f0 * H * W : nullptr,
                           labels8 ? labels8 + (int64_t)f0 * H * W : nullptr);
    }
}

// ---------------------------------------------------------------------------------
// Hot-path kernel: ONE workgroup per frame runs every phase of the labelling and the region
// properties, separated by workgroup barriers instead of kernel boundaries.
//
// The unit of the union-find is the horizontal RUN (maximal row segment of foreground), not the
// pixel: a sparse frame has a few hundred runs.  Two bitmaps in LDS describe the frame:
//   fgr    one bit per pixel in raster order -> run start / run end / "next foreground pixel in the
//          row above" are bit scans (clz / ffs) over one or two words;
//   rsbits one bit per RUN START in the id order of the numbering rule (2x2-block or raster), so
//          rank(run) = prefix popcount is monotone in the id, and the smallest id of a component is the
//          smallest run rank in it -- the union-find root.  Final label = 1 + rank of the root among roots.
// A run is joined to every run of the previous row it touches (8-way: columns cs-1..ce+1); region
// properties are accumulated per run (area += length, column sum = arithmetic series).  Frames with more
// than `cap` runs (noise) fall back to per-pixel union-find with parents in global memory.
// ---------------------------------------------------------------------------------
constexpr int kFrameThreads = 1024;

struct FrameLds {
    int area[256], r0[256], c0[256], r1[256], c1[256];
    unsigned long long sr[256], sc[256];
    int scan[kFrameThreads / 64];
    int wave_tot[4];
};

template <int VEC> __device__ __forceinline__ uint32_t load_word(const uint8_t *img, int i)
{
    if (VEC == 4) return ((const uint32_t *)img)[i];
    if (VEC == 2) return ((const uint16_t *)img)[i];
    return img[i];
}

// fn(word index, VEC-byte word) over the frame, thread-strided; four loads in flight per trip
template <int VEC, typename Fn>
__device__ __forceinline__ void for_each_word(const uint8_t *__restrict__ img, int nwords, Fn fn)
{
    int i = threadIdx.x;
    for (; i + 3 * kFrameThreads < nwords; i += 4 * kFrameThreads) {
        const uint32_t a = load_word<VEC>(img, i), b = load_word<VEC>(img, i + kFrameThreads);
        const uint32_t c = load_word<VEC>(img, i + 2 * kFrameThreads), d = load_word<VEC>(img, i + 3 * kFrameThreads);
        fn(i, a); fn(i + kFrameThreads, b); fn(i + 2 * kFrameThreads, c); fn(i + 3 * kFrameThreads, d);
    }
    for (; i < nwords; i += kFrameThreads) fn(i, load_word<VEC>(img, i));
}

template <int VEC, typename Fn>
__device__ __forceinline__ void for_each_fg(const uint8_t *__restrict__ img, int nwords, Fn fn)
{
    for_each_word<VEC>(img, nwords, [&](int i, uint32_t v) {
        if (!v) return;
#pragma unroll
        for (int k = 0; k < VEC; ++k)
            if ((v >> (8 * k)) & 0xffu) fn(VEC * i + k);
    });
}

// Block-wide exclusive prefix of popcounts over `nw` bitmap words, kept per group of kPfx words (the rank
// lookup adds the popcounts of up to kPfx-1 words itself: 4x less LDS).  Returns the total.
constexpr int kPfx = 4;
__device__ __forceinline__ int bitmap_prefix(const uint32_t *bits, int nw, int *prefix, int *scan)
{
    const int tid = threadIdx.x;
    const int ng = (nw + kPfx - 1) / kPfx;                       // groups of kPfx words
    const int per = (ng + kFrameThreads - 1) / kFrameThreads;    // groups per thread
    const int g0 = tid * per, g1 = g0 + per < ng ? g0 + per : ng;
    int cnt = 0;
    for (int w = g0 * kPfx; w < g1 * kPfx && w < nw; ++w) cnt += __popc(bits[w]);
    int inc = cnt;
    const int lane = tid & 63, wv = tid >> 6;
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(inc, off);
        if (lane >= off) inc += v;
    }
    if (lane == 63) scan[wv] = inc;
    __syncthreads();
    int base = 0, total = 0;
    for (int i = 0; i < kFrameThreads / 64; ++i) { const int t = scan[i]; if (i < wv) base += t; total += t; }
    int run = base + inc - cnt;
    for (int g = g0; g < g1; ++g) {
        prefix[g] = run;
        for (int w = g * kPfx; w < (g + 1) * kPfx && w < nw; ++w) run += __popc(bits[w]);
    }
    __syncthreads();
    return total;
}

// number of set bits below position `pos` of the bitmap, from the grouped prefix
__device__ __forceinline__ int bitmap_rank(const uint32_t *bits, const int *prefix, int pos)
{
    const int w = pos >> 5, g = w / kPfx;
    int r = prefix[g];
    for (int k = g * kPfx; k < w; ++k) r += __popc(bits[k]);
    return r + __popc(bits[w] & ((1u << (pos & 31)) - 1u));
}

__device__ __forceinline__ int lds_find(const int *par, int i)
{
    for (;;) { const int p = par[i]; if (p == i) return i; i = p; }
}

__device__ __forceinline__ void lds_union(int *par, int a, int b)
{
    for (;;) {
        a = lds_find(par, a);
        b = lds_find(par, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(&par[a], b);
        if (old == a) return;
        a = old;
    }
}

// ---- bit scans over the raster bitmap (bit b = pixel b of the frame) ----
__device__ __forceinline__ uint32_t range_mask(int lo, int hi)          // bits lo..hi of a word, 0 <= lo <= hi <= 31
{
    return (hi == 31 ? 0xffffffffu : ((1u << (hi + 1)) - 1u)) & ~((1u << lo) - 1u);
}
// first pixel of the run containing foreground pixel b; rb = first pixel of its row
__device__ __forceinline__ int run_first(const uint32_t *fgr, int b, int rb)
{
    int pos = b;
    for (;;) {
        const int wi = pos >> 5, w0 = wi << 5;
        const int lo = rb > w0 ? rb - w0 : 0;
        const uint32_t inv = ~fgr[wi] & range_mask(lo, pos - w0);
        if (inv) return w0 + (31 - __clz(inv)) + 1;
        if (w0 <= rb) return rb;
        pos = w0 - 1;
    }
}
// last pixel of the run containing foreground pixel b; re = last pixel of its row
__device__ __forceinline__ int run_last(const uint32_t *fgr, int b, int re)
{
    int pos = b;
    for (;;) {
        const int wi = pos >> 5, w0 = wi << 5;
        const int hi = re < w0 + 31 ? re - w0 : 31;
        const uint32_t inv = ~fgr[wi] & range_mask(pos - w0, hi);
        if (inv) return w0 + __ffs(inv) - 2;
        if (w0 + 31 >= re) return re;
        pos = w0 + 32;
    }
}
// first foreground pixel in [b, be], or -1
__device__ __forceinline__ int next_fg(const uint32_t *fgr, int b, int be)
{
    int pos = b;
    while (pos <= be) {
        const int wi = pos >> 5, w0 = wi << 5;
        const int hi = be < w0 + 31 ? be - w0 : 31;
        const uint32_t m = fgr[wi] & range_mask(pos - w0, hi);
        if (m) return w0 + __ffs(m) - 1;
        pos = w0 + 32;
    }
    return -1;
}

template <int VEC, bool PROPS>
__global__ __launch_bounds__(kFrameThreads) void k_ccl_frame(const uint8_t *__restrict__ src, int H, int W, int order, int conn8,
                                                             int *__restrict__ parent, int Pp, int words, int cap,
                                                             int32_t *__restrict__ labels32, uint8_t *__restrict__ labels8,
                                                             int32_t *__restrict__ ncomp, int seg_cap,
                                                             swk_segment *__restrict__ segs, int32_t *__restrict__ nseg)
{
    extern __shared__ unsigned char lds_raw[];
    FrameLds *L = (FrameLds *)lds_raw;
    const int P = H * W, Wb = (W + 1) / 2, rwords = (P + 31) >> 5, nwords = P / VEC;
    uint32_t *fgr = (uint32_t *)(lds_raw + sizeof(FrameLds));          // [rwords] raster foreground bitmap
    uint32_t *rsbits = fgr + rwords;                                    // [words]  run starts, id order (roots in the fallback)
    int *rsprefix = (int *)(rsbits + words);                            // [words / kPfx]
    int *run_rc = rsprefix + (words + kPfx - 1) / kPfx;                 // [cap]    row << 16 | first column
    int *run_ce = run_rc + cap;                                         // [cap]    last column
    int *cpar = run_ce + cap;                                           // [cap]    union-find parents over run ranks
    int *rlab = cpar + cap;                                             // [cap]    final label of each run
    uint32_t *rootbits = (uint32_t *)(rlab + cap);                      // [cap/32]
    int *rootprefix = (int *)(rootbits + cap / 32);                     // [cap/32]
    const int f = blockIdx.x, tid = threadIdx.x;
    const uint8_t *img = src + (int64_t)f * P;
    int *par = parent + (int64_t)f * Pp;

    for (int i = tid; i < rwords; i += kFrameThreads) fgr[i] = 0u;
    for (int i = tid; i < words; i += kFrameThreads) rsbits[i] = 0u;
    for (int i = tid; i < cap / 32; i += kFrameThreads) rootbits[i] = 0u;
    if (PROPS && tid < 256) {
        L->area[tid] = 0; L->r0[tid] = INT_MAX; L->c0[tid] = INT_MAX; L->r1[tid] = -1; L->c1[tid] = -1;
        L->sr[tid] = 0; L->sc[tid] = 0;
    }
    __syncthreads();
    // ---- raster foreground bitmap: one LDS atomic per nonzero word ----
    for_each_word<VEC>(img, nwords, [&](int i, uint32_t v) {
        if (!v) return;
        uint32_t nib = 0;
#pragma unroll
        for (int k = 0; k < VEC; ++k) nib |= ((v >> (8 * k)) & 0xffu) ? (1u << k) : 0u;
        const int b = VEC * i;
        atomicOr(&fgr[b >> 5], nib << (b & 31));
    });
    __syncthreads();
    // ---- run starts, in the id order of the numbering rule ----
    // a foreground bit starts a run when the bit before it is clear or it sits in column 0: found a word at a time
    auto for_each_run_start = [&](auto fn) {
        for (int wi = tid; wi < rwords; wi += kFrameThreads) {
            const uint32_t m = fgr[wi];
            if (!m) continue;
            const int w0 = wi << 5;
            uint32_t col0 = 0;                                  // bits of this word that are the first pixel of a row
            for (int b0 = ((w0 + W - 1) / W) * W; b0 < w0 + 32; b0 += W) col0 |= 1u << (b0 - w0);
            const uint32_t prev = (m << 1) | (wi > 0 ? fgr[wi - 1] >> 31 : 0u);
            uint32_t st = m & (~prev | col0);
            while (st) {
                const int bit = __ffs(st) - 1;
                st &= st - 1;
                const int b = w0 + bit;
                const int r = b / W;
                fn(b, r, b - r * W);
            }
        }
    };
    for_each_run_start([&](int b, int r, int c) {
        const int id = pix_id(r, c, W, Wb, order);
        atomicOr(&rsbits[id >> 5], 1u << (id & 31));
    });
    __syncthreads();
    const int nruns = bitmap_prefix(rsbits, words, rsprefix, L->scan);
    const bool by_runs = nruns <= cap;
    auto rank_of = [&](int id) -> int { return bitmap_rank(rsbits, rsprefix, id); };

    if (by_runs) {
        for_each_run_start([&](int b, int r, int c) {
            const int run = rank_of(pix_id(r, c, W, Wb, order));
            run_rc[run] = (r << 16) | c;
            run_ce[run] = run_last(fgr, b, r * W + W - 1) - r * W;
            cpar[run] = run;
        });
        __syncthreads();
        // ---- join every run with the runs it touches in the row above ----
        for (int run = tid; run < nruns; run += kFrameThreads) {
            const int r = run_rc[run] >> 16, cs = run_rc[run] & 0xffff, ce = run_ce[run];
            if (r == 0) continue;
            const int rb = (r - 1) * W, re = rb + W - 1;
            int b = rb + (cs - conn8 > 0 ? cs - conn8 : 0);
            const int be = rb + (ce + conn8 < W - 1 ? ce + conn8 : W - 1);
            while (b <= be) {
                const int hit = next_fg(fgr, b, be);
                if (hit < 0) break;
                const int us = run_first(fgr, hit, rb);
                lds_union(cpar, run, rank_of(pix_id(r - 1, us - rb, W, Wb, order)));
                b = run_last(fgr, hit, re) + 2;           // the pixel right after a run is background
            }
        }
        __syncthreads();
        for (int i = tid; i < nruns; i += kFrameThreads)
            if (lds_find(cpar, i) == i) atomicOr(&rootbits[i >> 5], 1u << (i & 31));
        __syncthreads();
        for (int i = tid; i < nruns; i += kFrameThreads) cpar[i] = lds_find(cpar, i);
        __syncthreads();
        const int nroots = bitmap_prefix(rootbits, (nruns + 31) >> 5, rootprefix, L->scan);
        if (ncomp && tid == 0) ncomp[f] = nroots;
        // ---- label of every run; region properties per run ----
        for (int run = tid; run < nruns; run += kFrameThreads) {
            const int root = cpar[run];
            const int label = bitmap_rank(rootbits, rootprefix, root) + 1;
            rlab[run] = label;
            if (PROPS) {
                const int v = label & 0xff;                        // astype(np.uint8), image_filtering.py:329
                if (v) {
                    const int r = run_rc[run] >> 16, cs = run_rc[run] & 0xffff, ce = run_ce[run];
                    const int len = ce - cs + 1;
                    atomicAdd(&L->area[v], len);
                    atomicMin(&L->r0[v], r); atomicMin(&L->c0[v], cs);
                    atomicMax(&L->r1[v], r); atomicMax(&L->c1[v], ce);
                    atomicAdd(&L->sr[v], (unsigned long long)r * (unsigned long long)len);
                    atomicAdd(&L->sc[v], ((unsigned long long)(cs + ce) * (unsigned long long)len) >> 1);
                }
            }
        }
        __syncthreads();
    } else {
        // ---- dense frame: per-pixel union-find, parents in global memory ----
        for_each_fg<VEC>(img, nwords, [&](int idx) {
            const int r = idx / W, c = idx - r * W;
            const int id = pix_id(r, c, W, Wb, order);
            par[id] = id;
        });
        __syncthreads();
        for_each_fg<VEC>(img, nwords, [&](int idx) {
            const int r = idx / W, c = idx - r * W;
            const int id = pix_id(r, c, W, Wb, order);
            if (c > 0 && img[idx - 1]) uf_union(par, id, pix_id(r, c - 1, W, Wb, order));
            if (r > 0) {
                const uint8_t *up = img + idx - W;
                if (up[0]) uf_union(par, id, pix_id(r - 1, c, W, Wb, order));
                else if (conn8) {
                    // with the pixel above set, NW and NE are already joined to it through their own W/E links
                    if (c > 0 && up[-1]) uf_union(par, id, pix_id(r - 1, c - 1, W, Wb, order));
                    if (c + 1 < W && up[1]) uf_union(par, id, pix_id(r - 1, c + 1, W, Wb, order));
                }
            }
        });
        __syncthreads();
        for (int i = tid; i < words; i += kFrameThreads) rsbits[i] = 0u;       // becomes the root bitmap
        __syncthreads();
        for_each_fg<VEC>(img, nwords, [&](int idx) {
            const int r = idx / W, c = idx - r * W;
            const int id = pix_id(r, c, W, Wb, order);
            const int root = uf_find(par, id);
            if (root == id) atomicOr(&rsbits[id >> 5], 1u << (id & 31));
            else __hip_atomic_store(&par[id], root, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        });
        __syncthreads();
        const int nroots = bitmap_prefix(rsbits, words, rsprefix, L->scan);
        if (ncomp && tid == 0) ncomp[f] = nroots;
    }
    // ---- label plane ----
    auto label_of = [&](int idx) -> int {
        const int r = idx / W, c = idx - r * W;                   // dense-frame path only (runs paint their own labels)
        const int id = pix_id(r, c, W, Wb, order);
        int root = __hip_atomic_load(&par[id], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (;;) {
            const int q = __hip_atomic_load(&par[root], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (q == root) break;
            root = q;
        }
        const int label = bitmap_rank(rsbits, rsprefix, root) + 1;
        if (PROPS) {
            const int v = label & 0xff;
            if (v) {
                atomicAdd(&L->area[v], 1);
                atomicMin(&L->r0[v], r); atomicMin(&L->c0[v], c);
                atomicMax(&L->r1[v], r); atomicMax(&L->c1[v], c);
                atomicAdd(&L->sr[v], (unsigned long long)r);
                atomicAdd(&L->sc[v], (unsigned long long)c);
            }
        }
        return label;
    };
    if (by_runs) {
        // sparse frame: clear the plane with full-width stores, then every run paints its own span
        if (labels8) {
            uint8_t *o = labels8 + (int64_t)f * P;
            if ((P & 15) == 0 && (((uintptr_t)o) & 15) == 0) {
                for (int i = tid; i < P / 16; i += kFrameThreads) ((uint4 *)o)[i] = make_uint4(0u, 0u, 0u, 0u);
            } else {
                for (int i = tid; i < P; i += kFrameThreads) o[i] = 0;
            }
        }
        if (labels32) {
            int32_t *o = labels32 + (int64_t)f * P;
            for (int i = tid; i < P; i += kFrameThreads) o[i] = 0;
        }
        __syncthreads();
        for (int run = tid; run < nruns; run += kFrameThreads) {
            const int r = run_rc[run] >> 16, cs = run_rc[run] & 0xffff, ce = run_ce[run];
            const int label = rlab[run];
            if (labels8) {
                uint8_t *o = labels8 + (int64_t)f * P + r * W;
                for (int c = cs; c <= ce; ++c) o[c] = (uint8_t)label;
            }
            if (labels32) {
                int32_t *o = labels32 + (int64_t)f * P + r * W;
                for (int c = cs; c <= ce; ++c) o[c] = label;
            }
        }
    } else {
    for_each_word<VEC>(img, nwords, [&](int i, uint32_t v) {
        uint32_t packed = 0;
        int lab[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            lab[k] = ((v >> (8 * k)) & 0xffu) ? label_of(VEC * i + k) : 0;
            packed |= (uint32_t)(lab[k] & 0xff) << (8 * k);
        }
        if (labels8) {
            uint8_t *o = labels8 + (int64_t)f * P;
            if (VEC == 4) ((uint32_t *)o)[i] = packed;
            else if (VEC == 2) ((uint16_t *)o)[i] = (uint16_t)packed;
            else o[i] = (uint8_t)packed;
        }
        if (labels32) {
            int32_t *o = labels32 + (int64_t)f * P + VEC * i;
#pragma unroll
            for (int k = 0; k < VEC; ++k) o[k] = lab[k];
        }
    });
    }
    if (!PROPS) return;
    __syncthreads();
    // ---- ascending-label segment list ----
    if (tid < 256) {
        const bool live = tid > 0 && L->area[tid] > 0;
        const unsigned long long mask = __ballot(live);
        if ((tid & 63) == 0) L->wave_tot[tid >> 6] = __popcll(mask);
    }
    __syncthreads();
    if (tid < 256) {
        const bool live = tid > 0 && L->area[tid] > 0;
        const unsigned long long mask = __ballot(live);
        const int lane = tid & 63, wv = tid >> 6;
        int base = 0;
        for (int i = 0; i < wv; ++i) base += L->wave_tot[i];
        const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
        if (segs && live && pos < seg_cap) {
            swk_segment sg;
            sg.label = tid; sg.r0 = L->r0[tid]; sg.c0 = L->c0[tid]; sg.r1 = L->r1[tid] + 1; sg.c1 = L->c1[tid] + 1;
            sg.reserved_ = 0; sg.area = L->area[tid];
            sg.sum_r = (int64_t)L->sr[tid]; sg.sum_c = (int64_t)L->sc[tid];
            segs[(int64_t)f * seg_cap + pos] = sg;
        }
        if (nseg && tid == 0) nseg[f] = L->wave_tot[0] + L->wave_tot[1] + L->wave_tot[2] + L->wave_tot[3];
    }
}

constexpr int kRunCap = 1024;          // runs per frame handled in LDS (a sparse frame has a few hundred)
size_t ccl_frame_lds_bytes(int H, int W)
{
    const size_t P = (size_t)H * W;
    const size_t words = ccl_words(H, W);
    return sizeof(FrameLds) + ((P + 31) / 32) * 4 + words * 4 + ((words + kPfx - 1) / kPfx) * 4 + (size_t)kRunCap * 16 + (kRunCap / 32) * 8;
}
bool ccl_frame_supported(int H, int W) { return ccl_frame_lds_bytes(H, W) <= 150 * 1024 && H < 32768 && W < 65536; }

template <int VEC, bool PROPS>
static void launch_frame_t(hipStream_t s, const uint8_t *src, int F, int H, int W, int order, int conn8, const CclBuffers &b,
                           int32_t *labels32, uint8_t *labels8, int seg_cap, swk_segment *segs, int32_t *nseg)
{
    const size_t lds = ccl_frame_lds_bytes(H, W);
    static size_t attr_bytes = 0;
    if (lds > attr_bytes) {
        (void)hipFuncSetAttribute((const void *)k_ccl_frame<VEC, PROPS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_bytes = lds;
    }
    hipLaunchKernelGGL((k_ccl_frame<VEC, PROPS>), dim3(F), dim3(kFrameThreads), lds, s, src, H, W, order, conn8, b.parent, b.Pp,
                       b.words, kRunCap, labels32, labels8, b.ncomp, seg_cap, segs, nseg);
}

template <int VEC>
static void launch_frame_p(hipStream_t s, const uint8_t *src, int F, int H, int W, int order, int conn8, const CclBuffers &b,
                           int32_t *labels32, uint8_t *labels8, bool props, int seg_cap, swk_segment *segs, int32_t *nseg)
{
    if (props) launch_frame_t<VEC, true>(s, src, F, H, W, order, conn8, b, labels32, labels8, seg_cap, segs, nseg);
    else launch_frame_t<VEC, false>(s, src, F, H, W, order, conn8, b, labels32, labels8, seg_cap, segs, nseg);
}

void launch_ccl_frame(hipStream_t s, const uint8_t *src, int F, int H, int W, int connectivity, int order, const CclBuffers &b,
                      int32_t *labels32, uint8_t *labels8, bool props, int seg_cap, swk_segment *segs, int32_t *nseg)
{
    if (connectivity == 4) order = SWK_ORDER_RASTER;
    const int conn8 = connectivity == 8;
    const int P = H * W;
    const uintptr_t align = (uintptr_t)src | (labels8 ? (uintptr_t)labels8 : 0);
    if (P % 4 == 0 && align % 4 == 0) launch_frame_p<4>(s, src, F, H, W, order, conn8, b, labels32, labels8, props, seg_cap, segs, nseg);
    else if (P % 2 == 0 && align % 2 == 0) launch_frame_p<2>(s, src, F, H, W, order, conn8, b, labels32, labels8, props, seg_cap, segs, nseg);
    else launch_frame_p<1>(s, src, F, H, W, order, conn8, b, labels32, labels8, props, seg_cap, segs, nseg);
}

// ---------------------------------------------------------------------------------
// region properties of u8 label planes: per label value 1..255 area, bbox, sum of rows/cols.
// Workgroup = 16 rows of one frame; LDS table, then a handful of global atomics.
// ---------------------------------------------------------------------------------
constexpr int kPropRows = 16;

__global__ void k_props_init(int32_t *__restrict__ table, unsigned long long *__restrict__ sums, int64_t entries)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= entries) return;
    int32_t *t = table + i * 8;
    t[0] = 0; t[1] = INT_MAX; t[2] = INT_MAX; t[3] = -1; t[4] = -1; t[5] = 0; t[6] = 0; t[7] = 0;
    sums[i * 2] = 0; sums[i * 2 + 1] = 0;
}

__global__ __launch_bounds__(256) void k_props(const uint8_t *__restrict__ lab, int H, int W,
                                               int32_t *__restrict__ table, unsigned long long *__restrict__ sums)
{
    __shared__ int s_area[256], s_r0[256], s_c0[256], s_r1[256], s_c1[256];
    __shared__ unsigned long long s_sr[256], s_sc[256];
    const int f = blockIdx.y, t = threadIdx.x;
    s_area[t] = 0; s_r0[t] = INT_MAX; s_c0[t] = INT_MAX; s_r1[t] = -1; s_c1[t] = -1; s_sr[t] = 0; s_sc[t] = 0;
    __syncthreads();
    const int rbeg = blockIdx.x * kPropRows;
    const int rend = rbeg + kPropRows < H ? rbeg + kPropRows : H;
    const uint8_t *img = lab + (int64_t)f * H * W;
    for (int i = rbeg * W + t; i < rend * W; i += 256) {
        const int v = img[i];
        if (!v) continue;
        const int r = i / W, c = i % W;
        atomicAdd(&s_area[v], 1);
        atomicMin(&s_r0[v], r); atomicMin(&s_c0[v], c);
        atomicMax(&s_r1[v], r); atomicMax(&s_c1[v], c);
        atomicAdd(&s_sr[v], (unsigned long long)r);
        atomicAdd(&s_sc[v], (unsigned long long)c);
    }
    __syncthreads();
    if (t > 0 && s_area[t]) {
        int32_t *g = table + ((int64_t)f * 256 + t) * 8;
        atomicAdd(&g[0], s_area[t]);
        atomicMin(&g[1], s_r0[t]); atomicMin(&g[2], s_c0[t]);
        atomicMax(&g[3], s_r1[t]); atomicMax(&g[4], s_c1[t]);
        atomicAdd(&sums[((int64_t)f * 256 + t) * 2], s_sr[t]);
        atomicAdd(&sums[((int64_t)f * 256 + t) * 2 + 1], s_sc[t]);
    }
}

// compact the 255 table rows of a frame into its ascending-label segment list
__global__ __launch_bounds__(256) void k_props_compact(const int32_t *__restrict__ table, const unsigned long long *__restrict__ sums,
                                                       int seg_cap, swk_segment *__restrict__ segs, int32_t *__restrict__ nseg)
{
    __shared__ int s_wave[4];
    const int f = blockIdx.x, t = threadIdx.x;
    const int32_t *g = table + ((int64_t)f * 256 + t) * 8;
    const bool live = t > 0 && g[0] > 0;
    const unsigned long long mask = __ballot(live);
    const int lane = t & 63, wv = t >> 6;
    if (lane == 0) s_wave[wv] = __popcll(mask);
    __syncthreads();
    int base = 0;
    for (int i = 0; i < wv; ++i) base += s_wave[i];
    const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
    if (live && pos < seg_cap) {
        swk_segment sg;
        sg.label = t; sg.r0 = g[1]; sg.c0 = g[2]; sg.r1 = g[3] + 1; sg.c1 = g[4] + 1; sg.reserved_ = 0;
        sg.area = g[0];
        sg.sum_r = (int64_t)sums[((int64_t)f * 256 + t) * 2];
        sg.sum_c = (int64_t)sums[((int64_t)f * 256 + t) * 2 + 1];
        segs[(int64_t)f * seg_cap + pos] = sg;
    }
    if (t == 0) nseg[f] = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
}

void launch_regionprops(hipStream_t s, const uint8_t *labels8, int F, int H, int W, const CclBuffers &b,
                        int seg_cap, swk_segment *segs, int32_t *nseg)
{
    const int64_t entries = (int64_t)F * 256;
    hipLaunchKernelGGL(k_props_init, dim3((unsigned)((entries + 255) / 256)), dim3(256), 0, s, b.table, b.sums, entries);
    for (int f0 = 0; f0 < F; f0 += 32768) {
        const int fc = F - f0 < 32768 ? F - f0 : 32768;
        hipLaunchKernelGGL(k_props, dim3((H + kPropRows - 1) / kPropRows, fc), dim3(256), 0, s,
                           labels8 + (int64_t)f0 * H * W, H, W, b.table + (int64_t)f0 * 256 * 8, b.sums + (int64_t)f0 * 256 * 2);
    }
    hipLaunchKernelGGL(k_props_compact, dim3(F), dim3(256), 0, s, b.table, b.sums, seg_cap, segs, nseg);
}

}  // namespace swk
