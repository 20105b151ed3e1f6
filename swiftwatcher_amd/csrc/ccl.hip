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
    (void)hipMemsetAsync(b.parent, 0xFF, (size_t)F * b.Pp * sizeof(int32_t), s);
    (void)hipMemsetAsync(b.rootbits, 0, (size_t)F * b.words * sizeof(uint32_t), s);
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
// properties, separated by workgroup barriers instead of kernel boundaries:
//   init -> union -> flatten + root bitmap -> bitmap prefix -> labels (+ region table) -> segment list.
// The root bitmap, its per-word prefix and the 256-entry region table live in LDS; only the
// union-find parents (touched for foreground pixels only) are in global memory, and stay in L2.
// A frame is scanned 4 pixels per lane (one dword) when the row length allows it; the sparse image
// makes most dwords zero and those cost one compare.
// ---------------------------------------------------------------------------------
constexpr int kFrameThreads = 1024;

struct FrameLds {
    int area[256], r0[256], c0[256], r1[256], c1[256];
    unsigned long long sr[256], sc[256];
    int scan[kFrameThreads];
    int wave_tot[4];
};

template <int VEC, typename Fn>
__device__ __forceinline__ void for_each_fg(const uint8_t *__restrict__ img, int P, Fn fn)
{
    if (VEC == 4) {
        const uint32_t *w32 = (const uint32_t *)img;
        for (int i = threadIdx.x; i < (P >> 2); i += kFrameThreads) {
            const uint32_t v = w32[i];
            if (!v) continue;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if ((v >> (8 * k)) & 0xffu) fn(4 * i + k);
        }
    } else {
        for (int i = threadIdx.x; i < P; i += kFrameThreads)
            if (img[i]) fn(i);
    }
}

// block-wide exclusive prefix of popcounts over `nw` bitmap words; returns the total
__device__ __forceinline__ int bitmap_prefix(const uint32_t *bits, int nw, int *prefix, int *scan)
{
    const int tid = threadIdx.x;
    const int per = (nw + kFrameThreads - 1) / kFrameThreads;
    const int w0 = tid * per, w1 = w0 + per < nw ? w0 + per : nw;
    int cnt = 0;
    for (int w = w0; w < w1; ++w) cnt += __popc(bits[w]);
    // wave-level inclusive scan, then the 16 wave totals
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
    for (int w = w0; w < w1; ++w) { prefix[w] = run; run += __popc(bits[w]); }
    __syncthreads();
    return total;
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

// The frame's foreground is compacted first: a bitmap over the id space + its prefix give every
// foreground pixel a dense rank that is monotone in the id, so the whole union-find (parents, root
// bitmap) fits in LDS and "smallest id" stays "smallest rank".  Frames with more foreground pixels
// than `cap` fall back to parents in global memory (same results, slower).
template <int VEC, bool PROPS>
__global__ __launch_bounds__(kFrameThreads) void k_ccl_frame(const uint8_t *__restrict__ src, int H, int W, int order, int conn8,
                                                             int *__restrict__ parent, int Pp, int words, int cap,
                                                             int32_t *__restrict__ labels32, uint8_t *__restrict__ labels8,
                                                             int32_t *__restrict__ ncomp, int seg_cap,
                                                             swk_segment *__restrict__ segs, int32_t *__restrict__ nseg)
{
    extern __shared__ unsigned char lds_raw[];
    FrameLds *L = (FrameLds *)lds_raw;
    uint32_t *fgbits = (uint32_t *)(lds_raw + sizeof(FrameLds));       // [words]  also the root bitmap of the global path
    int *fgprefix = (int *)(fgbits + words);                            // [words]
    int *cpar = fgprefix + words;                                       // [cap]    compact parents
    uint32_t *rootbits = (uint32_t *)(cpar + cap);                      // [cap/32]
    int *rootprefix = (int *)(rootbits + cap / 32);                     // [cap/32]
    const int f = blockIdx.x, tid = threadIdx.x;
    const int P = H * W, Wb = (W + 1) / 2;
    const uint8_t *img = src + (int64_t)f * P;
    int *par = parent + (int64_t)f * Pp;

    for (int i = tid; i < words; i += kFrameThreads) fgbits[i] = 0u;
    for (int i = tid; i < cap / 32; i += kFrameThreads) rootbits[i] = 0u;
    if (PROPS && tid < 256) {
        L->area[tid] = 0; L->r0[tid] = INT_MAX; L->c0[tid] = INT_MAX; L->r1[tid] = -1; L->c1[tid] = -1;
        L->sr[tid] = 0; L->sc[tid] = 0;
    }
    __syncthreads();
    for_each_fg<VEC>(img, P, [&](int idx) {
        const int r = idx / W, c = idx - r * W;
        const int id = pix_id(r, c, W, Wb, order);
        atomicOr(&fgbits[id >> 5], 1u << (id & 31));
    });
    __syncthreads();
    const int nfg = bitmap_prefix(fgbits, words, fgprefix, L->scan);
    const bool compact = nfg <= cap;
    auto rank_of = [&](int id) -> int { return fgprefix[id >> 5] + __popc(fgbits[id >> 5] & ((1u << (id & 31)) - 1u)); };

    if (compact) {
        for (int i = tid; i < nfg; i += kFrameThreads) cpar[i] = i;
        __syncthreads();
        for_each_fg<VEC>(img, P, [&](int idx) {
            const int r = idx / W, c = idx - r * W;
            const int me = rank_of(pix_id(r, c, W, Wb, order));
            if (c > 0 && img[idx - 1]) lds_union(cpar, me, rank_of(pix_id(r, c - 1, W, Wb, order)));
            if (r > 0) {
                const uint8_t *up = img + idx - W;
                if (up[0]) lds_union(cpar, me, rank_of(pix_id(r - 1, c, W, Wb, order)));
                else if (conn8) {
                    // with the pixel above set, NW and NE are already joined to it through their own W/E links
                    if (c > 0 && up[-1]) lds_union(cpar, me, rank_of(pix_id(r - 1, c - 1, W, Wb, order)));
                    if (c + 1 < W && up[1]) lds_union(cpar, me, rank_of(pix_id(r - 1, c + 1, W, Wb, order)));
                }
            }
        });
        __syncthreads();
        for (int i = tid; i < nfg; i += kFrameThreads) {
            const int root = lds_find(cpar, i);
            if (root == i) atomicOr(&rootbits[i >> 5], 1u << (i & 31));
        }
        __syncthreads();
        // compress after every root is known (a concurrent find may still walk the old chains above)
        for (int i = tid; i < nfg; i += kFrameThreads) cpar[i] = lds_find(cpar, i);
        __syncthreads();
        const int nroots = bitmap_prefix(rootbits, (nfg + 31) >> 5, rootprefix, L->scan);
        if (ncomp && tid == 0) ncomp[f] = nroots;
    } else {
        // ---- global-memory parents (dense frames) ----
        for_each_fg<VEC>(img, P, [&](int idx) {
            const int r = idx / W, c = idx - r * W;
            const int id = pix_id(r, c, W, Wb, order);
            par[id] = id;
        });
        __syncthreads();
        for_each_fg<VEC>(img, P, [&](int idx) {
            const int r = idx / W, c = idx - r * W;
            const int id = pix_id(r, c, W, Wb, order);
            if (c > 0 && img[idx - 1]) uf_union(par, id, pix_id(r, c - 1, W, Wb, order));
            if (r > 0) {
                const uint8_t *up = img + idx - W;
                if (up[0]) uf_union(par, id, pix_id(r - 1, c, W, Wb, order));
                else if (conn8) {
                    if (c > 0 && up[-1]) uf_union(par, id, pix_id(r - 1, c - 1, W, Wb, order));
                    if (c + 1 < W && up[1]) uf_union(par, id, pix_id(r - 1, c + 1, W, Wb, order));
                }
            }
        });
        __syncthreads();
        // the foreground bitmap is no longer needed: turn it into the root bitmap
        for (int i = tid; i < words; i += kFrameThreads) fgbits[i] = 0u;
        __syncthreads();
        for_each_fg<VEC>(img, P, [&](int idx) {
            const int r = idx / W, c = idx - r * W;
            const int id = pix_id(r, c, W, Wb, order);
            const int root = uf_find(par, id);
            if (root == id) atomicOr(&fgbits[id >> 5], 1u << (id & 31));
            else __hip_atomic_store(&par[id], root, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        });
        __syncthreads();
        const int nroots = bitmap_prefix(fgbits, words, fgprefix, L->scan);
        if (ncomp && tid == 0) ncomp[f] = nroots;
    }
    // ---- labels for every pixel, region table for the foreground ----
    auto label_of = [&](int idx) -> int {
        const int r = idx / W, c = idx - r * W;
        const int id = pix_id(r, c, W, Wb, order);
        int label;
        if (compact) {
            const int root = cpar[rank_of(id)];
            label = rootprefix[root >> 5] + __popc(rootbits[root >> 5] & ((1u << (root & 31)) - 1u)) + 1;
        } else {
            int root = __hip_atomic_load(&par[id], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (;;) {
                const int q = __hip_atomic_load(&par[root], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (q == root) break;
                root = q;
            }
            label = fgprefix[root >> 5] + __popc(fgbits[root >> 5] & ((1u << (root & 31)) - 1u)) + 1;
        }
        if (PROPS) {
            const int v = label & 0xff;                        // astype(np.uint8), image_filtering.py:329
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
    if (VEC == 4) {
        const uint32_t *w32 = (const uint32_t *)img;
        for (int i = tid; i < (P >> 2); i += kFrameThreads) {
            const uint32_t v = w32[i];
            uint32_t packed = 0;
            int lab[4] = {0, 0, 0, 0};
            if (v) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if ((v >> (8 * k)) & 0xffu) { lab[k] = label_of(4 * i + k); packed |= (uint32_t)(lab[k] & 0xff) << (8 * k); }
            }
            if (labels8) ((uint32_t *)(labels8 + (int64_t)f * P))[i] = packed;
            if (labels32) {
                int32_t *o = labels32 + (int64_t)f * P + 4 * i;
                o[0] = lab[0]; o[1] = lab[1]; o[2] = lab[2]; o[3] = lab[3];
            }
        }
    } else {
        for (int i = tid; i < P; i += kFrameThreads) {
            const int lab = img[i] ? label_of(i) : 0;
            if (labels8) labels8[(int64_t)f * P + i] = (uint8_t)(lab & 0xff);
            if (labels32) labels32[(int64_t)f * P + i] = lab;
        }
    }
    if (!PROPS) return;
    __syncthreads();
    // ---- ascending-label segment list ----
    if (tid < 256) {
        const bool live = tid > 0 && L->area[tid] > 0;
        const unsigned long long mask = __ballot(live);
        const int lane = tid & 63, wv = tid >> 6;
        if (lane == 0) L->wave_tot[wv] = __popcll(mask);
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

constexpr int kCompactCap = 8192;     // foreground pixels per frame handled with LDS-resident parents
size_t ccl_frame_lds_bytes(int H, int W) { return sizeof(FrameLds) + ccl_words(H, W) * 8 + (size_t)kCompactCap * 4 + (kCompactCap / 32) * 8; }
bool ccl_frame_supported(int H, int W) { return ccl_frame_lds_bytes(H, W) <= 150 * 1024; }

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
                       b.words, kCompactCap, labels32, labels8, b.ncomp, seg_cap, segs, nseg);
}

void launch_ccl_frame(hipStream_t s, const uint8_t *src, int F, int H, int W, int connectivity, int order, const CclBuffers &b,
                      int32_t *labels32, uint8_t *labels8, bool props, int seg_cap, swk_segment *segs, int32_t *nseg)
{
    if (connectivity == 4) order = SWK_ORDER_RASTER;
    const int conn8 = connectivity == 8;
    const bool vec4 = (W % 4 == 0) && (((uintptr_t)src & 3) == 0) && (!labels8 || ((uintptr_t)labels8 & 3) == 0);
    if (vec4) {
        if (props) launch_frame_t<4, true>(s, src, F, H, W, order, conn8, b, labels32, labels8, seg_cap, segs, nseg);
        else launch_frame_t<4, false>(s, src, F, H, W, order, conn8, b, labels32, labels8, seg_cap, segs, nseg);
    } else {
        if (props) launch_frame_t<1, true>(s, src, F, H, W, order, conn8, b, labels32, labels8, seg_cap, segs, nseg);
        else launch_frame_t<1, false>(s, src, F, H, W, order, conn8, b, labels32, labels8, seg_cap, segs, nseg);
    }
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
