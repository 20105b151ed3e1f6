/*
 * swk_oracle.c -- CPU restatement of swiftwatcher's per-frame segmentation
 * stages.  TEST INFRASTRUCTURE ONLY: this file is the checker for the HIP path
 * (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).  Nothing in
 * swiftwatcher_amd/ may link, import or call it.
 *
 * Every function cites the reference line it restates (paths are relative to
 * the reference checkout, swiftwatcher/...).  Stages whose arithmetic lives in
 * opencv-python==4.1.0.25 (requirements.txt:8; not vendored, not installed
 * anywhere in the build image) are restated from OpenCV 4.1.0's published
 * imgproc algorithms and are marked PARITY UNPINNED: no golden vector from the
 * reference or from cv2 exists for them.  Stages that run on scipy / skimage /
 * numpy are pinned by tests/golden/ fixtures generated from the reference
 * itself (oracle/make_goldens.py).
 *
 * Plain C99, no dependencies.  Built by oracle/build.py into
 * oracle/_build/libswk_oracle.so.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* convert_grayscale: image_filtering.py:188-196 -> cv2.cvtColor(BGR2GRAY).   */
/* PARITY UNPINNED.  OpenCV 4.1.0 color_rgb.cpp, RGB2Gray<uchar>: fixed-point  */
/* weights B=1868 G=9617 R=4899, descale by 14 bits with rounding.             */
/* mode 1 = the 15-bit weights newer OpenCV builds use (3735/19235/9798).      */
/* ------------------------------------------------------------------------- */
ORC_API void orc_bgr2gray(const uint8_t *bgr, int H, int W, int64_t row_stride,
                          int mode, uint8_t *gray)
{
    for (int r = 0; r < H; ++r) {
        const uint8_t *src = bgr + (int64_t)r * row_stride;
        uint8_t *dst = gray + (int64_t)r * W;
        for (int c = 0; c < W; ++c) {
            int b = src[3 * c], g = src[3 * c + 1], rr = src[3 * c + 2];
            int y;
            if (mode == 0)
                y = (b * 1868 + g * 9617 + rr * 4899 + (1 << 13)) >> 14;
            else
                y = (b * 3735 + g * 19235 + rr * 9798 + (1 << 14)) >> 15;
            dst[c] = (uint8_t)y;
        }
    }
}

/* ------------------------------------------------------------------------- */
/* rpca() tail: image_filtering.py:244-245.  S = clip(-E, 0, 255).astype(u8)   */
/* (astype truncates toward zero).                                             */
/* ------------------------------------------------------------------------- */
ORC_API void orc_rpca_epilogue(const double *E, int64_t count, uint8_t *S)
{
    for (int64_t i = 0; i < count; ++i) {
        double v = -E[i];
        if (v < 0.0) v = 0.0;
        if (v > 255.0) v = 255.0;
        S[i] = (uint8_t)v;
    }
}

/* ------------------------------------------------------------------------- */
/* bilateral_blur: image_filtering.py:304-307 -> cv2.bilateralFilter(u8,d,sc,ss) */
/* PARITY UNPINNED.  OpenCV 4.1.0 bilateral_filter: radius=d/2, border          */
/* REFLECT_101, circular support (r <= radius), weights exp() in double cast    */
/* to float, float accumulation in tap order (row offset outer, column inner),  */
/* result cvRound(sum/wsum) (round half to even).  use_fma selects              */
/* sum = fma(val, w, sum) as OpenCV's v_muladd does on FMA3 builds.             */
/* ------------------------------------------------------------------------- */
static inline int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * (len - 1) - p;
    }
    return p;
}

ORC_API int orc_bilateral_tables(int d, double sigma_color, double sigma_space,
                                 float *color_w /*256*/, float *space_w /*d*d*/,
                                 int *tap_dr, int *tap_dc)
{
    if (sigma_color <= 0) sigma_color = 1;
    if (sigma_space <= 0) sigma_space = 1;
    double gc = -0.5 / (sigma_color * sigma_color);
    double gs = -0.5 / (sigma_space * sigma_space);
    int radius = d <= 0 ? (int)lrint(sigma_space * 1.5) : d / 2;
    if (radius < 1) radius = 1;
    for (int i = 0; i < 256; ++i) color_w[i] = (float)exp((double)(i * i) * gc);
    int k = 0;
    for (int i = -radius; i <= radius; ++i)
        for (int j = -radius; j <= radius; ++j) {
            double r = sqrt((double)i * i + (double)j * j);
            if (r > radius) continue;
            space_w[k] = (float)exp(r * r * gs);
            tap_dr[k] = i;
            tap_dc[k] = j;
            ++k;
        }
    return k;
}

ORC_API void orc_bilateral_u8(const uint8_t *src, int H, int W, int d,
                              double sigma_color, double sigma_space,
                              int use_fma, uint8_t *dst)
{
    float color_w[256], space_w[32 * 32];
    int tdr[32 * 32], tdc[32 * 32];
    int maxk = orc_bilateral_tables(d, sigma_color, sigma_space, color_w, space_w, tdr, tdc);
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < W; ++c) {
            int v0 = src[(int64_t)r * W + c];
            float sum = 0.f, wsum = 0.f;
            for (int k = 0; k < maxk; ++k) {
                int rr = reflect101(r + tdr[k], H), cc = reflect101(c + tdc[k], W);
                int v = src[(int64_t)rr * W + cc];
                float w = space_w[k] * color_w[abs(v - v0)];
                if (use_fma) sum = fmaf((float)v, w, sum);
                else sum += (float)v * w;   /* built with -ffp-contract=off */
                wsum += w;
            }
            dst[(int64_t)r * W + c] = (uint8_t)lrintf(sum / wsum);
        }
}

/* ------------------------------------------------------------------------- */
/* thresh_to_zero: image_filtering.py:310-316 -> cv2.threshold(THRESH_TOZERO)  */
/* dst = src > thresh ? src : 0.  PARITY UNPINNED (trivial integer rule).      */
/* ------------------------------------------------------------------------- */
ORC_API void orc_thresh_tozero_u8(const uint8_t *src, int64_t count, int thresh, uint8_t *dst)
{
    for (int64_t i = 0; i < count; ++i) dst[i] = src[i] > thresh ? src[i] : 0;
}

/* ------------------------------------------------------------------------- */
/* grayscale_opening: image_filtering.py:319-322 -> scipy.ndimage.grey_opening */
/* (size=(kh,kw)): flat erosion then flat dilation, border mode 'reflect'      */
/* (d c b a | a b c d | d c b a).  The reference uses (3,3) only              */
/* (data_structures.py:202); any size is accepted here (scipy's placement of   */
/* an even window restated in flat_minmax).  PINNED by golden fixtures made    */
/* with scipy through the reference's own function, odd and even sizes.        */
/* ------------------------------------------------------------------------- */
static inline int reflect_sym(int p, int len)
{
    while (p < 0 || p >= len) {
        if (p < 0) p = -p - 1;
        else p = 2 * len - 1 - p;
    }
    return p;
}

static void flat_minmax(const uint8_t *src, int H, int W, int kh, int kw, int is_max, uint8_t *dst)
{
    /* scipy.ndimage.grey_erosion(size=k) looks at offsets -k/2 .. k-1-k/2; grey_dilation at the mirrored ones (its origin is
       negated, and shifted by one for an even size): the same for an odd size, one apart for an even one */
    int lo_r = is_max ? -(kh - 1 - kh / 2) : -(kh / 2), hi_r = is_max ? kh / 2 : kh - 1 - kh / 2;
    int lo_c = is_max ? -(kw - 1 - kw / 2) : -(kw / 2), hi_c = is_max ? kw / 2 : kw - 1 - kw / 2;
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < W; ++c) {
            int acc = is_max ? 0 : 255;
            for (int i = lo_r; i <= hi_r; ++i)
                for (int j = lo_c; j <= hi_c; ++j) {
                    int v = src[(int64_t)reflect_sym(r + i, H) * W + reflect_sym(c + j, W)];
                    if (is_max) { if (v > acc) acc = v; }
                    else        { if (v < acc) acc = v; }
                }
            dst[(int64_t)r * W + c] = (uint8_t)acc;
        }
}

ORC_API int orc_grey_open_u8(const uint8_t *src, int H, int W, int kh, int kw, uint8_t *dst)
{
    if (kh < 1 || kw < 1) return -1;
    uint8_t *tmp = (uint8_t *)malloc((size_t)H * W);
    if (!tmp) return -2;
    flat_minmax(src, H, W, kh, kw, 0, tmp);
    flat_minmax(tmp, H, W, kh, kw, 1, dst);
    free(tmp);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* resize_frame: image_filtering.py:206-212 -> cv2.resize(frame, (w, h)),      */
/* INTER_LINEAR on 8-bit pixels.  PARITY UNPINNED (and dead code in the        */
/* reference: its two call sites are commented out, data_structures.py:179-181,*/
/* image_filtering.py:117-118).  OpenCV 4.1.0's generic 8u path restated:      */
/* source coordinate fx = (dx + 0.5) * (src / dst) - 0.5, clamped to the image */
/* (weight 0 on the missing neighbour); weights as 11-bit fixed point          */
/* (saturate_cast<short>(w * 2048)); horizontal sums in int32; vertical:       */
/* (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.             */
/* ------------------------------------------------------------------------- */
static void resize_axis(int src, int dst, int *idx, short *w)
{
    double scale = (double)src / dst;
    for (int d = 0; d < dst; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s0 = (int)floorf(f);
        f -= s0;
        if (s0 < 0) { s0 = 0; f = 0.f; }
        if (s0 >= src - 1) { s0 = src - 1; f = 0.f; }
        idx[d] = s0;
        w[2 * d] = (short)lrintf((1.f - f) * 2048.f);
        w[2 * d + 1] = (short)lrintf(f * 2048.f);
    }
}

ORC_API int orc_resize_linear_u8(const uint8_t *src, int H, int W, int ch, int dH, int dW, uint8_t *dst)
{
    int *xi = (int *)malloc(sizeof(int) * dW), *yi = (int *)malloc(sizeof(int) * dH);
    short *xw = (short *)malloc(sizeof(short) * 2 * dW), *yw = (short *)malloc(sizeof(short) * 2 * dH);
    if (!xi || !yi || !xw || !yw) { free(xi); free(yi); free(xw); free(yw); return -2; }
    resize_axis(W, dW, xi, xw);
    resize_axis(H, dH, yi, yw);
    for (int y = 0; y < dH; ++y) {
        int y0 = yi[y], y1 = y0 + 1 < H ? y0 + 1 : y0;
        for (int x = 0; x < dW; ++x) {
            int x0 = xi[x], x1 = x0 + 1 < W ? x0 + 1 : x0;
            for (int c = 0; c < ch; ++c) {
                int s0 = src[((int64_t)y0 * W + x0) * ch + c] * xw[2 * x] + src[((int64_t)y0 * W + x1) * ch + c] * xw[2 * x + 1];
                int s1 = src[((int64_t)y1 * W + x0) * ch + c] * xw[2 * x] + src[((int64_t)y1 * W + x1) * ch + c] * xw[2 * x + 1];
                int v = (((yw[2 * y] * (s0 >> 4)) >> 16) + ((yw[2 * y + 1] * (s1 >> 4)) >> 16) + 2) >> 2;
                dst[((int64_t)y * dW + x) * ch + c] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
            }
        }
    }
    free(xi); free(yi); free(xw); free(yw);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* cc_labeling: image_filtering.py:325-329 -> cv2.connectedComponents(frame,4) */
/* PARITY UNPINNED.  The literal 4 lands in the binding's `labels` slot, so    */
/* OpenCV runs its default: 8-connectivity, CCL_DEFAULT (= BBDT, Grana et al.) */
/* whose union-find roots are the smallest provisional label and whose         */
/* provisional labels are issued in 2x2-block raster order; flattening then    */
/* numbers components 1..K by the block-raster position of their first block.  */
/* For 4-connectivity OpenCV uses SAUF whose numbering is first-pixel raster   */
/* order (identical to scipy.ndimage.label).  Both rules are restated as       */
/* "rank components by a key", which is what the two algorithms compute:       */
/*   order 0 (raster):   key = r*W + c of the first pixel                      */
/*   order 1 (block2x2): key = (r>>1)*ceil(W/2) + (c>>1) of the first block    */
/* Foreground = nonzero.  Output int32 labels (0 = background).                */
/* ------------------------------------------------------------------------- */
typedef struct { int64_t key; int32_t comp; } orc_rank_t;

static int rank_cmp(const void *a, const void *b)
{
    int64_t ka = ((const orc_rank_t *)a)->key, kb = ((const orc_rank_t *)b)->key;
    return ka < kb ? -1 : (ka > kb ? 1 : 0);
}

ORC_API int orc_ccl_u8(const uint8_t *src, int H, int W, int connectivity, int order, int32_t *labels)
{
    int64_t P = (int64_t)H * W;
    if (connectivity == 4) order = 0;   /* OpenCV's 4-way algorithm (SAUF) numbers in pixel raster order;
                                           two 4-components can share a 2x2 block, so block order is 8-way only */
    int32_t *stack = (int32_t *)malloc(sizeof(int32_t) * (size_t)P);
    orc_rank_t *rk = (orc_rank_t *)malloc(sizeof(orc_rank_t) * (size_t)(P / 1 + 1));
    if (!stack || !rk) { free(stack); free(rk); return -2; }
    int Wb = (W + 1) / 2;
    for (int64_t i = 0; i < P; ++i) labels[i] = src[i] ? -1 : 0;
    int ncomp = 0;
    static const int dr8[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
    static const int dc8[8] = {-1, 0, 1, -1, 1, -1, 0, 1};
    static const int dr4[4] = {-1, 0, 0, 1};
    static const int dc4[4] = {0, -1, 1, 0};
    const int *dr = connectivity == 4 ? dr4 : dr8, *dc = connectivity == 4 ? dc4 : dc8;
    int nn = connectivity == 4 ? 4 : 8;
    for (int64_t s = 0; s < P; ++s) {
        if (labels[s] != -1) continue;
        int comp = ++ncomp;           /* provisional id in raster discovery order */
        int64_t best = INT64_MAX;
        int sp = 0;
        stack[sp++] = (int32_t)s;
        labels[s] = comp;
        while (sp) {
            int32_t p = stack[--sp];
            int r = p / W, c = p % W;
            int64_t key = order == 0 ? (int64_t)p : (int64_t)(r >> 1) * Wb + (c >> 1);
            if (key < best) best = key;
            for (int k = 0; k < nn; ++k) {
                int r2 = r + dr[k], c2 = c + dc[k];
                if (r2 < 0 || r2 >= H || c2 < 0 || c2 >= W) continue;
                int64_t q = (int64_t)r2 * W + c2;
                if (labels[q] == -1) { labels[q] = comp; stack[sp++] = (int32_t)q; }
            }
        }
        rk[comp - 1].key = best;
        rk[comp - 1].comp = comp;
    }
    qsort(rk, (size_t)ncomp, sizeof(orc_rank_t), rank_cmp);
    int32_t *remap = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ncomp + 1));
    if (!remap) { free(stack); free(rk); return -2; }
    remap[0] = 0;
    for (int i = 0; i < ncomp; ++i) remap[rk[i].comp] = i + 1;
    for (int64_t i = 0; i < P; ++i) labels[i] = remap[labels[i]];
    free(remap); free(stack); free(rk);
    return ncomp;
}

/* labeled_frame.astype(np.uint8): image_filtering.py:329 (wraps mod 256). */
ORC_API void orc_labels_to_u8(const int32_t *labels, int64_t count, uint8_t *out)
{
    for (int64_t i = 0; i < count; ++i) out[i] = (uint8_t)(labels[i] & 0xff);
}

/* ------------------------------------------------------------------------- */
/* get_segment_properties: image_filtering.py:332-335 ->                      */
/* skimage.measure.regionprops(u8 label image).  One region per distinct       */
/* nonzero label value, ascending; fields consumed downstream:                 */
/* label, bbox=(min_r,min_c,max_r+1,max_c+1), centroid=mean(coords), area.      */
/* PINNED by golden fixtures (skimage).  Centroid is returned as integer sums. */
/* ------------------------------------------------------------------------- */
typedef struct {
    int32_t label, r0, c0, r1, c1, pad_;
    int64_t area, sum_r, sum_c;
} orc_segment;

ORC_API int orc_regionprops_u8(const uint8_t *lab, int H, int W, orc_segment *out /*255*/)
{
    orc_segment t[256];
    for (int v = 0; v < 256; ++v) {
        t[v].label = v; t[v].r0 = INT32_MAX; t[v].c0 = INT32_MAX; t[v].r1 = -1; t[v].c1 = -1;
        t[v].pad_ = 0; t[v].area = 0; t[v].sum_r = 0; t[v].sum_c = 0;
    }
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < W; ++c) {
            int v = lab[(int64_t)r * W + c];
            if (!v) continue;
            orc_segment *s = &t[v];
            if (r < s->r0) s->r0 = r;
            if (c < s->c0) s->c0 = c;
            if (r + 1 > s->r1) s->r1 = r + 1;
            if (c + 1 > s->c1) s->c1 = c + 1;
            s->area++; s->sum_r += r; s->sum_c += c;
        }
    int n = 0;
    for (int v = 1; v < 256; ++v)
        if (t[v].area) out[n++] = t[v];
    return n;
}
