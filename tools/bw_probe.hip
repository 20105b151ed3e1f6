// Bandwidth probe for the IALM pass's access pattern (diagnostic, not part of the library).
// Streams two f64 arrays laid out as [frames][P] the way layout L does:
//   mode 0: 8 B/lane, a wave instruction covers 4 frame rows x 16 pixels (128-B segments)
//   mode 1: 16 B/lane, 4 frame rows x 32 pixels (256-B segments)
//   mode 2: 16 B/lane, fully contiguous 1 KiB per wave instruction
// Each element is read from A and Y and written back to A and Y (32 B/element of traffic).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void probe(double *__restrict__ A, double *__restrict__ Y, int n, int P, int tiles_per_wave, double *__restrict__ A2 = nullptr, double *__restrict__ Y2 = nullptr)
{
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int nwaves = gridDim.x * 4;
    if (MODE == 0) {
        const int ntiles = P / 16;
        for (int tile = wave; tile < ntiles; tile += nwaves) {
            const size_t o = (size_t)(lane >> 4) * P + tile * 16 + (lane & 15);
            double a[16], y[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) { a[t] = A[o + (size_t)4 * t * P]; y[t] = Y[o + (size_t)4 * t * P]; }
            double *Ao = A2 ? A2 : A, *Yo = Y2 ? Y2 : Y;
#pragma unroll
            for (int t = 0; t < 16; ++t) { Ao[o + (size_t)4 * t * P] = a[t] + y[t]; Yo[o + (size_t)4 * t * P] = a[t] - y[t]; }
        }
    } else if (MODE == 1) {
        const int ntiles = P / 32;
        for (int tile = wave; tile < ntiles; tile += nwaves) {
            const size_t o = (size_t)(lane >> 4) * P + tile * 32 + 2 * (lane & 15);
            double2 a[16], y[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) { a[t] = *(double2 *)&A[o + (size_t)4 * t * P]; y[t] = *(double2 *)&Y[o + (size_t)4 * t * P]; }
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                double2 s = {a[t].x + y[t].x, a[t].y + y[t].y}, d = {a[t].x - y[t].x, a[t].y - y[t].y};
                *(double2 *)&A[o + (size_t)4 * t * P] = s; *(double2 *)&Y[o + (size_t)4 * t * P] = d;
            }
        }
    } else {
        const size_t total = (size_t)n * P / 2;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
            double2 a = ((double2 *)A)[i], y = ((double2 *)Y)[i];
            double2 s = {a.x + y.x, a.y + y.y}, d = {a.x - y.x, a.y - y.y};
            ((double2 *)A)[i] = s; ((double2 *)Y)[i] = d;
        }
    }
}

int main()
{
    const int n = 64, P = 89888, W = 16;          // 16 windows back to back = one launch of the real kernel
    const size_t elems = (size_t)W * n * P;
    double *A, *Y;
    hipMalloc(&A, elems * 8); hipMalloc(&Y, elems * 8);
    hipMemset(A, 0, elems * 8); hipMemset(Y, 0, elems * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double *A2, *Y2;
    hipMalloc(&A2, elems * 8); hipMalloc(&Y2, elems * 8);
    for (int blocks : {4096, 16384}) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(e0);
            for (int w = 0; w < W; ++w) {
                size_t off = (size_t)w * n * P;
                if (rep & 1) probe<0><<<blocks / W, 256>>>(A2 + off, Y2 + off, n, P, 0, A + off, Y + off);
                else probe<0><<<blocks / W, 256>>>(A + off, Y + off, n, P, 0, A2 + off, Y2 + off);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("mode 0 OUT-OF-PLACE blocks %5d: %.3f ms  %.0f GB/s\n", blocks, best, elems * 32.0 / best / 1e6);
    }
    for (int mode = 0; mode < 3; ++mode)
        for (int blocks : {4096, 8192, 16384, 32768}) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                // treat the W windows as one long plane set: n rows of W*P would change strides; instead launch per window
                for (int w = 0; w < W; ++w) {
                    double *a = A + (size_t)w * n * P, *y = Y + (size_t)w * n * P;
                    if (mode == 0) probe<0><<<blocks / W > 0 ? blocks / W : 1, 256>>>(a, y, n, P, 0);
                    else if (mode == 1) probe<1><<<blocks / W > 0 ? blocks / W : 1, 256>>>(a, y, n, P, 0);
                    else probe<2><<<blocks / W > 0 ? blocks / W : 1, 256>>>(a, y, n, P, 0);
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("mode %d blocks %5d: %.3f ms  %.0f GB/s\n", mode, blocks, best, elems * 32.0 / best / 1e6);
        }
    return 0;
}
