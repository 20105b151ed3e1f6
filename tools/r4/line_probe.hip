// Streaming-read rate by how a wave-instruction's 64 x 16 B are laid over 128-byte lines (tools/r4: why the classifier's memory-bound kernels
// plateau near 4 TB/s inside a forward while a plain streaming read reaches 6).  "Pixels" of C floats (C * 4 bytes apart), a wave owns 32 of them
// per tile and reads 32 channels (one 128-byte line) of each per step:
//   mode 0: lane (r = l & 31, h = l >> 5) reads 16 B at channel 16 h + 4 v of pixel r, v = 0..3   (k_conv1x1_relu_place: 32 lines per
//           instruction, a quarter^H^H half of each line's two 64-byte halves ... every line is touched by all four instructions)
//   mode 1: lane (g = l >> 3, q = l & 7) reads 16 B at channel 4 q of pixel g + 8 v, v = 0..3        (8 whole lines per instruction)
// Both read exactly the same bytes.   hipcc -O3 --offload-arch=gfx950 tools/r4/line_probe.hip -o tools/r4/line_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int MODE>
__global__ __launch_bounds__(1024) void k_probe(const float *__restrict__ x, long npix, int C, float *__restrict__ out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const long ntiles = npix / 32, stride = (long)gridDim.x * nw;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long t = (long)blockIdx.x * nw + wave; t < ntiles; t += stride) {
        const float *base = x + t * 32 * C;
        for (int kb = 0; kb < C; kb += 64) {          // two chunks in flight
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float *p = MODE == 0 ? base + (long)(lane & 31) * C + kb + 32 * u + 16 * (lane >> 5) + 4 * i
                                               : base + (long)((lane >> 3) + 8 * i) * C + kb + 32 * u + 4 * (lane & 7);
                    v[4 * u + i] = *(const float4 *)p;
                }
#pragma unroll
            for (int i = 0; i < 8; ++i) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[threadIdx.x] = acc.x;
}

int main(int argc, char **argv)
{
    const int C = argc > 1 ? atoi(argv[1]) : 384;
    const long npix = (argc > 2 ? atol(argv[2]) : 8192L * 196) / 32 * 32;
    float *x, *out;
    const size_t bytes = (size_t)npix * C * 4;
    hipMalloc(&x, bytes); hipMalloc(&out, 4096);
    hipMemset(x, 0, bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int wg = 256; wg <= 1024; wg *= 2)
        for (int per_cu = 1; per_cu <= 1024 / wg * 2; per_cu *= 2)
            for (int mode = 0; mode < 2; ++mode) {
                float best = 1e9f;
                for (int rep = 0; rep < 5; ++rep) {
                    hipEventRecord(a);
                    if (mode == 0) hipLaunchKernelGGL(k_probe<0>, dim3(256 * per_cu), dim3(wg), 0, 0, x, npix, C, out);
                    else hipLaunchKernelGGL(k_probe<1>, dim3(256 * per_cu), dim3(wg), 0, 0, x, npix, C, out);
                    hipEventRecord(b); hipEventSynchronize(b);
                    float ms; hipEventElapsedTime(&ms, a, b);
                    if (ms < best) best = ms;
                }
                printf("C %d  %.2f GB  wg %4d x %d per CU  mode %d: %.1f us  %.2f TB/s\n", C, bytes / 1e9, wg, per_cu, mode, best * 1e3, bytes / best / 1e9);
            }
    return 0;
}
