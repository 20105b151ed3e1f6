// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the IALM pass's access shapes (diagnostic, not part of
// the library).  MI355X_MICROARCH.md (HBM): FETCH_SIZE tallies 128-B requests at 64 B; "other access widths are
// uncalibrated: calibrate on a known byte count in your own access pattern".  Each kernel below moves a KNOWN
// number of bytes in layout L (a wave instruction covers 4 frame rows x 16 pixels, lane = pixel (l&15), row l>>4):
//   rd8 / rd32 / rd64 : read  a [windows][64][P] array of u8 / f32 / f64 once
//   wr8 / wr32 / wr64 : write the same arrays once
// Run under   rocprofv3 --kernel-trace --pmc FETCH_SIZE ...   and   --pmc WRITE_SIZE ...  ; the factor of a shape is
// known bytes / (counter x 1024).   Build: hipcc --offload-arch=gfx950 -O3 -o tools/fetch_calib tools/fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <typename T, bool WRITE>
__global__ __launch_bounds__(256) void k_shape(T *__restrict__ buf, int P, int ntiles, T *__restrict__ sink)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T *base = buf + (size_t)blockIdx.y * 64 * P;
    T acc = 0;
    // same tile -> wave mapping as k_ialm_pass_v3: groups of 8 tiles per block, two per wave
    const int nlg = (ntiles + 7) >> 3;
    for (int lg = blockIdx.x; lg < nlg; lg += gridDim.x)
        for (int h = 0; h < 2; ++h) {
            const int tile = lg * 8 + wave * 2 + h;
            if (tile >= ntiles) continue;
            const int p = tile * 16 + (lane & 15);
            if (p >= P) continue;
            const size_t o = (size_t)(lane >> 4) * P + p;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                if (WRITE) base[o + (size_t)4 * t * P] = (T)(t + lane);
                else acc += base[o + (size_t)4 * t * P];
            }
        }
    if (!WRITE && acc == (T)123457) sink[0] = acc;
}

template <typename T, bool WRITE>
static void run(const char *name, void *buf, int W, int P, void *sink)
{
    const int ntiles = (P + 15) / 16;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_shape<T, WRITE>), dim3(12, W), dim3(256), 0, 0, (T *)buf, P, ntiles, (T *)sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep == 2) printf("%-6s known_bytes %zu  %.3f ms  %.1f GB/s\n", name, (size_t)W * 64 * P * sizeof(T), ms,
                             (double)W * 64 * P * sizeof(T) / ms * 1e-6);
    }
}

int main()
{
    const int W = 128, P = 89888;
    const size_t elems = (size_t)W * 64 * P;
    void *buf, *sink;
    if (hipMalloc(&buf, elems * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&sink, 64);
    hipMemset(buf, 1, elems * 8);
    hipDeviceSynchronize();
    run<uint8_t, false>("rd8", buf, W, P, sink);
    run<float, false>("rd32", buf, W, P, sink);
    run<double, false>("rd64", buf, W, P, sink);
    run<uint8_t, true>("wr8", buf, W, P, sink);
    run<float, true>("wr32", buf, W, P, sink);
    run<double, true>("wr64", buf, W, P, sink);
    hipDeviceSynchronize();
    return 0;
}
