// Do f64 MFMAs and f64 VALU instructions overlap on one SIMD of gfx950, or do they share the execution units?
// Workgroups of 512 threads = 2 waves per SIMD, one workgroup per CU (grid = CU count).  Modes:
//   0: every wave only MFMA (v_mfma_f64_16x16x4_f64, 4 independent accumulators)
//   1: every wave only f64 VALU (v_fma_f64, 8 independent chains)
//   2: waves 0-3 MFMA, waves 4-7 (their SIMD partners) f64 VALU          -> overlap between waves?
//   3: every wave interleaves 1 MFMA : R f64 FMAs                         -> overlap inside a wave?
//   4: waves 0-3 MFMA, waves 4-7 f32 VALU (v_fma_f32)
// Prints milliseconds; if time(2) ~ max(time(0)/2, time(1)/2) the pipes are separate, if ~ sum they are shared.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512, 2) void probe(double *out, int iters, double seed)
{
    const int wave = threadIdx.x >> 6;
    d4 acc[4] = {{seed, 0, 0, 0}, {0, seed, 0, 0}, {0, 0, seed, 0}, {0, 0, 0, seed}};
    double v[8];
    float f[8];
    for (int i = 0; i < 8; ++i) { v[i] = seed + i + threadIdx.x; f[i] = (float)v[i]; }
    const double a = 1.0000001 + seed, b = 0.5;
    const bool do_mfma = MODE == 0 || MODE == 3 || ((MODE == 2 || MODE == 4) && wave < 4);
    const bool do_valu = MODE == 1 || MODE == 3 || (MODE == 2 && wave >= 4);
    const bool do_f32 = MODE == 4 && wave >= 4;
    for (int it = 0; it < iters; ++it) {
        if (do_mfma && !do_valu) {
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[k & 3], 0, 0, 0);
        } else if (do_valu && !do_mfma) {
#pragma unroll
            for (int k = 0; k < 64; ++k) v[k & 7] = __builtin_fma(v[k & 7], a, b);
        } else if (do_mfma && do_valu) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                acc[k & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[k & 3], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = __builtin_fma(v[j], a, b);
            }
        } else if (do_f32) {
#pragma unroll
            for (int k = 0; k < 64; ++k) f[k & 7] = __builtin_fmaf(f[k & 7], 1.0000001f, 0.5f);
        }
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i] + f[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE>
float run(double *out, int cus, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE><<<cus, 512>>>(out, iters, 0.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<MODE><<<cus, 512>>>(out, iters, 0.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, iters = 20000;
    double *out;
    hipMalloc(&out, (size_t)cus * 512 * 8);
    // per wave and iteration: 8 MFMAs (8 x 64 = 512 cycles of matrix pipe) and / or 64 f64 FMAs
    printf("CUs %d, iterations %d\n", cus, iters);
    printf("mode 0  all waves MFMA f64 (8 per iteration)                  %8.3f ms\n", run<0>(out, cus, iters));
    printf("mode 1  all waves f64 FMA (64 per iteration)                  %8.3f ms\n", run<1>(out, cus, iters));
    printf("mode 2  SIMD partners: one MFMA f64, one f64 FMA              %8.3f ms\n", run<2>(out, cus, iters));
    printf("mode 3  every wave: 8 x (1 MFMA f64 + 8 f64 FMA)              %8.3f ms\n", run<3>(out, cus, iters));
    printf("mode 4  SIMD partners: one MFMA f64, one f32 FMA              %8.3f ms\n", run<4>(out, cus, iters));
    return 0;
}
