// Do f32 MFMAs (v_mfma_f32_32x32x2_f32) and f32 VALU instructions overlap on one SIMD of gfx950?
// Workgroups of 512 threads = 2 waves per SIMD, one workgroup per CU (grid = CU count).  Modes:
//   0: every wave only MFMA, ONE accumulator (a dependent chain, like the Winograd kernel's M)
//   1: every wave only MFMA, 4 independent accumulators
//   2: every wave only v_fma_f32 (64 per iteration, 8 independent chains)
//   3: waves 0-3 MFMA (one accumulator), waves 4-7 (their SIMD partners) v_fma_f32      -> overlap between waves?
//   4: every wave: 8 x (1 MFMA on one accumulator + 8 v_fma_f32)                          -> overlap inside a wave?
//   5: every wave: 8 x (1 MFMA + 4 v_pk_fma_f32)
//   6: every wave: 8 MFMAs, then 64 v_fma_f32 (the two waves of a SIMD in step)           -> phases that coincide
// Prints milliseconds; per wave and iteration: 8 MFMAs = 512 cycles of the matrix pipe, 64 FMAs = 256 issue cycles.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(512, 2) void probe(float *out, int iters, float seed)
{
    const int wave = threadIdx.x >> 6;
    f16v acc[4];
    for (int k = 0; k < 4; ++k)
        for (int e = 0; e < 16; ++e) acc[k][e] = seed + k;
    float f[8];
    f2v g[4];
    for (int i = 0; i < 8; ++i) f[i] = seed + i + threadIdx.x;
    for (int i = 0; i < 4; ++i) g[i] = f2v{f[i], f[i + 4]};
    const float a = 1.0000001f + seed, b = 0.5f;
    const f2v a2 = {a, a}, b2 = {b, b};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || (MODE == 3 && wave < 4)) {
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
        } else if (MODE == 1) {
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k & 3], 0, 0, 0);
        } else if (MODE == 2 || (MODE == 3 && wave >= 4)) {
#pragma unroll
            for (int k = 0; k < 64; ++k) f[k & 7] = __builtin_fmaf(f[k & 7], a, b);
        } else if (MODE == 4) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = __builtin_fmaf(f[j], a, b);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
            }
        } else if (MODE == 5) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) g[j] = __builtin_elementwise_fma(g[j], a2, b2);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            }
        } else if (MODE == 6) {
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 64; ++k) f[k & 7] = __builtin_fmaf(f[k & 7], a, b);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    for (int i = 0; i < 8; ++i) s += f[i];
    for (int i = 0; i < 4; ++i) s += g[i][0] + g[i][1];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE>
float run(float *out, int cus, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE><<<cus, 512>>>(out, iters, 0.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<MODE><<<cus, 512>>>(out, iters, 0.0f);
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
    float *out;
    hipMalloc(&out, (size_t)cus * 512 * 4);
    printf("CUs %d, iterations %d; 2 waves per SIMD; per wave-iteration 8 MFMAs (512 pipe cycles) and/or 64 FMAs\n", cus, iters);
    printf("mode 0  all waves MFMA f32 32x32x2, one accumulator            %8.3f ms\n", run<0>(out, cus, iters));
    printf("mode 1  all waves MFMA, four accumulators                       %8.3f ms\n", run<1>(out, cus, iters));
    printf("mode 2  all waves v_fma_f32 (64 per iteration)                  %8.3f ms\n", run<2>(out, cus, iters));
    printf("mode 3  SIMD partners: one MFMA, one v_fma_f32                  %8.3f ms\n", run<3>(out, cus, iters));
    printf("mode 4  every wave: 8 x (1 MFMA + 8 v_fma_f32)                  %8.3f ms\n", run<4>(out, cus, iters));
    printf("mode 5  every wave: 8 x (1 MFMA + 4 v_pk_fma_f32)               %8.3f ms\n", run<5>(out, cus, iters));
    printf("mode 6  every wave: 8 MFMAs, then 64 v_fma_f32                  %8.3f ms\n", run<6>(out, cus, iters));
    return 0;
}
