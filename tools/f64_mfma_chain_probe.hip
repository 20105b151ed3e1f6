// How many independent accumulator chains does ONE wave need to issue v_mfma_f64_16x16x4_f64 at the pipe's rate
// (64 cycles per instruction per SIMD on gfx950)?  Blocks of 256 threads (1 wave per SIMD) or 512 (2 per SIMD),
// one block per CU; each wave issues `iters` x 16 MFMAs round-robin over CH accumulators; operands from registers.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int CH, int THREADS>
__global__ __launch_bounds__(THREADS, THREADS / 256) void probe(double *out, int iters, double seed)
{
    d4 acc[CH];
    for (int i = 0; i < CH; ++i) acc[i] = d4{seed + i, 0, 0, 0};
    double a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = 1.0 + seed * i; b[i] = 0.5 + seed + i + threadIdx.x * 1e-9; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k % CH] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[k & 3], b[(k >> 2) & 3], acc[k % CH], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < CH; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
}

template <int CH, int THREADS>
void run(double *out, int cus, int iters, double ghz_guess)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<CH, THREADS><<<cus, THREADS>>>(out, iters, 0.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<CH, THREADS><<<cus, THREADS>>>(out, iters, 0.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)iters * 16 * (THREADS / 256);
    printf("waves/SIMD %d  chains/wave %d   %8.3f ms   %6.1f ns per MFMA per SIMD  (~%5.1f cycles at %.1f GHz)\n", THREADS / 256, CH, ms,
           ms * 1e6 / per_simd, ms * 1e6 / per_simd * ghz_guess, ghz_guess);
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, iters = 20000;
    double *out;
    hipMalloc(&out, (size_t)cus * 512 * 8);
    run<1, 256>(out, cus, iters, 2.4); run<2, 256>(out, cus, iters, 2.4); run<4, 256>(out, cus, iters, 2.4); run<8, 256>(out, cus, iters, 2.4);
    run<1, 512>(out, cus, iters, 2.4); run<2, 512>(out, cus, iters, 2.4); run<4, 512>(out, cus, iters, 2.4);
    return 0;
}
