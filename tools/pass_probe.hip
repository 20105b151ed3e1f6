// Memory-shape probe for the M-state IALM pass (diagnostic, not part of the library): moves exactly the pass's
// bytes (X u8 + M f64 + U f16 read, M f64 + U f16 + S u8 written in place) with no arithmetic, in two shapes:
//   shape 0: as k_ialm_pass_v3 -- X/S frame-major planes (16-byte row segments), M/U in the pass's chunk layout
//            (one contiguous 512-B / 128-B piece per wave instruction)
//   shape 1: X and S as one dword per lane over a 64-pixel super-tile (64-B row segments), U and M per tile as before
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/pass_probe tools/pass_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int SHAPE>
__global__ __launch_bounds__(256, 2) void k_probe(const uint8_t *__restrict__ X, uint8_t *__restrict__ S, double *__restrict__ M,
                                                   unsigned short *__restrict__ U, int P, int ntiles)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pl = lane & 15, fr0 = lane >> 4;
    const size_t wofs = (size_t)blockIdx.y * 64 * P;
    const size_t wofs_c = (size_t)blockIdx.y * 64 * ((size_t)(P + 127) / 128 * 128);      // chunk layout: whole groups
    X += wofs; S += wofs; M += (SHAPE == 0 ? wofs_c : wofs); U += (SHAPE == 0 ? wofs_c : wofs);
    if (SHAPE == 0) {
        const int nlg = (ntiles + 7) >> 3;
        for (int lg = blockIdx.x; lg < nlg; lg += gridDim.x)
            for (int h = 0; h < 2; ++h) {
                const int tile = lg * 8 + wave * 2 + h;
                if (tile >= ntiles) continue;
                const size_t o = (size_t)fr0 * P + tile * 16 + pl;
                // M and U as the pass keeps them: [group of 128 px][k-step t][tile][frame 4t + 0..3][16 px]
                const size_t oc = (size_t)(tile >> 3) * 64 * 128 + (size_t)(tile & 7) * 64 + fr0 * 16 + pl;
                int x[16]; double m[16]; unsigned short u[16];
#pragma unroll
                for (int t = 0; t < 16; ++t) { x[t] = X[o + (size_t)4 * t * P]; m[t] = M[oc + (size_t)t * 512]; u[t] = U[oc + (size_t)t * 512]; }
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    M[oc + (size_t)t * 512] = m[t] + (double)u[t];
                    U[oc + (size_t)t * 512] = (unsigned short)(u[t] + x[t]);
                    S[o + (size_t)4 * t * P] = (uint8_t)(x[t] + 1);
                }
            }
    } else {
        // super-tile = 64 pixels = 4 tiles, one wave; 4 waves of a block take 4 consecutive super-tiles (2 lines)
        const int nst = (ntiles + 3) >> 2;
        for (int st = blockIdx.x * 4 + wave; st < nst; st += gridDim.x * 4) {
            const size_t ox = (size_t)fr0 * P + st * 64 + 4 * pl;
            uint32_t xd[16], sd[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) { xd[t] = *(const uint32_t *)(X + ox + (size_t)4 * t * P); sd[t] = 0; }
            for (int q = 0; q < 4; ++q) {
                const int tile = st * 4 + q;
                if (tile >= ntiles) break;
                const size_t o = (size_t)fr0 * P + tile * 16 + pl;
                double m[16]; unsigned short u[16];
#pragma unroll
                for (int t = 0; t < 16; ++t) { m[t] = M[o + (size_t)4 * t * P]; u[t] = U[o + (size_t)4 * t * P]; }
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int x = (xd[t] >> (8 * q)) & 255;
                    M[o + (size_t)4 * t * P] = m[t] + (double)u[t];
                    U[o + (size_t)4 * t * P] = (unsigned short)(u[t] + x);
                    sd[t] |= (uint32_t)((x + 1) & 255) << (8 * q);
                }
            }
#pragma unroll
            for (int t = 0; t < 16; ++t) *(uint32_t *)(S + ox + (size_t)4 * t * P) = sd[t];
        }
    }
}

int main()
{
    const int W = 128, P = 89888;           // P % 64 == 32: the last super-tile is half full (bytes past P land in the next plane: harmless here)
    const size_t elems = (size_t)W * 64 * P;
    uint8_t *X, *S; double *M; unsigned short *U;
    if (hipMalloc(&X, elems + 256) != hipSuccess || hipMalloc(&S, elems + 256) != hipSuccess || hipMalloc(&M, (size_t)W * 64 * ((P + 127) / 128 * 128) * 8) != hipSuccess ||
        hipMalloc(&U, (size_t)W * 64 * ((P + 127) / 128 * 128) * 2) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(X, 1, elems); (void)hipMemset(S, 0, elems); (void)hipMemset(M, 0, elems * 8); (void)hipMemset(U, 0, elems * 2);
    const int ntiles = (P + 15) / 16;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int shape = 0; shape < 2; ++shape)
        for (int rep = 0; rep < 4; ++rep) {
            (void)hipEventRecord(e0);
            if (shape == 0) hipLaunchKernelGGL(k_probe<0>, dim3(12, W), dim3(256), 0, 0, X, S, M, U, P, ntiles);
            else hipLaunchKernelGGL(k_probe<1>, dim3(12, W), dim3(256), 0, 0, X, S, M, U, P, ntiles);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("shape %d  %.3f ms  %.1f GB/s algorithmic (22 B/element)\n", shape, ms, (double)elems * 22 / ms * 1e-6);
        }
    return 0;
}
