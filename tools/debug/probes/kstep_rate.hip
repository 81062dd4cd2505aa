// Probe: what limits ONE wave's k-step of the f16x3 GEMM (12 MFMAs + 8 ds_read_b128 + 2 global 1-KiB loads)?  Each variant drops
// one ingredient.  One workgroup per CU; 4 waves (one per SIMD) or 8 (two per SIMD); time by events, per k-step of a wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
template <bool LDS, bool GLD, int THREADS>
__global__ __launch_bounds__(THREADS) void k(const uint4* __restrict__ w, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) _Float16 sX[2 * 128 * 264];
    for (int i = threadIdx.x; i < 2 * 128 * 264; i += THREADS) sX[i] = (_Float16)(0.001f * (i % 97));
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc[4];
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    h8 xh[4], xl[4], wh, wl;
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 8; ++r) { xh[c][r] = (_Float16)0.01f; xl[c][r] = (_Float16)0.001f; }
    for (int r = 0; r < 8; ++r) { wh[r] = (_Float16)0.5f; wl[r] = (_Float16)0.0001f; }
    const _Float16* bh = sX + (lane & 31) * 264 + 8 * (lane >> 5);
    const _Float16* bl = bh + 128 * 264;
    const uint4* wp = w + (size_t)(blockIdx.x % 4) * 65536 + wave * 2048 + lane;
    uint4 ring[4][2];
    for (int u = 0; u < 4; ++u) { ring[u][0] = wp[0]; ring[u][1] = wp[64]; wp += 128; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (LDS) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    xh[nt] = *reinterpret_cast<const h8*>(bh + nt * 32 * 264 + ((4 * i + u) & 15) * 16);
                    xl[nt] = *reinterpret_cast<const h8*>(bl + nt * 32 * 264 + ((4 * i + u) & 15) * 16);
                }
            }
            if (GLD) { wh = __builtin_bit_cast(h8, ring[u][0]); wl = __builtin_bit_cast(h8, ring[u][1]); }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = MFMA(wl, xh[nt], acc[nt]);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = MFMA(wh, xl[nt], acc[nt]);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = MFMA(wh, xh[nt], acc[nt]);
            if (GLD) { ring[u][0] = wp[0]; ring[u][1] = wp[64]; wp += 128; if ((i & 7) == 7 && u == 3) wp -= 128 * 32; }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
}
// the same k-step with the reads staggered: lo fragments requested in front of MFMA 1 (used from MFMA 9), the next k-step's hi
// fragments behind MFMA 8 (used from the next MFMA 1): product order Wl.xh, Wh.xh, Wh.xl, no second register set
template <int THREADS>
__global__ __launch_bounds__(THREADS) void kst(const uint4* __restrict__ w, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) _Float16 sX[2 * 128 * 264];
    for (int i = threadIdx.x; i < 2 * 128 * 264; i += THREADS) sX[i] = (_Float16)(0.001f * (i % 97));
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc[4];
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    h8 xh[4], xl[4], wh, wl;
    const _Float16* bh = sX + (lane & 31) * 264 + 8 * (lane >> 5);
    const _Float16* bl = bh + 128 * 264;
    const uint4* wp = w + (size_t)(blockIdx.x % 4) * 65536 + wave * 2048 + lane;
    uint4 ring[4][2];
    for (int u = 0; u < 4; ++u) { ring[u][0] = wp[0]; ring[u][1] = wp[64]; wp += 128; }
    for (int nt = 0; nt < 4; ++nt) xh[nt] = *reinterpret_cast<const h8*>(bh + nt * 32 * 264);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) xl[nt] = *reinterpret_cast<const h8*>(bl + nt * 32 * 264 + ((4 * i + u) & 15) * 16);
            wh = __builtin_bit_cast(h8, ring[u][0]); wl = __builtin_bit_cast(h8, ring[u][1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = MFMA(wl, xh[nt], acc[nt]);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = MFMA(wh, xh[nt], acc[nt]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) xh[nt] = *reinterpret_cast<const h8*>(bh + nt * 32 * 264 + ((4 * i + u + 1) & 15) * 16);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = MFMA(wh, xl[nt], acc[nt]);
            ring[u][0] = wp[0]; ring[u][1] = wp[64]; wp += 128; if ((i & 7) == 7 && u == 3) wp -= 128 * 32;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
}
template <int THREADS> void run_st(const char* name, const uint4* w) {
    float* d; hipMalloc(&d, 256 * THREADS * 4);
    const int iters = 2000;
    kst<THREADS><<<256, THREADS>>>(w, d, iters); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); kst<THREADS><<<256, THREADS>>>(w, d, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ksteps = 4.0 * iters, wps = THREADS / 256.0;
    printf("%-62s %7.1f ns per k-step of a wave; pipe %.0f %% busy\n", name, ms * 1e6 / ksteps, 100.0 * (12 * 17.1 * wps) / (ms * 1e6 / ksteps));
    hipFree(d);
}
template <bool LDS, bool GLD, int THREADS> void run(const char* name, const uint4* w) {
    float* d; hipMalloc(&d, 256 * THREADS * 4);
    const int iters = 2000;
    k<LDS, GLD, THREADS><<<256, THREADS>>>(w, d, iters); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k<LDS, GLD, THREADS><<<256, THREADS>>>(w, d, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ksteps = 4.0 * iters, wps = THREADS / 256.0;
    printf("%-62s %7.1f ns per k-step of a wave (12 MFMAs: 205 ns at the measured MFMA rate); pipe %.0f %% busy\n", name,
           ms * 1e6 / ksteps, 100.0 * (12 * 17.1 * wps) / (ms * 1e6 / ksteps));
    hipFree(d);
}
int main() {
    uint4* w; hipMalloc(&w, 4 * 65536 * 16 + (1 << 20)); hipMemset(w, 0, 4 * 65536 * 16 + (1 << 20));
    run<false, false, 256>("MFMAs only, 1 wave/SIMD", w);
    run<true, false, 256>("+ 8 ds_read_b128 per k-step, 1 wave/SIMD", w);
    run<false, true, 256>("+ 2 global 1-KiB loads per k-step (4-deep ring), 1 wave/SIMD", w);
    run<true, true, 256>("+ both, 1 wave/SIMD", w);
    run<true, true, 512>("+ both, 2 waves/SIMD", w);
    run<true, false, 512>("+ LDS reads only, 2 waves/SIMD", w);
    run_st<256>("staggered reads + loads, 1 wave/SIMD", w);
    run_st<512>("staggered reads + loads, 2 waves/SIMD", w);
    return 0;
}
