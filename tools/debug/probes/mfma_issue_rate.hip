// Probe: how fast can ONE wave issue v_mfma_f32_32x32x16_f16 (independent accumulators)?  Variants: accumulators in
// architectural VGPRs vs accumulation registers, 4 vs 8 independent chains, one or two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O2 mfma_issue_rate.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CH, bool AGPR, int THREADS>
__global__ __launch_bounds__(THREADS) void k(float* out, unsigned long long* cyc, int iters) {
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = (float)threadIdx.x;
    h8 a, b;
    for (int r = 0; r < 8; ++r) { a[r] = (_Float16)(0.001f * threadIdx.x); b[r] = (_Float16)(0.002f * r); }
    if (AGPR) for (int c = 0; c < CH; ++c) asm volatile("" : "+a"(acc[c]));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 12 / CH * CH; ++u) {
            if (AGPR) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[u % CH]) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[u % CH]) : "v"(a), "v"(b));
        }
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int CH, bool AGPR, int THREADS> void run(const char* name) {
    float* d; unsigned long long* c; hipMalloc(&d, 256 * THREADS * 4); hipMalloc(&c, 8);
    const int iters = 2000;
    k<CH, AGPR, THREADS><<<256, THREADS>>>(d, c, iters); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k<CH, AGPR, THREADS><<<256, THREADS>>>(d, c, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    const double per_wave = (double)iters * (12 / CH * CH), waves_per_simd = THREADS / 256.0;
    // one workgroup per CU (256 CUs): MFMAs per SIMD = per_wave * waves_per_simd; ns per MFMA of a SIMD's pipe:
    printf("%-58s %6.1f s_memtime ticks per MFMA of one wave; %6.2f ns per MFMA per SIMD by events = %.1f TFLOP/s chip-wide\n", name,
           (double)h / per_wave, ms * 1e6 / (per_wave * waves_per_simd), per_wave * waves_per_simd * 1024 * 32768.0 / (ms * 1e-3) / 1e12);
    hipFree(d); hipFree(c);
}
int main() {
    run<4, false, 256>("4 chains, VGPR accumulators, 1 wave/SIMD");
    run<4, true, 256>("4 chains, AGPR accumulators, 1 wave/SIMD");
    run<2, false, 256>("2 chains, VGPR accumulators, 1 wave/SIMD");
    run<1, false, 256>("1 chain (dependent), VGPR accumulators, 1 wave/SIMD");
    run<4, false, 512>("4 chains, VGPR accumulators, 2 waves/SIMD (per wave)");
    run<4, true, 512>("4 chains, AGPR accumulators, 2 waves/SIMD (per wave)");
    return 0;
}
