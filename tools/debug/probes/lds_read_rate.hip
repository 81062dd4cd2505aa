// Probe: LDS read bandwidth per CU for the GEMM's B-operand pattern (ds_read_b128, 528-byte rows, lane & 31 -> row, lane >> 5 -> +16 B),
// no MFMAs: 4 or 8 waves per CU, 8 or 16 reads in flight per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
template <int THREADS, int BATCH>
__global__ __launch_bounds__(THREADS) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) _Float16 sX[2 * 128 * 264];
    for (int i = threadIdx.x; i < 2 * 128 * 264; i += THREADS) sX[i] = (_Float16)(0.001f * (i % 97));
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const _Float16* bh = sX + (lane & 31) * 264 + 8 * (lane >> 5);
    h8 acc = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
        h8 v[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) v[u] = *reinterpret_cast<const h8*>(bh + (u & 7) * 32 * 264 / 2 * 1 + ((i + u) & 15) * 16);
#pragma unroll
        for (int u = 0; u < BATCH; ++u) acc += v[u];
    }
    out[blockIdx.x * THREADS + threadIdx.x] = (float)acc[0];
}
template <int THREADS, int BATCH> void run(const char* name) {
    float* d; hipMalloc(&d, 256 * THREADS * 4);
    const int iters = 4000;
    k<THREADS, BATCH><<<256, THREADS>>>(d, iters); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k<THREADS, BATCH><<<256, THREADS>>>(d, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes_per_cu = (double)iters * BATCH * (THREADS / 64) * 1024;
    printf("%-40s %7.1f GB/s per CU = %5.1f B/clk at 2.05 GHz\n", name, bytes_per_cu / (ms * 1e-3) / 1e9, bytes_per_cu / (ms * 1e-3) / 2.05e9);
    hipFree(d);
}
int main() {
    run<256, 8>("4 waves, 8 reads in flight");
    run<512, 8>("8 waves, 8 reads in flight");
    run<512, 16>("8 waves, 16 reads in flight");
    run<1024, 8>("16 waves, 8 reads in flight");
    return 0;
}
