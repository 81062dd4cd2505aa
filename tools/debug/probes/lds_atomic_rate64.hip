// Probe 2: 64-bit LDS atomics (ds_add_f64, ds_add_u64) and the float add as a compare-and-swap loop, lane-linear addresses.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ double s[256 * 8];
    for (int i = threadIdx.x; i < 256 * 8; i += 256) s[i] = 0.0;
    __syncthreads();
    double* p = s + threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            if (MODE == 0) atomicAdd(p + 256 * c, 1.0);
            else if (MODE == 1) atomicAdd(reinterpret_cast<unsigned long long*>(p + 256 * c), 1ull);
            else if (MODE == 2) unsafeAtomicAdd(reinterpret_cast<float*>(p + 256 * c), 1.0f);
            else __hip_atomic_fetch_add(reinterpret_cast<float*>(p + 256 * c), 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (float)s[5];
}
template <int MODE> void run(const char* name) {
    float* d; hipMalloc(&d, 4096 * 4);
    const int iters = 2000, blocks = 1024;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, iters); hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE><<<blocks, 256>>>(d, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)blocks * 256 * iters * 5;
    printf("%-44s %8.3f ms  %8.1f G lane-ops/s  = %.2f lane-ops per clock per CU\n", name, ms, ops / ms / 1e6, ops / (ms * 1e-3) / 256 / 2.1e9);
    hipFree(d);
}
int main() {
    run<0>("atomicAdd(double) in LDS");
    run<1>("atomicAdd(unsigned long long) in LDS");
    run<2>("unsafeAtomicAdd(float) in LDS");
    run<3>("__hip_atomic_fetch_add(float, workgroup scope)");
    return 0;
}
