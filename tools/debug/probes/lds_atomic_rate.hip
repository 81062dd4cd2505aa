// Probe: throughput of LDS atomics per CU -- ds_add_f32 vs ds_add_u32 vs plain ds_write, conflict-free addresses
// (lane-linear) and the splat accumulator's pattern (stride 5 floats).   hipcc --offload-arch=gfx950 -O2 lds_atomic_rate.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE, int STRIDE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ float s[256 * 8 * 5];
    for (int i = threadIdx.x; i < 256 * 8 * 5; i += 256) s[i] = 0.f;
    __syncthreads();
    float* p = s + (threadIdx.x * STRIDE) % (256 * 8 * 5 - 64);
    unsigned* q = reinterpret_cast<unsigned*>(p);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            if (MODE == 0) atomicAdd(p + c, 1.0f);
            else if (MODE == 1) atomicAdd(q + c, 1u);
            else p[c] = (float)i;
        }
        p += 5 * 64 * (i & 1) - 5 * 32;          // wander a little (stays inside the array)
        q = reinterpret_cast<unsigned*>(p);
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = s[5];
}
template <int MODE, int STRIDE> void run(const char* name) {
    float* d; hipMalloc(&d, 4096 * 4);
    const int iters = 2000, blocks = 1024;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, STRIDE><<<blocks, 256>>>(d, iters); hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE, STRIDE><<<blocks, 256>>>(d, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)blocks * 256 * iters * 5;
    printf("%-34s %8.3f ms  %8.1f G lane-ops/s  = %.2f lane-ops per clock per CU (256 CUs, 2.1 GHz)\n", name, ms, ops / ms / 1e6,
           ops / (ms * 1e-3) / 256 / 2.1e9);
    hipFree(d);
}
int main() {
    run<0, 1>("ds_add_f32, lane-linear");
    run<1, 1>("ds_add_u32, lane-linear");
    run<2, 1>("ds_write_b32, lane-linear");
    run<0, 5>("ds_add_f32, stride 5 floats");
    run<1, 5>("ds_add_u32, stride 5 floats");
    run<0, 40>("ds_add_f32, stride 40 floats");
    return 0;
}
