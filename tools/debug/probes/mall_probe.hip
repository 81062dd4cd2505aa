// Probe (round 6): does a buffer that one kernel just WROTE come back faster when the next kernel reads it, as long
// as it is smaller than the 256 MB Infinity Cache (MALL)?  If it does, running a field node's backward in ray blocks whose
// pre-activation-gradient fragments fit the cache would take their re-read by the weight-gradient GEMM off the HBM.
//   hipcc --offload-arch=gfx950 -O2 mall_probe.hip -o /tmp/mall && /tmp/mall
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void wr(float4* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__global__ void rd(const float4* p, size_t n, float* out) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = p[i];
        acc += v.x + v.w;
    }
    if (acc == 12345.678f) *out = acc;
}
int main() {
    const size_t max_bytes = 2048ull << 20;
    float4* buf; float* out;
    hipMalloc(&buf, max_bytes); hipMalloc(&out, 4);
    float4* other; hipMalloc(&other, max_bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%10s %14s %14s %18s\n", "MB", "write TB/s", "read-after-write TB/s", "read cold TB/s (another 2 GB touched in between)");
    for (size_t mb : {16, 32, 64, 128, 192, 256, 384, 512, 1024, 2048}) {
        const size_t n = (mb << 20) / 16;
        float tw = 0, tr = 0, tc = 0;
        const int reps = 5;
        for (int r = 0; r < reps + 1; ++r) {
            float a, b, c;
            hipEventRecord(e0); wr<<<2048, 256>>>(buf, n); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&a, e0, e1);
            hipEventRecord(e0); rd<<<2048, 256>>>(buf, n, out); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&b, e0, e1);
            wr<<<2048, 256>>>(other, max_bytes / 16);                 // evict
            hipEventRecord(e0); rd<<<2048, 256>>>(buf, n, out); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&c, e0, e1);
            if (r) { tw += a; tr += b; tc += c; }
        }
        const double gb = (double)(mb << 20) / 1e12 * 1e3 * reps;      // TB per ms-sum
        printf("%10zu %14.2f %14.2f %18.2f\n", mb, gb / tw, gb / tr, gb / tc);
    }
    return 0;
}
