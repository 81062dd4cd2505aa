// Probe: what does ds_read_b64_tr_b16 hand each lane?  LDS image s[pt][col] = 64 * pt + col (row stride 264 halfs).
// Lane i of a 16-lane group supplies the address of 4 consecutive columns of point (i >> 2).
//   hipcc --offload-arch=gfx950 -O2 tr_read_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __fp16 fp4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__global__ void k(float* out) {
    __shared__ _Float16 s[16 * 264];
    for (int i = threadIdx.x; i < 16 * 264; i += 64) s[i] = (_Float16)(float)(64 * (i / 264) + (i % 264));
    __syncthreads();
    const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
    const _Float16* p = s + ((i >> 2) + 8 * (g >> 1)) * 264 + 4 * (i & 3) + 16 * (g & 1);
    fp4 r = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp4*)p);
    h4 v = __builtin_bit_cast(h4, r);
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (float)v[j];
}
int main() {
    float* d; hipMalloc(&d, 256 * 4);
    k<<<1, 64>>>(d);
    float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int g = l >> 4, c = l & 15;
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) {
            const int v = (int)h[l * 4 + j];
            printf("  (pt %2d col %2d)", v / 64, v % 64);
            if (v / 64 != j + 8 * (g >> 1) || v % 64 != c + 16 * (g & 1)) ++bad;      // expected: element j = point j, column = lane's
        }
        printf("\n");
    }
    printf("expected layout (elem j = point j of the group's four, column = 16 * (g & 1) + lane %% 16): %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
    return 0;
}
