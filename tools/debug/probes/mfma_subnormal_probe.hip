// Does v_mfma_f32_32x32x16_f16 read fp16 SUBNORMAL inputs, or as zero?  (and do v_pk_mul_f16 / v_cvt_f16_f32 keep them?)
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_sub tools/debug/probes/mfma_subnormal_probe.hip && /tmp/mfma_sub
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void probe(const float* in, float* out) {
    const float a = in[0], b = in[1];
    h8 av, bv;
    for (int t = 0; t < 8; ++t) { av[t] = (_Float16)a; bv[t] = (_Float16)b; }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
    if (threadIdx.x == 0) {
        out[0] = acc[0];
        out[1] = (float)av[0];
        _Float16 p = av[0] * (_Float16)0.5f;
        out[2] = (float)p;
    }
}
int main() {
    float *in, *out;
    hipMalloc(&in, 8); hipMalloc(&out, 16);
    const float cases[][2] = {{9.5367431640625e-07f, 1024.f}, {3.0517578125e-05f, 1024.f}, {6.103515625e-05f, 1024.f}, {1024.f, 9.5367431640625e-07f}, {5.9604644775390625e-08f, 65504.f}};
    for (auto& c : cases) {
        hipMemcpy(in, c, 8, hipMemcpyHostToDevice);
        probe<<<1, 64>>>(in, out);
        float h[3];
        hipMemcpy(h, out, 12, hipMemcpyDeviceToHost);
        printf("a %.10g (fp16 %s) b %.10g : mfma acc %.10g  expected %.10g   fp16(a) reads back %.10g   a * 0.5 in fp16 %.10g\n", c[0],
               (c[0] < 6.103515625e-05f || c[1] < 6.103515625e-05f) ? "subnormal" : "normal", c[1], h[0], 16.0 * (double)c[0] * c[1], h[1], h[2]);
    }
    return 0;
}
