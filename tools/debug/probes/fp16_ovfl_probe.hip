// Probe: which fp16-producing instructions honour MODE.FP16_OVFL (bit 23: "an overflowed FP16 VALU result is clamped to +-MAX_FP16")
// on gfx950?  hipcc --offload-arch=gfx950 -O2 fp16_ovfl_probe.hip -o /tmp/ovfl_probe && /tmp/ovfl_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out, float big) {
    unsigned a, b, c, d, e;
    float nbig = -big;
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1\n\t"
                 "s_nop 4\n\t"
                 "v_cvt_pk_f16_f32 %0, %5, %6\n\t"          // gfx950's packed round-to-nearest conversion
                 "v_cvt_f16_f32 %1, %5\n\t"                 // the scalar one
                 "v_cvt_pkrtz_f16_f32 %2, %5, %6\n\t"       // round toward zero: never overflows by construction
                 "v_mov_b32 %3, 0x7bff7bff\n\t"
                 "v_pk_add_f16 %3, %3, %3\n\t"              // 65504 + 65504
                 "v_mov_b32 %4, 0x7bff7bff\n\t"
                 "v_pk_mul_f16 %4, %4, %4\n\t"
                 "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 0\n\t"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d), "=&v"(e) : "v"(big), "v"(nbig));
    out[0] = a; out[1] = b; out[2] = c; out[3] = d; out[4] = e;
    unsigned a2, b2;
    asm volatile("v_cvt_pk_f16_f32 %0, %2, %3\n\tv_cvt_f16_f32 %1, %2\n\t" : "=&v"(a2), "=&v"(b2) : "v"(big), "v"(nbig));
    out[5] = a2; out[6] = b2;
}
int main() {
    unsigned* d; hipMalloc(&d, 64);
    k<<<1, 1>>>(d, 1.0e6f);
    unsigned h[8]; hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
    printf("FP16_OVFL = 1:  v_cvt_pk_f16_f32(1e6, -1e6) = %08x   v_cvt_f16_f32(1e6) = %04x   v_cvt_pkrtz = %08x   pk_add(max,max) = %08x   pk_mul(max,max) = %08x\n",
           h[0], h[1] & 0xffff, h[2], h[3], h[4]);
    printf("FP16_OVFL = 0:  v_cvt_pk_f16_f32(1e6, -1e6) = %08x   v_cvt_f16_f32(1e6) = %04x      (7c00 = inf, 7bff = 65504)\n", h[5], h[6] & 0xffff);
    return 0;
}
