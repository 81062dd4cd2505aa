// Row N1: the optimizer step of the reference's training loop -- torch.optim.Adam(lr, eps=1e-8, weight_decay) as
// utils/__init__.py:45-47 (get_optimizer) builds it for train.py:140-146 -- on ONE flat buffer per tensor kind.
// Bound: HBM (28 B per parameter: read p, g, m, v; write p, m, v); 2.3 M parameters -> one launch of a few microseconds
// instead of ~10 multi-tensor launches (eager) or ~200 per-parameter scalar kernels (capturable torch Adam inside a
// hipGraph).  Step count and learning rate are device scalars, so the launch pair is capturable.
#include <hip/hip_runtime.h>
#include <cstdint>

#include "../../include/nsff_render.h"
#include "nsff_common.h"

namespace {

// state[0] = step count t (as float, exact up to 2^24), state[1] = lr / (1 - beta1^t), state[2] = sqrt(1 - beta2^t)
__global__ void adam_tick_kernel(float* state, const float* lr, double beta1, double beta2) {
    const float t = state[0] + 1.0f;
    state[0] = t;
    // double on purpose: torch evaluates the bias corrections in Python floats
    const double bc1 = 1.0 - pow(beta1, (double)t), bc2 = 1.0 - pow(beta2, (double)t);
    state[1] = (float)((double)*lr / bc1);
    state[2] = (float)sqrt(bc2);
}

__global__ __launch_bounds__(256) void adam_step_kernel(float4* __restrict__ p, const float4* __restrict__ g, float4* __restrict__ m,
                                                         float4* __restrict__ v, long long n4, const float* __restrict__ state,
                                                         float w1, float beta2, float w2, float eps, float wd) {
    const float step_size = state[1], sqrt_bc2 = state[2];             // w1 = 1 - beta1, w2 = 1 - beta2 (rounded from double)
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float4 pp = p[i], mm = m[i], vv = v[i];
        const float4 gg = g[i];
        float* P = &pp.x; float* M = &mm.x; float* V = &vv.x; const float* G = &gg.x;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float gr = G[c] + wd * P[c];                         // L2 penalty folded into the gradient (torch Adam)
            M[c] = M[c] + w1 * (gr - M[c]);                           // exp_avg.lerp_(grad, 1 - beta1)
            V[c] = V[c] * beta2 + w2 * gr * gr;                       // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
            const float denom = sqrtf(V[c]) / sqrt_bc2 + eps;
            P[c] = P[c] - step_size * (M[c] / denom);
        }
        p[i] = pp; m[i] = mm; v[i] = vv;
    }
}

}  // namespace

extern "C" int nsff_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float* state,
                              const float* lr, double beta1, double beta2, double eps, double weight_decay, void* stream) {
    if (!state || !lr || (n > 0 && (!param || !grad || !exp_avg || !exp_avg_sq))) return NSFF_ERR_NULL;
    if (n < 0 || (n & 3)) return NSFF_ERR_INVALID;
    if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return NSFF_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, st, state, lr, beta1, beta2);
    if (n > 0) {
        const long long n4 = n / 4;
        const long long blocks = std::min<long long>((n4 + 255) / 256, 2048);
        hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<float4*>(param),
                           reinterpret_cast<const float4*>(grad), reinterpret_cast<float4*>(exp_avg),
                           reinterpret_cast<float4*>(exp_avg_sq), n4, state, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2),
                           (float)eps, (float)weight_decay);
    }
    return nsff_launch_status();
}
