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

// index of the parameter tensor that owns flat element e: seg_start[k] <= e < seg_start[k + 1] (seg_start ascending, n_seg + 1 entries)
__device__ __forceinline__ int seg_of(const long long* __restrict__ seg_start, int n_seg, long long e) {
    int lo = 0, hi = n_seg;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (seg_start[mid] <= e) lo = mid; else hi = mid; }
    return lo;
}

// used[k] = 1 when any gradient element of parameter tensor k is non-zero this step (torch.optim.Adam skips a parameter whose
// .grad is None -- train.py's unused heads; on the flat buffer such a parameter's gradient slice is exactly zero)
__global__ __launch_bounds__(256) void adam_used_kernel(const float4* __restrict__ g, long long n4, const long long* __restrict__ seg_start,
                                                         int n_seg, int* __restrict__ used) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 gg = g[i];
        const float* G = &gg.x;
        if (gg.x == 0.f && gg.y == 0.f && gg.z == 0.f && gg.w == 0.f) continue;
        const int s0 = seg_of(seg_start, n_seg, 4 * i), s3 = seg_of(seg_start, n_seg, 4 * i + 3);
        if (s0 == s3) { used[s0] = 1; continue; }
#pragma unroll
        for (int c = 0; c < 4; ++c) if (G[c] != 0.f) used[seg_of(seg_start, n_seg, 4 * i + c)] = 1;
    }
}

// seg_start / used (both or neither): elements of a parameter tensor with used[k] == 0 are left untouched (value and moments)
__global__ __launch_bounds__(256) void adam_step_kernel(float4* __restrict__ p, const float4* __restrict__ g, float4* __restrict__ m,
                                                         float4* __restrict__ v, long long n4, const float* __restrict__ state,
                                                         float w1, float beta2, float w2, float eps, float wd,
                                                         const long long* __restrict__ seg_start, int n_seg, const int* __restrict__ used) {
    const float step_size = state[1], sqrt_bc2 = state[2];             // w1 = 1 - beta1, w2 = 1 - beta2 (rounded from double)
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        unsigned live = 0xFu;                                          // bit c: element c of this float4 is updated
        if (used != nullptr) {
            const int s0 = seg_of(seg_start, n_seg, 4 * i), s3 = seg_of(seg_start, n_seg, 4 * i + 3);
            if (s0 == s3) live = used[s0] ? 0xFu : 0u;
            else {
                live = 0u;
#pragma unroll
                for (int c = 0; c < 4; ++c) if (used[seg_of(seg_start, n_seg, 4 * i + c)]) live |= 1u << c;
            }
            if (live == 0u) continue;
        }
        float4 pp = p[i], mm = m[i], vv = v[i];
        const float4 gg = g[i];
        float* P = &pp.x; float* M = &mm.x; float* V = &vv.x; const float* G = &gg.x;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (!((live >> c) & 1u)) continue;
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
    return nsff_adam_step_segments(param, grad, exp_avg, exp_avg_sq, n, state, lr, beta1, beta2, eps, weight_decay, nullptr, 0,
                                   nullptr, stream);
}

extern "C" int nsff_adam_step_segments(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float* state,
                                       const float* lr, double beta1, double beta2, double eps, double weight_decay,
                                       const int64_t* seg_start, int n_seg, int32_t* seg_used, void* stream) {
    if (!state || !lr || (n > 0 && (!param || !grad || !exp_avg || !exp_avg_sq))) return NSFF_ERR_NULL;
    if (n < 0 || (n & 3)) return NSFF_ERR_INVALID;
    if ((seg_start == nullptr) != (seg_used == nullptr) || (seg_start && n_seg <= 0)) return NSFF_ERR_INVALID;
    if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return NSFF_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, st, state, lr, beta1, beta2);
    if (n > 0) {
        const long long n4 = n / 4;
        const long long blocks = std::min<long long>((n4 + 255) / 256, 2048);
        if (seg_start) {
            hipError_t e = hipMemsetAsync(seg_used, 0, sizeof(int32_t) * (size_t)n_seg, st);
            if (e != hipSuccess) return nsff_hip_fail(e);
            hipLaunchKernelGGL(adam_used_kernel, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<const float4*>(grad), n4,
                               reinterpret_cast<const long long*>(seg_start), n_seg, seg_used);
        }
        hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<float4*>(param),
                           reinterpret_cast<const float4*>(grad), reinterpret_cast<float4*>(exp_avg),
                           reinterpret_cast<float4*>(exp_avg_sq), n4, state, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2),
                           (float)eps, (float)weight_decay, reinterpret_cast<const long long*>(seg_start), n_seg, seg_used);
    }
    return nsff_launch_status();
}
