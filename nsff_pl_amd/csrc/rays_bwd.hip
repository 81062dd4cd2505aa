// Backward of the sigma->alpha compositing (training): gradients of the per-ray / per-sample
// outputs of composite_kernel (rays.hip; reference models/rendering.py:122-140,200-298) w.r.t. the raw field
// records of the main pass and of the two flow-warped re-queries.
//
// One wavefront per ray, lanes over samples (chunks of 64).  Sweep 1 walks the chunks front to back and leaves the
// four exclusive transmittance products (blend, static-only, fw-warp, bw-warp) in a scratch buffer; sweep 2 walks
// them back to front: every product chain T_i = prod_{j<i} om_j is differentiated with the exact reverse recurrence
//      R_j = dT_{j+1} + om_{j+1} R_{j+1},   d om_j = T_j R_j
// (a suffix scan of affine maps across the wave, carried between chunks) -- no division, so saturated samples
// (om = 0) need no special case.  Memory-bound: ~250 B per sample.
#include "nsff_common.h"

namespace {

constexpr int WAVES_PER_BLOCK = 4;
constexpr int RS = NSFF_RAW_STRIDE;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ float wave_scan_mul(float v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float u = __shfl_up(v, off);
        if (lane >= off) v *= u;
    }
    return v;
}
__device__ __forceinline__ float softplus(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoidf(float x) { return x > 20.f ? 1.f : 1.f / (1.f + expf(-x)); }   // softplus'

// exclusive product of om over the lanes, continued from `carry` (updated to include this chunk)
__device__ __forceinline__ float excl_prod(float om, float& carry, int lane) {
    const float inc = wave_scan_mul(om, lane);
    float exc = __shfl_up(inc, 1);
    if (lane == 0) exc = 1.f;
    const float T = carry * exc;
    carry *= __shfl(inc, 63);
    return T;
}

// R_j = c_j + m_j R_{j+1} over the lanes (R beyond lane 63 = carry); returns R of this lane, carry := R of lane 0
__device__ __forceinline__ float rev_affine(float m, float c, float& carry, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float m2 = __shfl_down(m, off), c2 = __shfl_down(c, off);
        if (lane + off < 64) { c = fmaf(m, c2, c); m *= m2; }
    }
    const float R = fmaf(m, carry, c);
    carry = __shfl(R, 0);
    return R;
}

struct Fwd {           // forward quantities of one sample, recomputed from the raw records
    float z, d_s, d_t;
    float xs, sig_s, al_s, xt, sig_t, al_t, alpha;
    float rgb_s[3], rgb_t[3];
    float xw[2], al_w[2], rgb_w[2][3];     // warps: 0 = fw, 1 = bw
};

__device__ __forceinline__ void load_fwd(Fwd& f, const NsffCompositeBwdArgs& a, long long idx, bool last, bool tr, bool warps) {
    f.z = a.zs[idx];
    const float dz = last ? 0.f : a.zs[idx + 1] - f.z;
    f.d_s = last ? 100.f : dz;
    f.d_t = last ? 1e-3f : dz;
    const float* rec = a.raw + idx * RS;
    f.xs = rec[3] + (a.noise_static ? a.noise_static[idx] * a.noise_std : 0.f);
    f.sig_s = softplus(f.xs);
    f.al_s = 1.f - expf(-f.d_s * f.sig_s);
    f.rgb_s[0] = rec[0]; f.rgb_s[1] = rec[1]; f.rgb_s[2] = rec[2];
    f.xt = 0.f; f.sig_t = 0.f; f.al_t = 0.f; f.alpha = f.al_s;
    f.rgb_t[0] = f.rgb_t[1] = f.rgb_t[2] = 0.f;
    if (tr) {
        f.xt = rec[7] + (a.noise_transient ? a.noise_transient[idx] * a.noise_std : 0.f);
        f.sig_t = softplus(f.xt);
        f.al_t = 1.f - expf(-f.d_t * f.sig_t);
        f.alpha = 1.f - (1.f - f.al_s) * (1.f - f.al_t);
        f.rgb_t[0] = rec[4]; f.rgb_t[1] = rec[5]; f.rgb_t[2] = rec[6];
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        f.xw[k] = 0.f; f.al_w[k] = 0.f;
        f.rgb_w[k][0] = f.rgb_w[k][1] = f.rgb_w[k][2] = 0.f;
        if (warps) {
            const float* rw = (k == 0 ? a.raw_fw : a.raw_bw) + idx * RS;
            const float* nz = k == 0 ? a.noise_fw : a.noise_bw;
            f.xw[k] = rw[7] + (nz ? nz[idx] * a.noise_std : 0.f);
            f.al_w[k] = 1.f - expf(-f.d_t * softplus(f.xw[k]));
            f.rgb_w[k][0] = rw[4]; f.rgb_w[k][1] = rw[5]; f.rgb_w[k][2] = rw[6];
        }
    }
}

__device__ __forceinline__ float ld(const float* p, long long i) { return p ? p[i] : 0.f; }

__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void composite_bwd_kernel(const NsffCompositeBwdArgs a) {
    const int lane = threadIdx.x & 63;
    const long long ray = (long long)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (ray >= a.n_rays) return;
    const int S = a.n_samples;
    const bool tr = a.has_transient != 0;
    const bool warps = tr && a.flow_mode >= 2;
    const bool flows = tr && a.flow_mode >= 1;
    const int n_chunks = (S + 63) / 64;

    // ---- sweep 1: the exclusive products, front to back ----
    {
        float cT = 1.f, cTs = 1.f, cTw0 = 1.f, cTw1 = 1.f;
        for (int ch = 0; ch < n_chunks; ++ch) {
            const int i = ch * 64 + lane;
            const bool on = i < S;
            const long long idx = ray * S + (on ? i : S - 1);
            Fwd f;
            load_fwd(f, a, idx, i >= S - 1, tr, warps);
            const float T = excl_prod(on ? 1.f - f.alpha : 1.f, cT, lane);
            float Ts = 1.f, Tw0 = 1.f, Tw1 = 1.f;
            if (tr) Ts = excl_prod(on ? 1.f - f.al_s : 1.f, cTs, lane);
            if (warps) {
                Tw0 = excl_prod(on ? (1.f - f.al_s) * (1.f - f.al_w[0]) : 1.f, cTw0, lane);
                Tw1 = excl_prod(on ? (1.f - f.al_s) * (1.f - f.al_w[1]) : 1.f, cTw1, lane);
            }
            if (on) *reinterpret_cast<float4*>(a.scratch + idx * 4) = make_float4(T, Ts, Tw0, Tw1);
        }
    }

    // ---- per-ray output gradients ----
    const float g_depth = ld(a.g_depth, ray), g_ta = ld(a.g_transient_alpha, ray), g_sod = ld(a.g_so_depth, ray);
    float g_rgb[3], g_trgb[3], g_sorgb[3], g_xyz[3], g_ffw[3], g_fbw[3], g_rw[2][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        g_rgb[c] = ld(a.g_rgb, ray * 3 + c); g_trgb[c] = ld(a.g_transient_rgb, ray * 3 + c);
        g_sorgb[c] = ld(a.g_so_rgb, ray * 3 + c); g_xyz[c] = ld(a.g_xyz_exp, ray * 3 + c);
        g_ffw[c] = ld(a.g_flow_fw_exp, ray * 3 + c); g_fbw[c] = ld(a.g_flow_bw_exp, ray * 3 + c);
        g_rw[0][c] = ld(a.g_rgb_fw, ray * 3 + c); g_rw[1][c] = ld(a.g_rgb_bw, ray * 3 + c);
    }
    const float g_trgb_sum = g_trgb[0] + g_trgb[1] + g_trgb[2];

    // ---- sweep 2: back to front ----
    float cR = 0.f, cRs = 0.f, cRw[2] = {0.f, 0.f};                 // R of the first sample of the chunk behind
    float nx_dT = 0.f, nx_om = 0.f, nx_dTs = 0.f, nx_oms = 0.f;    // dT / om of that sample (lane 63's "next")
    float nx_dTw[2] = {0.f, 0.f}, nx_omw[2] = {0.f, 0.f};
    for (int ch = n_chunks - 1; ch >= 0; --ch) {
        const int i = ch * 64 + lane;
        const bool on = i < S;
        const long long idx = ray * S + (on ? i : S - 1);
        Fwd f;
        load_fwd(f, a, idx, i >= S - 1, tr, warps);
        const float4 Ts4 = *reinterpret_cast<const float4*>(a.scratch + idx * 4);
        const float T = Ts4.x, Ts = Ts4.y;
        const float Tw[2] = {Ts4.z, Ts4.w};

        float d_rgb_s[3] = {0, 0, 0}, d_rgb_t[3] = {0, 0, 0}, d_al_s = 0.f, d_al_t = 0.f;
        float xyzv[3] = {0, 0, 0}, ffw[3] = {0, 0, 0}, fbw[3] = {0, 0, 0};
        if (flows && on) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { xyzv[c] = a.xyz[idx * 3 + c]; ffw[c] = a.f_fw[idx * 3 + c]; fbw[c] = a.f_bw[idx * 3 + c]; }
        }
        // -- blend chain --
        float dw = ld(a.g_weights, idx) + g_depth * f.z;
        if (flows) {
#pragma unroll
            for (int c = 0; c < 3; ++c) dw += g_xyz[c] * xyzv[c] + g_ffw[c] * ffw[c] + g_fbw[c] * fbw[c];
        }
        float dT, d_alpha;
        const float w = f.alpha * T, w_s = f.al_s * T, w_t = f.al_t * T;
        if (tr) {
            float dws = ld(a.g_static_weights, idx), dwt = ld(a.g_transient_weights, idx) + g_ta - 0.8f * g_trgb_sum;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                dws += g_rgb[c] * f.rgb_s[c];
                dwt += (g_rgb[c] + g_trgb[c]) * f.rgb_t[c];
                d_rgb_s[c] += g_rgb[c] * w_s;
                d_rgb_t[c] += (g_rgb[c] + g_trgb[c]) * w_t;
            }
            dT = dw * f.alpha + dws * f.al_s + dwt * f.al_t;
            d_alpha = dw * T;
            d_al_s += dws * T;
            d_al_t += dwt * T;
        } else {
            dw += ld(a.g_static_weights, idx);
#pragma unroll
            for (int c = 0; c < 3; ++c) { dw += g_rgb[c] * f.rgb_s[c]; d_rgb_s[c] += g_rgb[c] * w; }
            dT = dw * f.alpha;
            d_alpha = dw * T;
        }
        if (!on) dT = 0.f;
        const float om = on ? 1.f - f.alpha : 1.f;
        {
            float c_ = __shfl_down(dT, 1), m_ = __shfl_down(om, 1);
            if (lane == 63) { c_ = nx_dT; m_ = nx_om; }
            if (i >= S - 1) { c_ = 0.f; m_ = on ? 0.f : 1.f; }
            const float dT0 = __shfl(dT, 0), om0 = __shfl(om, 0);
            const float R = rev_affine(m_, c_, cR, lane);
            nx_dT = dT0; nx_om = om0;
            d_alpha -= T * R;
        }
        if (tr) { d_al_s += d_alpha * (1.f - f.al_t); d_al_t += d_alpha * (1.f - f.al_s); }
        else d_al_s += d_alpha;

        if (tr) {
            // -- static field alone --
            float dso = g_sod * f.z;
#pragma unroll
            for (int c = 0; c < 3; ++c) { dso += g_sorgb[c] * f.rgb_s[c]; d_rgb_s[c] += g_sorgb[c] * f.al_s * Ts; }
            float dTs = on ? dso * f.al_s : 0.f;
            d_al_s += dso * Ts;
            const float oms = on ? 1.f - f.al_s : 1.f;
            float c_ = __shfl_down(dTs, 1), m_ = __shfl_down(oms, 1);
            if (lane == 63) { c_ = nx_dTs; m_ = nx_oms; }
            if (i >= S - 1) { c_ = 0.f; m_ = on ? 0.f : 1.f; }
            const float d0 = __shfl(dTs, 0), o0 = __shfl(oms, 0);
            const float R = rev_affine(m_, c_, cRs, lane);
            nx_dTs = d0; nx_oms = o0;
            d_al_s -= Ts * R;
        }
        float d_xw[2] = {0.f, 0.f}, d_rgb_w[2][3] = {{0, 0, 0}, {0, 0, 0}};
        if (warps) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                float A = 0.f, B = 0.f;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    A += g_rw[k][c] * f.rgb_s[c]; B += g_rw[k][c] * f.rgb_w[k][c];
                    d_rgb_s[c] += g_rw[k][c] * f.al_s * Tw[k];
                    d_rgb_w[k][c] = g_rw[k][c] * f.al_w[k] * Tw[k];
                }
                float dTw = on ? A * f.al_s + B * f.al_w[k] : 0.f;
                d_al_s += A * Tw[k];
                float d_al_w = B * Tw[k];
                const float omw = on ? (1.f - f.al_s) * (1.f - f.al_w[k]) : 1.f;
                float c_ = __shfl_down(dTw, 1), m_ = __shfl_down(omw, 1);
                if (lane == 63) { c_ = nx_dTw[k]; m_ = nx_omw[k]; }
                if (i >= S - 1) { c_ = 0.f; m_ = on ? 0.f : 1.f; }
                const float d0 = __shfl(dTw, 0), o0 = __shfl(omw, 0);
                const float R = rev_affine(m_, c_, cRw[k], lane);
                nx_dTw[k] = d0; nx_omw[k] = o0;
                const float d_al = -Tw[k] * R;                      // al = 1 - (1-al_s)(1-al_w), om_w = 1 - al
                d_al_s += d_al * (1.f - f.al_w[k]);
                d_al_w += d_al * (1.f - f.al_s);
                d_xw[k] = d_al_w * f.d_t * (1.f - f.al_w[k]) * sigmoidf(f.xw[k]);
            }
        }
        if (!on) continue;
        // -- back through alpha = 1 - exp(-delta softplus(x)) --
        const float d_sig_s = ld(a.g_static_sigmas, idx) + d_al_s * f.d_s * (1.f - f.al_s);
        float* dr = a.d_raw + idx * RS;
        float4 o0 = make_float4(d_rgb_s[0], d_rgb_s[1], d_rgb_s[2], d_sig_s * sigmoidf(f.xs));
        float4 o1 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tr) {
            const float d_sig_t = ld(a.g_transient_sigmas, idx) + d_al_t * f.d_t * (1.f - f.al_t);
            o1 = make_float4(d_rgb_t[0], d_rgb_t[1], d_rgb_t[2], d_sig_t * sigmoidf(f.xt));
        }
        reinterpret_cast<float4*>(dr)[0] = o0;
        reinterpret_cast<float4*>(dr)[1] = o1;
        reinterpret_cast<float4*>(dr)[2] = make_float4(0.f, 0.f, 0.f, 0.f);
        reinterpret_cast<float4*>(dr)[3] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (flows) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (a.d_f_fw) a.d_f_fw[idx * 3 + c] = g_ffw[c] * w;
                if (a.d_f_bw) a.d_f_bw[idx * 3 + c] = g_fbw[c] * w;
            }
        }
        if (warps) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                float* dw_ = (k == 0 ? a.d_raw_fw : a.d_raw_bw) + idx * RS;
                reinterpret_cast<float4*>(dw_)[0] = make_float4(0.f, 0.f, 0.f, 0.f);
                reinterpret_cast<float4*>(dw_)[1] = make_float4(d_rgb_w[k][0], d_rgb_w[k][1], d_rgb_w[k][2], d_xw[k]);
                reinterpret_cast<float4*>(dw_)[2] = make_float4(0.f, 0.f, 0.f, 0.f);
                reinterpret_cast<float4*>(dw_)[3] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
}

}  // namespace

// Gradient of the flow glue (see the header): thread = point, the record leaves as four 16-byte stores (accumulate: the named
// columns are read-modify-written).
__global__ __launch_bounds__(256) void flow_grad_kernel(const NsffFlowGradArgs a) {
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= a.n_points) return;
    const float m = a.zs[p] > a.z_far ? 0.f : 1.f;
    float va[3] = {0.f, 0.f, 0.f}, vb[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (a.col_a >= 0 && a.g_a[k] != nullptr) { const float* g = a.g_a[k] + 3 * p; va[0] += g[0]; va[1] += g[1]; va[2] += g[2]; }
        if (a.col_b >= 0 && a.g_b[k] != nullptr) { const float* g = a.g_b[k] + 3 * p; vb[0] += g[0]; vb[1] += g[1]; vb[2] += g[2]; }
    }
    float* o = a.out + 16 * p;
    if (a.accumulate) {
        if (a.col_a >= 0) for (int c = 0; c < 3; ++c) o[a.col_a + c] += m * va[c];
        if (a.col_b >= 0) for (int c = 0; c < 3; ++c) o[a.col_b + c] += m * vb[c];
    } else {
        float rec[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            float v = 0.f;
            if (a.col_a >= 0 && c >= a.col_a && c < a.col_a + 3) v = m * va[c - a.col_a];
            if (a.col_b >= 0 && c >= a.col_b && c < a.col_b + 3) v = m * vb[c - a.col_b];
            rec[c] = v;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            reinterpret_cast<float4*>(o)[q] = make_float4(rec[4 * q], rec[4 * q + 1], rec[4 * q + 2], rec[4 * q + 3]);
    }
}

extern "C" int nsff_flow_grad(const NsffFlowGradArgs* args, void* stream) {
    if (!args) return NSFF_ERR_NULL;
    const NsffFlowGradArgs& a = *args;
    if (a.n_points < 0 || a.col_a > 13 || a.col_b > 13 || (a.col_a < 0 && a.col_b < 0)) return NSFF_ERR_INVALID;
    if (a.col_a >= 0 && a.col_b >= 0 && a.col_a + 3 > a.col_b && a.col_b + 3 > a.col_a) return NSFF_ERR_INVALID;     // overlapping groups
    if (a.n_points == 0) return NSFF_OK;
    if (!a.zs || !a.out) return NSFF_ERR_NULL;
    if ((uintptr_t)a.out & 15) return NSFF_ERR_ALIGN;
    const unsigned blocks = (unsigned)((a.n_points + 255) / 256);
    hipLaunchKernelGGL(flow_grad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    return nsff_launch_status();
}

extern "C" int nsff_composite_backward(const NsffCompositeBwdArgs* args, void* stream) {
    if (!args) return NSFF_ERR_NULL;
    const NsffCompositeBwdArgs& a = *args;
    if (a.n_rays < 0 || a.n_samples < 1 || a.flow_mode < 0 || a.flow_mode > 2) return NSFF_ERR_INVALID;
    if (a.flow_mode && !a.has_transient) return NSFF_ERR_INVALID;
    if (a.n_rays == 0) return NSFF_OK;
    if (!a.raw || !a.zs || !a.scratch || !a.d_raw) return NSFF_ERR_NULL;
    if (a.flow_mode >= 1 && (!a.xyz || !a.f_fw || !a.f_bw)) return NSFF_ERR_NULL;
    if (a.flow_mode == 2 && (!a.raw_fw || !a.raw_bw || !a.d_raw_fw || !a.d_raw_bw)) return NSFF_ERR_NULL;
    if (((uintptr_t)a.scratch | (uintptr_t)a.d_raw | (uintptr_t)a.d_raw_fw | (uintptr_t)a.d_raw_bw) & 15) return NSFF_ERR_ALIGN;
    const unsigned blocks = (unsigned)((a.n_rays + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK);
    hipLaunchKernelGGL(composite_bwd_kernel, dim3(blocks), dim3(64 * WAVES_PER_BLOCK), 0, (hipStream_t)stream, a);
    return nsff_launch_status();
}
