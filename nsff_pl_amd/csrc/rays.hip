// Per-ray stages of render_rays for gfx950: sample placement, inverse-CDF fine sampling
// with merge+sort, flow warping of the sample points, and sigma->alpha compositing.
//
// These stages are HBM/latency-bound (a few hundred bytes per sample, no reuse), so the
// design rules are the streaming ones: one 64-lane wavefront per ray, lane i owns sample
// i (+64, +128 ...) so every global access of the wave is a contiguous run, transmittance
// is an exclusive product scan over the wave with a carry between 64-sample chunks, and
// per-ray sums are butterfly reductions.  Compiled with -ffp-contract=off so that
// o + d*z, w*z ... round like the reference's separate torch multiply / add kernels.
//
// Reference: models/rendering.py:10-49 (sample_pdf), :98-140 (render_transient_warping),
// :187-188, :202-298 (inference), :314-324, :332-348, :359 (render_rays).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "nsff_common.h"

namespace {

constexpr int WAVES_PER_BLOCK = 4;
constexpr int RS = NSFF_RAW_STRIDE;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// Inclusive product scan across the 64 lanes of a wave.
__device__ __forceinline__ float wave_scan_mul(float v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float u = __shfl_up(v, off);
        if (lane >= off) v *= u;
    }
    return v;
}
__device__ __forceinline__ float wave_scan_add(float v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float u = __shfl_up(v, off);
        if (lane >= off) v += u;
    }
    return v;
}

// torch.nn.Softplus(beta=1, threshold=20)
__device__ __forceinline__ float softplus(float x) { return x > 20.f ? x : log1pf(expf(x)); }

// ---------------------------------------------------------------------------------
__global__ void coarse_samples_kernel(const float* __restrict__ rays, long long n_rays,
                                      const float* __restrict__ z_lin, int S, float perturb,
                                      const float* __restrict__ rnd, float* __restrict__ zs,
                                      float* __restrict__ xyz) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_rays * S) return;
    const long long n = idx / S;
    const int i = (int)(idx - n * S);
    float z = z_lin[i];
    if (perturb > 0.f) {
        const float lower = i > 0 ? 0.5f * (z_lin[i - 1] + z_lin[i]) : z_lin[0];
        const float upper = i < S - 1 ? 0.5f * (z_lin[i] + z_lin[i + 1]) : z_lin[S - 1];
        z = lower + (upper - lower) * (perturb * rnd[idx]);
    }
    zs[idx] = z;
    const float* r = rays + n * 6;
#pragma unroll
    for (int c = 0; c < 3; ++c) xyz[idx * 3 + c] = r[c] + r[3 + c] * z;
}

// ---------------------------------------------------------------------------------
// One ray's inverse-CDF draw.  `w` points at the M interior weights, `bins` at M+1 bin
// edges (LDS or global), `cdf` is an LDS scratch of M+1 floats owned by this wave.
// Every lane of the wave must call this (it contains block barriers).
__device__ __forceinline__ void sample_pdf_ray(const float* w, const float* bins, int M, float eps,
                                               const float* u, int n_imp, float* cdf,
                                               float* out_a, float* out_b, int lane, bool active) {
    // pdf = (w+eps) / sum(w+eps);  cdf = [0, cumsum(pdf)]
    float part = 0.f;
    for (int j = lane; j < M; j += 64) part += (active ? w[j] : 0.f) + eps;
    const float total = wave_sum(part);
    float carry = 0.f;
    if (lane == 0) cdf[0] = 0.f;
    for (int base = 0; base < M; base += 64) {
        const int j = base + lane;
        const float pdf = j < M ? ((active ? w[j] : 0.f) + eps) / total : 0.f;
        const float inc = wave_scan_add(pdf, lane) + carry;
        if (j < M) cdf[j + 1] = inc;
        carry = __shfl(inc, 63);
    }
    __syncthreads();
    for (int k = lane; k < n_imp; k += 64) {
        const float uk = active ? u[k] : 0.f;
        // searchsorted(cdf, u, right=True): number of entries <= u
        int lo = 0, hi = M + 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] <= uk) lo = mid + 1; else hi = mid;
        }
        const int below = max(lo - 1, 0), above = min(lo, M);
        const float cb = cdf[below], ca = cdf[above];
        const float bb = bins[below], ba = bins[above];
        float denom = ca - cb;
        if (denom < eps) denom = 1.f;
        const float s = bb + (uk - cb) / denom * (ba - bb);
        if (active) {
            if (out_a) out_a[k] = s;
            if (out_b) out_b[k] = s;
        }
    }
    __syncthreads();
}

struct FineArgs {
    const float* rays; long long n_rays; const float* z_lin; const float* zs_coarse;
    int S, n_imp;
    const float* w_static; const float* w_transient;
    const float* u_static; const float* u_transient; int u_per_ray;
    float* samples_static; float* samples_transient;
    float* zs_fine; float* xyz_fine;
};

// One ray per workgroup of four waves: the inverse-CDF draws of the static and the transient weights run side by side (wave 0 /
// wave 1, the arithmetic of either is one wave's as before: the same scan order, the same results), the merge sort ranks one
// element per thread, and a CU holds eight rays' workgroups at a time -- the one-wave-per-ray form of earlier rounds (four waves
// per CU, three elements' ranks per lane) took 24 us per 1024 rays, a sequence of exposed latencies.
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void fine_samples_kernel(const FineArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tid = threadIdx.x;
    constexpr int T = 64 * WAVES_PER_BLOCK;
    const long long ray = blockIdx.x;
    const int S = a.S, n_imp = a.n_imp;
    const int k_sets = a.w_transient ? 2 : 1;
    const int Sf = S + k_sets * n_imp;
    float* cdf = smem + wave * S;                              // S-1 used; one scratch per wave (sample_pdf_ray writes it unconditionally)
    float* bins = smem + WAVES_PER_BLOCK * S;                  // S-1 used: interval mid points (rendering.py:315)
    float* merged = bins + ((S + 3) & ~3);                     // Sf, 16-byte aligned
    for (int j = tid; j < S - 1; j += T) bins[j] = 0.5f * (a.z_lin[j] + a.z_lin[j + 1]);
    for (int j = tid; j < S; j += T) merged[j] = a.zs_coarse[ray * S + j];
    __syncthreads();
    {
        const bool mine = wave < k_sets;                       // (the other waves go through the function's barriers with nothing to do)
        const int set = mine ? wave : 0;
        const float* w = (set == 0 ? a.w_static : a.w_transient) + ray * S + 1;   // weights[:, 1:-1]
        const float* ub = set == 0 ? a.u_static : a.u_transient;
        const float* u = a.u_per_ray ? ub + ray * n_imp : ub;
        float* keep = set == 0 ? a.samples_static : a.samples_transient;
        sample_pdf_ray(w, bins, S - 2, 1e-5f, u, n_imp, cdf, merged + S + set * n_imp,
                       keep ? keep + ray * n_imp : nullptr, lane, mine);
    }
    // torch.sort(cat([zs, zs_static, zs_transient]))[0]: stable rank sort, one element per thread, four candidates per LDS read
    const float* r = a.rays + ray * 6;
    const float o0 = r[0], o1 = r[1], o2 = r[2], d0 = r[3], d1 = r[4], d2 = r[5];
    for (int i = tid; i < Sf; i += T) {
        const float v = merged[i];
        int rank = 0;
        int j = 0;
        for (; j + 4 <= Sf; j += 4) {
            const float4 o = *reinterpret_cast<const float4*>(merged + j);
            rank += (o.x < v || (o.x == v && j < i)) ? 1 : 0;
            rank += (o.y < v || (o.y == v && j + 1 < i)) ? 1 : 0;
            rank += (o.z < v || (o.z == v && j + 2 < i)) ? 1 : 0;
            rank += (o.w < v || (o.w == v && j + 3 < i)) ? 1 : 0;
        }
        for (; j < Sf; ++j) {
            const float o = merged[j];
            rank += (o < v || (o == v && j < i)) ? 1 : 0;
        }
        const long long dst = ray * Sf + rank;
        a.zs_fine[dst] = v;
        a.xyz_fine[dst * 3 + 0] = o0 + d0 * v;
        a.xyz_fine[dst * 3 + 1] = o1 + d1 * v;
        a.xyz_fine[dst * 3 + 2] = o2 + d2 * v;
    }
}

struct PdfArgs {
    const float* bins; const float* weights; long long n_rays; int M;
    const float* u; int n_imp; int u_per_ray; float eps; float* samples;
};

__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void sample_pdf_kernel(const PdfArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long ray_raw = (long long)blockIdx.x * WAVES_PER_BLOCK + wave;
    const bool active = ray_raw < a.n_rays;
    const long long ray = active ? ray_raw : a.n_rays - 1;
    float* cdf = smem + wave * (a.M + 1);
    const float* u = a.u_per_ray ? a.u + ray * a.n_imp : a.u;
    sample_pdf_ray(a.weights + ray * a.M, a.bins + ray * (a.M + 1), a.M, a.eps, u, a.n_imp, cdf,
                   active ? a.samples + ray * a.n_imp : nullptr, nullptr, lane, true);
}

// ---------------------------------------------------------------------------------
// Camera rays of a pinhole frame, straight to NDC (reference datasets/ray_utils.py:7-36
// get_ray_directions, :39-59 get_rays, :62-106 get_ndc_rays; called from monocular.py:268-276).
struct FrameRayArgs {
    float fx, fy, cx, cy;
    float c2w[12];           // row-major (3,4)
    int W;
    float near, shift_near;
    long long first, count;
    float* rays;             // (count, 6)
};

__global__ void frame_rays_kernel(const FrameRayArgs a) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.count) return;
    const long long pix = a.first + idx;
    const float i = (float)(pix % a.W), j = (float)(pix / a.W);      // no +0.5 pixel centring (ray_utils.py:26)
    const float dc[3] = {(i - a.cx) / a.fx, -(j - a.cy) / a.fy, -1.0f};
    float d[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) d[r] = dc[0] * a.c2w[4 * r + 0] + dc[1] * a.c2w[4 * r + 1] + dc[2] * a.c2w[4 * r + 2];
    const float nrm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
#pragma unroll
    for (int r = 0; r < 3; ++r) d[r] /= nrm;
    float o[3] = {a.c2w[3], a.c2w[7], a.c2w[11]};
    // shift the origin to the near plane, then project
    const float t = -(a.shift_near + o[2]) / d[2];
#pragma unroll
    for (int r = 0; r < 3; ++r) o[r] = o[r] + t * d[r];
    const float ox_oz = o[0] / o[2], oy_oz = o[1] / o[2];
    const float sx = -1.0f / (a.cx / a.fx), sy = -1.0f / (a.cy / a.fy);
    const float o2 = 1.0f + 2.0f * a.near / o[2];
    float* r = a.rays + idx * 6;
    r[0] = sx * ox_oz;
    r[1] = sy * oy_oz;
    r[2] = o2;
    r[3] = sx * (d[0] / d[2] - ox_oz);
    r[4] = sy * (d[1] / d[2] - oy_oz);
    r[5] = 1.0f - o2;
}

// ---------------------------------------------------------------------------------
__global__ void warp_points_kernel(const float* __restrict__ raw, const float* __restrict__ xyz,
                                   const float* __restrict__ zs, long long n_points, float z_far,
                                   float* __restrict__ xyz_fw, float* __restrict__ xyz_bw) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_points * 3) return;
    const long long p = idx / 3;
    const int c = (int)(idx - p * 3);
    const bool far = zs[p] > z_far;
    const float x = xyz[idx];
    xyz_fw[idx] = x + (far ? 0.f : raw[p * RS + 8 + c]);
    xyz_bw[idx] = x + (far ? 0.f : raw[p * RS + 11 + c]);
}

// ---------------------------------------------------------------------------------
// a6: number of the training cameras of frame ts[0] that see an NDC point (reference rendering.py:190-200).
// ndc2world (ray_utils.py:127-151): Rz = 2 / (z - 1 - eps), Rx = -Rz x cx / fx, Ry = -Rz y cy / fy; then per camera
// (ray_utils.py:154-181): cam = R x_w + t with [R | t] = inverse pose, in front <=> cam_z < 0, axes -> right-down-front,
// pixel = K cam, inside <=> 0 <= u < W and 0 <= v < H.  This file is compiled with -ffp-contract=off: products and sums
// round like the reference's separate torch kernels; the 3-term dot products are fma chains like a BLAS matmul's.
__device__ __forceinline__ float frustum_count(const NsffFrustumArgs& v, float x, float y, float z) {
    const float fx = v.K4[0], fy = v.K4[1], cx = v.K4[2], cy = v.K4[3];
    const float rz = 2.0f / (z - 1.0f - 1e-6f);
    const float rx = -rz * x * cx / fx;
    const float ry = -rz * y * cy / fy;
    const long long frame = v.ts[0];
    float count = 0.f;
    if (frame < 0 || frame >= v.n_frames) return count;          // a frame the table does not hold is seen by no camera
    for (int c = 0; c < v.n_cams; ++c) {
        const float* m = v.w2c + ((long long)c * v.n_frames + frame) * 12;
        const float cam0 = fmaf(m[2], rz, fmaf(m[1], ry, m[0] * rx)) + m[3];
        const float cam1 = fmaf(m[6], rz, fmaf(m[5], ry, m[4] * rx)) + m[7];
        const float cam2 = fmaf(m[10], rz, fmaf(m[9], ry, m[8] * rx)) + m[11];
        if (!(cam2 < 0.f)) continue;                            // front is the negative z axis
        const float c1 = -cam1, c2 = -cam2;
        const float u = fmaf(cx, c2, fx * cam0) / c2, w = fmaf(cy, c2, fy * c1) / c2;
        if (u >= 0.f && u < (float)v.W && w >= 0.f && w < (float)v.H) count += 1.f;
    }
    return count;
}

__global__ __launch_bounds__(256) void frustum_visibility_kernel(const NsffFrustumArgs v, const float* __restrict__ xyz,
                                                                 long long n_points, float* __restrict__ out) {
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p < n_points) out[p] = frustum_count(v, xyz[3 * p], xyz[3 * p + 1], xyz[3 * p + 2]);
}

__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void composite_kernel(const NsffCompositeArgs a) {
    const int lane = threadIdx.x & 63;
    const long long ray = (long long)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (ray >= a.n_rays) return;     // whole wave exits; no block-level barrier below
    const int S = a.n_samples;
    const bool tr = a.has_transient != 0;
    const bool flows = tr && a.flow_mode >= 1;
    const bool warps = tr && a.flow_mode >= 2;

    float cT = 1.f, cTs = 1.f, cTfw = 1.f, cTbw = 1.f;     // running products between chunks
    float s_depth = 0.f, s_rgb[3] = {0, 0, 0}, s_talpha = 0.f, s_trgb[3] = {0, 0, 0};
    float s_sorgb[3] = {0, 0, 0}, s_sodepth = 0.f;
    float s_xyz[3] = {0, 0, 0}, s_ffw[3] = {0, 0, 0}, s_fbw[3] = {0, 0, 0};
    float s_rgbfw[3] = {0, 0, 0}, s_rgbbw[3] = {0, 0, 0}, s_occfw = 0.f, s_occbw = 0.f;

    for (int base = 0; base < S; base += 64) {
        const int i = base + lane;
        const bool on = i < S;
        const long long idx = ray * S + (on ? i : S - 1);
        const float z = a.zs[idx];
        const bool last = i >= S - 1;
        const float dz = last ? 0.f : a.zs[idx + 1] - z;
        const float d_s = last ? 100.f : dz;        // rendering.py:203
        const float d_t = last ? 1e-3f : dz;        // rendering.py:204
        const float* rec = a.raw + idx * RS;

        float sig_s = rec[3];
        if (a.noise_static) sig_s += a.noise_static[idx] * a.noise_std;
        sig_s = softplus(sig_s);
        const float al_s = 1.f - expf(-d_s * sig_s);
        float rgb_s[3] = {0, 0, 0};
        if (a.has_rgb) { rgb_s[0] = rec[0]; rgb_s[1] = rec[1]; rgb_s[2] = rec[2]; }

        float sig_t = 0.f, al_t = 0.f, alpha = al_s;
        float rgb_t[3] = {0, 0, 0};
        if (tr) {
            sig_t = rec[7];
            if (a.visibility && a.visibility[idx] == 0.f) sig_t = -10.f;     // rendering.py:200
            if (a.vis.w2c && frustum_count(a.vis, a.xyz[idx * 3], a.xyz[idx * 3 + 1], a.xyz[idx * 3 + 2]) == 0.f) sig_t = -10.f;
            if (a.noise_transient) sig_t += a.noise_transient[idx] * a.noise_std;
            sig_t = softplus(sig_t);
            al_t = 1.f - expf(-d_t * sig_t);
            alpha = 1.f - (1.f - al_s) * (1.f - al_t);
            if (a.has_rgb) { rgb_t[0] = rec[4]; rgb_t[1] = rec[5]; rgb_t[2] = rec[6]; }
        }
        // exclusive cumprod of (1 - alpha), no epsilon (rendering.py:234-235)
        const float om = on ? 1.f - alpha : 1.f;
        const float inc = wave_scan_mul(om, lane);
        float exc = __shfl_up(inc, 1);
        if (lane == 0) exc = 1.f;
        const float T = cT * exc;
        cT *= __shfl(inc, 63);
        const float w = alpha * T, w_s = al_s * T, w_t = al_t * T;

        if (on) {
            if (a.static_sigmas) a.static_sigmas[idx] = sig_s;
            if (a.has_rgb && a.static_rgbs) {
                a.static_rgbs[idx * 3 + 0] = rgb_s[0]; a.static_rgbs[idx * 3 + 1] = rgb_s[1];
                a.static_rgbs[idx * 3 + 2] = rgb_s[2];
            }
            if (tr) {
                if (a.transient_sigmas) a.transient_sigmas[idx] = sig_t;
                if (a.has_rgb && a.transient_rgbs) {
                    a.transient_rgbs[idx * 3 + 0] = rgb_t[0]; a.transient_rgbs[idx * 3 + 1] = rgb_t[1];
                    a.transient_rgbs[idx * 3 + 2] = rgb_t[2];
                }
                if (a.static_alphas) a.static_alphas[idx] = al_s;
                if (a.transient_alphas) a.transient_alphas[idx] = al_t;
                if (a.static_weights) a.static_weights[idx] = w_s;
                if (a.transient_weights) a.transient_weights[idx] = w_t;
                if (a.weights) a.weights[idx] = w;
            } else {
                if (a.static_weights) a.static_weights[idx] = w;     // rendering.py:248
                if (a.weights) a.weights[idx] = w;
            }
        }
        if (!a.has_rgb) continue;     // sigma-only coarse pass stops at the weights (rendering.py:253)

        if (on) {
            s_depth += w * z;
            if (tr) {
#pragma unroll
                for (int c = 0; c < 3; ++c) { s_rgb[c] += w_s * rgb_s[c]; s_trgb[c] += w_t * rgb_t[c]; }
                s_talpha += w_t;
            } else {
#pragma unroll
                for (int c = 0; c < 3; ++c) s_rgb[c] += w * rgb_s[c];
            }
        }
        if (!tr) continue;

        // static field alone, with its own transmittance (rendering.py:270-278)
        {
            const float oms = on ? 1.f - al_s : 1.f;
            const float incs = wave_scan_mul(oms, lane);
            float excs = __shfl_up(incs, 1);
            if (lane == 0) excs = 1.f;
            const float Ts = cTs * excs;
            cTs *= __shfl(incs, 63);
            const float wso = al_s * Ts;
            if (on) {
#pragma unroll
                for (int c = 0; c < 3; ++c) s_sorgb[c] += wso * rgb_s[c];
                s_sodepth += wso * z;
            }
        }
        if (!flows) continue;

        const bool far = z > a.z_far;
        float ffw[3], fbw[3], x[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            ffw[c] = far ? 0.f : rec[8 + c];       // rendering.py:187-188
            fbw[c] = far ? 0.f : rec[11 + c];
            x[c] = a.xyz[idx * 3 + c];
        }
        if (on) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (a.flows_fw) a.flows_fw[idx * 3 + c] = ffw[c];
                if (a.flows_bw) a.flows_bw[idx * 3 + c] = fbw[c];
                s_xyz[c] += w * x[c]; s_ffw[c] += w * ffw[c]; s_fbw[c] += w * fbw[c];
            }
        }
        if (!warps) continue;

        // re-composite with the warped transient field and the CURRENT static field
        // (render_transient_warping, rendering.py:98-140)
#pragma unroll
        for (int dir = 0; dir < 2; ++dir) {
            const float* recw = (dir == 0 ? a.raw_fw : a.raw_bw) + idx * RS;
            const float* nz = dir == 0 ? a.noise_fw : a.noise_bw;
            float sg = recw[7];
            if (nz) sg += nz[idx] * a.noise_std;
            const float al_tw = 1.f - expf(-d_t * softplus(sg));
            const float al_w = 1.f - (1.f - al_s) * (1.f - al_tw);
            const float omw = on ? 1.f - al_w : 1.f;
            const float incw = wave_scan_mul(omw, lane);
            float excw = __shfl_up(incw, 1);
            if (lane == 0) excw = 1.f;
            float& cw = dir == 0 ? cTfw : cTbw;
            const float Tw = cw * excw;
            cw *= __shfl(incw, 63);
            const float ws_w = al_s * Tw, wt_w = al_tw * Tw;
            if (on) {
                float* srgb = dir == 0 ? s_rgbfw : s_rgbbw;
#pragma unroll
                for (int c = 0; c < 3; ++c) srgb[c] += ws_w * rgb_s[c] + wt_w * recw[4 + c];
                // cycle point: the warped query's flow back (fw warp -> 'bw' head and vice versa)
                const float* xw = (dir == 0 ? a.xyz_fw : a.xyz_bw) + idx * 3;
                float* cyc = dir == 0 ? a.xyzs_fw_bw : a.xyzs_bw_fw;
                const int head = dir == 0 ? 11 : 8;
                if (cyc) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) cyc[idx * 3 + c] = xw[c] + (far ? 0.f : recw[head + c]);
                }
                const float occ = wt_w - w_t;              // rendering.py:290-291
                if (dir == 0) { s_occfw += occ; if (a.disoccs_fw) a.disoccs_fw[idx] = 1.f - fabsf(occ); }
                else          { s_occbw += occ; if (a.disoccs_bw) a.disoccs_bw[idx] = 1.f - fabsf(occ); }
            }
        }
    }
    if (!a.has_rgb) return;

    s_depth = wave_sum(s_depth);
#pragma unroll
    for (int c = 0; c < 3; ++c) s_rgb[c] = wave_sum(s_rgb[c]);
    if (tr) {
        s_talpha = wave_sum(s_talpha); s_sodepth = wave_sum(s_sodepth);
#pragma unroll
        for (int c = 0; c < 3; ++c) { s_trgb[c] = wave_sum(s_trgb[c]); s_sorgb[c] = wave_sum(s_sorgb[c]); }
    }
    if (flows) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { s_xyz[c] = wave_sum(s_xyz[c]); s_ffw[c] = wave_sum(s_ffw[c]); s_fbw[c] = wave_sum(s_fbw[c]); }
    }
    if (warps) {
        s_occfw = wave_sum(s_occfw); s_occbw = wave_sum(s_occbw);
#pragma unroll
        for (int c = 0; c < 3; ++c) { s_rgbfw[c] = wave_sum(s_rgbfw[c]); s_rgbbw[c] = wave_sum(s_rgbbw[c]); }
    }
    if (lane != 0) return;
    if (a.depth) a.depth[ray] = s_depth;
    if (!tr) {
        if (a.rgb) for (int c = 0; c < 3; ++c) a.rgb[ray * 3 + c] = s_rgb[c];
        return;
    }
    for (int c = 0; c < 3; ++c) {
        if (a.rgb) a.rgb[ray * 3 + c] = s_rgb[c] + s_trgb[c];                                   // :262
        if (a.transient_rgb) a.transient_rgb[ray * 3 + c] = s_trgb[c] + 0.8f * (1.f - s_talpha);  // :264-265
        if (a.static_only_rgb) a.static_only_rgb[ray * 3 + c] = s_sorgb[c];
    }
    if (a.transient_alpha) a.transient_alpha[ray] = s_talpha;
    if (a.static_only_depth) a.static_only_depth[ray] = s_sodepth;
    if (flows) {
        for (int c = 0; c < 3; ++c) {
            if (a.xyz_exp) a.xyz_exp[ray * 3 + c] = s_xyz[c];
            if (a.flow_fw_exp) a.flow_fw_exp[ray * 3 + c] = s_ffw[c];
            if (a.flow_bw_exp) a.flow_bw_exp[ray * 3 + c] = s_fbw[c];
            if (a.xyz_fw_exp) a.xyz_fw_exp[ray * 3 + c] = s_xyz[c] + s_ffw[c];
            if (a.xyz_bw_exp) a.xyz_bw_exp[ray * 3 + c] = s_xyz[c] + s_fbw[c];
        }
    }
    if (warps) {
        for (int c = 0; c < 3; ++c) {
            if (a.rgb_fw) a.rgb_fw[ray * 3 + c] = s_rgbfw[c];
            if (a.rgb_bw) a.rgb_bw[ray * 3 + c] = s_rgbbw[c];
        }
        if (a.disocc_fw) a.disocc_fw[ray] = 1.f - fabsf(s_occfw);
        if (a.disocc_bw) a.disocc_bw[ray] = 1.f - fabsf(s_occbw);
    }
}

// The same for 64 < n_samples <= 256 (the fine pass: 192): ONE ray per workgroup, wave c = samples [64 c, 64 c + 64).  The serial
// form above walks a ray's chunks one after the other -- three exposed memory latencies per ray on four waves per CU; here every
// chunk's loads are in flight at once.  A chunk's transmittances start at the product of the chunks in front of it: each wave
// leaves its four products of (1 - alpha) in LDS, a chunk multiplies its predecessors' in the serial form's order (the same
// weights, bit for bit); the per-ray sums are wave sums per chunk added in chunk order.
template <int NCH>
__global__ __launch_bounds__(64 * NCH) void composite_kernel_chunks(const NsffCompositeArgs a) {
    __shared__ float tot[NCH][4];
    __shared__ float part[NCH][32];
    const int lane = threadIdx.x & 63, c = threadIdx.x >> 6;
    const long long ray = blockIdx.x;
    const int S = a.n_samples;
    const bool tr = a.has_transient != 0;
    const bool flows = tr && a.flow_mode >= 1;
    const bool warps = tr && a.flow_mode >= 2;

    const int i = 64 * c + lane;
    const bool on = i < S;
    const long long idx = ray * S + (on ? i : S - 1);
    const float z = a.zs[idx];
    const bool last = i >= S - 1;
    const float dz = last ? 0.f : a.zs[idx + 1] - z;
    const float d_s = last ? 100.f : dz;        // rendering.py:203
    const float d_t = last ? 1e-3f : dz;        // rendering.py:204
    const float* rec = a.raw + idx * RS;

    float sig_s = rec[3];
    if (a.noise_static) sig_s += a.noise_static[idx] * a.noise_std;
    sig_s = softplus(sig_s);
    const float al_s = 1.f - expf(-d_s * sig_s);
    float rgb_s[3] = {0, 0, 0};
    if (a.has_rgb) { rgb_s[0] = rec[0]; rgb_s[1] = rec[1]; rgb_s[2] = rec[2]; }

    float sig_t = 0.f, al_t = 0.f, alpha = al_s;
    float rgb_t[3] = {0, 0, 0};
    if (tr) {
        sig_t = rec[7];
        if (a.visibility && a.visibility[idx] == 0.f) sig_t = -10.f;     // rendering.py:200
        if (a.vis.w2c && frustum_count(a.vis, a.xyz[idx * 3], a.xyz[idx * 3 + 1], a.xyz[idx * 3 + 2]) == 0.f) sig_t = -10.f;
        if (a.noise_transient) sig_t += a.noise_transient[idx] * a.noise_std;
        sig_t = softplus(sig_t);
        al_t = 1.f - expf(-d_t * sig_t);
        alpha = 1.f - (1.f - al_s) * (1.f - al_t);
        if (a.has_rgb) { rgb_t[0] = rec[4]; rgb_t[1] = rec[5]; rgb_t[2] = rec[6]; }
    }
    // the chunk's own scans: blend, static field alone (rendering.py:270-278), the two warped renders (rendering.py:98-140)
    const float inc = wave_scan_mul(on ? 1.f - alpha : 1.f, lane);
    float incs = 1.f, incw[2] = {1.f, 1.f}, al_tw[2] = {0.f, 0.f};
    if (a.has_rgb && tr) incs = wave_scan_mul(on ? 1.f - al_s : 1.f, lane);
    if (a.has_rgb && warps) {
#pragma unroll
        for (int dir = 0; dir < 2; ++dir) {
            const float* recw = (dir == 0 ? a.raw_fw : a.raw_bw) + idx * RS;
            const float* nz = dir == 0 ? a.noise_fw : a.noise_bw;
            float sg = recw[7];
            if (nz) sg += nz[idx] * a.noise_std;
            al_tw[dir] = 1.f - expf(-d_t * softplus(sg));
            const float al_w = 1.f - (1.f - al_s) * (1.f - al_tw[dir]);
            incw[dir] = wave_scan_mul(on ? 1.f - al_w : 1.f, lane);
        }
    }
    if (lane == 63) { tot[c][0] = inc; tot[c][1] = incs; tot[c][2] = incw[0]; tot[c][3] = incw[1]; }
    __syncthreads();
    float cT = 1.f, cTs = 1.f, cTw[2] = {1.f, 1.f};
    for (int cc = 0; cc < c; ++cc) { cT *= tot[cc][0]; cTs *= tot[cc][1]; cTw[0] *= tot[cc][2]; cTw[1] *= tot[cc][3]; }

    // exclusive cumprod of (1 - alpha), no epsilon (rendering.py:234-235)
    float exc = __shfl_up(inc, 1);
    if (lane == 0) exc = 1.f;
    const float T = cT * exc;
    const float w = alpha * T, w_s = al_s * T, w_t = al_t * T;
    if (on) {
        if (a.static_sigmas) a.static_sigmas[idx] = sig_s;
        if (a.has_rgb && a.static_rgbs) {
            a.static_rgbs[idx * 3 + 0] = rgb_s[0]; a.static_rgbs[idx * 3 + 1] = rgb_s[1];
            a.static_rgbs[idx * 3 + 2] = rgb_s[2];
        }
        if (tr) {
            if (a.transient_sigmas) a.transient_sigmas[idx] = sig_t;
            if (a.has_rgb && a.transient_rgbs) {
                a.transient_rgbs[idx * 3 + 0] = rgb_t[0]; a.transient_rgbs[idx * 3 + 1] = rgb_t[1];
                a.transient_rgbs[idx * 3 + 2] = rgb_t[2];
            }
            if (a.static_alphas) a.static_alphas[idx] = al_s;
            if (a.transient_alphas) a.transient_alphas[idx] = al_t;
            if (a.static_weights) a.static_weights[idx] = w_s;
            if (a.transient_weights) a.transient_weights[idx] = w_t;
            if (a.weights) a.weights[idx] = w;
        } else {
            if (a.static_weights) a.static_weights[idx] = w;     // rendering.py:248
            if (a.weights) a.weights[idx] = w;
        }
    }
    if (!a.has_rgb) return;     // sigma-only coarse pass stops at the weights (rendering.py:253); uniform: no barrier is skipped by a part of the workgroup

    // this chunk's terms of the per-ray sums (lanes past the ray's end contribute zeros), in the order of part[]'s columns
    enum { P_DEPTH, P_RGB, P_TRGB = P_RGB + 3, P_TALPHA = P_TRGB + 3, P_SORGB, P_SODEPTH = P_SORGB + 3, P_XYZ, P_FFW = P_XYZ + 3,
           P_FBW = P_FFW + 3, P_RGBFW = P_FBW + 3, P_RGBBW = P_RGBFW + 3, P_OCCFW = P_RGBBW + 3, P_OCCBW, P_N };
    static_assert(P_N <= 32, "part[] columns");
    float p[P_N];
#pragma unroll
    for (int k = 0; k < P_N; ++k) p[k] = 0.f;
    if (on) {
        p[P_DEPTH] = w * z;
        if (tr) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { p[P_RGB + k] = w_s * rgb_s[k]; p[P_TRGB + k] = w_t * rgb_t[k]; }
            p[P_TALPHA] = w_t;
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) p[P_RGB + k] = w * rgb_s[k];
        }
    }
    if (tr) {
        float excs = __shfl_up(incs, 1);
        if (lane == 0) excs = 1.f;
        const float wso = al_s * (cTs * excs);
        if (on) {
#pragma unroll
            for (int k = 0; k < 3; ++k) p[P_SORGB + k] = wso * rgb_s[k];
            p[P_SODEPTH] = wso * z;
        }
    }
    if (flows) {
        const bool far = z > a.z_far;
        float ffw[3], fbw[3], x[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            ffw[k] = far ? 0.f : rec[8 + k];       // rendering.py:187-188
            fbw[k] = far ? 0.f : rec[11 + k];
            x[k] = a.xyz[idx * 3 + k];
        }
        if (on) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (a.flows_fw) a.flows_fw[idx * 3 + k] = ffw[k];
                if (a.flows_bw) a.flows_bw[idx * 3 + k] = fbw[k];
                p[P_XYZ + k] = w * x[k]; p[P_FFW + k] = w * ffw[k]; p[P_FBW + k] = w * fbw[k];
            }
        }
        if (warps) {
#pragma unroll
            for (int dir = 0; dir < 2; ++dir) {
                const float* recw = (dir == 0 ? a.raw_fw : a.raw_bw) + idx * RS;
                float excw = __shfl_up(incw[dir], 1);
                if (lane == 0) excw = 1.f;
                const float Tw = cTw[dir] * excw;
                const float ws_w = al_s * Tw, wt_w = al_tw[dir] * Tw;
                if (on) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) p[(dir == 0 ? P_RGBFW : P_RGBBW) + k] = ws_w * rgb_s[k] + wt_w * recw[4 + k];
                    // cycle point: the warped query's flow back (fw warp -> 'bw' head and vice versa)
                    const float* xw = (dir == 0 ? a.xyz_fw : a.xyz_bw) + idx * 3;
                    float* cyc = dir == 0 ? a.xyzs_fw_bw : a.xyzs_bw_fw;
                    const int head = dir == 0 ? 11 : 8;
                    if (cyc) {
#pragma unroll
                        for (int k = 0; k < 3; ++k) cyc[idx * 3 + k] = xw[k] + (far ? 0.f : recw[head + k]);
                    }
                    const float occ = wt_w - w_t;              // rendering.py:290-291
                    if (dir == 0) { p[P_OCCFW] = occ; if (a.disoccs_fw) a.disoccs_fw[idx] = 1.f - fabsf(occ); }
                    else          { p[P_OCCBW] = occ; if (a.disoccs_bw) a.disoccs_bw[idx] = 1.f - fabsf(occ); }
                }
            }
        }
    }
    const int n_used = warps ? (int)P_N : (flows ? (int)P_RGBFW : (tr ? (int)P_XYZ : (int)P_TRGB));
#pragma unroll
    for (int k = 0; k < P_N; ++k) {
        if (k < n_used) {
            const float v = wave_sum(p[k]);
            if (lane == 0) part[c][k] = v;
        }
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    float s[P_N];
#pragma unroll
    for (int k = 0; k < P_N; ++k) {
        s[k] = 0.f;
        if (k < n_used) for (int cc = 0; cc < NCH; ++cc) s[k] += part[cc][k];
    }
    if (a.depth) a.depth[ray] = s[P_DEPTH];
    if (!tr) {
        if (a.rgb) for (int k = 0; k < 3; ++k) a.rgb[ray * 3 + k] = s[P_RGB + k];
        return;
    }
    for (int k = 0; k < 3; ++k) {
        if (a.rgb) a.rgb[ray * 3 + k] = s[P_RGB + k] + s[P_TRGB + k];                                      // :262
        if (a.transient_rgb) a.transient_rgb[ray * 3 + k] = s[P_TRGB + k] + 0.8f * (1.f - s[P_TALPHA]);    // :264-265
        if (a.static_only_rgb) a.static_only_rgb[ray * 3 + k] = s[P_SORGB + k];
    }
    if (a.transient_alpha) a.transient_alpha[ray] = s[P_TALPHA];
    if (a.static_only_depth) a.static_only_depth[ray] = s[P_SODEPTH];
    if (flows) {
        for (int k = 0; k < 3; ++k) {
            if (a.xyz_exp) a.xyz_exp[ray * 3 + k] = s[P_XYZ + k];
            if (a.flow_fw_exp) a.flow_fw_exp[ray * 3 + k] = s[P_FFW + k];
            if (a.flow_bw_exp) a.flow_bw_exp[ray * 3 + k] = s[P_FBW + k];
            if (a.xyz_fw_exp) a.xyz_fw_exp[ray * 3 + k] = s[P_XYZ + k] + s[P_FFW + k];
            if (a.xyz_bw_exp) a.xyz_bw_exp[ray * 3 + k] = s[P_XYZ + k] + s[P_FBW + k];
        }
    }
    if (warps) {
        for (int k = 0; k < 3; ++k) {
            if (a.rgb_fw) a.rgb_fw[ray * 3 + k] = s[P_RGBFW + k];
            if (a.rgb_bw) a.rgb_bw[ray * 3 + k] = s[P_RGBBW + k];
        }
        if (a.disocc_fw) a.disocc_fw[ray] = 1.f - fabsf(s[P_OCCFW]);
        if (a.disocc_bw) a.disocc_bw[ray] = 1.f - fabsf(s[P_OCCBW]);
    }
}

// rows of the time-code table for the neighbouring frames: next[r] = E[min(ts[r] + 1, max_t)], prev[r] = E[max(ts[r] - 1, 0)]
// (reference rendering.py:218,224: embedding_t(torch.clamp(ts +- 1, ...))) -- one launch instead of two add / clamp / gather
// triples; either output may be NULL.  One thread per float4 (width % 4 == 0) or per float.
__global__ __launch_bounds__(256) void time_rows_kernel(const float* __restrict__ table, long long n_table, int width,
                                                         const long long* __restrict__ ts, long long n, long long max_t,
                                                         float* __restrict__ next, float* __restrict__ prev, int vec) {
    const int per_row = vec ? width / 4 : width;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * per_row) return;
    const long long r = i / per_row;
    const int c = (int)(i - r * per_row);
    const long long t = ts[r];
    long long tn = t + 1 < max_t ? t + 1 : max_t, tp = t - 1 > 0 ? t - 1 : 0;
    tn = tn < 0 ? 0 : (tn > n_table - 1 ? n_table - 1 : tn);
    tp = tp > n_table - 1 ? n_table - 1 : tp;
    if (vec) {
        if (next) reinterpret_cast<float4*>(next)[i] = reinterpret_cast<const float4*>(table + tn * width)[c];
        if (prev) reinterpret_cast<float4*>(prev)[i] = reinterpret_cast<const float4*>(table + tp * width)[c];
    } else {
        if (next) next[i] = table[tn * width + c];
        if (prev) prev[i] = table[tp * width + c];
    }
}

// Gradient of the three time-code gathers of a training step, E[ts], E[min(ts + 1, max_t)], E[max(ts - 1, 0)], w.r.t. the table:
// d_table[r] = sum of the rows of g_cur / g_next / g_prev whose (clamped) index is r.  One workgroup per table row: wave 0
// compacts, per gather, the rays that hit the row into an LDS list in ascending order (ballot + popcount), the four waves then
// add the listed rows -- wave w the entries w, w + 4, ... -- and their partial sums are added in a fixed order: deterministic.
// (A batch hits ~30 distinct rows: torch's embedding backward serialises on atomics, ~200 us per gather; the one-hot GEMM that
// replaced it in round 1 was 4 kernels per gather; a first form of this kernel that let every thread walk all rays took 110 us.)
constexpr int TRB_CHUNK = 2048;          // rays per compaction pass
constexpr int TRB_COLS = 4;              // 64-column groups held in registers (width <= 256; wider tables loop over groups of 4)
__global__ __launch_bounds__(256) void time_rows_bwd_kernel(const float* __restrict__ g_cur, const float* __restrict__ g_next,
                                                             const float* __restrict__ g_prev, const long long* __restrict__ ts,
                                                             long long n, long long max_t, long long n_table, int width,
                                                             float* __restrict__ d_table) {
    __shared__ int sList[3][TRB_CHUNK];
    __shared__ int sTs[TRB_CHUNK];           // the chunk's frame indices (clamped to int range), staged by all four waves
    __shared__ int sCnt[3];
    __shared__ float sPart[4][64];
    const long long r = blockIdx.x;
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const float* gs[3] = {g_cur, g_next, g_prev};
    for (int cg = 0; cg < width; cg += 64 * TRB_COLS) {
        float acc[TRB_COLS];
#pragma unroll
        for (int j = 0; j < TRB_COLS; ++j) acc[j] = 0.f;
        for (long long base = 0; base < n; base += TRB_CHUNK) {
            const long long end = base + TRB_CHUNK < n ? base + TRB_CHUNK : n;
            for (long long i = base + threadIdx.x; i < end; i += 256) {
                const long long t = ts[i];
                sTs[i - base] = (int)(t < -2 ? -2 : (t > 0x7ffffff0LL ? 0x7ffffff0LL : t));
            }
            __syncthreads();
            if (part == 0) {
                int cnt[3] = {0, 0, 0};
                for (long long i0 = base; i0 < end; i0 += 64) {
                    const long long i = i0 + lane;
                    bool hit[3] = {false, false, false};
                    if (i < end) {
                        const long long t = sTs[i - base];
                        long long tn = t + 1 < max_t ? t + 1 : max_t, tp = t - 1 > 0 ? t - 1 : 0;    // (nsff_time_rows' own index arithmetic)
                        tn = tn < 0 ? 0 : (tn > n_table - 1 ? n_table - 1 : tn);
                        tp = tp > n_table - 1 ? n_table - 1 : tp;
                        hit[0] = t == r; hit[1] = tn == r; hit[2] = tp == r;
                    }
#pragma unroll
                    for (int l = 0; l < 3; ++l) {
                        const unsigned long long m = __ballot(hit[l] && gs[l] != nullptr);
                        if (hit[l] && gs[l] != nullptr) sList[l][cnt[l] + __popcll(m & ((1ull << lane) - 1ull))] = (int)(i - base);
                        cnt[l] += __popcll(m);
                    }
                }
                if (lane == 0) { sCnt[0] = cnt[0]; sCnt[1] = cnt[1]; sCnt[2] = cnt[2]; }
            }
            __syncthreads();
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                const float* __restrict__ g = gs[l];
                const int cnt = sCnt[l];
                for (int k0 = part; k0 < cnt; k0 += 16) {            // four rows requested before the first add (fixed order)
                    float v[4][TRB_COLS];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int k = k0 + 4 * u;
                        const float* row = g + (base + sList[l][k < cnt ? k : k0]) * width + cg + lane;
#pragma unroll
                        for (int j = 0; j < TRB_COLS; ++j)
                            v[u][j] = (k < cnt && cg + 64 * j + lane < width) ? row[64 * j] : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int j = 0; j < TRB_COLS; ++j) acc[j] += v[u][j];
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < TRB_COLS; ++j) {
            const int c = cg + 64 * j + lane;
            if (cg + 64 * j >= width) break;
            sPart[part][lane] = acc[j];
            __syncthreads();
            if (part == 0 && c < width) d_table[r * width + c] = (sPart[0][lane] + sPart[1][lane]) + (sPart[2][lane] + sPart[3][lane]);
            __syncthreads();
        }
    }
}

}  // namespace

extern "C" {

int nsff_time_rows_backward(const float* g_cur, const float* g_next, const float* g_prev, const int64_t* ts, int64_t n,
                            int64_t max_t, int64_t n_table, int32_t width, float* d_table, void* stream) {
    if (n < 0 || n_table < 1 || n_table > 0x7fffffffLL || width < 1) return NSFF_ERR_INVALID;
    if (!d_table || (n > 0 && !ts)) return NSFF_ERR_NULL;
    hipLaunchKernelGGL(time_rows_bwd_kernel, dim3((unsigned)n_table), dim3(256), 0, (hipStream_t)stream, g_cur, g_next, g_prev,
                       reinterpret_cast<const long long*>(ts), (long long)n, (long long)max_t, (long long)n_table, (int)width,
                       d_table);
    return nsff_launch_status();
}

int nsff_time_rows(const float* table, int64_t n_table, int32_t width, const int64_t* ts, int64_t n, int64_t max_t,
                   float* next, float* prev, void* stream) {
    if (n < 0 || n_table < 1 || width < 1) return NSFF_ERR_INVALID;
    if (n == 0 || (!next && !prev)) return NSFF_OK;
    if (!table || !ts) return NSFF_ERR_NULL;
    const int vec = (width % 4 == 0) && !(((uintptr_t)table | (uintptr_t)next | (uintptr_t)prev) & 15);
    const long long total = n * (vec ? width / 4 : width);
    hipLaunchKernelGGL(time_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, table,
                       (long long)n_table, (int)width, reinterpret_cast<const long long*>(ts), (long long)n, (long long)max_t,
                       next, prev, vec);
    return nsff_launch_status();
}


int nsff_coarse_samples(const float* rays, int64_t n_rays, const float* z_lin, int32_t n_samples,
                        float perturb, const float* perturb_rand, float* zs, float* xyz, void* stream) {
    if (n_rays < 0 || n_samples < 1) return NSFF_ERR_INVALID;
    if (n_rays == 0) return NSFF_OK;
    if (!rays || !z_lin || !zs || !xyz) return NSFF_ERR_NULL;
    if (perturb > 0.f && !perturb_rand) return NSFF_ERR_NULL;
    const long long total = (long long)n_rays * n_samples;
    hipLaunchKernelGGL(coarse_samples_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, rays, (long long)n_rays, z_lin, n_samples, perturb,
                       perturb_rand, zs, xyz);
    return nsff_launch_status();
}

int nsff_fine_samples(const float* rays, int64_t n_rays, const float* z_lin, const float* zs_coarse,
                      int32_t n_samples, int32_t n_importance,
                      const float* weights_static, const float* weights_transient,
                      const float* u_static, const float* u_transient, int32_t u_per_ray,
                      float* samples_static, float* samples_transient,
                      float* zs_fine, float* xyz_fine, void* stream) {
    if (n_rays < 0 || n_samples < 3 || n_importance < 1) return NSFF_ERR_INVALID;
    if (n_rays == 0) return NSFF_OK;
    if (!rays || !z_lin || !zs_coarse || !weights_static || !u_static || !zs_fine || !xyz_fine) return NSFF_ERR_NULL;
    if (weights_transient && !u_transient) return NSFF_ERR_NULL;
    FineArgs a{rays, (long long)n_rays, z_lin, zs_coarse, n_samples, n_importance,
               weights_static, weights_transient, u_static, u_transient, u_per_ray,
               samples_static, samples_transient, zs_fine, xyz_fine};
    const int Sf = n_samples + (weights_transient ? 2 : 1) * n_importance;
    const size_t lds = (size_t)(WAVES_PER_BLOCK * n_samples + ((n_samples + 3) & ~3) + Sf) * sizeof(float);
    if (lds > 64 * 1024 || n_rays > 0x7fffffffLL) return NSFF_ERR_INVALID;
    const unsigned blocks = (unsigned)n_rays;                   // one ray per workgroup
    hipLaunchKernelGGL(fine_samples_kernel, dim3(blocks), dim3(64 * WAVES_PER_BLOCK), lds,
                       (hipStream_t)stream, a);
    return nsff_launch_status();
}

int nsff_sample_pdf(const float* bins, const float* weights, int64_t n_rays, int32_t n_bins_minus1,
                    const float* u, int32_t n_importance, int32_t u_per_ray, float eps,
                    float* samples, void* stream) {
    if (n_rays < 0 || n_bins_minus1 < 1 || n_importance < 1) return NSFF_ERR_INVALID;
    if (n_rays == 0) return NSFF_OK;
    if (!bins || !weights || !u || !samples) return NSFF_ERR_NULL;
    PdfArgs a{bins, weights, (long long)n_rays, n_bins_minus1, u, n_importance, u_per_ray, eps, samples};
    const size_t lds = (size_t)WAVES_PER_BLOCK * (n_bins_minus1 + 1) * sizeof(float);
    if (lds > 64 * 1024) return NSFF_ERR_INVALID;
    const unsigned blocks = (unsigned)((n_rays + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK);
    hipLaunchKernelGGL(sample_pdf_kernel, dim3(blocks), dim3(64 * WAVES_PER_BLOCK), lds,
                       (hipStream_t)stream, a);
    return nsff_launch_status();
}

int nsff_warp_points(const float* raw, const float* xyz, const float* zs, int64_t n_points,
                     float z_far, float* xyz_fw, float* xyz_bw, void* stream) {
    if (n_points < 0) return NSFF_ERR_INVALID;
    if (n_points == 0) return NSFF_OK;
    if (!raw || !xyz || !zs || !xyz_fw || !xyz_bw) return NSFF_ERR_NULL;
    const long long total = (long long)n_points * 3;
    hipLaunchKernelGGL(warp_points_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, raw, xyz, zs, (long long)n_points, z_far, xyz_fw, xyz_bw);
    return nsff_launch_status();
}

int nsff_frame_rays(const float* K4_host, const float* c2w_host, int32_t H, int32_t W, float near, float shift_near,
                    int64_t first_pixel, int64_t n_pixels, float* rays, void* stream) {
    if (!K4_host || !c2w_host) return NSFF_ERR_NULL;
    if (H < 1 || W < 1 || first_pixel < 0 || n_pixels < 0 || first_pixel + n_pixels > (int64_t)H * W) return NSFF_ERR_INVALID;
    if (n_pixels == 0) return NSFF_OK;
    if (!rays) return NSFF_ERR_NULL;
    FrameRayArgs a{};
    a.fx = K4_host[0]; a.fy = K4_host[1]; a.cx = K4_host[2]; a.cy = K4_host[3];
    for (int i = 0; i < 12; ++i) a.c2w[i] = c2w_host[i];
    a.W = W; a.near = near; a.shift_near = shift_near; a.first = first_pixel; a.count = n_pixels; a.rays = rays;
    hipLaunchKernelGGL(frame_rays_kernel, dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return nsff_launch_status();
}

int nsff_composite(const NsffCompositeArgs* args, void* stream) {
    if (!args) return NSFF_ERR_NULL;
    const NsffCompositeArgs& a = *args;
    if (a.n_rays < 0 || a.n_samples < 1 || a.flow_mode < 0 || a.flow_mode > 2) return NSFF_ERR_INVALID;
    if (a.n_rays == 0) return NSFF_OK;
    if (!a.raw || !a.zs) return NSFF_ERR_NULL;
    if (a.flow_mode && (!a.has_transient || !a.has_rgb)) return NSFF_ERR_INVALID;
    if (a.flow_mode >= 1 && !a.xyz) return NSFF_ERR_NULL;
    if (a.flow_mode == 2 && (!a.raw_fw || !a.raw_bw || !a.xyz_fw || !a.xyz_bw)) return NSFF_ERR_NULL;
    if (a.vis.w2c && (!a.vis.ts || !a.xyz)) return NSFF_ERR_NULL;
    if (a.vis.w2c && (a.vis.n_cams < 1 || a.vis.n_frames < 1)) return NSFF_ERR_INVALID;
    // 65..256 samples per ray: one ray per workgroup, one wave per 64-sample chunk (NSFF_SERIAL_COMPOSITE=1: the serial form, A/B)
    const int nch = (a.n_samples + 63) / 64;
    if (nch >= 2 && nch <= 4 && a.n_rays <= 0x7fffffffLL && getenv("NSFF_SERIAL_COMPOSITE") == nullptr) {
        const dim3 grid((unsigned)a.n_rays);
        if (nch == 2) hipLaunchKernelGGL(composite_kernel_chunks<2>, grid, dim3(128), 0, (hipStream_t)stream, a);
        else if (nch == 3) hipLaunchKernelGGL(composite_kernel_chunks<3>, grid, dim3(192), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL(composite_kernel_chunks<4>, grid, dim3(256), 0, (hipStream_t)stream, a);
        return nsff_launch_status();
    }
    const unsigned blocks = (unsigned)((a.n_rays + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK);
    hipLaunchKernelGGL(composite_kernel, dim3(blocks), dim3(64 * WAVES_PER_BLOCK), 0,
                       (hipStream_t)stream, a);
    return nsff_launch_status();
}

int nsff_frustum_visibility(const NsffFrustumArgs* vis, const float* xyz, int64_t n_points, float* vis_count,
                            void* stream) {
    if (!vis || !vis->w2c || !vis->ts) return NSFF_ERR_NULL;
    if (n_points < 0 || vis->n_cams < 1 || vis->n_frames < 1) return NSFF_ERR_INVALID;
    if (n_points == 0) return NSFF_OK;
    if (!xyz || !vis_count) return NSFF_ERR_NULL;
    const long long blocks = (n_points + 255) / 256;
    if (blocks > 0x7fffffffLL) return NSFF_ERR_INVALID;
    hipLaunchKernelGGL(frustum_visibility_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *vis, xyz,
                       (long long)n_points, vis_count);
    return nsff_launch_status();
}

}  // extern "C"
