// N1: the NSFF training objective as six small launches forward + two backward (reference losses.py:8-28
// shiftscale_invariant_depthloss, :31-171 NeRFWLoss) instead of the ~580 elementwise / reduction kernels the torch expression of
// its forward and backward needs.  All eleven terms of the train-mode NSFF configuration (flows + disocclusion present), with
// every reduction the reference's constructor can ask for: plain means, per-ray `weights` (hard sampling), top-k mining
// (--topk < 1: mean of the int(topk * M) largest per-ray values, losses.py:162-169) and the dilated cross entropy
// (--thickness > 1, losses.py:91-95).
//
//   loss_sums_kernel    grid-stride sums of the per-sample disocclusion weights (their means normalise cyc_l)
//   loss_median_kernel  medians of depth_fine / depth_coarse / -disp by rank counting in LDS (torch.median = lower median)
//   loss_stats_kernel   ONE workgroup: mean absolute deviations around them, means of the per-ray disocclusion weights
//   loss_rays_kernel<1> one wavefront per ray, lane = sample: the ray's value of every term (x its weight) -> per_ray
//   loss_select_kernel  per term: rank of every ray's value inside the term's population (LDS rank counting), K = int(topk M),
//                       coef[k][n] = weight_n [rank < K] / K, term_k = sum of the selected values / K
//   loss_stats2_kernel  (backward) the second-level sums the gradient of the depth term needs, with those coefficients
//   loss_rays_kernel<2> the gradient of sum_k w_k * term_k w.r.t. every consumed render tensor
// Bound: HBM (reads ~70 B/sample, writes ~50 B/sample in mode 2); a few microseconds per launch at 1024 x 192.
#include <hip/hip_runtime.h>
#include <algorithm>
#include "nsff_common.h"

namespace {

constexpr int MAXN = 4096;                 // rays per call the single-workgroup statistics kernel handles
enum { T_COL = 0, T_DISP, T_ENT, T_CE, T_FLOW_FW, T_FLOW_BW, T_PHO, T_CYC, T_TEMP, T_MIN, T_SP, N_TERMS };
// stats buffer (floats)
enum { ST_MED = 0,      // [3] median of depth_fine, depth_coarse, -disp
       ST_MAD = 3,      // [3] mean |x - median|
       ST_IDX = 6,      // [3] index of the median element (int bits)
       ST_SG = 9,       // [2] sum_n g_n           (g = d disp_l / d y, fine / coarse)
       ST_SGY = 11,     // [2] sum_n g_n y_n
       ST_SSGN = 13,    // [2] sum_n sign(x_n - median)
       ST_DOCC = 15,    // [2] mean disocc_fw, disocc_bw
       ST_DOCCS = 17,   // [2] SUM of disoccs_fw, disoccs_bw (loss_sums_kernel; divided by N*S where used)
       ST_CNT = 19,     // [2] number of rays with a valid forward / backward flow projection
       ST_SIZE = 24 };

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

struct Cam { float cxfx, cyfy; };
// datasets/ray_utils.py:127-151
__device__ __forceinline__ void ndc2world(const float* x, const Cam& c, float* w) {
    const float wz = 2.0f / (x[2] - 1.0f - 1e-6f);
    w[0] = -wz * x[0] * c.cxfx; w[1] = -wz * x[1] * c.cyfy; w[2] = wz;
}
// gradient w.r.t. the NDC point given the gradient g w.r.t. the world point
__device__ __forceinline__ void ndc2world_bwd(const float* x, const Cam& c, const float* g, float* d) {
    const float wz = 2.0f / (x[2] - 1.0f - 1e-6f);
    const float hz = 0.5f * wz * wz;                       // -d wz / d z
    d[0] = -wz * c.cxfx * g[0];
    d[1] = -wz * c.cyfy * g[1];
    d[2] = hz * (x[0] * c.cxfx * g[0] + x[1] * c.cyfy * g[1] - g[2]);
}

__device__ __forceinline__ Cam ray_cam(const NsffLossArgs& a, long long n) {
    const long long cam = a.cam_ids ? a.cam_ids[n] : 0;
    const float* K = a.Ks + cam * 9;
    return Cam{K[2] / K[0], K[5] / K[4]};
}

// losses.py:66-70 _project of one expected end point; returns validity of the projection
struct Proj { float uv[2]; float uvd[3]; float den; bool ok; const float* P; };
__device__ __forceinline__ Proj project(const NsffLossArgs& a, long long n, const float* xyz, const Cam& c, bool forward) {
    const long long cam = a.cam_ids ? a.cam_ids[n] : 0;
    long long t = a.ts[n] + (forward ? 1 : -1);
    t = t < 0 ? 0 : (t > a.max_t ? a.max_t : t);
    Proj p;
    p.P = a.Ps + (cam * a.n_frames + t) * 12;
    float w[3];
    ndc2world(xyz, c, w);
#pragma unroll
    for (int r = 0; r < 3; ++r) p.uvd[r] = p.P[4 * r] * w[0] + p.P[4 * r + 1] * w[1] + p.P[4 * r + 2] * w[2] + p.P[4 * r + 3];
    p.den = fabsf(p.uvd[2]) + 1e-8f;
    p.uv[0] = p.uvd[0] / p.den; p.uv[1] = p.uvd[1] / p.den;
    p.ok = p.uvd[2] > 0.f && (forward ? a.ts[n] < a.max_t : a.ts[n] > 0);
    return p;
}

__global__ __launch_bounds__(256) void loss_sums_kernel(const NsffLossArgs a) {
    const long long total = a.n_rays * a.n_samples;
    float s_fw = 0.f, s_bw = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        s_fw += a.disoccs_fw[i]; s_bw += a.disoccs_bw[i];
    }
    s_fw = wave_sum(s_fw); s_bw = wave_sum(s_bw);
    if ((threadIdx.x & 63) == 0) { atomicAdd(a.stats + ST_DOCCS, s_fw); atomicAdd(a.stats + ST_DOCCS + 1, s_bw); }
}

// block-wide sum of up to four values (1024 threads)
__device__ __forceinline__ void block_sum4(float (&v)[4], float* sRed) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = wave_sum(v[k]);
    __syncthreads();
    if (lane == 0) { sRed[wave * 4 + 0] = v[0]; sRed[wave * 4 + 1] = v[1]; sRed[wave * 4 + 2] = v[2]; sRed[wave * 4 + 3] = v[3]; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += sRed[w * 4 + k];
        v[k] = t;
    }
}

// medians by rank counting (rank = # smaller, ties broken by index): block (b, v) ranks 256 elements of vector v
__global__ __launch_bounds__(256) void loss_median_kernel(const NsffLossArgs a) {
    __shared__ float sX[MAXN];
    const int N = (int)a.n_rays, v = blockIdx.y;
    const float* src = v == 0 ? a.depth_fine : (v == 1 ? a.depth_coarse : a.disps);
    if (src == nullptr) return;
    for (int i = threadIdx.x; i < N; i += 256) sX[i] = v == 2 ? -src[i] : src[i];
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float x = sX[i];
    int rank = 0;
    for (int j = 0; j < N; ++j) { const float y = sX[j]; rank += (y < x || (y == x && j < i)) ? 1 : 0; }
    if (rank == (N - 1) / 2) {                              // torch.median: the lower of the two middle elements
        a.stats[ST_MED + v] = x; a.stats[ST_IDX + v] = __int_as_float(i);
    }
}

__global__ __launch_bounds__(1024) void loss_stats_kernel(const NsffLossArgs a) {
    __shared__ float sRed[64];
    __shared__ float sMed[3], sMad[3];
    const int N = (int)a.n_rays, tid = threadIdx.x;
    // ---- mean absolute deviations around the medians of loss_median_kernel ----
    {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const float m0 = a.stats[ST_MED], m1 = a.stats[ST_MED + 1], m2 = a.stats[ST_MED + 2];
        for (int i = tid; i < N; i += 1024) {
            acc[0] += fabsf(a.depth_fine[i] - m0);
            if (a.depth_coarse != nullptr) acc[1] += fabsf(a.depth_coarse[i] - m1);
            acc[2] += fabsf(-a.disps[i] - m2);
        }
        block_sum4(acc, sRed);
        if (tid == 0) {
            for (int v = 0; v < 3; ++v) { sMed[v] = a.stats[ST_MED + v]; sMad[v] = acc[v] / (float)N; a.stats[ST_MAD + v] = sMad[v]; }
        }
    }
    __syncthreads();
    (void)sMed; (void)sMad;
    // ---- means of the per-ray disocclusion weights, counts of valid projections ----
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < N; i += 1024) {
        acc[0] += a.disocc_fw[i]; acc[1] += a.disocc_bw[i];
        const Cam c = ray_cam(a, i);
        acc[2] += project(a, i, a.xyz_fw + 3LL * i, c, true).ok ? 1.f : 0.f;
        acc[3] += project(a, i, a.xyz_bw + 3LL * i, c, false).ok ? 1.f : 0.f;
    }
    block_sum4(acc, sRed);
    if (tid == 0) {
        a.stats[ST_DOCC] = acc[0] / (float)N; a.stats[ST_DOCC + 1] = acc[1] / (float)N;
        a.stats[ST_CNT] = acc[2]; a.stats[ST_CNT + 1] = acc[3];
    }
}

// Reduction of one term (blockIdx.y) over its population: rank counting like the medians (value descending, ties by
// index), K = int(topk * M) (all M for topk >= 1), coef = weight [rank < K] / K, term = sum of the selected values / K.
// per_ray holds the WEIGHTED values; a negative entry marks a ray outside the population (flow terms: invalid projection).
__global__ __launch_bounds__(256) void loss_select_kernel(const NsffLossArgs a) {
    __shared__ float sX[MAXN];
    __shared__ float sRed[8];
    const int N = (int)a.n_rays, k = blockIdx.y;
    const bool masked = k == T_FLOW_FW || k == T_FLOW_BW;
    const float* v = a.per_ray + (long long)k * N;
    int m_loc = 0;
    for (int i = threadIdx.x; i < N; i += 256) {
        const float x = v[i];
        sX[i] = x;
        m_loc += (masked && x < 0.f) ? 0 : 1;
    }
    float mf = wave_sum((float)m_loc);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sRed[threadIdx.x >> 6] = mf;
    __syncthreads();
    const int M = (int)(sRed[0] + sRed[1] + sRed[2] + sRed[3]);
    const long long K = a.topk >= 1.0 ? (long long)M : (long long)(a.topk * (double)M);
    const int i = blockIdx.x * 256 + threadIdx.x;
    float mine = 0.f;
    if (i < N) {
        const float x = sX[i];
        const bool in_pop = !(masked && x < 0.f);
        bool sel = in_pop && K > 0;
        if (sel && K < M) {
            int rank = 0;
            for (int j = 0; j < N; ++j) {
                const float y = sX[j];
                if (masked && y < 0.f) continue;
                rank += (y > x || (y == x && j < i)) ? 1 : 0;
            }
            sel = rank < K;
        }
        const float w = a.weights != nullptr ? a.weights[i] : 1.0f;
        a.coef[(long long)k * N + i] = sel ? w / (float)K : 0.f;
        mine = sel ? x / (float)K : 0.f;
    }
    mine = wave_sum(mine);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sRed[4 + (threadIdx.x >> 6)] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = sRed[4] + sRed[5] + sRed[6] + sRed[7];
        if (t != 0.f) atomicAdd(a.terms + k, t);
    }
}

// (backward) second-level sums of the depth term: y = (x - m) / s, target t = (-disp - m_t) / s_t, g_n = 2 (y - t) lambda c_n
// with c_n = coef[disp_l][n] (1 / N for the plain mean)
__global__ __launch_bounds__(1024) void loss_stats2_kernel(const NsffLossArgs a) {
    __shared__ float sRed[64];
    const int N = (int)a.n_rays, tid = threadIdx.x;
    const float lam = a.hyper[0];
    const float* cf = a.coef + (long long)T_DISP * N;
    for (int v = 0; v < 2; ++v) {
        const float* src = v == 0 ? a.depth_fine : a.depth_coarse;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (src != nullptr) {
            const float med = a.stats[ST_MED + v], mad = a.stats[ST_MAD + v];
            for (int i = tid; i < N; i += 1024) {
                const float y = (src[i] - med) / mad;
                const float t = (-a.disps[i] - a.stats[ST_MED + 2]) / a.stats[ST_MAD + 2];
                const float g = 2.0f * (y - t) * lam * cf[i];
                acc[0] += g; acc[1] += g * y; acc[2] += sgn(src[i] - med);
            }
        }
        block_sum4(acc, sRed);
        if (tid == 0) { a.stats[ST_SG + v] = acc[0]; a.stats[ST_SGY + v] = acc[1]; a.stats[ST_SSGN + v] = acc[2]; }
    }
}

// mode 1: the ray's (weighted) value of every term -> a.per_ray (11, N)
// mode 2: gradients of sum_k w[k] * term_k, term_k = sum_n coef[k][n] value_k[n] / weight_n
template <int MODE>
__global__ __launch_bounds__(256) void loss_rays_kernel(const NsffLossArgs a) {
    const int lane = threadIdx.x & 63;
    const int S = a.n_samples;
    const long long NR = a.n_rays;
    const float N = (float)a.n_rays;
    for (long long n = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); n < a.n_rays; n += (long long)gridDim.x * 4) {
    const float* st = a.stats;
    const float lam_d = a.hyper[0], lam_f = a.hyper[1], cross_w = a.hyper[2], lam_reg = a.hyper[3], lam_ent = a.hyper[4];
    // mode 2: c[k] = (upstream scalar of term k) * d term_k / d (this ray's value) -- weight, selection and 1 / K in one number
    float c[N_TERMS];
#pragma unroll
    for (int k = 0; k < N_TERMS; ++k) c[k] = MODE == 2 ? a.term_w[k] * a.coef[k * NR + n] : 0.f;
    float part[N_TERMS];
#pragma unroll
    for (int k = 0; k < N_TERMS; ++k) part[k] = 0.f;
    bool flow_ok[2] = {false, false};
    const Cam cam = ray_cam(a, n);

    // ---------------- per-ray terms (lane 0) ----------------
    if (lane == 0) {
        const float* tg = a.rgbs + 3 * n;
        // col_l (losses.py:74-76)
        float e = 0.f;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float d = a.rgb_fine[3 * n + ch] - tg[ch];
            e += d * d;
            if (MODE == 2) a.g_rgb_fine[3 * n + ch] = c[T_COL] * 2.0f * d / 3.0f;
            if (a.rgb_coarse != nullptr) {
                const float dc = a.rgb_coarse[3 * n + ch] - tg[ch];
                e += 0.1f * dc * dc;
                if (MODE == 2) a.g_rgb_coarse[3 * n + ch] = c[T_COL] * 0.2f * dc / 3.0f;
            }
        }
        part[T_COL] = e / 3.0f;
        // disp_l (losses.py:8-28, 77-80): gradient through the median and the mean absolute deviation (loss_stats2_kernel)
        const float t = (-a.disps[n] - st[ST_MED + 2]) / st[ST_MAD + 2];
        for (int v = 0; v < 2; ++v) {
            const float* src = v == 0 ? a.depth_fine : a.depth_coarse;
            if (src == nullptr) continue;
            const float m = st[ST_MED + v], sd = st[ST_MAD + v];
            const float y = (src[n] - m) / sd;
            part[T_DISP] += lam_d * (y - t) * (y - t);
            if (MODE == 2) {
                const float g = 2.0f * (y - t) * lam_d * a.coef[T_DISP * NR + n];
                const bool is_med = __float_as_int(st[ST_IDX + v]) == (int)n;
                float d = g / sd - st[ST_SGY + v] / sd * (sgn(src[n] - m) / N);
                if (is_med) d += -st[ST_SG + v] / sd + st[ST_SGY + v] / sd * (st[ST_SSGN + v] / N);
                (v == 0 ? a.g_depth_fine : a.g_depth_coarse)[n] = a.term_w[T_DISP] * d;
            }
        }
        // flow_fw_l / flow_bw_l (losses.py:96-110): per-ray value lambda_f / 2 * mean |uv - uv_target| over the valid rays
        for (int dir = 0; dir < 2; ++dir) {
            const float* x = (dir == 0 ? a.xyz_fw : a.xyz_bw) + 3 * n;
            const float* uvt = (dir == 0 ? a.uv_fw : a.uv_bw) + 2 * n;
            const Proj p = project(a, n, x, cam, dir == 0);
            float gx[3] = {0.f, 0.f, 0.f};
            flow_ok[dir] = p.ok;
            if (p.ok) {
                const float e0 = p.uv[0] - uvt[0], e1 = p.uv[1] - uvt[1];
                part[T_FLOW_FW + dir] = lam_f * 0.25f * (fabsf(e0) + fabsf(e1));
                if (MODE == 2) {
                    const float coef = c[T_FLOW_FW + dir] * lam_f * 0.25f;
                    const float gu0 = sgn(e0) * coef, gu1 = sgn(e1) * coef;
                    float guvd[3];
                    guvd[0] = gu0 / p.den; guvd[1] = gu1 / p.den;
                    guvd[2] = -(gu0 * p.uvd[0] + gu1 * p.uvd[1]) / (p.den * p.den) * sgn(p.uvd[2]);
                    float gw[3];
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) gw[ch] = p.P[ch] * guvd[0] + p.P[4 + ch] * guvd[1] + p.P[8 + ch] * guvd[2];
                    ndc2world_bwd(x, cam, gw, gx);
                }
            }
            if (MODE == 2) {
                float* g = (dir == 0 ? a.g_xyz_fw : a.g_xyz_bw) + 3 * n;
                g[0] = gx[0]; g[1] = gx[1]; g[2] = gx[2];
            }
        }
        // pho_l (losses.py:113-116)
        float ph = 0.f;
        for (int dir = 0; dir < 2; ++dir) {
            const float dc = (dir == 0 ? a.disocc_fw : a.disocc_bw)[n] / st[ST_DOCC + dir];
            const float* r = (dir == 0 ? a.rgb_fw : a.rgb_bw) + 3 * n;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const float d = r[ch] - tg[ch];
                ph += dc * d * d;
                if (MODE == 2) (dir == 0 ? a.g_rgb_fw : a.g_rgb_bw)[3 * n + ch] = c[T_PHO] * 2.0f * dc * d / 3.0f;
            }
        }
        part[T_PHO] = ph / 3.0f;
    }

    // ---------------- per-sample terms (lane = sample) ----------------
    const int n_keep = a.n_keep;
    const float c_cyc_fw = (N * S) / st[ST_DOCCS], c_cyc_bw = (N * S) / st[ST_DOCCS + 1];      // 1 / mean(disoccs)
    const float c_reg = lam_reg / (3.0f * (float)n_keep);
    const float c_sp = n_keep > 1 ? lam_reg / (3.0f * (float)(n_keep - 1)) : 0.f;
    const int th = a.thickness, pad = (th - 1) / 2;
    const long long base = n * S;
    for (int s = lane; s < S; s += 64) {
        const long long e = base + s;
        // entropy_l / cross_entropy_l (losses.py:83-95); the cross entropy sees the transient weights through a
        // 1 x thickness box filter (zero padded, detached)
        const float tw = a.t_weights[e], sw = a.s_weights[e];
        float dil = tw;
        if (th > 1) {
            dil = 0.f;
            for (int j = s - pad; j <= s + th - 1 - pad; ++j) if (j >= 0 && j < S) dil += a.t_weights[base + j];
        }
        part[T_ENT] += -tw * logf(tw + 1e-8f) * lam_ent;
        part[T_CE] += cross_w * dil * logf(sw + 1e-8f);
        if (MODE == 2) {
            a.g_t_weights[e] = c[T_ENT] * (-lam_ent) * (logf(tw + 1e-8f) + tw / (tw + 1e-8f));
            a.g_s_weights[e] = c[T_CE] * cross_w * dil / (sw + 1e-8f);
        }
        // cyc_l (losses.py:117-120)
        const float* x0 = a.xyzs_fine + 3 * e;
        {
            const float df = a.disoccs_fw[e] * c_cyc_fw, db = a.disoccs_bw[e] * c_cyc_bw;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const float ef = a.xyzs_fw_bw[3 * e + ch] - x0[ch], eb = a.xyzs_bw_fw[3 * e + ch] - x0[ch];
                part[T_CYC] += (df * fabsf(ef) + db * fabsf(eb)) / (3.0f * S);
                if (MODE == 2) {
                    a.g_xyzs_fw_bw[3 * e + ch] = c[T_CYC] * df * sgn(ef) / (3.0f * S);
                    a.g_xyzs_bw_fw[3 * e + ch] = c[T_CYC] * db * sgn(eb) / (3.0f * S);
                }
            }
        }
        // scene-flow regularisers in world space, near samples only (losses.py:122-133)
        float gf[3] = {0.f, 0.f, 0.f}, gb[3] = {0.f, 0.f, 0.f};
        if (s < n_keep) {
            float p0[3], pf[3], pb[3];
            ndc2world(x0, cam, p0); ndc2world(a.xyzs_fw + 3 * e, cam, pf); ndc2world(a.xyzs_bw + 3 * e, cam, pb);
            float gwf[3], gwb[3];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const float tsm = pf[ch] + pb[ch] - 2.0f * p0[ch], mf = pf[ch] - p0[ch], mb = pb[ch] - p0[ch];
                part[T_TEMP] += c_reg * fabsf(tsm);
                part[T_MIN] += c_reg * (fabsf(mf) + fabsf(mb));
                gwf[ch] = (c[T_TEMP] * sgn(tsm) + c[T_MIN] * sgn(mf)) * c_reg;
                gwb[ch] = (c[T_TEMP] * sgn(tsm) + c[T_MIN] * sgn(mb)) * c_reg;
            }
            // spatial smoothness: pairs (s-1, s) and (s, s+1)
            for (int side = 0; side < 2; ++side) {
                const int s1 = side == 0 ? s - 1 : s;                 // pair (s1, s1+1)
                if (s1 < 0 || s1 + 1 >= n_keep) continue;
                const long long e1 = base + s1, e2 = e1 + 1;
                float a0[3], a1[3], f0[3], f1[3], b0[3], b1[3];
                ndc2world(a.xyzs_fine + 3 * e1, cam, a0); ndc2world(a.xyzs_fine + 3 * e2, cam, a1);
                ndc2world(a.xyzs_fw + 3 * e1, cam, f0); ndc2world(a.xyzs_fw + 3 * e2, cam, f1);
                ndc2world(a.xyzs_bw + 3 * e1, cam, b0); ndc2world(a.xyzs_bw + 3 * e2, cam, b1);
                const float dx = a1[0] - a0[0], dy = a1[1] - a0[1], dz = a1[2] - a0[2];
                const float near = expf(-2.0f * sqrtf(dx * dx + dy * dy + dz * dz));
                const float sg = side == 0 ? 1.0f : -1.0f;            // this sample is the second / the first of the pair
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    const float df = (f1[ch] - a1[ch]) - (f0[ch] - a0[ch]), db = (b1[ch] - a1[ch]) - (b0[ch] - a0[ch]);
                    if (side == 1) part[T_SP] += c_sp * (fabsf(df) + fabsf(db)) * near;      // each pair counted once
                    gwf[ch] += c[T_SP] * sg * sgn(df) * near * c_sp;
                    gwb[ch] += c[T_SP] * sg * sgn(db) * near * c_sp;
                }
            }
            if (MODE == 2) {
                ndc2world_bwd(a.xyzs_fw + 3 * e, cam, gwf, gf);
                ndc2world_bwd(a.xyzs_bw + 3 * e, cam, gwb, gb);
            }
        }
        if (MODE == 2) {
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) { a.g_xyzs_fw[3 * e + ch] = gf[ch]; a.g_xyzs_bw[3 * e + ch] = gb[ch]; }
        }
    }
    if (MODE == 1) {      // this ray's value of every term, times its weight; a negative entry = outside a flow term's population
        const float wn = a.weights != nullptr ? a.weights[n] : 1.0f;
        const bool ok_fw = __shfl(flow_ok[0] ? 1 : 0, 0) != 0, ok_bw = __shfl(flow_ok[1] ? 1 : 0, 0) != 0;
#pragma unroll
        for (int k = 0; k < N_TERMS; ++k) {
            float v = wave_sum(part[k]) * wn;
            if (k == T_FLOW_FW && !ok_fw) v = -1.0f;
            if (k == T_FLOW_BW && !ok_bw) v = -1.0f;
            if (lane == 0) a.per_ray[k * NR + n] = v;
        }
    }
    }   // rays of this wave
}

}  // namespace

extern "C" {

int nsff_nerfw_loss(const NsffLossArgs* args, int mode, void* stream) {
    if (!args) return NSFF_ERR_NULL;
    const NsffLossArgs& a = *args;
    if (a.n_rays < 1 || a.n_rays > MAXN || a.n_samples < 2 || a.n_frames < 1 || a.max_t < 0) return NSFF_ERR_INVALID;
    if (a.n_keep < 1 || a.n_keep > a.n_samples) return NSFF_ERR_INVALID;
    if (mode != 1 && mode != 2) return NSFF_ERR_INVALID;
    if (!a.rgb_fine || !a.depth_fine || !a.rgbs || !a.disps || !a.ts || !a.Ks || !a.Ps || !a.uv_fw || !a.uv_bw ||
        !a.t_weights || !a.s_weights || !a.xyz_fw || !a.xyz_bw || !a.rgb_fw || !a.rgb_bw || !a.disocc_fw || !a.disocc_bw ||
        !a.disoccs_fw || !a.disoccs_bw || !a.xyzs_fw_bw || !a.xyzs_bw_fw || !a.xyzs_fine || !a.xyzs_fw || !a.xyzs_bw ||
        !a.stats || !a.hyper) return NSFF_ERR_NULL;
    if ((a.rgb_coarse == nullptr) != (a.depth_coarse == nullptr)) return NSFF_ERR_INVALID;
    if (!a.per_ray || !a.coef || a.thickness < 1 || !(a.topk > 0.0)) return NSFF_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const unsigned blocks = (unsigned)((a.n_rays + 3) / 4);
    if (mode == 1) {
        if (!a.terms) return NSFF_ERR_NULL;
        hipError_t e = hipMemsetAsync(a.stats, 0, ST_SIZE * sizeof(float), st);
        if (e == hipSuccess) e = hipMemsetAsync(a.terms, 0, N_TERMS * sizeof(float), st);
        if (e != hipSuccess) return nsff_hip_fail(e);
        const long long total = a.n_rays * a.n_samples;
        hipLaunchKernelGGL(loss_sums_kernel, dim3((unsigned)std::min<long long>((total + 2047) / 2048, 256)), dim3(256), 0, st, a);
        hipLaunchKernelGGL(loss_median_kernel, dim3((unsigned)((a.n_rays + 255) / 256), 3), dim3(256), 0, st, a);
        hipLaunchKernelGGL(loss_stats_kernel, dim3(1), dim3(1024), 0, st, a);
        hipLaunchKernelGGL(loss_rays_kernel<1>, dim3(blocks), dim3(256), 0, st, a);
        hipLaunchKernelGGL(loss_select_kernel, dim3((unsigned)((a.n_rays + 255) / 256), N_TERMS), dim3(256), 0, st, a);
    } else {
        if (!a.term_w || !a.g_rgb_fine || !a.g_depth_fine || !a.g_t_weights || !a.g_s_weights || !a.g_xyz_fw || !a.g_xyz_bw ||
            !a.g_rgb_fw || !a.g_rgb_bw || !a.g_xyzs_fw_bw || !a.g_xyzs_bw_fw || !a.g_xyzs_fw || !a.g_xyzs_bw) return NSFF_ERR_NULL;
        if (a.rgb_coarse && (!a.g_rgb_coarse || !a.g_depth_coarse)) return NSFF_ERR_NULL;
        hipLaunchKernelGGL(loss_stats2_kernel, dim3(1), dim3(1024), 0, st, a);
        hipLaunchKernelGGL(loss_rays_kernel<2>, dim3(blocks), dim3(256), 0, st, a);
    }
    return nsff_launch_status();
}

}  // extern "C"
