// N1: the NSFF training objective as four small launches forward + one backward (reference losses.py:8-28 shiftscale_invariant_depthloss,
// :31-171 NeRFWLoss) instead of the ~580 elementwise / reduction kernels the torch expression of its forward and
// backward needs.  All eleven terms, train-mode NSFF configuration (flows + disocclusion present, topk == 1, no
// per-ray weights, thickness == 1); every term is reduced to its scalar mean like the reference does.
//
//   loss_sums_kernel   grid-stride sums of the per-sample disocclusion weights (their means normalise cyc_l)
//   loss_median_kernel medians of depth_fine / depth_coarse / -disp by rank counting in LDS (torch.median = lower median)
//   loss_stats_kernel  ONE workgroup: mean absolute deviations around them, the second-level sums their gradients
//                      need, means of the per-ray disocclusion weights, counts of valid flow projections
//   loss_rays_kernel   one wavefront per ray, lane = sample: term sums (mode 1) or the gradient of
//                      sum_k w_k * term_k w.r.t. every consumed render tensor (mode 2, w = upstream scalars)
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
    // ---- second-level sums of the depth terms: y = (x - m) / s, target t = (-disp - m_t) / s_t, g = 2 (y - t) lambda / N ----
    const float lam = a.hyper[0];
    for (int v = 0; v < 2; ++v) {
        const float* src = v == 0 ? a.depth_fine : a.depth_coarse;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (src != nullptr) {
            for (int i = tid; i < N; i += 1024) {
                const float y = (src[i] - sMed[v]) / sMad[v];
                const float t = (-a.disps[i] - sMed[2]) / sMad[2];
                const float g = 2.0f * (y - t) * lam / (float)N;
                acc[0] += g; acc[1] += g * y; acc[2] += sgn(src[i] - sMed[v]);
            }
        }
        block_sum4(acc, sRed);
        if (tid == 0) { a.stats[ST_SG + v] = acc[0]; a.stats[ST_SGY + v] = acc[1]; a.stats[ST_SSGN + v] = acc[2]; }
    }
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

// mode 1: term sums -> a.terms (11 floats, atomically accumulated; the buffer is zeroed by the host side)
// mode 2: gradients of sum_k w[k] * term_k
template <int MODE>
__global__ __launch_bounds__(256) void loss_rays_kernel(const NsffLossArgs a) {
    const int lane = threadIdx.x & 63;
    const int S = a.n_samples;
    const float N = (float)a.n_rays;
    __shared__ float sPart[4][N_TERMS];
    float tot[N_TERMS];
#pragma unroll
    for (int k = 0; k < N_TERMS; ++k) tot[k] = 0.f;
    for (long long n = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); n < a.n_rays; n += (long long)gridDim.x * 4) {
    const float* st = a.stats;
    const float lam_d = a.hyper[0], lam_f = a.hyper[1], cross_w = a.hyper[2], lam_reg = a.hyper[3], lam_ent = a.hyper[4];
    float w[N_TERMS];
#pragma unroll
    for (int k = 0; k < N_TERMS; ++k) w[k] = MODE == 2 ? a.term_w[k] : 1.0f;
    float part[N_TERMS];
#pragma unroll
    for (int k = 0; k < N_TERMS; ++k) part[k] = 0.f;
    const Cam cam = ray_cam(a, n);

    // ---------------- per-ray terms (lane 0) ----------------
    if (lane == 0) {
        const float* tg = a.rgbs + 3 * n;
        // col_l (losses.py:74-76)
        float e = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float d = a.rgb_fine[3 * n + c] - tg[c];
            e += d * d;
            if (MODE == 2) a.g_rgb_fine[3 * n + c] = w[T_COL] * 2.0f * d / (3.0f * N);
            if (a.rgb_coarse != nullptr) {
                const float dc = a.rgb_coarse[3 * n + c] - tg[c];
                e += 0.1f * dc * dc;
                if (MODE == 2) a.g_rgb_coarse[3 * n + c] = w[T_COL] * 0.2f * dc / (3.0f * N);
            }
        }
        part[T_COL] = e / 3.0f;
        // disp_l (losses.py:8-28, 77-80): gradient through the median and the mean absolute deviation
        const float t = (-a.disps[n] - st[ST_MED + 2]) / st[ST_MAD + 2];
        for (int v = 0; v < 2; ++v) {
            const float* src = v == 0 ? a.depth_fine : a.depth_coarse;
            if (src == nullptr) continue;
            const float m = st[ST_MED + v], s = st[ST_MAD + v];
            const float y = (src[n] - m) / s;
            part[T_DISP] += lam_d * (y - t) * (y - t);
            if (MODE == 2) {
                const float g = 2.0f * (y - t) * lam_d / N;
                const bool is_med = __float_as_int(st[ST_IDX + v]) == (int)n;
                float d = g / s - st[ST_SGY + v] / s * (sgn(src[n] - m) / N);
                if (is_med) d += -st[ST_SG + v] / s + st[ST_SGY + v] / s * (st[ST_SSGN + v] / N);
                (v == 0 ? a.g_depth_fine : a.g_depth_coarse)[n] = w[T_DISP] * d;
            }
        }
        // flow_fw_l / flow_bw_l (losses.py:96-110), masked means
        for (int dir = 0; dir < 2; ++dir) {
            const float* x = (dir == 0 ? a.xyz_fw : a.xyz_bw) + 3 * n;
            const float* uvt = (dir == 0 ? a.uv_fw : a.uv_bw) + 2 * n;
            const Proj p = project(a, n, x, cam, dir == 0);
            const float cnt = fmaxf(st[ST_CNT + dir], 1.0f);
            float gx[3] = {0.f, 0.f, 0.f};
            if (p.ok) {
                const float e0 = p.uv[0] - uvt[0], e1 = p.uv[1] - uvt[1];
                part[T_FLOW_FW + dir] = (fabsf(e0) + fabsf(e1));           // scaled at the end
                if (MODE == 2) {
                    const float coef = w[T_FLOW_FW + dir] * lam_f / (4.0f * cnt);
                    const float gu0 = sgn(e0) * coef, gu1 = sgn(e1) * coef;
                    float guvd[3];
                    guvd[0] = gu0 / p.den; guvd[1] = gu1 / p.den;
                    guvd[2] = -(gu0 * p.uvd[0] + gu1 * p.uvd[1]) / (p.den * p.den) * sgn(p.uvd[2]);
                    float gw[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) gw[c] = p.P[c] * guvd[0] + p.P[4 + c] * guvd[1] + p.P[8 + c] * guvd[2];
                    ndc2world_bwd(x, cam, gw, gx);
                }
            }
            if (MODE == 2) {
                float* g = (dir == 0 ? a.g_xyz_fw : a.g_xyz_bw) + 3 * n;
                g[0] = gx[0]; g[1] = gx[1]; g[2] = gx[2];
            }
            part[T_FLOW_FW + dir] *= lam_f / (4.0f * cnt) * N;              // (the common 1/N is applied below)
        }
        // pho_l (losses.py:113-116)
        float ph = 0.f;
        for (int dir = 0; dir < 2; ++dir) {
            const float dc = (dir == 0 ? a.disocc_fw : a.disocc_bw)[n] / st[ST_DOCC + dir];
            const float* r = (dir == 0 ? a.rgb_fw : a.rgb_bw) + 3 * n;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float d = r[c] - tg[c];
                ph += dc * d * d;
                if (MODE == 2) (dir == 0 ? a.g_rgb_fw : a.g_rgb_bw)[3 * n + c] = w[T_PHO] * 2.0f * dc * d / (3.0f * N);
            }
        }
        part[T_PHO] = ph / 3.0f;
    }

    // ---------------- per-sample terms (lane = sample) ----------------
    const int n_keep = a.n_keep;
    const float c_cyc_fw = (N * S) / st[ST_DOCCS], c_cyc_bw = (N * S) / st[ST_DOCCS + 1];      // 1 / mean(disoccs)
    const float c_reg = lam_reg / (3.0f * (float)n_keep);                                    // per ray; 1/N below
    const float c_sp = n_keep > 1 ? lam_reg / (3.0f * (float)(n_keep - 1)) : 0.f;
    const long long base = n * S;
    for (int s = lane; s < S; s += 64) {
        const long long e = base + s;
        // entropy_l / cross_entropy_l (losses.py:83-95)
        const float tw = a.t_weights[e], sw = a.s_weights[e];
        part[T_ENT] += -tw * logf(tw + 1e-8f) * lam_ent;
        part[T_CE] += cross_w * tw * logf(sw + 1e-8f);
        if (MODE == 2) {
            a.g_t_weights[e] = w[T_ENT] * (-lam_ent / N) * (logf(tw + 1e-8f) + tw / (tw + 1e-8f));
            a.g_s_weights[e] = w[T_CE] * (cross_w / N) * tw / (sw + 1e-8f);
        }
        // cyc_l (losses.py:117-120)
        const float* x0 = a.xyzs_fine + 3 * e;
        {
            const float df = a.disoccs_fw[e] * c_cyc_fw, db = a.disoccs_bw[e] * c_cyc_bw;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float ef = a.xyzs_fw_bw[3 * e + c] - x0[c], eb = a.xyzs_bw_fw[3 * e + c] - x0[c];
                part[T_CYC] += (df * fabsf(ef) + db * fabsf(eb)) / (3.0f * S);
                if (MODE == 2) {
                    a.g_xyzs_fw_bw[3 * e + c] = w[T_CYC] * df * sgn(ef) / (3.0f * S * N);
                    a.g_xyzs_bw_fw[3 * e + c] = w[T_CYC] * db * sgn(eb) / (3.0f * S * N);
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
            for (int c = 0; c < 3; ++c) {
                const float tsm = pf[c] + pb[c] - 2.0f * p0[c], mf = pf[c] - p0[c], mb = pb[c] - p0[c];
                part[T_TEMP] += c_reg * fabsf(tsm);
                part[T_MIN] += c_reg * (fabsf(mf) + fabsf(mb));
                gwf[c] = (w[T_TEMP] * sgn(tsm) + w[T_MIN] * sgn(mf)) * c_reg / N;
                gwb[c] = (w[T_TEMP] * sgn(tsm) + w[T_MIN] * sgn(mb)) * c_reg / N;
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
                for (int c = 0; c < 3; ++c) {
                    const float df = (f1[c] - a1[c]) - (f0[c] - a0[c]), db = (b1[c] - a1[c]) - (b0[c] - a0[c]);
                    if (side == 1) part[T_SP] += c_sp * (fabsf(df) + fabsf(db)) * near;      // each pair counted once
                    gwf[c] += w[T_SP] * sg * sgn(df) * near * c_sp / N;
                    gwb[c] += w[T_SP] * sg * sgn(db) * near * c_sp / N;
                }
            }
            if (MODE == 2) {
                ndc2world_bwd(a.xyzs_fw + 3 * e, cam, gwf, gf);
                ndc2world_bwd(a.xyzs_bw + 3 * e, cam, gwb, gb);
            }
        }
        if (MODE == 2) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { a.g_xyzs_fw[3 * e + c] = gf[c]; a.g_xyzs_bw[3 * e + c] = gb[c]; }
        }
    }
    if (MODE == 1) {
#pragma unroll
        for (int k = 0; k < N_TERMS; ++k) tot[k] += part[k];
    }
    }   // rays of this wave
    if (MODE == 1) {      // one atomic per term and WORKGROUP (they serialise in the L2: ~13 ns each)
#pragma unroll
        for (int k = 0; k < N_TERMS; ++k) {
            const float t = wave_sum(tot[k]) / N;
            if (lane == 0) sPart[threadIdx.x >> 6][k] = t;
        }
        __syncthreads();
        if (threadIdx.x < N_TERMS) {
            const float t = sPart[0][threadIdx.x] + sPart[1][threadIdx.x] + sPart[2][threadIdx.x] + sPart[3][threadIdx.x];
            if (t != 0.f) atomicAdd(a.terms + threadIdx.x, t);
        }
    }
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
        hipLaunchKernelGGL(loss_rays_kernel<1>, dim3(std::min(blocks, 64u)), dim3(256), 0, st, a);
    } else {
        if (!a.term_w || !a.g_rgb_fine || !a.g_depth_fine || !a.g_t_weights || !a.g_s_weights || !a.g_xyz_fw || !a.g_xyz_bw ||
            !a.g_rgb_fw || !a.g_rgb_bw || !a.g_xyzs_fw_bw || !a.g_xyzs_bw_fw || !a.g_xyzs_fw || !a.g_xyzs_bw) return NSFF_ERR_NULL;
        if (a.rgb_coarse && (!a.g_rgb_coarse || !a.g_depth_coarse)) return NSFF_ERR_NULL;
        hipLaunchKernelGGL(loss_rays_kernel<2>, dim3(blocks), dim3(256), 0, st, a);
    }
    return nsff_launch_status();
}

}  // extern "C"
