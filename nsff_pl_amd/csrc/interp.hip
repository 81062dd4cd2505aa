// N2: plane splatting + MPI compositing for time interpolation (reference models/rendering.py:365-460,
// models/softsplat.py:6-44,303-326).  HBM/atomic-bound scatter work: one thread per (pixel, plane) with the
// plane index fastest so that source reads are fully coalesced and the 5 atomics of one bilinear corner fall
// into one 32-byte sector of the (pixel, plane, 8) accumulator; compositing is one wavefront per pixel with a
// segmented product scan over the planes (same scheme as composite_kernel in rays.hip).
#include "nsff_common.h"

namespace {

constexpr int WAVES_PER_BLOCK = 4;

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

// datasets/ray_utils.py:127-151, (N,3) branch, same operation order
__device__ __forceinline__ void ndc2world(const float x, const float y, const float z, const float* K4, float* w) {
    const float rz = 2.0f / (z - 1.0f - 1e-6f);
    w[0] = -rz * x * K4[2] / K4[0];
    w[1] = -rz * y * K4[3] / K4[1];
    w[2] = rz;
}

__global__ __launch_bounds__(256) void splat_planes_kernel(const NsffSplatArgs a) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;       // (pixel, plane), plane fastest
    const long long total = (long long)a.H * a.W * a.n_planes;
    if (idx >= total) return;
    const int S = a.n_planes;
    const long long pix = idx / S;
    const int s = (int)(idx - pix * S);
    const int px = (int)(pix % a.W), py = (int)(pix / a.W);

    const float* xp = a.xyz + idx * 3;
    const float* fp = a.flow + idx * 3;
    const float x = xp[0], y = xp[1], z = xp[2];
    float pw[3], qw[3];
    ndc2world(x, y, z, a.K4, pw);
    ndc2world(x + fp[0], y + fp[1], z + fp[2], a.K4, qw);
#pragma unroll
    for (int c = 0; c < 3; ++c) qw[c] = pw[c] + a.scale * (qw[c] - pw[c]);
    float uvd[3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
        uvd[r] = (a.P[4 * r] * qw[0] + a.P[4 * r + 1] * qw[1] + a.P[4 * r + 2] * qw[2]) + a.P[4 * r + 3];
    // optical flow on this plane, then the splat target (rendering.py:411-414, softsplat.py:16-17)
    const float ox = (float)px + (uvd[0] / uvd[2] - (float)px);
    const float oy = (float)py + (uvd[1] / uvd[2] - (float)py);
    const float flx = floorf(ox), fly = floorf(oy);
    if (!(flx >= -2.0f && flx <= (float)a.W && fly >= -2.0f && fly <= (float)a.H)) return;   // all 4 corners outside (or NaN)
    const int nwx = (int)flx, nwy = (int)fly;
    const float sex = (float)(nwx + 1), sey = (float)(nwy + 1);
    const float wnw = (sex - ox) * (sey - oy), wne = (ox - (float)nwx) * (sey - oy);
    const float wsw = (sex - ox) * (oy - (float)nwy), wse = (ox - (float)nwx) * (oy - (float)nwy);

    const float* cp = a.rgb + idx * 3;
    const float src[5] = {cp[0], cp[1], cp[2], a.alpha[idx], 1.0f};
    auto corner = [&](int cx, int cy, float wgt) {
        if (cx < 0 || cx >= a.W || cy < 0 || cy >= a.H) return;
        float* dst = a.accum + (((long long)cy * a.W + cx) * S + s) * 8;
#pragma unroll
        for (int c = 0; c < 5; ++c) unsafeAtomicAdd(dst + c, src[c] * wgt);
    };
    corner(nwx, nwy, wnw);
    corner(nwx + 1, nwy, wne);
    corner(nwx, nwy + 1, wsw);
    corner(nwx + 1, nwy + 1, wse);
}

__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void mpi_composite_kernel(const NsffMpiArgs a) {
    const int lane = threadIdx.x & 63;
    const long long pix = (long long)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (pix >= (long long)a.H * a.W) return;
    const int S = a.n_planes;
    const float dt = a.dt, omdt = 1.0f - a.dt;
    float carry = 1.0f;                       // transmittance 1 - A in front of this chunk
    float acc[4] = {0.f, 0.f, 0.f, 0.f};      // r, g, b, depth
    for (int s0 = 0; s0 < S; s0 += 64) {
        const int s = s0 + lane;
        float c_rgb[3] = {0.f, 0.f, 0.f}, c_a = 0.f, z = 0.f;
        if (s < S) {
            const long long e = pix * S + s;
            const float4 f0 = *reinterpret_cast<const float4*>(a.accum_fw + e * 8);
            const float4 b0 = *reinterpret_cast<const float4*>(a.accum_bw + e * 8);
            float fn = a.accum_fw[e * 8 + 4], bn = a.accum_bw[e * 8 + 4];
            fn = fn == 0.0f ? 1.0f : fn;
            bn = bn == 0.0f ? 1.0f : bn;
            const float fa = f0.w / fn, ba = b0.w / bn;
            const float sa = a.static_alpha[e];
            const float* sr = a.static_rgb + e * 3;
            const float fr[3] = {f0.x / fn, f0.y / fn, f0.z / fn}, br[3] = {b0.x / bn, b0.y / bn, b0.z / bn};
#pragma unroll
            for (int c = 0; c < 3; ++c) c_rgb[c] = fr[c] * fa * omdt + br[c] * ba * dt + sr[c] * sa;
            c_a = 1.0f - (1.0f - (fa * omdt + ba * dt)) * (1.0f - sa);
            z = a.zs[e];
        }
        const float incl = wave_scan_mul(1.0f - c_a, lane);
        float excl = __shfl_up(incl, 1);
        if (lane == 0) excl = 1.0f;
        const float T = carry * excl;
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] += T * c_rgb[c];
        acc[3] += T * c_a * z;
        carry *= __shfl(incl, 63);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = wave_sum(acc[c]);
    if (lane == 0) {
        a.rgb[pix * 3 + 0] = acc[0];
        a.rgb[pix * 3 + 1] = acc[1];
        a.rgb[pix * 3 + 2] = acc[2];
        a.depth[pix] = acc[3];
    }
}

}  // namespace

extern "C" {

int nsff_splat_planes(const NsffSplatArgs* args, void* stream) {
    if (!args) return NSFF_ERR_NULL;
    const NsffSplatArgs& a = *args;
    if (a.H < 1 || a.W < 1 || a.n_planes < 1) return NSFF_ERR_INVALID;
    if (!a.xyz || !a.flow || !a.rgb || !a.alpha || !a.accum) return NSFF_ERR_NULL;
    if (reinterpret_cast<uintptr_t>(a.accum) & 15) return NSFF_ERR_ALIGN;
    const long long total = (long long)a.H * a.W * a.n_planes;
    hipError_t e = hipMemsetAsync(a.accum, 0, (size_t)total * 8 * sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) return nsff_hip_fail(e);
    hipLaunchKernelGGL(splat_planes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return nsff_launch_status();
}

int nsff_mpi_composite(const NsffMpiArgs* args, void* stream) {
    if (!args) return NSFF_ERR_NULL;
    const NsffMpiArgs& a = *args;
    if (a.H < 1 || a.W < 1 || a.n_planes < 1) return NSFF_ERR_INVALID;
    if (!a.accum_fw || !a.accum_bw || !a.static_rgb || !a.static_alpha || !a.zs || !a.rgb || !a.depth) return NSFF_ERR_NULL;
    if ((reinterpret_cast<uintptr_t>(a.accum_fw) | reinterpret_cast<uintptr_t>(a.accum_bw)) & 15) return NSFF_ERR_ALIGN;
    const long long pixels = (long long)a.H * a.W;
    hipLaunchKernelGGL(mpi_composite_kernel, dim3((unsigned)((pixels + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK)),
                       dim3(64 * WAVES_PER_BLOCK), 0, (hipStream_t)stream, a);
    return nsff_launch_status();
}

}  // extern "C"
