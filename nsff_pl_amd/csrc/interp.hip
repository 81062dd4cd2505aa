// N2: plane splatting + MPI compositing for time interpolation (reference models/rendering.py:365-460,
// models/softsplat.py:6-44,303-326).  Scatter work: output tiles are OWNED by workgroups that accumulate the
// samples landing in them in LDS (ds_add_f32) and write them with plain stores; only samples that move farther
// than the halo use global atomics.  Compositing is one wavefront per pixel with a segmented product scan over
// the planes (same scheme as composite_kernel in rays.hip).
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

// Where sample (pixel px,py; plane s) of the source frame lands: bilinear corner (nwx, nwy) and the four weights.
struct Landing { int nwx, nwy; float w[4]; bool finite; };

__device__ __forceinline__ Landing project_sample(const NsffSplatArgs& a, long long idx, int px, int py) {
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
    Landing L;
    L.finite = flx >= -2.0f && flx <= (float)a.W && fly >= -2.0f && fly <= (float)a.H;   // false for NaN / far outside
    L.nwx = L.finite ? (int)flx : -4;
    L.nwy = L.finite ? (int)fly : -4;
    const float sex = (float)(L.nwx + 1), sey = (float)(L.nwy + 1);
    L.w[0] = (sex - ox) * (sey - oy);                 // north-west
    L.w[1] = (ox - (float)L.nwx) * (sey - oy);        // north-east
    L.w[2] = (sex - ox) * (oy - (float)L.nwy);        // south-west
    L.w[3] = (ox - (float)L.nwx) * (oy - (float)L.nwy);
    return L;
}

// A sample is "near" when its landing cell is within HALO pixels of its own pixel: then every output tile it
// touches sees it inside its halo and accumulates it in LDS.  Everything else goes through global atomics.
constexpr int TILE_X = 32, TILE_Y = 8, HALO = 4, PL = 8;           // output tile, halo, planes per workgroup
constexpr int REG_X = TILE_X + 2 * HALO, REG_Y = TILE_Y + 2 * HALO;
__device__ __forceinline__ bool is_near(const Landing& L, int px, int py) {
    const int dx = L.nwx - px, dy = L.nwy - py;
    return L.finite && dx >= -HALO && dx < HALO && dy >= -HALO && dy < HALO;
}

// Pass 1: each workgroup OWNS a TILE_X x TILE_Y block of output pixels for PL consecutive planes, accumulates the
// near samples of the surrounding (tile + halo) source region in LDS (ds_add_f32) and writes the block with plain
// stores -- no global atomics, no memset; the 2.5x redundant projection work is cheap next to the atomics it saves.
__global__ __launch_bounds__(256) void splat_tiles_kernel(const NsffSplatArgs a) {
    __shared__ float sAcc[TILE_X * TILE_Y * PL * 5];
    const int S = a.n_planes;
    const int tiles_x = (a.W + TILE_X - 1) / TILE_X;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int x0 = tx * TILE_X, y0 = ty * TILE_Y, s0 = blockIdx.y * PL;
    for (int i = threadIdx.x; i < TILE_X * TILE_Y * PL * 5; i += 256) sAcc[i] = 0.f;
    __syncthreads();
    for (int item = threadIdx.x; item < REG_X * REG_Y * PL; item += 256) {
        const int j = item % PL, rp = item / PL;
        const int px = x0 - HALO + rp % REG_X, py = y0 - HALO + rp / REG_X, s = s0 + j;
        if (px < 0 || px >= a.W || py < 0 || py >= a.H || s >= S) continue;
        const long long idx = ((long long)py * a.W + px) * S + s;
        const Landing L = project_sample(a, idx, px, py);
        if (!is_near(L, px, py)) continue;
        const float* cp = a.rgb + idx * 3;
        const float src[5] = {cp[0], cp[1], cp[2], a.alpha[idx], 1.0f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int cx = L.nwx + (k & 1) - x0, cy = L.nwy + (k >> 1) - y0;        // inside the owned tile?
            if (cx < 0 || cx >= TILE_X || cy < 0 || cy >= TILE_Y) continue;
            if (L.nwx + (k & 1) >= a.W || L.nwy + (k >> 1) >= a.H) continue;        // (tile may overhang the image)
            float* dst = sAcc + ((cy * TILE_X + cx) * PL + j) * 5;
#pragma unroll
            for (int c = 0; c < 5; ++c) atomicAdd(dst + c, src[c] * L.w[k]);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < TILE_X * TILE_Y * PL; i += 256) {
        const int j = i % PL, pix = i / PL;
        const int px = x0 + pix % TILE_X, py = y0 + pix / TILE_X, s = s0 + j;
        if (px >= a.W || py >= a.H || s >= S) continue;
        const float* v = sAcc + i * 5;
        float4* dst = reinterpret_cast<float4*>(a.accum + (((long long)py * a.W + px) * S + s) * 8);
        dst[0] = make_float4(v[0], v[1], v[2], v[3]);
        dst[1] = make_float4(v[4], 0.f, 0.f, 0.f);
    }
}

// Pass 2: samples that move farther than the halo (rare for a trained flow field) are added with global atomics.
__global__ __launch_bounds__(256) void splat_far_kernel(const NsffSplatArgs a) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;       // (pixel, plane), plane fastest
    const long long total = (long long)a.H * a.W * a.n_planes;
    if (idx >= total) return;
    const int S = a.n_planes;
    const long long pix = idx / S;
    const int s = (int)(idx - pix * S);
    const int px = (int)(pix % a.W), py = (int)(pix / a.W);
    const Landing L = project_sample(a, idx, px, py);
    if (!L.finite || is_near(L, px, py)) return;
    const float* cp = a.rgb + idx * 3;
    const float src[5] = {cp[0], cp[1], cp[2], a.alpha[idx], 1.0f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int cx = L.nwx + (k & 1), cy = L.nwy + (k >> 1);
        if (cx < 0 || cx >= a.W || cy < 0 || cy >= a.H) continue;
        float* dst = a.accum + (((long long)cy * a.W + cx) * S + s) * 8;
#pragma unroll
        for (int c = 0; c < 5; ++c) unsafeAtomicAdd(dst + c, src[c] * L.w[k]);
    }
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
    const unsigned tiles = (unsigned)(((a.W + TILE_X - 1) / TILE_X) * ((a.H + TILE_Y - 1) / TILE_Y));
    hipLaunchKernelGGL(splat_tiles_kernel, dim3(tiles, (unsigned)((a.n_planes + PL - 1) / PL)), dim3(256), 0,
                       (hipStream_t)stream, a);
    hipLaunchKernelGGL(splat_far_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
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
