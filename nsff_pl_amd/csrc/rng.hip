// The random draws of one render_rays call in ONE launch.  The reference draws with torch.rand / torch.randn where it needs the
// numbers (models/rendering.py:321, 207, 213, 128 twice, 338-340, per model pass): nine launches of a few microseconds per C2
// call, each of them behind a 0.6 ms field launch in the stream.  Their shapes depend on the call's arguments alone, so the whole
// plan is known when the call starts.  This kernel reproduces torch's own generator bit for bit -- the same Philox4x32-10
// counters (hiprand's device API, which is what torch's kernels are built from), the same subsequence per thread of the same
// launch geometry (ATen's distribution_elementwise_grid_stride_kernel: 256 threads, min(#CU x 8, ceil(numel / 256)) blocks, four
// values per generator call, grid-stride over 4 x grid x 256 elements), the same transforms -- and the host advances the
// generator's offset by what torch would have consumed: a seeded run draws the numbers it drew before, the generator state
// behind the call is the one the reference leaves (tests/test_gpu_parity.py::test_fused_draws_are_torchs).
#include <hip/hip_runtime.h>
#include <hiprand/hiprand_kernel.h>
#include <cstdint>

#include "../../include/nsff_render.h"
#include "nsff_common.h"

namespace {

struct RngArgs {
    NsffRngJob job[NSFF_MAX_RNG_JOBS];
    unsigned first_block[NSFF_MAX_RNG_JOBS + 1];
    int n;
    unsigned long long seed;
    NsffRngCoarse coarse;       // (job < 0: none)
};

// The coarse depths from the perturbation draw, where it is made (reference rendering.py:314-324, 332; csrc/rays.hip's
// coarse_samples_kernel is the stand-alone form, same arithmetic: separate multiplies and adds as torch's kernels round them).
__device__ __forceinline__ void coarse_from_draw(const NsffRngCoarse& c, long long idx, float rnd) {
#pragma clang fp contract(off)
    const int S = c.n_samples;
    const long long n = idx / S;
    const int i = (int)(idx - n * S);
    const float* __restrict__ z_lin = c.z_lin;
    const float lower = i > 0 ? 0.5f * (z_lin[i - 1] + z_lin[i]) : z_lin[0];
    const float upper = i < S - 1 ? 0.5f * (z_lin[i] + z_lin[i + 1]) : z_lin[S - 1];
    const float t = c.perturb * rnd;
    const float span = upper - lower;
    const float z = lower + span * t;
    c.zs[idx] = z;
    const float* r = c.rays + n * 6;
#pragma unroll
    for (int k = 0; k < 3; ++k) { const float dz = r[3 + k] * z; c.xyz[idx * 3 + k] = r[k] + dz; }
}

__global__ __launch_bounds__(256) void nsff_rng_kernel(const RngArgs a) {
    int j = 0;
    while (j + 1 < a.n && blockIdx.x >= a.first_block[j + 1]) ++j;
    const NsffRngJob& J = a.job[j];
    const long long idx = (long long)(blockIdx.x - a.first_block[j]) * 256 + threadIdx.x;
    hiprandStatePhilox4_32_10_t state;
    hiprand_init(a.seed, (unsigned long long)idx, J.offset, &state);
    const long long stride = 256LL * J.grid, numel = J.numel;
    const long long rounded = ((numel - 1) / (stride * 4) + 1) * (stride * 4);
    for (long long li = idx; li < rounded; li += stride * 4) {
        const float4 r = J.kind ? hiprand_normal4(&state) : hiprand_uniform4(&state);
        const float v[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const long long l = li + stride * ii;
            if (l < numel) {
                // torch.randn: rand * std + mean with (1, 0); torch.rand: rand * (to - from) + from, the bound (0, 1] reversed
                const float x = v[ii] * 1.0f + 0.0f;
                const float y = J.kind ? x : (x == 1.0f ? 0.0f : x);
                if (J.out) J.out[l] = y;
                if (j == a.coarse.job) coarse_from_draw(a.coarse, l, y);
            }
        }
    }
}

}  // namespace

extern "C" int nsff_rng_draws(const NsffRngJob* jobs, int32_t n_jobs, uint64_t seed, void* stream) {
    return nsff_rng_draws_coarse(jobs, n_jobs, seed, nullptr, stream);
}

extern "C" int nsff_rng_draws_coarse(const NsffRngJob* jobs, int32_t n_jobs, uint64_t seed, const NsffRngCoarse* coarse, void* stream) {
    if (n_jobs == 0) return coarse ? NSFF_ERR_INVALID : NSFF_OK;
    if (!jobs) return NSFF_ERR_NULL;
    if (n_jobs < 0 || n_jobs > NSFF_MAX_RNG_JOBS) return NSFF_ERR_INVALID;
    RngArgs a{};
    a.coarse.job = -1;
    if (coarse) {
        const NsffRngCoarse& c = *coarse;
        if (c.job < 0 || c.job >= n_jobs || c.n_samples < 1 || !(c.perturb > 0.f)) return NSFF_ERR_INVALID;
        if (!c.rays || !c.z_lin || !c.zs || !c.xyz) return NSFF_ERR_NULL;
        if (jobs[c.job].kind != 0 || jobs[c.job].numel % c.n_samples != 0) return NSFF_ERR_INVALID;
        a.coarse = c;
    }
    unsigned total = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const NsffRngJob& J = jobs[j];
        if (!J.out && j != a.coarse.job) return NSFF_ERR_NULL;       // (the perturbation draw may live in the coarse depths alone)
        if (J.numel <= 0 || J.grid == 0 || (J.kind != 0 && J.kind != 1) || (J.offset & 3ull)) return NSFF_ERR_INVALID;
        if ((long long)J.grid > (J.numel + 255) / 256 || total + J.grid < total) return NSFF_ERR_INVALID;
        a.job[j] = J;
        a.first_block[j] = total;
        total += J.grid;
    }
    a.first_block[n_jobs] = total;
    a.n = n_jobs;
    a.seed = seed;
    hipLaunchKernelGGL(nsff_rng_kernel, dim3(total), dim3(256), 0, (hipStream_t)stream, a);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? NSFF_OK : nsff_hip_fail(e);
}
