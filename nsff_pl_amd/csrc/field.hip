// Fused NSFF field query for gfx950 (MI355X):  encode -> 8+1 layer trunk(s) -> heads.
//
// Replaces, per point, the reference's PosEmbedding.forward (models/nerf.py:17-30),
// the torch.cat/repeat input assembly (models/rendering.py:153-172) and NeRF.forward
// (models/nerf.py:118-213).  One workgroup (4 waves) owns a tile of 64 points whose
// activations never leave LDS; weights stream from L2 as pre-packed MFMA B tiles.
//
// Tiling (see DESIGN.md "field kernel"):
//   * activations  sX[64][260] fp32 in LDS (66,560 B -> two workgroups per CU; the
//     second workgroup's VALU/LDS phases hide under the first one's MFMAs);
//   * wave w computes output columns [64w, 64w+64) for all 64 rows: 2x2 tiles of
//     v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains), 64 accumulator VGPRs;
//   * per 8 input columns: 2 ds_read_b128 (A), 2 global_load_dwordx4 (B), 16 MFMAs;
//   * the skip layer consumes the hidden segment first, then the tile's network
//     input is re-encoded into the (now free) LDS buffer for the second segment.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <vector>
#include <mutex>
#include "nsff_layout.h"
#include "nsff_common.h"
#include "nsff_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int TM = 64;     // points per workgroup
constexpr int LDA = 260;   // LDS row stride (floats): 260/4 odd -> conflict-free ds_read_b128
constexpr int NTHREADS = 256;

struct FieldKArgs {
    NsffLayout L;
    const float* packed;
    const float* xyz;
    const float* x_emb;
    const float* dir_emb;
    const float* a_emb;
    const float* t_emb;
    float* raw;
    long long n_points;
    int pts_per_ray;
    int static_mode, transient_mode;
    int D; unsigned skip_mask;
    int in_xyz, in_dir, in_a, in_t;
    int use_viewdir;
    int fold;                // evaluate the heads that read *_final with the pre-multiplied rows; the layer is not executed
    float flow_scale;
    int n_freqs;
    float freqs[NSFF_MAX_FREQS];
    int ld_emb, off_xyz, off_dir, off_a, off_t;
};

#define MFMA_F32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ void mma8(f32x16 (&acc)[2][2], const float4& a0, const float4& a1,
                                     const float4& b0, const float4& b1) {
    acc[0][0] = MFMA_F32(a0.x, b0.x, acc[0][0]);
    acc[0][1] = MFMA_F32(a0.x, b1.x, acc[0][1]);
    acc[1][0] = MFMA_F32(a1.x, b0.x, acc[1][0]);
    acc[1][1] = MFMA_F32(a1.x, b1.x, acc[1][1]);
    acc[0][0] = MFMA_F32(a0.y, b0.y, acc[0][0]);
    acc[0][1] = MFMA_F32(a0.y, b1.y, acc[0][1]);
    acc[1][0] = MFMA_F32(a1.y, b0.y, acc[1][0]);
    acc[1][1] = MFMA_F32(a1.y, b1.y, acc[1][1]);
    acc[0][0] = MFMA_F32(a0.z, b0.z, acc[0][0]);
    acc[0][1] = MFMA_F32(a0.z, b1.z, acc[0][1]);
    acc[1][0] = MFMA_F32(a1.z, b0.z, acc[1][0]);
    acc[1][1] = MFMA_F32(a1.z, b1.z, acc[1][1]);
    acc[0][0] = MFMA_F32(a0.w, b0.w, acc[0][0]);
    acc[0][1] = MFMA_F32(a0.w, b1.w, acc[0][1]);
    acc[1][0] = MFMA_F32(a1.w, b0.w, acc[1][0]);
    acc[1][1] = MFMA_F32(a1.w, b1.w, acc[1][1]);
}

// acc += sX[:, 0:8*nkb] . seg^T for this wave's two 32-column tiles.
//   sA : LDS address of this lane's A fragment  = sX + (lane&31)*LDA + (lane>>5)*4
//   wB : packed segment + (2*wave*nkb*64 + lane) float4s
__device__ __forceinline__ void gemm_seg(f32x16 (&acc)[2][2], const float* sA,
                                         const float4* __restrict__ wB, int nkb) {
    const float4* __restrict__ w0 = wB;
    const float4* __restrict__ w1 = wB + nkb * 64;
    float4 b0 = w0[0], b1 = w1[0];
    float4 a0 = *reinterpret_cast<const float4*>(sA);
    float4 a1 = *reinterpret_cast<const float4*>(sA + 32 * LDA);
    for (int kb = 1; kb < nkb; ++kb) {
        const float4 nb0 = w0[kb * 64], nb1 = w1[kb * 64];
        const float4 na0 = *reinterpret_cast<const float4*>(sA + kb * 8);
        const float4 na1 = *reinterpret_cast<const float4*>(sA + 32 * LDA + kb * 8);
        mma8(acc, a0, a1, b0, b1);
        a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
    }
    mma8(acc, a0, a1, b0, b1);
}

__device__ __forceinline__ void acc_init(f32x16 (&acc)[2][2], const float* __restrict__ bias,
                                         int wave, int lane) {
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const float b = bias[(2 * wave + n) * 32 + (lane & 31)];
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][n][r] = b; acc[1][n][r] = b; }
    }
}

// C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
template <bool RELU>
__device__ __forceinline__ void acc_store(float* sX, const f32x16 (&acc)[2][2], int wave, int lane) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int col = (2 * wave + n) * 32 + (lane & 31);
                float v = acc[m][n][r];
                if (RELU) v = fmaxf(v, 0.0f);
                sX[row * LDA + col] = v;
            }
}

// Network input of the tile -> LDS columns [0,k0s) (+ [k0s,k0s+kt) with the time code).
__device__ __forceinline__ void build_input(float* sX, const FieldKArgs& a, long long p0, bool with_t) {
    const int r = threadIdx.x & 63, q = threadIdx.x >> 6;
    const long long p = p0 + r;
    const bool valid = p < a.n_points;
    float* row = sX + r * LDA;
    const int k0s = (int)a.L.k0s;
    if (a.xyz != nullptr) {
        float x[3] = {0.f, 0.f, 0.f};
        if (valid) { x[0] = a.xyz[p * 3 + 0]; x[1] = a.xyz[p * 3 + 1]; x[2] = a.xyz[p * 3 + 2]; }
        if (q == 0) {
            row[0] = x[0]; row[1] = x[1]; row[2] = x[2];
            for (int c = a.in_xyz; c < k0s; ++c) row[c] = 0.f;
        }
        const int nf3 = 3 * a.n_freqs;
        for (int j = q; j < nf3; j += 4) {
            const int f = j / 3, c = j - 3 * f;
            float s, co;
            sincosf(a.freqs[f] * x[c], &s, &co);     // full-range accurate (args reach ~600 rad)
            row[3 + 6 * f + c] = s;
            row[3 + 6 * f + 3 + c] = co;
        }
    } else {
        const float* src = a.x_emb + p * a.ld_emb + a.off_xyz;
        for (int c = q; c < k0s; c += 4) row[c] = (valid && c < a.in_xyz) ? src[c] : 0.f;
    }
    if (with_t) {
        const float* src = nullptr;
        if (valid) src = (a.xyz != nullptr) ? a.t_emb + (p / a.pts_per_ray) * a.in_t
                                            : a.x_emb + p * a.ld_emb + a.off_t;
        const int kt = (int)a.L.kt;
        for (int c = q; c < kt; c += 4) row[k0s + c] = (valid && c < a.in_t) ? src[c] : 0.f;
    }
}

// Side input of static_dir_encoding: [dir | a] -> LDS columns [0, side_k).
__device__ __forceinline__ void build_side(float* sX, const FieldKArgs& a, long long p0) {
    const int r = threadIdx.x & 63, q = threadIdx.x >> 6;
    const long long p = p0 + r;
    const bool valid = p < a.n_points;
    float* row = sX + r * LDA;
    const float* sd = nullptr; const float* sa = nullptr;
    if (valid) {
        if (a.xyz != nullptr) {
            const long long ray = p / a.pts_per_ray;
            sd = a.dir_emb + ray * a.in_dir;
            if (a.in_a > 0) sa = a.a_emb + ray * a.in_a;
        } else {
            sd = a.x_emb + p * a.ld_emb + a.off_dir;
            if (a.in_a > 0) sa = a.x_emb + p * a.ld_emb + a.off_a;
        }
    }
    const int sk = (int)a.L.side_k;
    for (int c = q; c < sk; c += 4) {
        float v = 0.f;
        if (valid) {
            if (c < a.in_dir) v = sd[c];
            else if (c < a.in_dir + a.in_a) v = sa[c - a.in_dir];
        }
        row[c] = v;
    }
}

enum { ACT_NONE = 0, ACT_SIGMOID = 1, ACT_FLOW = 2 };

// Narrow output heads on the VALU: out[o] = act(h . w[o] + b[o]); wave w owns rows w, w+4, w+8.
// `kinds` packs one ACT_* code per row, 2 bits each.
__device__ __forceinline__ void heads(const float* sX, const float* __restrict__ w,
                                      const float* __restrict__ b, int n_rows, unsigned kinds,
                                      float flow_scale, float* raw_rec, int slot0, bool valid,
                                      int wave, int lane) {
    if (wave >= n_rows) return;
    const float* row = sX + lane * LDA;
    float acc[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < NSFF_W; k += 4) {
        const float4 h = *reinterpret_cast<const float4*>(row + k);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int o = wave + 4 * i;
            if (o < n_rows) {
                const float* wr = w + o * NSFF_W + k;
                acc[i] = fmaf(h.x, wr[0], acc[i]);
                acc[i] = fmaf(h.y, wr[1], acc[i]);
                acc[i] = fmaf(h.z, wr[2], acc[i]);
                acc[i] = fmaf(h.w, wr[3], acc[i]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int o = wave + 4 * i;
        if (o < n_rows) {
            float v = acc[i] + b[o];
            const unsigned kind = (kinds >> (2 * o)) & 3u;
            if (kind == ACT_SIGMOID) v = 1.0f / (1.0f + expf(-v));
            else if (kind == ACT_FLOW) v = flow_scale * tanhf(v);
            if (valid) raw_rec[slot0 + o] = v;
        }
    }
}

__global__ __launch_bounds__(NTHREADS, 2) void nsff_field_kernel(const FieldKArgs a) {
    __shared__ __attribute__((aligned(16))) float sX[TM * LDA];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long p0 = (long long)blockIdx.x * TM;
    const float* __restrict__ pk = a.packed;
    const float* sA = sX + (lane & 31) * LDA + (lane >> 5) * 4;
    const bool valid = true;      // (every record of the tile's LDS image is filled; the tail is cut when it is written out)
    // raw records of the tile: filled by the heads, written out as whole 64-byte rows at the end (scattered 4-byte
    // stores made the HBM side read-modify-write every record); slots this launch does not evaluate are written as 0
    __shared__ __attribute__((aligned(16))) float sRaw[TM * NSFF_RAW_STRIDE];
    for (int i = threadIdx.x; i < TM * NSFF_RAW_STRIDE; i += NTHREADS) sRaw[i] = 0.f;
    float* raw_rec = sRaw + lane * NSFF_RAW_STRIDE;

    f32x16 acc[2][2];
    auto seg = [&](uint32_t off, int nkb) {
        return reinterpret_cast<const float4*>(pk + off) + (2 * wave * nkb) * 64 + lane;
    };
    auto trunk = [&](const NsffTrunkLayout& T, bool with_t) {
        const int nk0 = (int)T.k0 / 8;
        for (int l = 0; l < a.D; ++l) {
            acc_init(acc, pk + T.bias[l], wave, lane);
            if (l == 0) {
                gemm_seg(acc, sA, seg(T.seg_x[0], nk0), nk0);
            } else {
                gemm_seg(acc, sA, seg(T.seg_h[l], NSFF_W / 8), NSFF_W / 8);
                if ((a.skip_mask >> l) & 1u) {
                    __syncthreads();
                    build_input(sX, a, p0, with_t);
                    __syncthreads();
                    gemm_seg(acc, sA, seg(T.seg_x[l], nk0), nk0);
                }
            }
            __syncthreads();
            acc_store<true>(sX, acc, wave, lane);
            __syncthreads();
        }
    };
    auto final_layer = [&](const NsffTrunkLayout& T) {
        acc_init(acc, pk + T.final_b, wave, lane);
        gemm_seg(acc, sA, seg(T.final_w, NSFF_W / 8), NSFF_W / 8);
        __syncthreads();
        acc_store<false>(sX, acc, wave, lane);
        __syncthreads();
    };

    if (a.static_mode != 0) {
        build_input(sX, a, p0, false);
        __syncthreads();
        trunk(a.L.st, false);
        if (a.static_mode == 2 && a.fold && !a.use_viewdir) {
            // rgb rows pre-multiplied with static_xyz_encoding_final (a Linear without activation, nerf.py:170) + the sigma row
            heads(sX, pk + a.L.s_fold_w, pk + a.L.s_fold_b, 4, 0x15u /* 3x sigmoid, sigma raw */, 0.f, raw_rec, 0, valid, wave, lane);
        } else {
        // sigma reads the last trunk activation, before *_final (nerf.py:169)
        heads(sX, pk + a.L.s_sigma_w, pk + a.L.s_sigma_b, 1, ACT_NONE, 0.f, raw_rec, 3, valid, wave, lane);
        if (a.static_mode == 2) {
            final_layer(a.L.st);
            if (a.use_viewdir) {
                acc_init(acc, pk + a.L.dir_b, wave, lane);
                gemm_seg(acc, sA, seg(a.L.dir_h, NSFF_W / 8), NSFF_W / 8);
                __syncthreads();
                build_side(sX, a, p0);
                __syncthreads();
                const int nks = (int)a.L.side_k / 8;
                gemm_seg(acc, sA, seg(a.L.dir_x, nks), nks);
                __syncthreads();
                acc_store<true>(sX, acc, wave, lane);
                __syncthreads();
            }
            heads(sX, pk + a.L.s_rgb_w, pk + a.L.s_rgb_b, 3, 0x15u /* 3x sigmoid */, 0.f,
                  raw_rec, 0, valid, wave, lane);
        }
        }
        __syncthreads();
    }
    if (a.transient_mode != 0) {
        build_input(sX, a, p0, true);
        __syncthreads();
        trunk(a.L.tr, true);
        if (!a.fold) final_layer(a.L.tr);
        // rows: rgb(3) sigmoid, sigma raw, fw(3) / bw(3) = flow_scale*tanh (nerf.py:197-208); folded: the same rows
        // pre-multiplied with transient_xyz_encoding_final (nerf.py:195), applied to the last trunk activation
        const unsigned kinds = 0x15u | (0xAAAu << 8);
        heads(sX, pk + (a.fold ? a.L.t_fold_w : a.L.t_head_w), pk + (a.fold ? a.L.t_fold_b : a.L.t_head_b),
              (int)a.L.t_head_rows, kinds, a.flow_scale, raw_rec, 4, valid, wave, lane);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < TM * (NSFF_RAW_STRIDE / 4); i += NTHREADS) {
        if (p0 + i / (NSFF_RAW_STRIDE / 4) < a.n_points)
            reinterpret_cast<float4*>(a.raw)[p0 * (NSFF_RAW_STRIDE / 4) + i] = reinterpret_cast<const float4*>(sRaw)[i];
    }
}

// ---------------------------------------------------------------------------------
// weight pack
struct PackSeg {
    const float* src;   // weight (rows, ld) or flat array
    uint32_t dst;       // float offset in packed buffer
    int32_t tiled;      // 1: MFMA tile segment, 0: flat copy
    int32_t ld;         // source row length
    int32_t kpad;       // padded segment width (tiled) / element count (flat)
    int32_t n0, s0;     // segment cols [0,n0)      <- source cols [s0, s0+n0)
    int32_t p1, n1, s1; // segment cols [p1,p1+n1)  <- source cols [s1, s1+n1)
};
constexpr int PACK_BATCH = 16;
struct PackArgs { PackSeg seg[PACK_BATCH]; float* dst; };

__global__ void nsff_pack_kernel(const PackArgs a) {
    const PackSeg& s = a.seg[blockIdx.y];
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (s.tiled) {
        const int nkb = s.kpad / 8;
        if (idx >= 8 * nkb * 64) return;
        const int lane = idx & 63, kb = (idx >> 6) % nkb, ntile = (idx >> 6) / nkb;
        const int n = ntile * 32 + (lane & 31);
        float v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int c = kb * 8 + (lane >> 5) * 4 + t;
            float x = 0.f;
            if (c < s.n0) x = s.src[(long long)n * s.ld + s.s0 + c];
            else if (c >= s.p1 && c < s.p1 + s.n1) x = s.src[(long long)n * s.ld + s.s1 + (c - s.p1)];
            v[t] = x;
        }
        reinterpret_cast<float4*>(a.dst + s.dst)[idx] = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        if (idx < s.kpad) a.dst[s.dst + idx] = s.src[idx];
    }
}

// ---------------------------------------------------------------------------------
// standalone PosEmbedding.forward
struct PosencArgs { const float* x; float* out; long long n_rows; int n_freqs; float freqs[NSFF_MAX_FREQS]; };

__global__ void nsff_posenc_kernel(const PosencArgs a) {
    const int per_row = 3 * a.n_freqs + 3;   // one thread per (row, coord, freq) + raw copies
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.n_rows * per_row) return;
    const long long row = idx / per_row;
    const int j = (int)(idx - row * per_row);
    const int C = 6 * a.n_freqs + 3;
    float* o = a.out + row * C;
    if (j < 3) { o[j] = a.x[row * 3 + j]; return; }
    const int f = (j - 3) / 3, c = (j - 3) - 3 * f;
    float s, co;
    sincosf(a.freqs[f] * a.x[row * 3 + c], &s, &co);
    o[3 + 6 * f + c] = s;
    o[3 + 6 * f + 3 + c] = co;
}

// ---------------------------------------------------------------------------------
int g_last_field_kernel = 0, g_last_field_grid = 0;
struct ProfRec { hipEvent_t e0, e1; double flops, executed; int span_slot; };
std::mutex g_prof_mu;
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
// Shader clock under load: a sample of the workgroups of a profiled f16 launch add their lifetimes in shader-clock ticks and
// in wall-clock ticks to a slot of this pool (field_h3.hip: span_begin / span_end); nsff_prof_collect_clock turns the ratio
// into GHz with the device's wall-clock rate.  Allocated at the first nsff_prof_enable(1), never on a product path (no
// profiling -> kernels get a null pointer and measure nothing).
constexpr int SPAN_SLOTS = 4096;
unsigned long long* g_span_pool = nullptr;
int g_span_next = 0;
int span_pool_reset() {
    const size_t bytes = (size_t)SPAN_SLOTS * NSFF_SPAN_WORDS * 8;
    if (!g_span_pool) {
        hipError_t e = hipMalloc(&g_span_pool, bytes);
        if (e != hipSuccess) { g_span_pool = nullptr; return nsff_hip_fail(e); }
    }
    hipError_t e = hipMemset(g_span_pool, 0, bytes);
    g_span_next = 0;
    return e == hipSuccess ? NSFF_OK : nsff_hip_fail(e);
}

double field_flops_per_point(const NsffModelDesc& d, int static_mode, int transient_mode, int flow_heads) {
    const double W = d.W;
    double macs = 0;
    if (static_mode) {
        macs += d.in_xyz * W + (d.D - 2) * W * W + (d.in_xyz + W) * W + W;           // trunk + sigma
        if (static_mode == 2) {
            macs += W * W + 3 * W;
            if (d.use_viewdir) macs += (W + d.in_dir + d.in_a) * W;
        }
    }
    if (transient_mode) {
        const double in = d.in_xyz + d.in_t;
        macs += in * W + (d.D - 2) * W * W + (in + W) * W + W * W + W;               // trunk + final + sigma
        if (transient_mode == 2) macs += 3 * W + 3 * W * flow_heads;
    }
    return 2.0 * macs;
}

}  // namespace

extern "C" {

int nsff_abi_version(void) { return NSFF_ABI_VERSION; }
const char* nsff_last_hip_error(void) { return hipGetErrorString(g_nsff_last_err); }

int nsff_packed_bytes(const NsffModelDesc* desc, int precision, size_t* bytes) {
    if (!desc || !bytes) return NSFF_ERR_NULL;
    if (precision == NSFF_PREC_F16X3) return nsff_h3_packed_bytes(desc, bytes);
    if (precision != NSFF_PREC_F32) return NSFF_ERR_INVALID;
    NsffLayout L;
    const int rc = nsff_make_layout(*desc, L);
    if (rc) return rc;
    *bytes = (size_t)L.total * sizeof(float);
    return NSFF_OK;
}

int nsff_param_count(const NsffModelDesc* d) {
    if (!d) return NSFF_ERR_NULL;
    int n = 2 * (d->D + 1) + (d->use_viewdir ? 2 : 0) + 4;
    if (d->has_transient) n += 2 * (d->D + 1) + 4 + (d->has_flow ? 4 : 0);
    return n;
}

int nsff_pack_weights(const NsffModelDesc* desc, int precision, const float* const* params, void* packed_v,
                      void* stream) {
    return nsff_pack_weights_ex(desc, precision, params, packed_v, 0, stream);
}

int nsff_fold_heads(const NsffModelDesc* desc, int precision, const float* const* params, void* packed_v, void* stream) {
    if (!desc || !params || !packed_v) return NSFF_ERR_NULL;
    if ((uintptr_t)packed_v & 15) return NSFF_ERR_ALIGN;
    if (precision == NSFF_PREC_F16X3)
        return nsff_h3_fold_heads(desc, params, packed_v, (hipStream_t)stream);
    if (precision != NSFF_PREC_F32) return NSFF_ERR_INVALID;
    NsffLayout L;
    const int rc = nsff_make_layout(*desc, L);
    if (rc) return rc;
    float* packed = reinterpret_cast<float*>(packed_v);
    return nsff_fold_rows_f32(desc, params, packed + L.s_fold_w, packed + L.s_fold_b,
                              desc->has_transient ? packed + L.t_fold_w : nullptr,
                              desc->has_transient ? packed + L.t_fold_b : nullptr, (hipStream_t)stream);
}

int nsff_pack_weights_ex(const NsffModelDesc* desc, int precision, const float* const* params, void* packed_v,
                         int32_t flags, void* stream) {
    if (!desc || !params || !packed_v) return NSFF_ERR_NULL;
    if ((uintptr_t)packed_v & 15) return NSFF_ERR_ALIGN;
    if (flags & ~NSFF_PACK_SKIP_FOLD) return NSFF_ERR_INVALID;
    if (precision == NSFF_PREC_F16X3)
        return nsff_h3_pack_weights(desc, params, packed_v, !(flags & NSFF_PACK_SKIP_FOLD), (hipStream_t)stream);
    if (precision != NSFF_PREC_F32) return NSFF_ERR_INVALID;
    float* packed = reinterpret_cast<float*>(packed_v);
    NsffLayout L;
    const int rc = nsff_make_layout(*desc, L);
    if (rc) return rc;
    const NsffModelDesc& d = *desc;
    std::vector<PackSeg> segs;
    int pi = 0;
    auto tiled = [&](const float* src, uint32_t dst, int ld, int kpad, int n0, int s0, int p1, int n1, int s1) {
        segs.push_back(PackSeg{src, dst, 1, ld, kpad, n0, s0, p1, n1, s1});
    };
    auto flat = [&](const float* src, uint32_t dst, int count) {
        segs.push_back(PackSeg{src, dst, 0, 0, count, 0, 0, 0, 0, 0});
    };
    // input-segment column map: xyz cols -> [0,in_xyz), time-code cols -> [k0s, k0s+in_t)
    auto trunk = [&](const NsffTrunkLayout& T, int in_t) {
        const int in = d.in_xyz + in_t;
        for (int l = 0; l < d.D; ++l) {
            const float* w = params[pi++]; const float* b = params[pi++];
            if (l == 0) {
                tiled(w, T.seg_x[0], in, (int)T.k0, d.in_xyz, 0, (int)L.k0s, in_t, d.in_xyz);
            } else if ((nsff_skip_layers(&d) >> l) & 1u) {
                tiled(w, T.seg_x[l], in + NSFF_W, (int)T.k0, d.in_xyz, 0, (int)L.k0s, in_t, d.in_xyz);
                tiled(w, T.seg_h[l], in + NSFF_W, NSFF_W, NSFF_W, in, 0, 0, 0);
            } else {
                tiled(w, T.seg_h[l], NSFF_W, NSFF_W, NSFF_W, 0, 0, 0, 0);
            }
            flat(b, T.bias[l], NSFF_W);
        }
        const float* w = params[pi++]; const float* b = params[pi++];
        tiled(w, T.final_w, NSFF_W, NSFF_W, NSFF_W, 0, 0, 0, 0);
        flat(b, T.final_b, NSFF_W);
    };
    trunk(L.st, 0);
    if (d.use_viewdir) {
        const float* w = params[pi++]; const float* b = params[pi++];
        const int ld = NSFF_W + d.in_dir + d.in_a;
        tiled(w, L.dir_h, ld, NSFF_W, NSFF_W, 0, 0, 0, 0);
        tiled(w, L.dir_x, ld, (int)L.side_k, d.in_dir + d.in_a, NSFF_W, 0, 0, 0);
        flat(b, L.dir_b, NSFF_W);
    }
    { const float* w = params[pi++]; const float* b = params[pi++]; flat(w, L.s_sigma_w, NSFF_W); flat(b, L.s_sigma_b, 1); }
    { const float* w = params[pi++]; const float* b = params[pi++]; flat(w, L.s_rgb_w, 3 * NSFF_W); flat(b, L.s_rgb_b, 3); }
    if (d.has_transient) {
        trunk(L.tr, d.in_t);
        const float* ws = params[pi++]; const float* bs = params[pi++];   // transient_sigma
        const float* wc = params[pi++]; const float* bc = params[pi++];   // transient_rgb
        flat(wc, L.t_head_w, 3 * NSFF_W); flat(bc, L.t_head_b, 3);
        flat(ws, L.t_head_w + 3 * NSFF_W, NSFF_W); flat(bs, L.t_head_b + 3, 1);
        if (d.has_flow) {
            const float* wf = params[pi++]; const float* bf = params[pi++];
            const float* wb = params[pi++]; const float* bb = params[pi++];
            flat(wf, L.t_head_w + 4 * NSFF_W, 3 * NSFF_W); flat(bf, L.t_head_b + 4, 3);
            flat(wb, L.t_head_w + 7 * NSFF_W, 3 * NSFF_W); flat(bb, L.t_head_b + 7, 3);
        }
    }
    for (int i = 0; i < pi; ++i) if (!params[i]) return NSFF_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    for (size_t base = 0; base < segs.size(); base += PACK_BATCH) {
        PackArgs pa{};
        pa.dst = packed;
        const int n = (int)std::min<size_t>(PACK_BATCH, segs.size() - base);
        int max_threads = 0;
        for (int i = 0; i < n; ++i) {
            pa.seg[i] = segs[base + i];
            const int thr = pa.seg[i].tiled ? 64 * pa.seg[i].kpad : pa.seg[i].kpad;   // 8*(k/8)*64
            max_threads = std::max(max_threads, thr);
        }
        dim3 grid((max_threads + 255) / 256, n);
        hipLaunchKernelGGL(nsff_pack_kernel, grid, dim3(256), 0, st, pa);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return nsff_hip_fail(e);
    if (flags & NSFF_PACK_SKIP_FOLD) return NSFF_OK;
    return nsff_fold_rows_f32(desc, params, packed + L.s_fold_w, packed + L.s_fold_b,
                              d.has_transient ? packed + L.t_fold_w : nullptr, d.has_transient ? packed + L.t_fold_b : nullptr, st);
}

int nsff_posenc(const float* x, int64_t n_rows, const float* freqs_host, int n_freqs, float* out, void* stream) {
    if (!x || !out || !freqs_host) return n_rows == 0 ? NSFF_OK : NSFF_ERR_NULL;
    if (n_freqs < 0 || n_freqs > NSFF_MAX_FREQS || n_rows < 0) return NSFF_ERR_INVALID;
    if (n_rows == 0) return NSFF_OK;
    PosencArgs a{};
    a.x = x; a.out = out; a.n_rows = n_rows; a.n_freqs = n_freqs;
    for (int i = 0; i < n_freqs; ++i) a.freqs[i] = freqs_host[i];
    const long long total = n_rows * (3 * n_freqs + 3);
    hipLaunchKernelGGL(nsff_posenc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? NSFF_OK : nsff_hip_fail(e);
}

int nsff_field_query(const NsffModelDesc* desc, const void* packed_v, const NsffFieldArgs* args, void* stream) {
    const float* packed = reinterpret_cast<const float*>(packed_v);
    if (!desc || !packed || !args) return NSFF_ERR_NULL;
    const NsffModelDesc& d = *desc;
    const NsffFieldArgs& g = *args;
    FieldKArgs k{};
    int rc = nsff_make_layout(d, k.L);
    if (rc) return rc;
    if (g.n_points < 0 || g.pts_per_ray < 1) return NSFF_ERR_INVALID;
    if (g.static_mode < 0 || g.static_mode > 2 || g.transient_mode < 0 || g.transient_mode > 2) return NSFF_ERR_INVALID;
    if (g.static_mode == 0 && g.transient_mode == 0) return NSFF_ERR_INVALID;
    if (g.flow_heads < 0 || g.flow_heads > 2 || (g.flow_heads && !d.has_flow)) return NSFF_ERR_INVALID;
    if (g.precision != NSFF_PREC_F32 && g.precision != NSFF_PREC_F16X3) return NSFF_ERR_INVALID;
    if (g.tile_points != 0 && g.tile_points != 64 && g.tile_points != 130 && g.tile_points != 131) return NSFF_ERR_INVALID;
    if (g.transient_mode && !d.has_transient) return NSFF_ERR_INVALID;
    if (g.launch_form != 0 && g.launch_form != 1) return NSFF_ERR_INVALID;
    if (g.n_points == 0) return NSFF_OK;
    if (!g.raw) return NSFF_ERR_NULL;
    if ((uintptr_t)packed & 15) return NSFF_ERR_ALIGN;
    if ((g.save_acts || g.save_xin || g.save_side) && (g.precision != NSFF_PREC_F16X3 || !g.xyz)) return NSFF_ERR_INVALID;
    if (((uintptr_t)g.save_acts | (uintptr_t)g.save_xin | (uintptr_t)g.save_side) & 15) return NSFF_ERR_ALIGN;
    const bool need_side = g.static_mode == 2 && d.use_viewdir;
    if (g.xyz) {
        if (g.x_emb) return NSFF_ERR_INVALID;
        if (g.n_freqs < 0 || g.n_freqs > NSFF_MAX_FREQS || 3 + 6 * g.n_freqs != d.in_xyz) return NSFF_ERR_INVALID;
        if (g.transient_mode && !g.t_emb) return NSFF_ERR_NULL;
        if (need_side && (!g.dir_emb || (d.in_a > 0 && !g.a_emb))) return NSFF_ERR_NULL;
    } else {
        if (!g.x_emb) return NSFF_ERR_NULL;
        if (g.off_xyz < 0 || g.ld_emb < d.in_xyz) return NSFF_ERR_INVALID;
        if (g.transient_mode && g.off_t < 0) return NSFF_ERR_INVALID;
        if (need_side && (g.off_dir < 0 || (d.in_a > 0 && g.off_a < 0))) return NSFF_ERR_INVALID;
    }
    k.packed = packed;
    k.xyz = g.xyz; k.x_emb = g.x_emb; k.dir_emb = g.dir_emb; k.a_emb = g.a_emb; k.t_emb = g.t_emb;
    k.raw = g.raw; k.n_points = g.n_points; k.pts_per_ray = g.pts_per_ray;
    k.static_mode = g.static_mode; k.transient_mode = g.transient_mode;
    k.D = d.D; k.skip_mask = nsff_skip_layers(&d);
    k.in_xyz = d.in_xyz; k.in_dir = d.in_dir; k.in_a = d.in_a; k.in_t = d.in_t;
    k.use_viewdir = d.use_viewdir; k.flow_scale = d.flow_scale;
    k.fold = 1;                                     // (the fp32 kernel has no training variant: every launch is inference)
    k.n_freqs = g.n_freqs;
    for (int i = 0; i < NSFF_MAX_FREQS; ++i) k.freqs[i] = g.freqs[i];
    k.ld_emb = g.ld_emb; k.off_xyz = g.off_xyz; k.off_dir = g.off_dir; k.off_a = g.off_a; k.off_t = g.off_t;

    const long long tiles = (g.n_points + TM - 1) / TM;
    if (tiles > 0x7fffffffLL) return NSFF_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    ProfRec pr{};
    pr.span_slot = -1;
    bool prof = false;
    unsigned long long* span = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        prof = g_prof_on;
        if (prof && g_span_pool && g_span_next < SPAN_SLOTS && g.precision != NSFF_PREC_F32) {
            pr.span_slot = g_span_next++;
            span = g_span_pool + (size_t)pr.span_slot * NSFF_SPAN_WORDS;
        }
    }
    if (prof) {
        hipError_t e = hipEventCreate(&pr.e0); if (e != hipSuccess) return nsff_hip_fail(e);
        e = hipEventCreate(&pr.e1); if (e != hipSuccess) return nsff_hip_fail(e);
        hipEventRecord(pr.e0, st);
    }
    hipError_t e = hipSuccess;
    if (g.precision == NSFF_PREC_F16X3) {
        // default tiling: 128 points as eight waves of 32 neurons (one workgroup per CU, every weight byte fetched once per
        // 128 points, no spilled registers: 41 MB instead of 79 MB of HBM traffic per C2 launch and -0.8 % time); launches
        // too small to give every CU such a tile keep the 64-point tiling (two workgroups per CU)
        const int tile_default = g.n_points >= 128LL * 256 ? 130 : 64;
        const int rc3 = nsff_h3_field_query(desc, packed_v, args, g.tile_points ? g.tile_points : tile_default, st, span);
        if (rc3 != NSFF_OK) return rc3;
    } else {
        hipLaunchKernelGGL(nsff_field_kernel, dim3((unsigned)tiles), dim3(NTHREADS), 0, st, k);
        e = hipGetLastError();
        g_last_field_kernel = NSFF_KERNEL_F32;
    }
    if (g.precision != NSFF_PREC_F32) g_last_field_kernel = g_nsff_last_h3_kernel;
    g_last_field_grid = g.precision != NSFF_PREC_F32 ? g_nsff_last_h3_grid : 0;
    if (prof) {
        hipEventRecord(pr.e1, st);
        pr.flops = field_flops_per_point(d, g.static_mode, g.transient_mode,
                                         g.transient_mode == 2 ? g.flow_heads : 0) * (double)g.n_points;
        // the f16 kernels fold the activation-free *_final layers into their head rows -- training forwards too (they are f16x3
        // launches); the exact-fp32 kernel folds in its inference launches (it has no saving form)
        const bool h3 = g.precision == NSFF_PREC_F16X3;
        const bool folds = h3 || !(g.save_acts || g.save_xin || g.save_masks || g.save_side);
        const int folded = folds ? ((g.static_mode == 2 && (h3 || !d.use_viewdir)) ? 1 : 0) + (g.transient_mode ? 1 : 0) : 0;
        pr.executed = pr.flops - 2.0 * d.W * d.W * folded * (double)g.n_points;
        std::lock_guard<std::mutex> lk(g_prof_mu);
        g_prof.push_back(pr);
    }
    return e == hipSuccess ? NSFF_OK : nsff_hip_fail(e);
}

int nsff_last_field_kernel(void) { return g_last_field_kernel; }
int nsff_last_field_grid(void) { return g_last_field_grid; }

int nsff_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (on && !g_prof_on) {
        const int rc = span_pool_reset();          // (synchronous: profiling is switched on outside timed regions)
        if (rc != NSFF_OK) return rc;
    }
    g_prof_on = on != 0;
    return NSFF_OK;
}

int nsff_prof_collect(int64_t* launches, double* total_ms, double* total_flops, double* executed_flops) {
    return nsff_prof_collect_clock(launches, total_ms, total_flops, executed_flops, nullptr, nullptr);
}

int nsff_prof_collect_clock(int64_t* launches, double* total_ms, double* total_flops, double* executed_flops,
                            double* shader_ticks, double* ticks_ms) {
    std::vector<ProfRec> recs;
    std::vector<unsigned long long> spans;
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        recs.swap(g_prof);
    }
    double ms = 0, fl = 0, ex = 0, ticks = 0, tms = 0;
    // every launch first: the span pool is copied once, when ALL profiled launches are complete (a copy taken behind the first
    // record's event would read the slots of later launches -- on non-blocking streams hipMemcpy does not wait for them -- half-written)
    for (auto& r : recs) {
        hipError_t e = hipEventSynchronize(r.e1);
        if (e != hipSuccess) return nsff_hip_fail(e);
    }
    for (auto& r : recs) {
        float t = 0;
        hipError_t e = hipEventElapsedTime(&t, r.e0, r.e1);
        if (e != hipSuccess) return nsff_hip_fail(e);
        ms += t; fl += r.flops; ex += r.executed;
        hipEventDestroy(r.e0); hipEventDestroy(r.e1);
        if (r.span_slot >= 0 && g_span_pool) {
            if (spans.empty()) {                   // (every event of this collection is complete: so is every stamp)
                spans.resize((size_t)SPAN_SLOTS * NSFF_SPAN_WORDS);
                if (hipMemcpy(spans.data(), g_span_pool, spans.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) spans.clear();
            }
            if (!spans.empty()) {
                const unsigned long long* sp = spans.data() + (size_t)r.span_slot * NSFF_SPAN_WORDS;
                if (sp[1] > 0) { ticks += (double)sp[0]; tms += (double)sp[1]; }
            }
        }
    }
    if (!spans.empty()) {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        span_pool_reset();
    }
    if (tms > 0) {                                  // wall-clock ticks -> milliseconds (the rate is reported in kHz)
        int dev = 0, khz = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) == hipSuccess && khz > 0)
            tms = tms / (double)khz;
        else
            ticks = tms = 0;
    }
    if (shader_ticks) *shader_ticks = ticks;
    if (ticks_ms) *ticks_ms = tms;
    if (launches) *launches = (int64_t)recs.size();
    if (total_ms) *total_ms = ms;
    if (total_flops) *total_flops = fl;
    if (executed_flops) *executed_flops = ex;
    return NSFF_OK;
}

}  // extern "C"
