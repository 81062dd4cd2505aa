// Backward of the fused NSFF field query for gfx950 (training).
//
//   nsff_field_backward  (K1)  data-gradient chain  d_raw -> heads -> *_final -> trunk layers D..1 -> trunk input,
//                              one workgroup per 64-point tile, the gradient tile lives in LDS as the fp16
//                              B operand, the TRANSPOSED weights stream from L2 as pre-packed A tiles
//                              (v_mfma_f32_32x32x16_f16, fp32 accumulate) -- the mirror image of field_h3.hip;
//   nsff_weight_grad     (K2)  dW = dY^T . X with K = all points: batched split-K GEMMs that stream the
//                              fragment-major activation / gradient tiles written by the forward (SAVE variant)
//                              and by K1; partial sums per split, bias gradients as row sums of the same fragments.
//
// Mixed precision: fp16 operands, fp32 accumulation.  Each point's gradient row is normalised by its own power of
// two (block floating point: the chain is linear per point), so fp16's range is spent on the 256 entries of a
// row, not on the 1e6:1 spread between points; what K2 consumes is re-expressed on one global scale G.
// Reference semantics: autograd of models/nerf.py:118-213 (+ PosEmbedding :17-30 on the host side).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <algorithm>
#include <vector>
#include "nsff_layout_h3.h"
#include "nsff_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

#ifdef BWD_TIMING
// debug build only (make timing): s_memtime stamps of the first 256 workgroups, 6 per step and wave + the end of the workgroup
__device__ unsigned g_bwd_timing[256 * 4 * 32 * 6];
#define BWD_STAMP(k) do { if (lane == 0 && blockIdx.x < 256 && i < 31) \
    g_bwd_timing[((blockIdx.x * 4 + wave) * 32 + i) * 6 + (k)] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
#define BWD_STAMP_END() do { if (lane == 0 && blockIdx.x < 256) \
    g_bwd_timing[((blockIdx.x * 4 + wave) * 32 + 31) * 6] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define BWD_STAMP(k) do {} while (0)
#define BWD_STAMP_END() do {} while (0)
#endif

namespace {

constexpr int LDH = 264;           // halfs per LDS row (528 B: conflict-free ds_read_b128)
constexpr int NT = 2;              // 32-point column tiles per workgroup (64 points)
constexpr int MAX_BSTEPS = 40;
#ifndef BWD_STORE_INTERLEAVE
#define BWD_STORE_INTERLEAVE 1     // the HBM copy of a step's tile rides inside the next step's GEMM (see the kernel)
#endif
#define MFMA_H(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
#define B_PIN() __builtin_amdgcn_sched_barrier(0)

// ------------------------------------------------------------------------------------------------
// Layout of the transposed weight pack (offsets in 4-byte words).  A segment holds Wt (256 rows k, K columns n)
// as A-operand tiles in streaming order  seg[wave][ks][mt][lane][8 halfs]:
//     value = Wt[64*wave + 32*mt + (lane&31)][16*ks + 8*(lane>>5) + t]
struct TrunkLayoutB {
    uint32_t head;                       // rows = the heads' input, K = head rows (16, padded to 64).  The heads' input is the LAST TRUNK
                                         // ACTIVATION h: *_xyz_encoding_final is a Linear without activation (nerf.py:170,195), so the
                                         // training forward evaluates W_head (W_final h + b_final) as (W_head W_final) h like the
                                         // inference launches do, and this tile holds (W_head W_final)^T -- the 256 x 256 *_final layer
                                         // is executed neither forward nor backward (its gradients: field_grad._folded_grads).
                                         // View-direction static trunk: static_rgb reads static_dir_encoding -- its own transpose.
    uint32_t layer[NSFF_MAX_LAYERS];     // l = 1..D-1: rows = activation of layer l-1, K = 256 (pre-activations of l)
    uint32_t x0;                         // rows = trunk input (xin_rows used), K = 256 (pre-activations of layer 0)
    uint32_t xskip[NSFF_MAX_LAYERS];     // the same for every skip layer l (NSFF_NONE elsewhere)
};
struct LayoutB {
    TrunkLayoutB st, tr;
    uint32_t s_sigma;                    // 256 fp32: static_sigma.weight (rank-1 term of the static head)
    uint32_t dir_h, dir_side;            // use_viewdir: static_dir_encoding transposed -- dir_h: rows = last trunk activation (256), the
                                         // FOLDED product (W_dir[:, :256] W_final)^T / dir_side: rows = [dir | a] inputs (side_rows used);
                                         // K = 256 (its pre-activations)
    uint32_t total;
    // trunk-input rows as the forward saves them and d_xin returns them: [0, k0s) position embedding, [k0s, k0s + kt) time
    // code, padded to xin_rows = 128 or 256 (the row counts the weight-gradient GEMM is built for); side_rows likewise
    int32_t k0s, xin_rows, side_rows;
};

inline bool is_skip(const NsffModelDesc& d, int l) { return (nsff_skip_layers(&d) >> l) & 1u; }

inline int make_layout_b(const NsffModelDesc& d, LayoutB& L) {
    NsffLayoutH3 f;
    const int rc = nsff_make_layout_h3(d, f);
    if (rc) return rc;
    L.k0s = (int32_t)f.k0s;
    L.xin_rows = (f.k0s + f.kt) <= 128 ? 128 : 256;          // (k0s + kt <= 256 is the forward layout's own limit)
    L.side_rows = f.side_k <= 128 ? 128 : 256;
    uint32_t off = 0;
    auto take = [&](uint32_t halfs) { uint32_t o = off; off += halfs / 2; return o; };
    auto trunk = [&](TrunkLayoutB& T, bool xparts) {
        T.head = take(256 * 64);
        for (int l = 0; l < NSFF_MAX_LAYERS; ++l) T.layer[l] = NSFF_NONE;
        for (int l = 1; l < d.D; ++l) T.layer[l] = take(256 * 256);
        T.x0 = NSFF_NONE;
        for (int l = 0; l < NSFF_MAX_LAYERS; ++l) T.xskip[l] = NSFF_NONE;
        if (xparts) {
            T.x0 = take(256 * 256);
            for (int l = 1; l < d.D; ++l) if (is_skip(d, l)) T.xskip[l] = take(256 * 256);
        }
    };
    trunk(L.st, false);
    L.tr = TrunkLayoutB{};
    if (d.has_transient) trunk(L.tr, true);
    L.s_sigma = off; off += 256;
    L.dir_h = L.dir_side = NSFF_NONE;
    if (d.use_viewdir) { L.dir_h = take(256 * 256); L.dir_side = take(256 * 256); }
    L.total = off;
    return NSFF_OK;
}

// ---- pack -------------------------------------------------------------------------------------
struct PackSegB {
    const float* src[4];     // kind 2: up to four head tensors; otherwise src[0]
    int32_t r0[4], nr[4];    // kind 2: destination K range of each head tensor
    uint32_t dst;
    int32_t kind;            // 0 flat fp32 copy (count = nks), 1 transposed Linear, 2 heads, 3 trunk-input rows, 4 side-input rows,
                             // 5 heads folded with *_final (fin), 6 static_dir_encoding[:, :256] folded with *_final
    const float* fin;        // kinds 5 / 6: *_xyz_encoding_final.weight (256, 256)
    int32_t ld, c0;          // kind 1: Wt[k][n] = W[n][c0 + k];  kind 3: W[n][xmap(k)]
    int32_t nks;
    int32_t in_xyz, in_t, k0s;
};
constexpr int PACKB_BATCH = 32;
struct PackArgsB { PackSegB seg[PACKB_BATCH]; uint32_t* dst; };

__global__ void pack_kernel_b(const PackArgsB a) {
    const PackSegB& s = a.seg[blockIdx.y];
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (s.kind == 0) {
        if (idx < s.nks) reinterpret_cast<float*>(a.dst + s.dst)[idx] = s.src[0][idx];
        return;
    }
    if (idx >= 4 * s.nks * 2 * 64) return;                 // chunks [wave][ks][mt][lane]
    const int lane = idx & 63, mt = (idx >> 6) & 1, wk = idx >> 7;
    const int ks = wk % s.nks, k = 64 * (wk / s.nks) + 32 * mt + (lane & 31);
    h8 out;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int n = ks * 16 + 8 * (lane >> 5) + t;
        float v = 0.f;
        if (s.kind == 1) {
            v = s.src[0][(long long)n * s.ld + s.c0 + k];
        } else if (s.kind == 2) {
            for (int j = 0; j < 4; ++j)
                if (s.src[j] != nullptr && n >= s.r0[j] && n < s.r0[j] + s.nr[j]) v = s.src[j][(long long)(n - s.r0[j]) * 256 + k];
        } else if (s.kind == 5) {                          // Wt[k][n] = sum_o W_head[n][o] W_final[o][k]
            for (int j = 0; j < 4; ++j)
                if (s.src[j] != nullptr && n >= s.r0[j] && n < s.r0[j] + s.nr[j]) {
                    const float* wh = s.src[j] + (long long)(n - s.r0[j]) * 256;
                    float t0 = 0.f;
                    for (int o = 0; o < 256; ++o) t0 = fmaf(wh[o], s.fin[(long long)o * 256 + k], t0);
                    v = t0;
                }
        } else if (s.kind == 6) {                          // Wt[k][n] = sum_o W_dir[n][o] W_final[o][k]
            const float* wd = s.src[0] + (long long)n * s.ld;
            float t0 = 0.f;
            for (int o = 0; o < 256; ++o) t0 = fmaf(wd[o], s.fin[(long long)o * 256 + k], t0);
            v = t0;
        } else if (s.kind == 4) {                          // Wt[k][n] = W_dir[n][256 + k], k < in_dir + in_a (passed in in_xyz)
            if (k < s.in_xyz) v = s.src[0][(long long)n * s.ld + 256 + k];
        } else {
            int c = -1;
            if (k < s.in_xyz) c = k;
            else if (k >= s.k0s && k < s.k0s + s.in_t) c = s.in_xyz + (k - s.k0s);
            if (c >= 0) v = s.src[0][(long long)n * s.ld + c];
        }
        out[t] = (_Float16)v;
    }
    reinterpret_cast<h8*>(a.dst + s.dst)[idx] = out;
}

// ---- K1 ---------------------------------------------------------------------------------------
enum { EPI_KEEP = 0, EPI_LINEAR = 1, EPI_MASK = 2, EPI_DXIN = 3 };
enum { F_CONTINUE = 1, F_STASH = 2, F_SIGMA = 4, F_FROM_STASH = 8, F_HALF_ROWS = 16, F_TO_SIDE = 32, F_ACCUM = 64 };
struct BStep {
    uint32_t w_off;
    uint8_t nks, epi, slot, flags;     // slot: dpre slot written (EPI_LINEAR / EPI_MASK); mask slot = slot too
};
struct BKArgs {
    BStep steps[MAX_BSTEPS];
    int n_steps, n_static_steps;       // [0, n_static_steps) static trunk, rest transient
    const uint32_t* packed;
    uint32_t s_sigma;
    const float* d_raw;
    const float* raw;
    const float* gmax;
    const unsigned long long* masks;
    _Float16* dpre;
    _Float16* dhead;
    float* d_xin;
    float* d_side;
    long long n_points, n_tiles;
    int D;
    int t_head_rows;                   // 4 or 10
    float flow_scale;
    int xin_rows, side_rows;           // row strides of d_xin / d_side (128 or 256)
};

struct WF1 { h8 w[2]; };
struct XF1 { h8 x[NT]; };
struct WRing1 { WF1 r[4]; };

__device__ __forceinline__ void load_w1(WF1& f, const uint4* __restrict__& wp) {
    const uint4 a0 = wp[0], a1 = wp[64];
    wp += 128;
    f.w[0] = __builtin_bit_cast(h8, a0); f.w[1] = __builtin_bit_cast(h8, a1);
}
__device__ __forceinline__ void load_x1(XF1& f, const _Float16* sB, int ks) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) f.x[nt] = *reinterpret_cast<const h8*>(sB + nt * 32 * LDH + ks * 16);
}
__device__ __forceinline__ void mma1(f32x16 (&acc)[2][NT], const WF1& w, const XF1& x) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = MFMA_H(w.w[mt], x.x[nt], acc[mt][nt]);
}

// first four k-steps of a segment; issued before the epilogue that precedes the segment's GEMM (L2 latency hidden)
__device__ __forceinline__ const uint4* prefetch_w1(WRing1& ring, const uint4* __restrict__ wp) {
    load_w1(ring.r[0], wp); load_w1(ring.r[1], wp); load_w1(ring.r[2], wp); load_w1(ring.r[3], wp);
    return wp;
}

// acc += Wt_seg . B^T for this wave's 64 output rows and the 64 points; nks is a multiple of 4; `ring` holds
// k-steps 0..3 and wp points at k-step 4.  `side(j)` is called once per group of four k-steps (j = 0, 1, ...) right behind
// the group's last weight refill, `side_rest(j)` once at the end: the caller's share of the HBM copy of the tile this very
// GEMM reads (see the kernel), so that its LDS reads / conversions issue in the shadow of the MFMAs and its stores sit
// BEHIND the refills in the wave's in-order memory queue (vmcnt counts loads and stores together on gfx9).
template <class Side, class Rest>
__device__ __forceinline__ void gemm1(f32x16 (&acc)[2][NT], WRing1& ring, const uint4* __restrict__ wp, const _Float16* sB, int nks,
                                      Side&& side, Rest&& side_rest) {
    XF1 x0, x1;
    load_x1(x0, sB, 0);
    int j = 0;
#pragma unroll 1
    for (int ks = 4; ks < nks; ks += 4) {
        load_x1(x1, sB, 1); mma1(acc, ring.r[0], x0); load_w1(ring.r[0], wp); B_PIN();
        load_x1(x0, sB, 2); mma1(acc, ring.r[1], x1); load_w1(ring.r[1], wp); B_PIN();
        load_x1(x1, sB, 3); mma1(acc, ring.r[2], x0); load_w1(ring.r[2], wp); B_PIN();
        load_x1(x0, sB, 4); mma1(acc, ring.r[3], x1); load_w1(ring.r[3], wp); B_PIN();
        side(j++);
        B_PIN();
        sB += 64;
    }
    load_x1(x1, sB, 1); mma1(acc, ring.r[0], x0);
    load_x1(x0, sB, 2); mma1(acc, ring.r[1], x1);
    load_x1(x1, sB, 3); mma1(acc, ring.r[2], x0);
    mma1(acc, ring.r[3], x1);
    side_rest(j);
}

__device__ __forceinline__ float pow2_scale(float amax) {      // 2^k with amax * 2^k in [1024, 2048); 1 for amax == 0
    if (!(amax > 0.f) || !(amax < 3.0e38f)) return 1.0f;
    int e;
    frexpf(amax, &e);
    return ldexpf(1.0f, 11 - e);
}

// LDS tile (fp16, [point][LDH]) -> HBM fragments dst[ks][row block][lane][8 pts] (layout of field_h3.hip's
// tile_to_fragments), every point scaled by rel[p] = G / s_p, a power of two <= 1 that reaches down to 2^-40: it is applied
// as two packed fp16 factors r1 = max(rel, 2^-14) and r2 = rel / r1 (sRel1 / sRel2, one half per point).  The first product
// is exact unless it is subnormal, the second only ever shrinks it -- what comes out is the fp16 rounding of v * rel up to a
// second rounding in the subnormal range (2^-25 absolute on a scale whose maximum is 2^10), and no clamp is needed since
// |v * rel| <= |v|.  One unit of work = one 1 KiB block (16 points x 32 rows) = one wave-wide contiguous 16-byte store: two
// transposing LDS reads (ds_read_b64_tr_b16: lane i of a 16-lane group addresses four rows of point i / 4 and receives row
// i % 16 at the group's four points -- see field_h3.hip), two 8-byte reads per factor array, four packed multiplies.  A
// 256-row tile is 32 blocks = eight per wave.
typedef __fp16 fp4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__device__ __forceinline__ h4 lds_tr4(const _Float16* p) {
    return __builtin_bit_cast(h4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp4*)p));
}
__device__ __forceinline__ void fragment_block(const _Float16* sB, const _Float16* sRel1, const _Float16* sRel2, _Float16* dst,
                                               int n_rows, int blk, int lane) {
    const int rblocks = n_rows >> 5;
    const int rb = blk % rblocks, ks = blk / rblocks;
    const int i = lane & 15, g = lane >> 4;
    const int pt0 = 16 * ks + 8 * (g >> 1);                                  // first of this lane's eight points
    const int at = (pt0 + (i >> 2)) * LDH + 32 * rb + 16 * (g & 1) + 4 * (i & 3);
    const h4 v03 = lds_tr4(sB + at), v47 = lds_tr4(sB + at + 4 * LDH);
    const h4 p03 = (v03 * *reinterpret_cast<const h4*>(sRel1 + pt0)) * *reinterpret_cast<const h4*>(sRel2 + pt0);
    const h4 p47 = (v47 * *reinterpret_cast<const h4*>(sRel1 + pt0 + 4)) * *reinterpret_cast<const h4*>(sRel2 + pt0 + 4);
    h8 out;
#pragma unroll
    for (int t = 0; t < 4; ++t) { out[t] = p03[t]; out[4 + t] = p47[t]; }
    // written once, read once by nsff_weight_grad: keep the weights' L2 lines (-17 % per launch)
    __builtin_nontemporal_store(out, reinterpret_cast<h8*>(dst + (((long long)ks * rblocks + rb) * 64 + lane) * 8));
}
__device__ __forceinline__ void tile_to_fragments_scaled(const _Float16* sB, const _Float16* sRel1, const _Float16* sRel2,
                                                         _Float16* dst, int n_rows) {
    for (int blk = threadIdx.x >> 6; blk < (n_rows >> 5) * 4; blk += 4) fragment_block(sB, sRel1, sRel2, dst, n_rows, blk, threadIdx.x & 63);
}

__global__ __launch_bounds__(256, 2) void nsff_field_bwd_kernel(const BKArgs a) {
    __shared__ __attribute__((aligned(16))) _Float16 sB[64 * LDH];
    __shared__ __attribute__((aligned(16))) _Float16 sStash[64 * LDH];
    __shared__ __attribute__((aligned(16))) float sInv[64], sSig[64];
    __shared__ __attribute__((aligned(16))) _Float16 sRel1[64], sRel2[64];     // G / s as two fp16 factors (fragment_block)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long tile = blockIdx.x;
    const long long p0 = tile * 64;
    const uint32_t* __restrict__ pk = a.packed;
    const float G = pow2_scale(*a.gmax);
    const _Float16* sBw = sB + (lane & 31) * LDH + 8 * (lane >> 5);
    const _Float16* sSw = sStash + (lane & 31) * LDH + 8 * (lane >> 5);
    const long long slot_stride = a.n_tiles * (64 * NSFF_W);

    f32x16 acc[2][NT];
    WRing1 ring;
    auto seg = [&](const BStep& s_) {
        return reinterpret_cast<const uint4*>(pk + s_.w_off) + (wave * s_.nks) * 2 * 64 + lane;
    };
    const uint4* wnext = prefetch_w1(ring, seg(a.steps[0]));
    // The tile a step leaves in sB goes to HBM (fragment order, global scale) DURING the next step's GEMM, which reads the
    // same tile: one (row pair, 8 points) task per group of four k-steps, issued right behind the group's weight refills.
    // vmcnt counts loads and stores in one in-order queue, so a burst of stores in front of a GEMM holds its weight loads
    // back until the stores are acknowledged (measured in round 2: +240 us per launch = the whole store time serialised;
    // stores after the GEMM: -1.5 %); spread over the GEMM, every refill waits for at most two stores and has four k-steps
    // of MFMAs to do so, and the copy's LDS reads / conversions issue in the shadow of the MFMAs.
    int pending_slot = -1;
    auto pending_dst = [&]() { return a.dpre + (long long)pending_slot * slot_stride + tile * (64 * NSFF_W); };
    auto flush_tile = [&]() {
        if (pending_slot >= 0) tile_to_fragments_scaled(sB, sRel1, sRel2, pending_dst(), NSFF_W);
        pending_slot = -1;
    };
#pragma unroll 1
    for (int i = 0; i < a.n_steps; ++i) {
        const BStep st = a.steps[i];
        const bool is_static = i < a.n_static_steps;
        if (i == 0 || i == a.n_static_steps) {
            // ---- head stage of this trunk: activation derivatives, per-point scale, head gradients into the tile ----
            flush_tile();                                  // (the previous trunk's last tile, still in sB / sRel)
            __syncthreads();
            const int pt = threadIdx.x & 63, grp = threadIdx.x >> 6;
            const long long p = p0 + pt;
            const bool valid = p < a.n_points;
            float hv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) hv[r] = 0.f;
            if (valid) {
                const float4* dr = reinterpret_cast<const float4*>(a.d_raw + p * NSFF_RAW_STRIDE);
                const float4* yr = reinterpret_cast<const float4*>(a.raw + p * NSFF_RAW_STRIDE);
                if (is_static) {
                    const float4 d = dr[0], y = yr[0];
                    hv[0] = d.x * y.x * (1.f - y.x); hv[1] = d.y * y.y * (1.f - y.y); hv[2] = d.z * y.z * (1.f - y.z);
                    hv[3] = d.w;
                } else {
                    const float4 d = dr[1], y = yr[1];
                    hv[0] = d.x * y.x * (1.f - y.x); hv[1] = d.y * y.y * (1.f - y.y); hv[2] = d.z * y.z * (1.f - y.z);
                    hv[3] = d.w;
                    if (a.t_head_rows > 4) {
                        const float4 d2 = dr[2], d3 = dr[3], y2 = yr[2], y3 = yr[3];
                        const float fs = a.flow_scale, ifs = 1.0f / a.flow_scale;
                        const float df[6] = {d2.x, d2.y, d2.z, d2.w, d3.x, d3.y};
                        const float yf[6] = {y2.x, y2.y, y2.z, y2.w, y3.x, y3.y};
#pragma unroll
                        for (int c = 0; c < 6; ++c) hv[4 + c] = df[c] * (fs - yf[c] * yf[c] * ifs);
                    }
                }
            }
            float amax = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) amax = fmaxf(amax, fabsf(hv[r]));
            const float s = pow2_scale(amax);
            if (grp == 0) {
                sInv[pt] = 1.0f / s;
                const float rel = fminf(G / s, 1.0f);                // (> 1 only for an all-zero row: any factor will do)
                const float f1 = fmaxf(rel, 6.103515625e-05f);       // 2^-14: the smallest normal fp16
                sRel1[pt] = (_Float16)f1;
                sRel2[pt] = (_Float16)(rel / f1);
                sSig[pt] = is_static ? hv[3] * s : 0.f;     // the static sigma head reads the trunk, not *_final
                h8 lo, hi;
#pragma unroll
                for (int r = 0; r < 8; ++r) { lo[r] = (_Float16)(hv[r] * s); hi[r] = (_Float16)(hv[8 + r] * s); }
                if (is_static) lo[3] = (_Float16)0.f;
                *reinterpret_cast<h8*>(sB + pt * LDH) = lo;
                *reinterpret_cast<h8*>(sB + pt * LDH + 8) = hi;
            } else {
                const h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                *reinterpret_cast<h8*>(sB + pt * LDH + 16 * grp) = z;
                *reinterpret_cast<h8*>(sB + pt * LDH + 16 * grp + 8) = z;
            }
            // head gradients on the global scale, fragment order [ks][lane = row + 32*(pt/8 & 1)][8 pts]; rows 8*grp .. 8*grp+7.
            // Rows 0..15 carry fp16(g * G), rows 16..31 the rounding remainder g * G - fp16(g * G): the head weight / bias
            // gradients are the sum of both halves (a bias gradient is a plain sum over points that may cancel heavily).
            _Float16* dh = a.dhead + ((is_static ? 0 : a.n_tiles) + tile) * (64 * 32);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int row = 8 * grp + r;
                const float v = fminf(fmaxf(hv[row & 15] * G, -65504.f), 65504.f);
                const _Float16 hi = (_Float16)v;
                const _Float16 o = row < 16 ? hi : (_Float16)(v - (float)hi);
                dh[(((pt >> 4) * 64) + row + 32 * ((pt >> 3) & 1)) * 8 + (pt & 7)] = o;
            }
            __syncthreads();
        }

        BWD_STAMP(0);
        unsigned long long mbits = 0ull;
        if (st.epi == EPI_MASK)
            mbits = a.masks[((long long)st.slot * a.n_tiles + tile) * 256 + threadIdx.x];
        if (!(st.flags & F_CONTINUE)) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
        }
        if (!(st.flags & F_HALF_ROWS) || wave < 2) {
            // (one instantiation of the GEMM loop: the copy is switched by a wave-uniform flag)
            const bool copy = BWD_STORE_INTERLEAVE && pending_slot >= 0;
            _Float16* dst = pending_dst();
            gemm1(acc, ring, wnext, (st.flags & F_FROM_STASH) ? sSw : sBw, st.nks,
                  [&](int j) {                       // two of the wave's eight blocks behind each group of four k-steps
                      if (copy) {
                          fragment_block(sB, sRel1, sRel2, dst, NSFF_W, wave + 4 * (2 * j), lane);
                          fragment_block(sB, sRel1, sRel2, dst, NSFF_W, wave + 4 * (2 * j + 1), lane);
                      }
                  },
                  [&](int j) {
                      if (copy) {
#pragma unroll 1
                          for (int u = 2 * j; u < 8; ++u) { fragment_block(sB, sRel1, sRel2, dst, NSFF_W, wave + 4 * u, lane); B_PIN(); }
                      }
                  });
            if (copy) pending_slot = -1;
        }
        BWD_STAMP(1);
        if (i + 1 < a.n_steps) wnext = prefetch_w1(ring, seg(a.steps[i + 1]));    // flies during the epilogue
        if (st.flags & F_SIGMA) {
            // the static sigma head reads the trunk, not *_final: its rank-1 term joins the accumulators here (one step per
            // static trunk; wave-uniform branch)
            const float* ws = reinterpret_cast<const float*>(pk + a.s_sigma);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 w4 = *reinterpret_cast<const float4*>(ws + 64 * wave + 32 * mt + 8 * q + 4 * (lane >> 5));
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const float sg = sSig[32 * nt + (lane & 31)];
                        acc[mt][nt][4 * q + 0] = fmaf(w4.x, sg, acc[mt][nt][4 * q + 0]);
                        acc[mt][nt][4 * q + 1] = fmaf(w4.y, sg, acc[mt][nt][4 * q + 1]);
                        acc[mt][nt][4 * q + 2] = fmaf(w4.z, sg, acc[mt][nt][4 * q + 2]);
                        acc[mt][nt][4 * q + 3] = fmaf(w4.w, sg, acc[mt][nt][4 * q + 3]);
                    }
                }
        }
        flush_tile();                                     // the previous step's tile (sB is still intact)
        BWD_STAMP(2);
        if (st.epi == EPI_KEEP) continue;
        if (st.epi == EPI_DXIN) {
            float* dst_in = (st.flags & F_TO_SIDE) ? a.d_side : a.d_xin;
            const int ld_in = (st.flags & F_TO_SIDE) ? a.side_rows : a.xin_rows;
            const bool accum = st.flags & F_ACCUM;          // a skip layer's share arrived earlier: this one is added to it
            if ((!(st.flags & F_HALF_ROWS) || wave < 2) && dst_in != nullptr) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const int pt = 32 * nt + (lane & 31);
                        const float inv = sInv[pt];
                        if (p0 + pt < a.n_points) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                float4 v;
                                v.x = acc[mt][nt][4 * q + 0] * inv; v.y = acc[mt][nt][4 * q + 1] * inv;
                                v.z = acc[mt][nt][4 * q + 2] * inv; v.w = acc[mt][nt][4 * q + 3] * inv;
                                float4* dp = reinterpret_cast<float4*>(dst_in + (p0 + pt) * ld_in + 64 * wave + 32 * mt + 8 * q + 4 * (lane >> 5));
                                if (accum) { const float4 o = *dp; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }   // (rows owned by this workgroup)
                                *dp = v;
                            }
                        }
                    }
            }
            continue;
        }
        // ---- epilogue: (ReLU mask) -> clamp -> fp16 tile (B operand of the next step).  The mask is applied as a bit
        // pattern (sign-extended bit of the forward's sign word AND the value): ~3.5 VALU per value, no branches ----
        __syncthreads();                                  // every wave is done reading the tile
        BWD_STAMP(3);
        const unsigned mword[2] = {st.epi == EPI_MASK ? (unsigned)mbits : ~0u, st.epi == EPI_MASK ? (unsigned)(mbits >> 32) : ~0u};
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int pt = 32 * nt + (lane & 31);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int bit = nt * 16 + 4 * q + e;               // (mt selects the word)
                        const int keep = (int)(mword[mt] << (31 - bit)) >> 31;          // 0 or -1 (v_bfe_i32)
                        v[e] = __int_as_float(__float_as_int(acc[mt][nt][4 * q + e]) & keep);
                        v[e] = __builtin_amdgcn_fmed3f(v[e], -65504.f, 65504.f);
                    }
                    h4 hv;
                    hv[0] = (_Float16)v[0]; hv[1] = (_Float16)v[1]; hv[2] = (_Float16)v[2]; hv[3] = (_Float16)v[3];
                    *reinterpret_cast<h4*>(sB + pt * LDH + 64 * wave + 32 * mt + 8 * q + 4 * (lane >> 5)) = hv;
                }
            }
        BWD_STAMP(4);
        __syncthreads();
        BWD_STAMP(5);
        if (st.flags & F_STASH) {
            // the skip layer's pre-activation gradient is needed again by the last step of the trunk (its trunk-input
            // part): an LDS -> LDS copy of the finished tile, once per trunk (the next write to sB is two barriers away)
            for (int c = threadIdx.x; c < 64 * 32; c += 256) {
                const int off = (c >> 5) * LDH + (c & 31) * 8;
                *reinterpret_cast<h8*>(sStash + off) = *reinterpret_cast<const h8*>(sB + off);
            }
        }
        pending_slot = st.slot;
    }
    flush_tile();
    BWD_STAMP_END();
}


// ---- d(trunk input) -> d(points), d(per-ray time codes) -------------------------------------------------------
// One workgroup per ray: the ray's d_xin rows (fp32 [point][XR]: columns [0,in_xyz) position embedding, [t0, t0 + in_t)
// time code; XR = 128 or 256) pass through LDS in 64-point chunks; d_xyz = derivative of PosEmbedding (reference
// nerf.py:17-30: [x, sin(f0 x), cos(f0 x), ...]), d_t = sum over the ray's points (the code is repeated per sample,
// rendering.py:168).
struct InArgs {
    const float* d_xin; const float* xyz; float* d_xyz; float* d_t;
    long long n_rays; int pts_per_ray, n_freqs, in_t, t0;
    float freqs[NSFF_MAX_FREQS];
};
template <int XR>
__global__ __launch_bounds__(256) void field_input_bwd_kernel(const InArgs a) {
    constexpr int LDI = XR + 4;                            // floats per LDS row (rows 16 B aligned, stride odd in 16-B units)
    __shared__ __attribute__((aligned(16))) float sD[64 * LDI];
    const long long ray = blockIdx.x;
    const int S = a.pts_per_ray, tid = threadIdx.x;
    const long long p_ray = ray * S;
    float tsum = 0.f;
    for (int s0 = 0; s0 < S; s0 += 64) {
        const int cnt = S - s0 < 64 ? S - s0 : 64;
        __syncthreads();
        for (int i = tid; i < cnt * (XR / 4); i += 256) {   // float4 chunks, coalesced
            const int r = i / (XR / 4), c4 = i % (XR / 4);
            *reinterpret_cast<float4*>(sD + r * LDI + 4 * c4) =
                *reinterpret_cast<const float4*>(a.d_xin + (p_ray + s0 + r) * XR + 4 * c4);
        }
        __syncthreads();
        if (a.d_xyz != nullptr && tid < 192) {
            const int r = tid & 63, c = tid >> 6;           // point of the chunk, component
            if (r < cnt) {
                const long long p = p_ray + s0 + r;
                const float x = a.xyz[p * 3 + c];
                const float* d = sD + r * LDI;
                float g = d[c];
                for (int f = 0; f < a.n_freqs; ++f) {
                    float sn, cs;
                    sincosf(a.freqs[f] * x, &sn, &cs);
                    g += a.freqs[f] * (cs * d[3 + 6 * f + c] - sn * d[3 + 6 * f + 3 + c]);
                }
                a.d_xyz[p * 3 + c] = g;
            }
        }
        if (a.d_t != nullptr && tid < a.in_t) {
            for (int r = 0; r < cnt; ++r) tsum += sD[r * LDI + a.t0 + tid];
        }
    }
    if (a.d_t != nullptr && tid < a.in_t) a.d_t[ray * a.in_t + tid] = tsum;
}

// ---- K2 ---------------------------------------------------------------------------------------
constexpr int MAX_WJOBS = 48;
struct WJob { const _Float16* a; const _Float16* b; long long out_off; long long bias_off; };   // offsets into the scratch
struct WKArgs {
    WJob jobs[MAX_WJOBS];
    float* out;          // scratch partial sums
    float* bias;         // scratch partial row sums
    long long n_tiles;
    int n_splits;
    int n_jobs_total;
};
struct RJob { long long part_off, bias_part_off, out_off; int size, n_splits, job_index, pad; };
struct RKArgs {
    RJob jobs[MAX_WJOBS];
    const float* part; const float* bias_part; const float* gmax;
    float* out; float* bias;
    int n_jobs;
};

// direct-to-LDS DMA of 16 bytes per lane: LDS[lds_dst + 16*lane] <- *gsrc  (lds_dst wave-uniform)
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// Output block 32*NA*WR x 32*NB*WC per workgroup, waves arranged WR x WC, NA x NB accumulator tiles per wave.
// Every k-step (16 points) is one stage of (A rows + B rows)/32 one-KiB fragments, DMA'd straight into an
// 8-stage LDS ring (each wave fetches an equal share, counted vmcnt, one barrier per stage); the waves then read
// their fragments with conflict-free ds_read_b128 -- each HBM byte is fetched once per workgroup.
template <int NA, int NB, int WR, int WC>
__global__ __launch_bounds__(256, 1) void nsff_wgrad_kernel(const WKArgs a) {
    constexpr int A_BLK = NA * WR, B_BLK = NB * WC, CH = A_BLK + B_BLK, PER_WAVE = CH / 4, RING = 8, STAGE = CH * 1024;
    constexpr int A_ROWS = 32 * A_BLK, B_ROWS = 32 * B_BLK;
    static_assert(CH % 4 == 0, "fragments per stage must split evenly over the 4 waves");
    __shared__ __attribute__((aligned(16))) char ring[RING * STAGE];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave / WC, wc = wave % WC;
    const int j = blockIdx.x / a.n_splits, split = blockIdx.x % a.n_splits;
    const WJob job = a.jobs[j];
    const long long per = (a.n_tiles + a.n_splits - 1) / a.n_splits;
    const long long t0 = split * per, t1 = (t0 + per < a.n_tiles) ? t0 + per : a.n_tiles;
    const long long steps = t1 > t0 ? (t1 - t0) * 4 : 0, s0 = t0 * 4;

    f32x16 acc[NA][NB];
#pragma unroll
    for (int na = 0; na < NA; ++na)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[na][nb][r] = 0.f;
    float bsum[NA];
#pragma unroll
    for (int na = 0; na < NA; ++na) bsum[na] = 0.f;

    if (steps > 0) {
        const char* asrc = reinterpret_cast<const char*>(job.a) + lane * 16;
        const char* bsrc = reinterpret_cast<const char*>(job.b) + lane * 16;
        const unsigned ring_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)ring);
        auto issue = [&](long long s) {          // stage s of this split (re-fetches the last one past the end)
            const long long gs = s0 + (s < steps ? s : steps - 1);
            const unsigned dst = ring_base + (unsigned)(s % RING) * STAGE;
#pragma unroll
            for (int c = 0; c < PER_WAVE; ++c) {
                const int chunk = wave * PER_WAVE + c;
                const char* g = chunk < A_BLK ? asrc + (gs * A_BLK + chunk) * 1024 : bsrc + (gs * B_BLK + (chunk - A_BLK)) * 1024;
                dma16(g, __builtin_amdgcn_readfirstlane(dst + (unsigned)chunk * 1024u));
            }
        };
        for (int s = 0; s < RING - 1; ++s) issue(s);
        const h2 ones = {(_Float16)1.f, (_Float16)1.f};
#pragma unroll 1
        for (long long s = 0; s < steps; ++s) {
            // my share of stage s has landed (RING-2 younger stages may still fly); after the barrier everyone's has,
            // and everyone is done with stage s-1, whose slot is refilled with stage s+RING-1
            if constexpr (PER_WAVE == 4) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            else if constexpr (PER_WAVE == 3) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            issue(s + RING - 1);
            const char* st = ring + (s % RING) * STAGE + lane * 16;
            h8 af[NA], bf[NB];
#pragma unroll
            for (int na = 0; na < NA; ++na) af[na] = *reinterpret_cast<const h8*>(st + (NA * wr + na) * 1024);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) bf[nb] = *reinterpret_cast<const h8*>(st + (A_BLK + NB * wc + nb) * 1024);
#pragma unroll
            for (int na = 0; na < NA; ++na) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[na][nb] = MFMA_H(af[na], bf[nb], acc[na][nb]);
                if (wc == 0) {
                    const h8 v = af[na];
                    float t = bsum[na];
                    t = __builtin_amdgcn_fdot2(h2{v[0], v[1]}, ones, t, false);
                    t = __builtin_amdgcn_fdot2(h2{v[2], v[3]}, ones, t, false);
                    t = __builtin_amdgcn_fdot2(h2{v[4], v[5]}, ones, t, false);
                    t = __builtin_amdgcn_fdot2(h2{v[6], v[7]}, ones, t, false);
                    bsum[na] = t;
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // no DMA may outlive the workgroup's LDS
    }
    float* out = a.out + job.out_off + (long long)split * (A_ROWS * B_ROWS);
#pragma unroll
    for (int na = 0; na < NA; ++na)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * (NA * wr + na) + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                out[(long long)row * B_ROWS + 32 * (NB * wc + nb) + (lane & 31)] = acc[na][nb][r];
            }
    if (wc == 0) {
        float* bias = a.bias + job.bias_off + (long long)split * 256;
#pragma unroll
        for (int na = 0; na < NA; ++na) {
            const float t = bsum[na] + __shfl_xor(bsum[na], 32);
            if (lane < 32) bias[32 * (NA * wr + na) + lane] = t;
        }
    }
}

// Head rows (32 x 256 outputs, a handful of jobs): fragments straight from L2, no staging.
__global__ __launch_bounds__(256, 1) void nsff_wgrad_head_kernel(const WKArgs a) {
    constexpr int NB = 2, B_ROWS = 256;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = blockIdx.x / a.n_splits, split = blockIdx.x % a.n_splits;
    const WJob job = a.jobs[j];
    const long long per = (a.n_tiles + a.n_splits - 1) / a.n_splits;
    const long long t0 = split * per, t1 = (t0 + per < a.n_tiles) ? t0 + per : a.n_tiles;
    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
    float bsum = 0.f;
    const h2 ones = {(_Float16)1.f, (_Float16)1.f};
    const _Float16* ap = job.a + lane * 8;
    const _Float16* bp = job.b + (NB * wave) * 512 + lane * 8;
    // The loop is one memory latency per step unless the fragments of the next steps are in flight: DEPTH steps (a tile = four)
    // are requested ahead (round 5: 49 -> see EXPERIMENTS R5 us per launch; the B fragments come from HBM, nobody else reads them)
    constexpr int DEPTH = 8;
    const long long s_begin = t0 * 4, s_end = t1 * 4;
    h8 fa[DEPTH], fb[DEPTH][NB];
    auto fetch = [&](int slot, long long s) {
        const long long c = s < s_end ? s : s_end - 1;          // (past the end: a valid address, never multiplied)
        fa[slot] = __builtin_nontemporal_load(reinterpret_cast<const h8*>(ap + c * 512));
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) fb[slot][nb] = __builtin_nontemporal_load(reinterpret_cast<const h8*>(bp + c * (B_ROWS * 16) + nb * 512));
    };
    if (s_end > s_begin) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) fetch(d, s_begin + d);
#pragma unroll 1
        for (long long s = s_begin; s < s_end; s += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const h8 af = fa[d];
                h8 bf[NB];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) bf[nb] = fb[d][nb];
                fetch(d, s + DEPTH + d);
                if (s + d < s_end) {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) acc[nb] = MFMA_H(af, bf[nb], acc[nb]);
                    bsum = __builtin_amdgcn_fdot2(h2{af[0], af[1]}, ones, bsum, false);
                    bsum = __builtin_amdgcn_fdot2(h2{af[2], af[3]}, ones, bsum, false);
                    bsum = __builtin_amdgcn_fdot2(h2{af[4], af[5]}, ones, bsum, false);
                    bsum = __builtin_amdgcn_fdot2(h2{af[6], af[7]}, ones, bsum, false);
                }
            }
        }
    }
    float* out = a.out + job.out_off + (long long)split * (32 * B_ROWS);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            out[(long long)row * B_ROWS + 32 * (NB * wave + nb) + (lane & 31)] = acc[nb][r];
        }
    if (wave == 0) {
        float* bias = a.bias + job.bias_off + (long long)split * 256;
        const float t = bsum + __shfl_xor(bsum, 32);
        if (lane < 32) bias[lane] = t;
    }
}

// Sum the split-K partials, undo the global scale G and write the final gradients (one launch for every job).
__global__ __launch_bounds__(256) void nsff_wgrad_reduce_kernel(const RKArgs a) {
    const RJob job = a.jobs[blockIdx.y];
    const float inv_g = 1.0f / pow2_scale(*a.gmax);
    for (int e = blockIdx.x * 256 + threadIdx.x; e < job.size + 256; e += gridDim.x * 256) {
        if (e < job.size) {
            const float* p = a.part + job.part_off + e;
            float t = 0.f;
            for (int s = 0; s < job.n_splits; ++s) t += p[(long long)s * job.size];
            a.out[job.out_off + e] = t * inv_g;
        } else {
            const int r = e - job.size;
            const float* p = a.bias_part + job.bias_part_off + r;
            float t = 0.f;
            for (int s = 0; s < job.n_splits; ++s) t += p[(long long)s * 256];
            a.bias[(long long)job.job_index * 256 + r] = t * inv_g;
        }
    }
}

// Gather form of the same reduction: the destination is the parameters' own gradient memory.  Entry i names one
// gradient element (dst, floats from grad_base) and the one or two partial-sum elements it is made of (job, e; e >= size
// addresses the bias row sums); grad[dst] += (sum_A + sum_B) / G.  Deterministic: every destination has one owner.
struct GKArgs {
    RJob jobs[MAX_WJOBS];
    const float* part; const float* gmax;
    const NsffGradMapEntry* map; float* grad;
    long long n_map; int n_jobs;
    float* aux;                  // entries with dst < 0: aux[-(dst + 1)] = value (a store: dense sums for the folded parameters)
};

__global__ __launch_bounds__(256) void nsff_wgrad_accumulate_kernel(const GKArgs a) {
    __shared__ long long sOff[MAX_WJOBS], sBiasOff[MAX_WJOBS];
    __shared__ int sSize[MAX_WJOBS], sSplits[MAX_WJOBS];
    for (int j = threadIdx.x; j < a.n_jobs; j += 256) {
        sOff[j] = a.jobs[j].part_off; sBiasOff[j] = a.jobs[j].bias_part_off;
        sSize[j] = a.jobs[j].size; sSplits[j] = a.jobs[j].n_splits;
    }
    __syncthreads();
    const float inv_g = 1.0f / pow2_scale(*a.gmax);
    auto partial = [&](int job, int e) {
        const int size = sSize[job], n = sSplits[job];
        const bool w = e < size;
        const float* p = a.part + (w ? sOff[job] + e : sBiasOff[job] + (e - size));
        const long long stride = w ? size : 256;
        float t = 0.f;
        int s = 0;
        for (; s + 8 <= n; s += 8) {                       // eight loads in flight, summed in split order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(p + (s + u) * stride);
#pragma unroll
            for (int u = 0; u < 8; ++u) t += v[u];
        }
        for (; s < n; ++s) t += __builtin_nontemporal_load(p + s * stride);
        return t;
    };
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.n_map; i += (long long)gridDim.x * 256) {
        const NsffGradMapEntry m = a.map[i];
        float t = partial(m.job_a, m.e_a);
        if (m.job_b >= 0) t += partial(m.job_b, m.e_b);
        if (m.dst >= 0) a.grad[m.dst] += t * inv_g;
        else if (a.aux != nullptr) a.aux[-(m.dst + 1)] = t * inv_g;
    }
}

// Gradients of *_xyz_encoding_final and of the heads that read it, from the folded heads' gradient (see the header):
// one workgroup per output neuron o of *_final, one thread per input column i.
__global__ __launch_bounds__(256) void fold_grads_kernel(const NsffFoldGradArgs a) {
    const int o = blockIdx.x, i = threadIdx.x, R = a.n_rows;
    __shared__ float sRed[4];
    float gi[16], dwf = 0.f;
    const float wfi = a.w_final[(long long)o * 256 + i];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        gi[r] = 0.f;
        if (r < R) {
            gi[r] = a.g[r * 256 + i] + a.g[(16 + r) * 256 + i];          // fp16 value + rounding remainder rows
            dwf = fmaf(a.w_head[r][o], gi[r], dwf);
        }
    }
    a.d_w_final[(long long)o * 256 + i] += dwf;
    float dbf = 0.f;
    for (int r = 0; r < R; ++r) {
        // dW_head[r][o] = sum_i G[r][i] W_final[o][i] + gb[r] b_final[o]
        float t = gi[r] * wfi;
        for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
        __syncthreads();
        if ((i & 63) == 0) sRed[i >> 6] = t;
        __syncthreads();
        if (i == 0) {
            const float gbr = a.gb[r] + a.gb[16 + r];
            a.d_w_head[r][o] += ((sRed[0] + sRed[1]) + (sRed[2] + sRed[3])) + gbr * a.b_final[o];
            dbf = fmaf(a.w_head[r][o], gbr, dbf);
            if (o == 0) *a.d_b_head[r] += gbr;
        }
    }
    if (i == 0) a.d_b_final[o] += dbf;
}

// The dense form (NsffFoldDenseArgs).  Kernel H: four rows of the folded layer per workgroup, thread = neuron o of *_final:
//   d_w_head[r][o] = sum_i G[r][i] W_final[o][i] + gb[r] b_final[o]  -- the thread streams its row of W_final as float4s (a 16-byte
//   read per 1 KiB stride: every 64-byte sector is fetched by four consecutive iterations, the 256 KiB matrix stays in L2), the
//   four G rows are broadcast reads from LDS.
// Kernel F: one workgroup per neuron o of *_final, thread = input column i:  d_w_final[o][i] = sum_r W_head[r][o] G[r][i]  (G read
//   coalesced along i, W_head[., o] staged in LDS), d_b_final[o] by thread 0 in row order.
constexpr int FOLD_ROWS = 4;
__global__ __launch_bounds__(256) void fold_dense_head_kernel(const NsffFoldDenseArgs a) {
    __shared__ __attribute__((aligned(16))) float sG[FOLD_ROWS][256];
    __shared__ float sGb[FOLD_ROWS];
    const int r0 = blockIdx.x * FOLD_ROWS, o = threadIdx.x;
    for (int k = 0; k < FOLD_ROWS; ++k) {
        const int r = r0 + k;
        float v = 0.f;
        if (r < a.n_rows) { v = a.g[r * 256 + o]; if (a.g2) v += a.g2[r * 256 + o]; }
        sG[k][o] = v;
    }
    if (o < FOLD_ROWS) {
        const int r = r0 + o;
        float v = 0.f;
        if (r < a.n_rows) { v = a.gb[r]; if (a.gb2) v += a.gb2[r]; }
        sGb[o] = v;
    }
    __syncthreads();
    float acc[FOLD_ROWS];
#pragma unroll
    for (int k = 0; k < FOLD_ROWS; ++k) acc[k] = 0.f;
    const float4* wf = reinterpret_cast<const float4*>(a.w_final + (long long)o * 256);
#pragma unroll 4
    for (int i4 = 0; i4 < 64; ++i4) {
        const float4 w = wf[i4];
#pragma unroll
        for (int k = 0; k < FOLD_ROWS; ++k) {
            const float4 gq = *reinterpret_cast<const float4*>(&sG[k][4 * i4]);
            acc[k] = fmaf(gq.x, w.x, acc[k]); acc[k] = fmaf(gq.y, w.y, acc[k]);
            acc[k] = fmaf(gq.z, w.z, acc[k]); acc[k] = fmaf(gq.w, w.w, acc[k]);
        }
    }
    const float bf = a.b_final[o];
#pragma unroll
    for (int k = 0; k < FOLD_ROWS; ++k) {
        const int r = r0 + k;
        if (r >= a.n_rows) break;
        const float v = fmaf(sGb[k], bf, acc[k]);
        float* d = a.d_w_head + (long long)r * a.ld_dhead + o;
        *d = a.accumulate ? *d + v : v;
        if (o == 0) a.d_b_head[r] = a.accumulate ? a.d_b_head[r] + sGb[k] : sGb[k];
    }
}

__global__ __launch_bounds__(256) void fold_dense_final_kernel(const NsffFoldDenseArgs a) {
    __shared__ float sW[256];
    const int o = blockIdx.x, i = threadIdx.x, R = a.n_rows;
    sW[i] = i < R ? a.w_head[(long long)i * a.ld_head + o] : 0.f;
    __syncthreads();
    float acc = 0.f;
    if (a.g2) { for (int r = 0; r < R; ++r) acc = fmaf(sW[r], a.g[r * 256 + i] + a.g2[r * 256 + i], acc); }
    else { for (int r = 0; r < R; ++r) acc = fmaf(sW[r], a.g[r * 256 + i], acc); }
    float* d = a.d_w_final + (long long)o * 256 + i;
    *d = a.accumulate ? *d + acc : acc;
    if (i == 0) {
        float t = 0.f;
        for (int r = 0; r < R; ++r) t = fmaf(sW[r], a.gb[r] + (a.gb2 ? a.gb2[r] : 0.f), t);
        a.d_b_final[o] = a.accumulate ? a.d_b_final[o] + t : t;
    }
}

// max |x| as a device scalar (the global scale of nsff_field_backward): non-negative floats order like their bits
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long long n, unsigned* out) {
    float m = 0.f;
    const long long n4 = n >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 v = x4[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    for (long long i = (n4 << 2) + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        m = fmaxf(m, fabsf(x[i]));
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ float sM[4];
    if ((threadIdx.x & 63) == 0) sM[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out, __float_as_uint(fmaxf(fmaxf(sM[0], sM[1]), fmaxf(sM[2], sM[3]))));
}

}  // namespace

#ifdef BWD_TIMING
extern "C" int nsff_debug_read_bwd_timing(unsigned* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_bwd_timing), sizeof(unsigned) * n) == hipSuccess ? 0 : -4;
}
#endif

extern "C" {

int nsff_fold_grads(const NsffFoldGradArgs* args, void* stream) {
    if (!args) return NSFF_ERR_NULL;
    const NsffFoldGradArgs& a = *args;
    if (a.n_rows < 1 || a.n_rows > 16) return NSFF_ERR_INVALID;
    if (!a.g || !a.gb || !a.w_final || !a.b_final || !a.d_w_final || !a.d_b_final) return NSFF_ERR_NULL;
    for (int r = 0; r < a.n_rows; ++r) if (!a.w_head[r] || !a.d_w_head[r] || !a.d_b_head[r]) return NSFF_ERR_NULL;
    hipLaunchKernelGGL(fold_grads_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, a);
    return nsff_launch_status();
}

int nsff_fold_grads_dense(const NsffFoldDenseArgs* args, void* stream) {
    if (!args) return NSFF_ERR_NULL;
    const NsffFoldDenseArgs& a = *args;
    if (a.n_rows < 1 || a.n_rows > 256 || a.ld_head < 256 || a.ld_dhead < 256) return NSFF_ERR_INVALID;
    if (!a.g || !a.gb || !a.w_head || !a.w_final || !a.b_final || !a.d_w_head || !a.d_b_head || !a.d_w_final || !a.d_b_final) return NSFF_ERR_NULL;
    if ((uintptr_t)a.w_final & 15) return NSFF_ERR_ALIGN;
    hipLaunchKernelGGL(fold_dense_head_kernel, dim3((a.n_rows + FOLD_ROWS - 1) / FOLD_ROWS), dim3(256), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(fold_dense_final_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, a);
    return nsff_launch_status();
}

int nsff_absmax(const float* x, int64_t n, float* out, void* stream) {
    if (!out || (n > 0 && !x)) return NSFF_ERR_NULL;
    if (n < 0) return NSFF_ERR_INVALID;
    if ((uintptr_t)x & 15) return NSFF_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(out, 0, 4, st) != hipSuccess) return nsff_launch_status();
    if (n == 0) return NSFF_OK;
    const long long blocks = std::min<long long>((n / 4 + 255) / 256 + 1, 1024);
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, (long long)n, reinterpret_cast<unsigned*>(out));
    return nsff_launch_status();
}

int nsff_bwd_packed_bytes(const NsffModelDesc* desc, size_t* bytes) {
    if (!desc || !bytes) return NSFF_ERR_NULL;
    LayoutB L;
    const int rc = make_layout_b(*desc, L);
    if (rc) return rc;
    *bytes = (size_t)L.total * 4;
    return NSFF_OK;
}

int nsff_pack_weights_bwd(const NsffModelDesc* desc, const float* const* params, void* packed, void* stream) {
    return nsff_pack_weights_bwd_ex(desc, params, packed, nullptr, stream);
}

// fwd_packed (or null): the F16X3 pack of the SAME weights with folded heads (nsff_pack_weights): its fp32 scratch already
// holds W_head W_final (and W_dir[:, :256] W_final) -- the transposed tiles are then packed from there instead of being
// multiplied again (a 256-long dot product per element: 44 -> 8 us per pack).
int nsff_pack_weights_bwd_ex(const NsffModelDesc* desc, const float* const* params, void* packed, const void* fwd_packed, void* stream) {
    if (!desc || !params || !packed) return NSFF_ERR_NULL;
    if ((uintptr_t)packed & 15) return NSFF_ERR_ALIGN;
    const NsffModelDesc& d = *desc;
    LayoutB L;
    const int rc = make_layout_b(d, L);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    NsffLayoutH3 F;
    if (nsff_make_layout_h3(d, F) != NSFF_OK) return NSFF_ERR_INVALID;
    const float* fold32 = reinterpret_cast<const float*>(fwd_packed);      // fp32 view of the forward pack (word offsets of F)
    std::vector<PackSegB> segs;
    int pi = 0;
    auto lin = [&](const float* w, uint32_t dst, int ld, int c0) {
        PackSegB s{}; s.src[0] = w; s.dst = dst; s.kind = 1; s.ld = ld; s.c0 = c0; s.nks = 16; segs.push_back(s);
    };
    auto xrows = [&](const float* w, uint32_t dst, int ld, int in_t) {
        PackSegB s{}; s.src[0] = w; s.dst = dst; s.kind = 3; s.ld = ld; s.nks = 16; s.in_xyz = d.in_xyz; s.in_t = in_t; s.k0s = L.k0s;
        segs.push_back(s);
    };
    const float* fin[2] = {nullptr, nullptr};              // *_xyz_encoding_final.weight of the two trunks
    auto trunk = [&](int t, const TrunkLayoutB& T, int in_t) {
        const int in = d.in_xyz + in_t;
        for (int l = 0; l < d.D; ++l) {
            const float* w = params[pi]; pi += 2;
            if (l == 0) { if (T.x0 != NSFF_NONE) xrows(w, T.x0, in, in_t); }
            else if (is_skip(d, l)) { lin(w, T.layer[l], in + NSFF_W, in); if (T.xskip[l] != NSFF_NONE) xrows(w, T.xskip[l], in + NSFF_W, in_t); }
            else lin(w, T.layer[l], NSFF_W, 0);
        }
        fin[t] = params[pi]; pi += 2;
    };
    trunk(0, L.st, 0);
    if (d.use_viewdir) {
        const float* wdir = params[pi]; pi += 2;
        const int ld = NSFF_W + d.in_dir + d.in_a;
        if (fold32) lin(fold32 + F.dir_fold_f32, L.dir_h, NSFF_W, 0);
        else { PackSegB s{}; s.src[0] = wdir; s.dst = L.dir_h; s.kind = 6; s.ld = ld; s.nks = 16; s.fin = fin[0]; segs.push_back(s); }
        PackSegB s{}; s.src[0] = wdir; s.dst = L.dir_side; s.kind = 4; s.ld = ld; s.nks = 16; s.in_xyz = d.in_dir + d.in_a; segs.push_back(s);
    }
    {
        const float* wsig = params[pi]; pi += 2;
        const float* wrgb = params[pi]; pi += 2;
        PackSegB s{}; s.src[0] = wsig; s.dst = L.s_sigma; s.kind = 0; s.nks = 256; segs.push_back(s);
        // (the static rgb head reads *_final -- folded -- unless the view-direction layer sits in between)
        PackSegB h{}; h.src[0] = wrgb; h.r0[0] = 0; h.nr[0] = 3; h.dst = L.st.head; h.kind = d.use_viewdir ? 2 : 5; h.nks = 4; h.fin = fin[0];
        if (fold32 && !d.use_viewdir) { h.src[0] = fold32 + F.fold_f32; h.kind = 2; }     // (rows 0..2 of the forward's products; row 3 = sigma stays out)
        segs.push_back(h);
    }
    if (d.has_transient) {
        trunk(1, L.tr, d.in_t);
        const float* ws = params[pi]; pi += 2;
        const float* wc = params[pi]; pi += 2;
        PackSegB h{}; h.dst = L.tr.head; h.kind = 5; h.nks = 4; h.fin = fin[1];
        h.src[0] = wc; h.r0[0] = 0; h.nr[0] = 3;
        h.src[1] = ws; h.r0[1] = 3; h.nr[1] = 1;
        if (d.has_flow) {
            h.src[2] = params[pi]; pi += 2; h.r0[2] = 4; h.nr[2] = 3;
            h.src[3] = params[pi]; pi += 2; h.r0[3] = 7; h.nr[3] = 3;
        }
        if (fold32) {
            const float* rows = fold32 + F.fold_f32 + 32 * NSFF_W;
            h.kind = 2; h.src[0] = rows; h.r0[0] = 0; h.nr[0] = d.has_flow ? 10 : 4;
            h.src[1] = h.src[2] = h.src[3] = nullptr;
        }
        segs.push_back(h);
    }
    for (int i = 0; i < pi; ++i) if (!params[i]) return NSFF_ERR_NULL;
    for (size_t base = 0; base < segs.size(); base += PACKB_BATCH) {
        PackArgsB pa{};
        pa.dst = reinterpret_cast<uint32_t*>(packed);
        const int n = (int)std::min<size_t>(PACKB_BATCH, segs.size() - base);
        int max_threads = 0;
        for (int i = 0; i < n; ++i) {
            pa.seg[i] = segs[base + i];
            max_threads = std::max(max_threads, pa.seg[i].kind == 0 ? pa.seg[i].nks : 4 * pa.seg[i].nks * 2 * 64);
        }
        hipLaunchKernelGGL(pack_kernel_b, dim3((max_threads + 255) / 256, n), dim3(256), 0, st, pa);
    }
    return nsff_launch_status();
}

int nsff_field_backward(const NsffModelDesc* desc, const void* packed_bwd, const NsffFieldBwdArgs* args, void* stream) {
    if (!desc || !packed_bwd || !args) return NSFF_ERR_NULL;
    const NsffModelDesc& d = *desc;
    const NsffFieldBwdArgs& g = *args;
    LayoutB L;
    const int rc = make_layout_b(d, L);
    if (rc) return rc;
    if (g.n_points < 0 || (g.static_mode != 0 && g.static_mode != 2) || (g.transient_mode != 0 && g.transient_mode != 2)) return NSFF_ERR_INVALID;
    if (!g.static_mode && !g.transient_mode) return NSFF_ERR_INVALID;
    if (g.transient_mode && !d.has_transient) return NSFF_ERR_INVALID;
    if (g.n_points == 0) return NSFF_OK;
    if (!g.d_raw || !g.raw || !g.gmax || !g.masks || !g.dpre || !g.dhead) return NSFF_ERR_NULL;
    if (((uintptr_t)packed_bwd | (uintptr_t)g.d_raw | (uintptr_t)g.raw | (uintptr_t)g.dpre | (uintptr_t)g.dhead |
         (uintptr_t)g.d_xin | (uintptr_t)g.d_side | (uintptr_t)g.masks) & 15) return NSFF_ERR_ALIGN;
    BKArgs k{};
    k.packed = reinterpret_cast<const uint32_t*>(packed_bwd);
    k.s_sigma = L.s_sigma;
    k.d_raw = g.d_raw; k.raw = g.raw; k.gmax = g.gmax;
    k.masks = reinterpret_cast<const unsigned long long*>(g.masks);
    k.dpre = reinterpret_cast<_Float16*>(g.dpre);
    k.dhead = reinterpret_cast<_Float16*>(g.dhead);
    k.d_xin = g.d_xin;
    k.d_side = d.use_viewdir ? g.d_side : nullptr;
    k.n_points = g.n_points; k.n_tiles = (g.n_points + 63) / 64;
    k.D = d.D; k.t_head_rows = d.has_flow ? 10 : 4; k.flow_scale = d.flow_scale;
    k.xin_rows = L.xin_rows; k.side_rows = L.side_rows;
    if (k.n_tiles > 0x7fffffffLL) return NSFF_ERR_INVALID;
    int n = 0;
    auto push = [&](uint32_t w, int nks, int epi, int slot, int flags) {
        BStep& s = k.steps[n++];
        s.w_off = w; s.nks = (uint8_t)nks; s.epi = (uint8_t)epi; s.slot = (uint8_t)slot; s.flags = (uint8_t)flags;
    };
    // slots: trunk t (0 static, 1 transient), layer l -> t*(D+1) + l ; l = D is *_final.  Masks use the forward's
    // activation slots, which are numbered the same way.
    auto trunk = [&](const TrunkLayoutB& T, int t, bool want_xin) {
        const int base = t * (d.D + 1);
        // *_xyz_encoding_final is never differentiated as a layer: the heads (the view-direction layer) that read it are linear in
        // the last trunk activation h, so their transposes are FOLDED with *_final's (pack kinds 5 / 6) and the first GEMM of a
        // trunk already produces d h -- masked by the last trunk layer's ReLU, plus the rank-1 term of static_sigma, which reads h
        // itself.  What used to be slot base + D (d *_final) no longer exists; *_final's and the heads' weight gradients are
        // small matrix products of the folded gradients (nsff_pl_amd/field_grad.py::_folded_grads).
        int stash0 = -1;
        if (want_xin) for (int l = 1; l < d.D; ++l) if (is_skip(d, l)) { stash0 = l; break; }
        const int first_flags = (t == 0 ? F_SIGMA : 0) | ((d.D - 1 == stash0) ? F_STASH : 0);
        if (t == 0 && d.use_viewdir) {
            // static_rgb reads static_dir_encoding = relu(W_dir . [*_final | dir | a]) (nerf.py:183-186)
            push(T.head, 4, EPI_MASK, 2 * d.D + 2, 0);
            if (k.d_side != nullptr) push(L.dir_side, 16, EPI_DXIN, 0, (L.side_rows == 128 ? F_HALF_ROWS : 0) | F_TO_SIDE);
            push(L.dir_h, 16, EPI_MASK, base + d.D - 1, first_flags);
        } else {
            push(T.head, 4, EPI_MASK, base + d.D - 1, first_flags);
        }
        // Trunk-input gradient (dynamic trunk, want_xin): d_xin = Wx_0^T dpre_0 + sum over skip layers l of Wx_l^T dpre_l.
        // The LOWEST skip layer's tile is stashed in LDS and multiplied last, into the accumulators layer 0 leaves (no
        // memory traffic -- the reference's one-skip architecture); any further skip layer is multiplied right after its
        // tile appears and goes to d_xin through memory (first one stores, later ones add).
        const int xhalf = L.xin_rows == 128 ? F_HALF_ROWS : 0;
        int stash_l = -1;
        if (want_xin) for (int l = 1; l < d.D; ++l) if (is_skip(d, l)) { stash_l = l; break; }
        bool wrote = false;
        auto after_tile = [&](int l) {                       // the tile of layer l's pre-activation gradient just appeared
            if (want_xin && l >= 1 && is_skip(d, l) && l != stash_l) {
                push(T.xskip[l], 16, EPI_DXIN, 0, xhalf | (wrote ? F_ACCUM : 0));
                wrote = true;
            }
        };
        after_tile(d.D - 1);
        for (int l = d.D - 1; l >= 1; --l) {
            push(T.layer[l], 16, EPI_MASK, base + l - 1, (l - 1 == stash_l) ? F_STASH : 0);
            after_tile(l - 1);
        }
        if (want_xin) {
            if (stash_l >= 0) {
                push(T.x0, 16, EPI_KEEP, 0, xhalf);
                push(T.xskip[stash_l], 16, EPI_DXIN, 0, xhalf | F_CONTINUE | F_FROM_STASH | (wrote ? F_ACCUM : 0));
            } else {
                push(T.x0, 16, EPI_DXIN, 0, xhalf | (wrote ? F_ACCUM : 0));
            }
        }
    };
    if (g.static_mode) trunk(L.st, 0, false);
    k.n_static_steps = n;
    if (g.transient_mode) trunk(L.tr, 1, g.d_xin != nullptr);
    if (n > MAX_BSTEPS) return NSFF_ERR_INVALID;
    k.n_steps = n;
    hipLaunchKernelGGL(nsff_field_bwd_kernel, dim3((unsigned)k.n_tiles), dim3(256), 0, (hipStream_t)stream, k);
    return nsff_launch_status();
}

int nsff_field_input_backward(const float* d_xin, int32_t xin_rows, int32_t t_row0, const float* xyz, int64_t n_rays, int32_t pts_per_ray,
                              const float* freqs_host, int32_t n_freqs, int32_t in_t, float* d_xyz, float* d_t, void* stream) {
    if (n_rays < 0 || pts_per_ray < 1 || n_freqs < 0 || n_freqs > NSFF_MAX_FREQS || in_t < 0) return NSFF_ERR_INVALID;
    if ((xin_rows != 128 && xin_rows != 256) || t_row0 < 0 || t_row0 + in_t > xin_rows || 3 + 6 * n_freqs > xin_rows) return NSFF_ERR_INVALID;
    if (n_rays == 0 || (!d_xyz && !d_t)) return NSFF_OK;
    if (!d_xin || (d_xyz && (!xyz || !freqs_host))) return NSFF_ERR_NULL;
    if ((uintptr_t)d_xin & 15) return NSFF_ERR_ALIGN;
    if (n_rays > 0x7fffffffLL) return NSFF_ERR_INVALID;
    InArgs a{};
    a.d_xin = d_xin; a.xyz = xyz; a.d_xyz = d_xyz; a.d_t = d_t;
    a.n_rays = n_rays; a.pts_per_ray = pts_per_ray; a.n_freqs = n_freqs; a.in_t = in_t; a.t0 = t_row0;
    for (int i = 0; i < n_freqs; ++i) a.freqs[i] = freqs_host[i];
    if (xin_rows == 128) hipLaunchKernelGGL(field_input_bwd_kernel<128>, dim3((unsigned)n_rays), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(field_input_bwd_kernel<256>, dim3((unsigned)n_rays), dim3(256), 0, (hipStream_t)stream, a);
    return nsff_launch_status();
}

// Row geometry of the training buffers of a model (what save_xin / save_side / d_xin / d_side and the weight-gradient jobs of
// their layers are sized by): trunk-input rows [0, t_row0) position embedding, [t_row0, ...) time code, xin_rows = 128 or 256 in
// total; side_rows likewise for the [dir | a] input of static_dir_encoding.
int nsff_train_dims(const NsffModelDesc* desc, int32_t* xin_rows, int32_t* t_row0, int32_t* side_rows) {
    if (!desc || !xin_rows || !t_row0 || !side_rows) return NSFF_ERR_NULL;
    LayoutB L;
    const int rc = make_layout_b(*desc, L);
    if (rc) return rc;
    *xin_rows = L.xin_rows; *t_row0 = L.k0s; *side_rows = L.side_rows;
    return NSFF_OK;
}

// Split-K factor of every job.  The two GEMM classes run one workgroup per CU (their LDS ring takes 96-128 KiB), so a
// class of m jobs is cut into floor(256 r / m) splits each -- whole rounds of the 256 CUs -- with the smallest r that gives
// at least 3/4 of the requested n_splits (18 jobs x 32 splits = 576 workgroups would be 2.25 rounds: a quarter of the
// last one idle; 28 splits = 504 fill two).  The head jobs are tiny: 8x as many splits.
constexpr int WGRAD_CUS = 256;
static inline int wgrad_class(const NsffWgradJob& j) { return j.a_rows == 32 ? 2 : (j.b_rows == 128 ? 1 : 0); }
static void wgrad_plan(const NsffWgradJob* jobs, int32_t n_jobs, int64_t n_tiles, int32_t n_splits, int* splits) {
    int m[3] = {0, 0, 0};
    for (int j = 0; j < n_jobs; ++j) ++m[wgrad_class(jobs[j])];
    long long per_class[3];
    for (int c = 0; c < 2; ++c) {
        long long r = 1;
        while (m[c] > 0 && 4LL * (WGRAD_CUS * r / m[c]) < 3LL * n_splits) ++r;
        per_class[c] = m[c] > 0 ? WGRAD_CUS * r / m[c] : 1;
    }
    per_class[2] = 8LL * n_splits;
    for (int j = 0; j < n_jobs; ++j) {
        long long s = per_class[wgrad_class(jobs[j])];
        if (s > n_tiles) s = n_tiles;
        splits[j] = (int)(s < 1 ? 1 : s);
    }
}

int64_t nsff_weight_grad_scratch(const NsffWgradJob* jobs, int32_t n_jobs, int64_t n_tiles, int32_t n_splits) {
    if (!jobs || n_jobs < 0 || n_jobs > MAX_WJOBS || n_splits < 1) return -1;
    int splits[MAX_WJOBS];
    wgrad_plan(jobs, n_jobs, n_tiles, n_splits, splits);
    int64_t total = 0;
    for (int j = 0; j < n_jobs; ++j) total += (int64_t)splits[j] * ((int64_t)jobs[j].a_rows * jobs[j].b_rows + 256);
    return total;
}

// validates the jobs, lays the split-K partials out in `scratch` (r.jobs) and launches the GEMM classes
static int wgrad_gemms(const NsffWgradJob* jobs, int32_t n_jobs, int64_t n_tiles, int32_t n_splits, float* scratch,
                       RJob* rjobs, int* max_size_out, hipStream_t st) {
    for (int j = 0; j < n_jobs; ++j) {
        const bool ok = (jobs[j].a_rows == 256 || jobs[j].a_rows == 32) && (jobs[j].b_rows == 256 || jobs[j].b_rows == 128) &&
                        !(jobs[j].a_rows == 32 && jobs[j].b_rows == 128);
        if (!ok) return NSFF_ERR_INVALID;
        if (!jobs[j].a || !jobs[j].b) return NSFF_ERR_NULL;
    }
    long long off = 0;
    int max_size = 0;
    int splits[MAX_WJOBS];
    wgrad_plan(jobs, n_jobs, n_tiles, n_splits, splits);
    for (int j = 0; j < n_jobs; ++j) {
        const int sp = splits[j];
        const int size = jobs[j].a_rows * jobs[j].b_rows;
        rjobs[j].part_off = off; off += (long long)sp * size;
        rjobs[j].bias_part_off = off; off += (long long)sp * 256;
        rjobs[j].out_off = jobs[j].out_off; rjobs[j].size = size; rjobs[j].n_splits = sp; rjobs[j].job_index = j;
        max_size = std::max(max_size, size);
    }
    *max_size_out = max_size;
    for (int cls = 0; cls < 3; ++cls) {
        const int ar = cls == 2 ? 32 : 256, brw = cls == 1 ? 128 : 256;
        WKArgs k{};
        k.out = scratch; k.bias = scratch; k.n_tiles = n_tiles; k.n_jobs_total = n_jobs;
        int m = 0;
        for (int j = 0; j < n_jobs; ++j) {
            if (jobs[j].a_rows != ar || jobs[j].b_rows != brw) continue;
            k.jobs[m].a = reinterpret_cast<const _Float16*>(jobs[j].a);
            k.jobs[m].b = reinterpret_cast<const _Float16*>(jobs[j].b);
            k.jobs[m].out_off = rjobs[j].part_off; k.jobs[m].bias_off = rjobs[j].bias_part_off;
            k.n_splits = rjobs[j].n_splits;                 // equal within a class
            ++m;
        }
        if (m == 0) continue;
        const dim3 grid((unsigned)(m * k.n_splits));
        if (cls == 0) hipLaunchKernelGGL((nsff_wgrad_kernel<4, 4, 2, 2>), grid, dim3(256), 0, st, k);
        else if (cls == 1) hipLaunchKernelGGL((nsff_wgrad_kernel<4, 2, 2, 2>), grid, dim3(256), 0, st, k);
        else hipLaunchKernelGGL(nsff_wgrad_head_kernel, grid, dim3(256), 0, st, k);
    }
    return NSFF_OK;
}

int nsff_weight_grad(const NsffWgradJob* jobs, int32_t n_jobs, int64_t n_tiles, int32_t n_splits,
                     float* scratch, float* out, float* bias, const float* gmax, void* stream) {
    if (!jobs || !scratch || !out || !bias || !gmax) return NSFF_ERR_NULL;
    if (n_jobs < 0 || n_jobs > MAX_WJOBS || n_tiles < 0 || n_splits < 1) return NSFF_ERR_INVALID;
    if (n_jobs == 0 || n_tiles == 0) return NSFF_OK;
    hipStream_t st = (hipStream_t)stream;
    RKArgs r{};
    r.part = scratch; r.bias_part = scratch; r.gmax = gmax; r.out = out; r.bias = bias; r.n_jobs = n_jobs;
    int max_size = 0;
    const int rc = wgrad_gemms(jobs, n_jobs, n_tiles, n_splits, scratch, r.jobs, &max_size, st);
    if (rc) return rc;
    hipLaunchKernelGGL(nsff_wgrad_reduce_kernel, dim3((unsigned)std::min((max_size + 256 + 255) / 256, 64), (unsigned)n_jobs),
                       dim3(256), 0, st, r);
    return nsff_launch_status();
}

int nsff_weight_grad_accumulate(const NsffWgradJob* jobs, int32_t n_jobs, int64_t n_tiles, int32_t n_splits, float* scratch,
                                const NsffGradMapEntry* map, int64_t n_map, float* grad_base, const float* gmax, void* stream) {
    return nsff_weight_grad_accumulate_aux(jobs, n_jobs, n_tiles, n_splits, scratch, map, n_map, grad_base, nullptr, gmax, stream);
}

int nsff_weight_grad_accumulate_aux(const NsffWgradJob* jobs, int32_t n_jobs, int64_t n_tiles, int32_t n_splits, float* scratch,
                                    const NsffGradMapEntry* map, int64_t n_map, float* grad_base, float* aux, const float* gmax,
                                    void* stream) {
    if (!jobs || !scratch || !gmax || (n_map > 0 && (!map || !grad_base))) return NSFF_ERR_NULL;
    if (n_jobs < 0 || n_jobs > MAX_WJOBS || n_tiles < 0 || n_splits < 1 || n_map < 0) return NSFF_ERR_INVALID;
    if ((uintptr_t)map & 15) return NSFF_ERR_ALIGN;
    if (n_jobs == 0 || n_tiles == 0 || n_map == 0) return NSFF_OK;
    hipStream_t st = (hipStream_t)stream;
    GKArgs g{};
    g.part = scratch; g.gmax = gmax; g.map = map; g.grad = grad_base; g.n_map = n_map; g.n_jobs = n_jobs; g.aux = aux;
    int max_size = 0;
    const int rc = wgrad_gemms(jobs, n_jobs, n_tiles, n_splits, scratch, g.jobs, &max_size, st);
    if (rc) return rc;
    const long long blocks = std::min<long long>((n_map + 255) / 256, 2048);
    hipLaunchKernelGGL(nsff_wgrad_accumulate_kernel, dim3((unsigned)blocks), dim3(256), 0, st, g);
    return nsff_launch_status();
}

}  // extern "C"
