// fp16-split ("f16x3") variant of the fused NSFF field query for gfx950.
//
// Same function as field.hip (encode -> trunk(s) -> heads == PosEmbedding + NeRF.forward,
// reference models/nerf.py:17-30,118-213) but every fp32 GEMM operand is carried as two halfs
// (x = hi + lo, 22 significant bits) and each product is three f16 MFMAs accumulated in fp32:
//
//        W.x  ~=  Wh.xh + Wh.xl + Wl.xh            (the dropped Wl.xl term is < 2^-22 relative)
//
// which reproduces the fp32 result to fp32-rounding level (measured through the
// whole pipeline against the reference goldens) on the 16x faster f16 matrix pipe.
//
// Formulation is transposed w.r.t. field.hip:  D[neuron][point] = sum_k W[neuron][k] X[point][k]
//   * A operand = weights, streamed from L2 as pre-packed hi/lo tiles (nsff_layout_h3.h);
//   * B operand = activations, kept in LDS as two fp16 planes Xh/Xl [points][264] (528-B rows:
//     conflict-free ds_read_b128);
//   * the 32x32 accumulator then holds 4 consecutive neurons of one point per register quad; the
//     epilogue packs them to fp16 and lanes i / i+32 trade halves (v_permlane32_swap) so that every
//     lane owns 8 consecutive neurons = one 16-byte, bank-conflict-free ds_write_b128 per plane
//     (the direct 8-byte stores are 2-way conflicted with 528-byte rows).
// Wave w owns neurons [64w, 64w+64) for all points of the tile: 2 x NT accumulator tiles.
// NT = 4 (128 points, one workgroup per CU) halves the weight stream per FLOP; NT = 2
// (64 points, two workgroups per CU) hides epilogues behind the other workgroup's MFMAs.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include <vector>
#include "nsff_layout_h3.h"
#include "nsff_common.h"
#include "nsff_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __fp16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2f __attribute__((ext_vector_type(2)));

#ifdef H3_TIMING
// debug build only (make timing): s_memtime stamps of the first 256 workgroups, 6 per step and wave
__device__ unsigned g_h3_timing[256 * 8 * 32 * 6 + 256 * 4 * 256];    // (+ the records of nsff_field_kernel_h3a: H3A_TBASE)
#define H3A_TBASE (256 * 8 * 32 * 6)
__device__ unsigned long long g_h3_span[2] = {~0ull, 0ull};     // first / last s_memtime of the launch (tick calibration)
#define H3_SPAN(k) do { if (lane == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
    if (k) atomicMax(&g_h3_span[1], t_); else atomicMin(&g_h3_span[0], t_); } } while (0)
#define H3_STAMP(k) do { if (lane == 0 && blockIdx.x < 256 && i < 32) \
    g_h3_timing[((blockIdx.x * 8 + wave_id) * 32 + i) * 6 + (k)] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
// stamps inside the LAST head call of a workgroup and behind it: slot 30
#define H3_HSTAMP(k) do { if (lane == 0 && blockIdx.x < 256) \
    g_h3_timing[((blockIdx.x * 8 + wave) * 32 + 30) * 6 + (k)] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define H3_STAMP(k) do {} while (0)
#define H3_SPAN(k) do {} while (0)
#define H3_HSTAMP(k) do {} while (0)
#endif

namespace {

#ifndef H3_SAVE_INTERLEAVE
#define H3_SAVE_INTERLEAVE 1      // training forward: the activation saves ride inside the next GEMM (see the kernel)
#endif
constexpr int LDH = 264;          // halfs per LDS row (528 B = 33 x 16 B): conflict-free ds_read_b128
// 8 consecutive halfs of an LDS row (one B-operand fragment)
__device__ __forceinline__ h8 lds_h8(const _Float16* p) { return *reinterpret_cast<const h8*>(p); }

// The network is executed as a short program of GEMM steps (built on the host), so that the
// kernel holds exactly one copy of the GEMM loop, the epilogue, the input builders and the heads.
struct H3Step {
    uint32_t w_off;      // packed weight segment (words)
    uint32_t bias_off;   // bias that (re)starts the accumulator, or NSFF_NONE to keep accumulating
    uint16_t nks;        // k-steps of 16 columns (multiple of 4)
    uint8_t pre;         // PRE_*: rebuild the LDS tile before this GEMM
    uint8_t post;        // POST_*: epilogue after this GEMM
    uint8_t head;        // HEAD_*: narrow heads evaluated on the stored activations
    uint8_t save;        // training forward: 1 + activation slot this epilogue also writes to HBM, 0 = none
    uint8_t pad[2];
};
enum { PRE_NONE = 0, PRE_INPUT = 1, PRE_INPUT_T = 2, PRE_SIDE = 3 };
enum { POST_NONE = 0, POST_RELU = 1, POST_LINEAR = 2 };
enum { HEAD_NONE = 0, HEAD_S_SIGMA = 1, HEAD_S_RGB = 2, HEAD_T = 3, HEAD_S_FOLD = 4, HEAD_T_FOLD = 5 };
constexpr int MAX_STEPS = 28;

struct H3KArgs {
    NsffLayoutH3 L;
    H3Step steps[MAX_STEPS];
    int n_steps;
    int n_static_steps;   // steps [0, n_static_steps) = static trunk; the rest = dynamic trunk
    int split_trunks;     // 1: workgroups [0, grid_tiles) run the static trunk of tile b, [grid_tiles, 2*grid_tiles) the
                          // dynamic trunk of tile b - grid_tiles (see nsff_h3_field_query)
    long long grid_tiles; // tiles of this launch's tile size
    const uint32_t* packed;
    const float* xyz;
    const float* x_emb;
    const float* dir_emb;
    const float* a_emb;
    const float* t_emb;
    const float* t_bias;       // (rays, tb_rows, 256) fp32: bias + time-code part of the dynamic trunk's input layers (nsff_time_bias), or null
    int tb_rows;
    const float* s_bias;       // (rays, sb_rows, 256) fp32: bias + [dir | a] part of static_dir_encoding (nsff_side_bias), or null
    int sb_rows;
    float* raw;
    // training forward (SAVE variant), all fragment-major so that the weight-gradient GEMM streams them:
    long long lo_delta[3];     // elements from a saved tile to its remainder twin (save_acts, save_xin, save_side); 0: none (NsffFieldArgs::save_lo_delta)
    _Float16* save_acts;       // (slots, tiles, 4 ks, 256 rows, 16 pts) fp16 post-activation values, or null
    _Float16* save_xin;        // (tiles, 4 ks, 128 rows, 16 pts) fp16 trunk input [xyz emb | pad | t at row 64 | pad], or null
    unsigned long long* save_masks;   // (slots, tiles, 256 threads) ReLU sign bits in accumulator order, or null
    _Float16* save_side;       // (tiles, 4 ks, 128 rows, 16 pts) fp16 [dir | a] input of static_dir_encoding, or null
    long long save_stride;     // halfs per activation slot = tiles * 64 * 256
    int xin_rows, side_rows;   // rows of a save_xin / save_side tile (128 or 256)
    long long n_tiles;
    long long n_points;
    int pts_per_ray;
    int static_mode, transient_mode;
    int D, skip;
    int in_xyz, in_dir, in_a, in_t;
    int use_viewdir;
    float flow_scale;
    int n_freqs;
    int octave_freqs;                // freqs[f+1] == 2 freqs[f], n_freqs <= 10, one 64-column segment: the doubling encoder applies
    float freqs[NSFF_MAX_FREQS];
    int ld_emb, off_xyz, off_dir, off_a, off_t;
    unsigned long long* span;        // profiling only (else null): per-XCD first / last s_memtime tick of the launch
};

// Shader clock under load (profiling only, span != null): every 16th workgroup measures its own lifetime in shader-clock ticks
// (s_memtime) and in ticks of the constant-rate wall clock (s_memrealtime: hipDeviceAttributeWallClockRate) and adds both to
// the launch's slot -- a per-workgroup ratio, so counters of different CUs / XCDs need not share an origin (they do not:
// first-to-last s_memtime across the chip is garbage).  clock = sum(shader ticks) / sum(wall ticks) * wall rate.
struct SpanT { unsigned long long t, r; };
__device__ __forceinline__ SpanT span_begin(const unsigned long long* span) {
    SpanT s{0ull, 0ull};
#ifdef H3_TIMING
    s.t = __builtin_amdgcn_s_memtime();
#endif
    if (span != nullptr && (blockIdx.x & 15) == 0) { s.t = __builtin_amdgcn_s_memtime(); s.r = __builtin_amdgcn_s_memrealtime(); }
    return s;
}
__device__ __forceinline__ void span_end(unsigned long long* span, const SpanT& s0, unsigned tid) {
    if (span == nullptr || (blockIdx.x & 15) != 0 || tid != 0) return;
    const unsigned long long t = __builtin_amdgcn_s_memtime(), r = __builtin_amdgcn_s_memrealtime();
    atomicAdd(span, t - s0.t);
    atomicAdd(span + 1, r - s0.r);
}
__device__ __forceinline__ void span_end(unsigned long long* span, const SpanT& s0) { span_end(span, s0, threadIdx.x); }

#define MFMA_H(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
#define H3_PIN() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ void split_store(_Float16* xh, _Float16* xl, int idx, float v) {
    const _Float16 hi = (_Float16)v;
    xh[idx] = hi;
    xl[idx] = (_Float16)(v - (float)hi);
}

// MTW = 32-neuron tiles per wave: 2 (four waves x 64 neurons) or 1 (eight waves x 32 neurons, kernel variant <4,1,*,1>)
template <int MTW> struct WFrag { h8 wh[MTW], wl[MTW]; };   // weights (A operand) of one k-step: 8*MTW VGPRs
template <int NT> struct XFrag { h8 xh[NT], xl[NT]; };      // activations (B operand) of one k-step
template <int MTW> struct WRing { WFrag<MTW> r[4]; };       // four k-steps of weights in flight (L2 latency)
template <int MTW> struct BiasRegs { float4 b[MTW][4]; };

// Weights are read through a bumped pointer so that every load is base + small immediate
// ([ks][mt][part][lane] 16-byte chunks = 4 KiB per k-step; the lane offset is in the pointer).
template <int MTW>
__device__ __forceinline__ void load_w(WFrag<MTW>& f, const uint4* __restrict__& wp) {
    if constexpr (MTW == 2) {
        const uint4 a0 = wp[0], a1 = wp[64], a2 = wp[128], a3 = wp[192];
        wp += 256;
        f.wh[0] = __builtin_bit_cast(h8, a0); f.wl[0] = __builtin_bit_cast(h8, a1);
        f.wh[1] = __builtin_bit_cast(h8, a2); f.wl[1] = __builtin_bit_cast(h8, a3);
    } else {                                   // the wave's pointer already selects its mt half of the k-step
        const uint4 a0 = wp[0], a1 = wp[64];
        wp += 256;
        f.wh[0] = __builtin_bit_cast(h8, a0); f.wl[0] = __builtin_bit_cast(h8, a1);
    }
}

// B-operand rows of this lane: one LDS pointer per 32-point tile (hi plane, lo plane), row stride baked in.
template <int NT> struct BRows { const _Float16* h[NT]; const _Float16* l[NT]; };
template <int NT>
__device__ __forceinline__ BRows<NT> b_rows(const _Float16* bh, const _Float16* bl, int ld) {
    BRows<NT> b;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { b.h[nt] = bh + nt * 32 * ld; b.l[nt] = bl + nt * 32 * ld; }
    return b;
}

template <int NT>
__device__ __forceinline__ void load_x(XFrag<NT>& f, const BRows<NT>& b, int ks) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        f.xh[nt] = lds_h8(b.h[nt] + ks * 16);
        f.xl[nt] = lds_h8(b.l[nt] + ks * 16);
    }
}

// Issue the weight loads of the first four k-steps of a segment (called BEFORE the barriers /
// epilogue that precede the segment's GEMM, so the L2 round trip hides behind them).
// Returns the pointer of k-step 4.
template <int MTW>
__device__ __forceinline__ const uint4* prefetch_w(WRing<MTW>& ring, const uint4* __restrict__ w) {
    const uint4* __restrict__ wp = w;
    load_w(ring.r[0], wp);
    load_w(ring.r[1], wp);
    load_w(ring.r[2], wp);
    load_w(ring.r[3], wp);
    return wp;
}

template <int NT, int MTW>
__device__ __forceinline__ void mma_step(f32x16 (&acc)[MTW][NT], const WFrag<MTW>& w, const XFrag<NT>& x) {
    // three passes so that consecutive MFMAs never touch the same accumulator
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = MFMA_H(w.wl[mt], x.xh[nt], acc[mt][nt]);
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = MFMA_H(w.wh[mt], x.xl[nt], acc[mt][nt]);
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = MFMA_H(w.wh[mt], x.xh[nt], acc[mt][nt]);
}

// acc += W_seg . X[:, 0:16*nks]^T  for this wave's 64 neurons and all 32*NT points.
// `ring` already holds k-steps 0..3 and `wp` points at k-step 4 (prefetch_w); weights run four
// k-steps ahead of the MFMAs, activations (LDS) one.
// `side(j)` runs once per group of four k-steps (j = 0, 1, ...) right behind the group's last weight refill and
// `side_rest(j)` once at the end: the training forward's HBM copy of the very tile this GEMM reads (see the kernel).
struct NoSide { __device__ __forceinline__ void operator()(int) const {} };
// `next` (or null): the weight stream this wave multiplies NEXT (the following segment, or its share of a head tile); its
// first four k-steps refill the ring slots of the last four k-steps as those are consumed, so the stream never stops at a
// layer boundary (requested in one burst after the GEMM, 8 waves x 12 loads of 1 KiB serialise on the CU's 64 B/clk
// vector-memory path for ~1.5k cycles in front of the barrier -- round-3 stamps).  Returns the pointer of k-step 4 of `next`.
// NEXT_STRIDE: uint4s per k-step of that stream (256 trunk segments, 128 head tiles).
template <int NT, int MTW, class Side = NoSide, class Rest = NoSide>
__device__ __forceinline__ const uint4* gemm_seg(f32x16 (&acc)[MTW][NT], WRing<MTW>& ring, const uint4* __restrict__ wp,
                                                 BRows<NT> b, int nks, Side&& side = Side{}, Rest&& side_rest = Rest{},
                                                 const uint4* __restrict__ next = nullptr, int next_stride = 256) {
    // nks is a multiple of 4 (every K-segment is zero-padded to 64 columns): no per-step branches.
    XFrag<NT> x0, x1;
    load_x<NT>(x0, b, 0);
    int j = 0;
#pragma unroll 1
    for (int ks = 4; ks < nks; ks += 4) {        // every group but the last: refill the ring
        // sched_barrier pins each weight refill right behind the MFMAs that free its ring slot (left alone hipcc sinks all
        // 16 loads to the end of the group and the ring never runs ahead)
        load_x<NT>(x1, b, 1);
        mma_step<NT, MTW>(acc, ring.r[0], x0);
        load_w(ring.r[0], wp);
        H3_PIN();
        load_x<NT>(x0, b, 2);
        mma_step<NT, MTW>(acc, ring.r[1], x1);
        load_w(ring.r[1], wp);
        H3_PIN();
        load_x<NT>(x1, b, 3);
        mma_step<NT, MTW>(acc, ring.r[2], x0);
        load_w(ring.r[2], wp);
        H3_PIN();
        load_x<NT>(x0, b, 4);
        mma_step<NT, MTW>(acc, ring.r[3], x1);
        load_w(ring.r[3], wp);
        H3_PIN();
        if constexpr (!__is_same(Side, NoSide)) { side(j++); H3_PIN(); }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { b.h[nt] += 64; b.l[nt] += 64; }      // four k-steps of 16 halfs
    }
    // (the refill of a slot is pinned behind the MFMAs that read it, like in the loop)
    auto refill = [&](WFrag<MTW>& f) __attribute__((always_inline)) {
        if (next != nullptr) {
            const uint4* __restrict__ p = next;
            load_w(f, p);                                       // (advances by a trunk segment's 256)
            next += next_stride;
        }
        H3_PIN();
    };
    load_x<NT>(x1, b, 1);
    mma_step<NT, MTW>(acc, ring.r[0], x0);
    refill(ring.r[0]);
    load_x<NT>(x0, b, 2);
    mma_step<NT, MTW>(acc, ring.r[1], x1);
    refill(ring.r[1]);
    load_x<NT>(x1, b, 3);
    mma_step<NT, MTW>(acc, ring.r[2], x0);
    refill(ring.r[2]);
    mma_step<NT, MTW>(acc, ring.r[3], x1);
    refill(ring.r[3]);
    if constexpr (!__is_same(Rest, NoSide)) side_rest(j);
    return next;
}

// bias of row (neuron) 64w + 32mt + 8q + 4h + e, e = 0..3 -- loaded early, applied by acc_init
template <int MTW>
__device__ __forceinline__ void load_bias(BiasRegs<MTW>& br, const float* __restrict__ bias, int nb0, int lane) {
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            br.b[mt][q] = *reinterpret_cast<const float4*>(bias + nb0 + 32 * mt + 8 * q + 4 * (lane >> 5));
}

template <int NT, int MTW>
__device__ __forceinline__ void acc_init(f32x16 (&acc)[MTW][NT], const BiasRegs<MTW>& br) {
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 b = br.b[mt][q];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                acc[mt][nt][4 * q + 0] = b.x; acc[mt][nt][4 * q + 1] = b.y;
                acc[mt][nt][4 * q + 2] = b.z; acc[mt][nt][4 * q + 3] = b.w;
            }
        }
}

// `mask` (or null) receives the ReLU sign bits of this lane's accumulators: bit (mt*NT + nt)*16 + 4q + e.
// Two 8-byte groups (4 neurons x fp16 each) of lane i and lane i+32 -> one 16-byte group per lane: lanes 0..31 end up
// with neurons [8p, 8p+8) of the 32-neuron block, lanes 32..63 with [16+8p, 16+8p+8) (v_permlane32_swap exchanges the
// upper half of one register with the lower half of another).  `a` = group q = p, `b` = group q = p + 2.
typedef unsigned u2v __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u4v pair_halves(h4 a, h4 b) {
    const u2v ua = __builtin_bit_cast(u2v, a), ub = __builtin_bit_cast(u2v, b);
    const u2v s0 = __builtin_amdgcn_permlane32_swap(ua[0], ub[0], false, false);
    const u2v s1 = __builtin_amdgcn_permlane32_swap(ua[1], ub[1], false, false);
    u4v r;
    r[0] = s0[0]; r[1] = s1[0]; r[2] = s0[1]; r[3] = s1[1];
    return r;
}
// column (halfs) of the 16-byte group pair_halves() leaves in this lane
__device__ __forceinline__ int pair_col(int p, int lane) { return 8 * p + 16 * (lane >> 5); }

// v - (float)h[0] / v - (float)h[1] in one instruction each (v_fma_mix_f32 converts the f16 source on the fly)
__device__ __forceinline__ float minus_lo_half(h2 h, float v) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(v));
    return r;
}
__device__ __forceinline__ float minus_hi_half(h2 h, float v) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(v));
    return r;
}
__device__ __forceinline__ float relu1(float v) {      // one v_max (fmaxf would also canonicalise: two)
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(v));
    return r;
}

// min(bits of v, 1): 1 for every v > +0 once the ReLU has run (v is +0 or positive then).  Inline asm on purpose: written
// as an integer expression hipcc turns it into v_cmp_class_f32 + v_cndmask + hazard nops (7 VALU per value).
__device__ __forceinline__ unsigned positive_flag(float v) {
    unsigned r;
    asm("v_min_u32 %0, 1, %1" : "=v"(r) : "v"(v));
    return r;
}

// Epilogue of one half of a 32x32 accumulator tile: the register quads q = p and q = p + 2 (8 values: neurons
// 8p + 4h + e and 16 + 8p + 4h + e of the tile, h = lane / 32) -> [ReLU ->] hi/lo split -> LDS.  3 VALU per value: v_max, half
// a v_cvt_pkrtz (hi), v_fma_mix (v - hi), half a v_cvt_pkrtz (lo); lanes i and i + 32 then trade halves (v_permlane32_swap)
// so that every lane owns 8 consecutive neurons = one 16-byte, bank-conflict-free ds_write_b128 per plane.
// rowh / rowl: this lane's point row at the tile's first neuron.  MASKS: also collect the ReLU sign bits, bit 4q + e of
// `signs` (+ sign_shift) -- 2 VALU per value.
template <bool RELU, bool MASKS>
__device__ __forceinline__ void acc_store_unit(_Float16* rowh, _Float16* rowl, const f32x16& t, int p, int lane,
                                               unsigned& signs, int sign_shift) {
    h4 hq[2], lq[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = p + 2 * j;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = t[4 * q + e];
            if (RELU) v[e] = relu1(v[e]);
            if constexpr (MASKS) signs |= positive_flag(v[e]) << (sign_shift + 4 * q + e);
        }
        const h2 h01 = __builtin_amdgcn_cvt_pkrtz(v[0], v[1]);
        const h2 h23 = __builtin_amdgcn_cvt_pkrtz(v[2], v[3]);
        const h2 l01 = __builtin_amdgcn_cvt_pkrtz(minus_lo_half(h01, v[0]), minus_hi_half(h01, v[1]));
        const h2 l23 = __builtin_amdgcn_cvt_pkrtz(minus_lo_half(h23, v[2]), minus_hi_half(h23, v[3]));
        h4 hv, lv;
        hv[0] = (_Float16)h01[0]; hv[1] = (_Float16)h01[1]; hv[2] = (_Float16)h23[0]; hv[3] = (_Float16)h23[1];
        lv[0] = (_Float16)l01[0]; lv[1] = (_Float16)l01[1]; lv[2] = (_Float16)l23[0]; lv[3] = (_Float16)l23[1];
        hq[j] = hv; lq[j] = lv;
    }
    *reinterpret_cast<u4v*>(rowh + pair_col(p, lane)) = pair_halves(hq[0], hq[1]);
    *reinterpret_cast<u4v*>(rowl + pair_col(p, lane)) = pair_halves(lq[0], lq[1]);
}

// Whole-tile epilogue of a wave.  `mask` (MASKS) = the 256 sign words of the workgroup's first 64-point tile: the backward
// kernel (64-point tiles, four waves of 64 neurons x [mt][nt]) reads word [tile][64 w + lane], bit ((mt * 2 + nt) * 4 + q) * 4 + e.
template <int NT, bool RELU, int MTW, bool MASKS = false>
__device__ __forceinline__ void acc_store(_Float16* sXh, _Float16* sXl, const f32x16 (&acc)[MTW][NT], int nb0, int nt0, int lane,
                                          unsigned long long* mask = nullptr, bool two_tiles = true) {
    unsigned signs[2] = {0u, 0u};                  // two tiles of 16 values per 32-bit word
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int row = (32 * (nt0 + nt) + (lane & 31)) * LDH + nb0 + 32 * mt;
            const int ti = mt * NT + nt;
#pragma unroll
            for (int p = 0; p < 2; ++p)
                acc_store_unit<RELU, MASKS>(sXh + row, sXl + row, acc[mt][nt], p, lane, signs[ti >> 1], 16 * (ti & 1));
        }
    if constexpr (MASKS) {
        unsigned* m32 = reinterpret_cast<unsigned*>(mask);
        const int word = 64 * (nb0 >> 6) + lane;
        if constexpr (MTW == 2) {                  // this wave IS wave nb0 / 64 of the (one) 64-point tile
            m32[2 * word] = signs[0]; m32[2 * word + 1] = signs[1];
        } else {
            // eight waves of 32 neurons x 128 points: this wave holds (mt = its neuron half) x (nt = 0..3); point tiles 0, 1
            // belong to the first 64-point tile, 2, 3 to the second; each goes out as one 32-bit half of the word
            static_assert(NT == 4 || MTW == 2, "mask layout of the 32-neuron-per-wave tiling");
            const int mt_b = (nb0 >> 5) & 1;
            m32[2 * word + mt_b] = signs[0];
            if (two_tiles) m32[2 * (256 + word) + mt_b] = signs[1];
        }
    }
}

// Copy the tile's activations (hi + lo planes, rounded to nearest) to HBM in the fragment order of the
// weight-gradient GEMM (K = points): dst[ks][row block][lane][8 pts] -- every 1 KiB block is exactly one MFMA
// operand fragment (32 rows x 16 points) in lane order, lane = (row & 31) + 32 * (8-point group & 1); rows = neurons (n_rows,
// a multiple of 32), ks = 16-point groups (four per 64-point tile of the backward kernels; a 128-point workgroup writes two
// consecutive tiles).
// One unit of work = one such block = one wave-wide, fully contiguous 16-byte store: the LDS image is point-major, the
// fragment wants 8 consecutive POINTS of one row per lane, and ds_read_b64_tr_b16 does that transposition in the read -- lane
// i of a 16-lane group addresses four consecutive rows (neurons) of point i / 4, and receives row i % 16 of the group's
// sixteen at its four points (tools/debug/probes/tr_read_probe.hip prints the mapping).  Four reads (two per plane), four
// packed adds (hi + lo: the sum of the two halfs rounded to nearest even -- what the fp32 sum followed by a conversion
// gives, the fp32 sum being exact) and one store per block instead of 16 four-byte reads, 16 element inserts and two
// strided stores per (row pair, 8 points).  Same box, 131 072 points, both trunks: training forward 1000 -> 988 us, field
// backward (same scheme, field_bwd.hip) 492 -> 464 us.  What the copy costs the training forward is mostly not instructions:
// with its stores compiled out 963 -> 885 us, without the copy 792 us, without the sign words 934 us (inference kernel: 736);
// reading the tile a GEMM group ahead of the adds (so that a group of MFMAs hides the LDS latency) changed nothing.
// ks_end: 16-point groups that exist (the second 64-point tile of the last 128-point workgroup may not).
typedef __fp16 fp4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__device__ __forceinline__ h4 lds_tr4(const _Float16* p) {
    return __builtin_bit_cast(h4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp4*)p));
}
// lo_delta != 0 (training forward for the three-product backward): the remainder  (hi + lo) - fp16(hi + lo)  goes to the tile's twin
// lo_delta elements further on -- (hi - v) is exact in fp16 (they differ by at most one unit in the last place), + lo rounds at lo's level.
__device__ __forceinline__ void fragment_block(const _Float16* sXh, const _Float16* sXl, _Float16* dst, int n_rows, int rows_copied,
                                               int rblocks, int blk, int ks_end, int lane, long long lo_delta = 0) {
    const int rb = blk % rblocks, ks = blk / rblocks;
    if (ks >= ks_end) return;
    const int i = lane & 15, g = lane >> 4;
    const int at = (16 * ks + 8 * (g >> 1) + (i >> 2)) * LDH + 32 * rb + 16 * (g & 1) + 4 * (i & 3);
    const h4 h03 = lds_tr4(sXh + at), l03 = lds_tr4(sXl + at), h47 = lds_tr4(sXh + at + 4 * LDH), l47 = lds_tr4(sXl + at + 4 * LDH);
    const h4 lo03 = h03 + l03;                                                       // points 0..3 of the lane's eight
    const h4 lo47 = h47 + l47;                                                       // points 4..7
    h8 out;
#pragma unroll
    for (int t = 0; t < 4; ++t) { out[t] = lo03[t]; out[4 + t] = lo47[t]; }
    _Float16* d = dst + (((long long)ks * (n_rows >> 5) + rb) * 64 + lane) * 8;
    if (32 * rb + (lane & 31) < rows_copied) {
        __builtin_nontemporal_store(out, reinterpret_cast<h8*>(d));          // written once, read once by nsff_weight_grad
        if (lo_delta != 0) {
            const h4 r03 = (h03 - lo03) + l03, r47 = (h47 - lo47) + l47;
            h8 rem;
#pragma unroll
            for (int t = 0; t < 4; ++t) { rem[t] = r03[t]; rem[4 + t] = r47[t]; }
            __builtin_nontemporal_store(rem, reinterpret_cast<h8*>(d + lo_delta));
        }
    }
}
// blocks of a tile: (16-point group, 32-row block), row block fastest
template <int M> __device__ __forceinline__ int fragment_blocks(int rows_copied) { return ((rows_copied + 31) >> 5) * (M / 16); }
template <int THREADS, int M>
__device__ __forceinline__ void tile_to_fragments(const _Float16* sXh, const _Float16* sXl, _Float16* dst, int n_rows,
                                                  int rows_copied, int ks_end, long long lo_delta = 0) {
    const int rblocks = (rows_copied + 31) >> 5;
    for (int blk = threadIdx.x >> 6; blk < rblocks * (M / 16); blk += THREADS / 64)
        fragment_block(sXh, sXl, dst, n_rows, rows_copied, rblocks, blk, ks_end, threadIdx.x & 63, lo_delta);
}

// four consecutive columns of one row -> one 8-byte store per plane
__device__ __forceinline__ void split_store4(_Float16* xh, _Float16* xl, int idx, const float4 v) {
    h4 hv, lv;
    hv[0] = (_Float16)v.x; hv[1] = (_Float16)v.y; hv[2] = (_Float16)v.z; hv[3] = (_Float16)v.w;
    *reinterpret_cast<h4*>(xh + idx) = hv;
    {
        lv[0] = (_Float16)(v.x - (float)hv[0]); lv[1] = (_Float16)(v.y - (float)hv[1]);
        lv[2] = (_Float16)(v.z - (float)hv[2]); lv[3] = (_Float16)(v.w - (float)hv[3]);
        *reinterpret_cast<h4*>(xl + idx) = lv;
    }
}

// (Re)build the trunk input tile [xyz embedding | zero pad to k0s | time code | zero pad to kt] of this workgroup's
// points.  `x` = the point of row threadIdx % M, read once at kernel start (raw-position mode).  The time-code
// loads are issued first (16-byte loads when the rows allow it) so that their L2 latency hides behind the sincos work.
// Thread -> (point row, part) of the input builders.  With four threads per row, 32 consecutive lanes are 8 rows x 4 parts:
// row stride 132 dwords (= 4 banks) x 8 rows + part stride 9 dwords x 4 parts touch 32 different banks, so the narrow
// (2- and 4-byte) stores of the encoders are conflict-free; lanes = 32 rows of one part were 4-way conflicted.
template <int M, int THREADS>
__device__ __forceinline__ int build_row(int tid) {
    if constexpr (THREADS / M == 4) return 8 * (tid >> 5) + (tid & 7);
    else return tid % M;
}
template <int M, int THREADS>
__device__ __forceinline__ int build_part(int tid) {
    if constexpr (THREADS / M == 4) return (tid >> 3) & 3;
    else return tid / M;
}

// two consecutive columns (idx even) -> one 4-byte store per plane
__device__ __forceinline__ void split_store2(_Float16* xh, _Float16* xl, int idx, float v0, float v1) {
    const h2 h = __builtin_amdgcn_cvt_pkrtz(v0, v1);
    *reinterpret_cast<h2*>(xh + idx) = h;
    *reinterpret_cast<h2*>(xl + idx) = __builtin_amdgcn_cvt_pkrtz(minus_lo_half(h, v0), minus_hi_half(h, v1));
}

// OCTAVE: allow the angle-doubling encoder (inference).  The training forward keeps one exact sincos per column: its
// gradients are compared with autograd of the reference network, whose ReLU pattern -- a function of sin(512 x) eight
// layers deep -- answers a 4-ulp change of the encoding with per-cent changes of single weight gradients.
// sin / cos of an embedding angle (|a| < 2^15; here up to ~800 rad) without the library's argument classification: three-step
// Cody-Waite reduction by pi/2 (the first two constants have so few bits that k C1, k C2 are exact; the fmas make the rest
// exact to the last step) and the cephes single-precision kernels on [-pi/4, pi/4] -- ~25 VALU instead of the library's two code
// paths under a per-lane branch; absolute error ~1e-7 (the encoder is compared with the reference's at 2e-6).
__device__ __forceinline__ void sincos_cw(float a, float* s, float* c) {
    const float k = __builtin_rintf(a * 0.63661977236758134f);
    float r = fmaf(k, -1.5703125f, a);
    r = fmaf(k, -4.837512969970703125e-4f, r);
    r = fmaf(k, -7.54978995489188216e-8f, r);
    const float r2 = r * r;
    float sp = fmaf(fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f);
    sp = fmaf(sp * r2, r, r);
    float cp = fmaf(fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f);
    cp = fmaf(cp * r2, r2, fmaf(-0.5f, r2, 1.0f));
    const int q = (int)k;
    const float ss = (q & 1) ? cp : sp, cc = (q & 1) ? sp : cp;
    *s = (q & 2) ? -ss : ss;
    *c = ((q + 1) & 2) ? -cc : cc;
}

// ILP: the hand-scheduled kernel's encoder -- one wave per SIMD with the whole register file to itself, dependent VALU chains are
// what it waits for: the three axes' range reductions may interleave and use sincos_cw; the eight-wave kernels keep the library
// call, one reduction at a time, for their register budget
template <int M, int THREADS, bool OCTAVE = true, bool ILP = false, class KA = H3KArgs>
__device__ __forceinline__ void build_input(_Float16* sXh, _Float16* sXl, const KA& a, long long p0, bool with_t,
                                            const float (&x)[3], int tid) {
    constexpr int G = THREADS / M;               // threads per point row
    constexpr int CH = 16 / G;                   // float4 chunks of a 64-column time-code segment per thread
    const int r = build_row<M, THREADS>(tid), q = build_part<M, THREADS>(tid);
    const long long p = p0 + r;
    const bool valid = p < a.n_points;
    const int base = r * LDH;
    const int k0s = (int)a.L.k0s, kt = (int)a.L.kt;
    const float* tsrc = nullptr;
    if (with_t && valid) tsrc = (a.xyz != nullptr) ? a.t_emb + (p / a.pts_per_ray) * a.in_t
                                                  : a.x_emb + p * a.ld_emb + a.off_t;
    // rows of 16-byte-aligned float4s: per-ray time codes with in_t % 4 == 0 (the pointer itself is checked on the host)
    const bool vec_t = with_t && a.xyz != nullptr && kt == 64 && (a.in_t & 3) == 0;
    // the first (up to) four chunks are requested before the sincos work, so that their L2 latency hides behind it
    constexpr int CH0 = CH < 4 ? CH : 4;
    float4 tv[CH0];
    if (vec_t) {
#pragma unroll
        for (int j = 0; j < CH0; ++j) {
            const int c = 4 * (q + j * G);
            tv[j] = (valid && c < a.in_t) ? *reinterpret_cast<const float4*>(tsrc + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    if (OCTAVE && (G == 4 || G == 2) && a.xyz != nullptr && a.octave_freqs) {
        // Part q encodes octaves [OPP q, OPP (q+1)) of all three axes = columns [3 + 6 OPP q, 3 + 6 OPP (q+1)): one sincos per
        // axis at its first octave, the following ones by angle doubling (freqs[f+1] == 2 freqs[f], checked on the host;
        // the parity-grade kernels re-anchor after two doublings, < 4 ulp); the values leave as 4-byte stores per plane
        // (the first and the last one of a part as 2-byte stores: parts start on odd columns).
        constexpr int OPP = G == 4 ? 3 : 5;                         // 12 / 10 octaves of capacity (n_freqs <= 10)
        float sn[3], cs[3];
        const int f0 = OPP * q;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            sn[c] = 0.f; cs[c] = 0.f;
            if (f0 < a.n_freqs) {
                if constexpr (ILP) sincos_cw(a.freqs[f0] * x[c], &sn[c], &cs[c]);
                else sincosf(a.freqs[f0] * x[c], &sn[c], &cs[c]);
            }
            if constexpr (!ILP) __builtin_amdgcn_sched_barrier(0);  // one range reduction at a time (register pressure)
        }
        const int c0 = 3 + 6 * OPP * q;                             // odd: first value alone, then even-aligned pairs
        float carry = 0.f;                                          // cos of the last axis waits for the next octave's first sin
#pragma unroll
        for (int k = 0; k < OPP; ++k) {
            if (f0 + k >= a.n_freqs) { sn[0] = sn[1] = sn[2] = cs[0] = cs[1] = cs[2] = 0.f; }   // (zero padding up to k0s)
            const int ck = c0 + 6 * k;                              // [sin x3 | cos x3] of octave f0 + k
            if (k == 0) split_store(sXh, sXl, base + ck, sn[0]);
            else if (ck - 1 < k0s) split_store2(sXh, sXl, base + ck - 1, carry, sn[0]);
            if (ck + 1 < k0s) split_store2(sXh, sXl, base + ck + 1, sn[1], sn[2]);
            if (ck + 3 < k0s) split_store2(sXh, sXl, base + ck + 3, cs[0], cs[1]);
            carry = cs[2];
            if (k + 1 < OPP) {
                if (k == 2 && f0 + 3 < a.n_freqs) {        // (five octaves per part only)
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        if constexpr (ILP) sincos_cw(a.freqs[f0 + 3] * x[c], &sn[c], &cs[c]);
                        else sincosf(a.freqs[f0 + 3] * x[c], &sn[c], &cs[c]);
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float s2 = 2.f * sn[c] * cs[c], c2 = fmaf(-2.f * sn[c], sn[c], 1.f);
                        sn[c] = s2; cs[c] = c2;
                    }
                }
            }
        }
        if (c0 + 6 * OPP - 1 < k0s) split_store(sXh, sXl, base + c0 + 6 * OPP - 1, carry);
        if (q == G - 1)                                             // columns past the last part's range (G = 2: column 63)
            for (int c = 3 + 6 * OPP * G; c < k0s; ++c) split_store(sXh, sXl, base + c, 0.f);
        if (q == 0) {
            split_store2(sXh, sXl, base + 0, x[0], x[1]);
            split_store(sXh, sXl, base + 2, x[2]);
        }
    } else if (a.xyz != nullptr) {
        if (q == 0) {
            split_store(sXh, sXl, base + 0, x[0]); split_store(sXh, sXl, base + 1, x[1]);
            split_store(sXh, sXl, base + 2, x[2]);
            for (int c = a.in_xyz; c < k0s; ++c) split_store(sXh, sXl, base + c, 0.f);
        }
        const int nf3 = 3 * a.n_freqs;
        for (int j = q; j < nf3; j += G) {
            const int f = j / 3, c = j - 3 * f;
            float s, co;
            sincosf(a.freqs[f] * x[c], &s, &co);
            split_store(sXh, sXl, base + 3 + 6 * f + c, s);
            split_store(sXh, sXl, base + 3 + 6 * f + 3 + c, co);
        }
    } else {
        const float* src = a.x_emb + p * a.ld_emb + a.off_xyz;
        for (int c = q; c < k0s; c += G) split_store(sXh, sXl, base + c, (valid && c < a.in_xyz) ? src[c] : 0.f);
    }
    if (vec_t) {
#pragma unroll
        for (int j = 0; j < CH0; ++j) split_store4(sXh, sXl, base + k0s + 4 * (q + j * G), tv[j]);
        if constexpr (CH > CH0) {                     // two threads per row: the second half of the 16 chunks
#pragma unroll
            for (int j = CH0; j < CH; ++j) {
                const int c = 4 * (q + j * G);
                tv[j - CH0] = (valid && c < a.in_t) ? *reinterpret_cast<const float4*>(tsrc + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int j = CH0; j < CH; ++j) split_store4(sXh, sXl, base + k0s + 4 * (q + j * G), tv[j - CH0]);
        }
    } else if (with_t) {
        for (int c = q; c < kt; c += G)
            split_store(sXh, sXl, base + k0s + c, (valid && c < a.in_t) ? tsrc[c] : 0.f);
    }
}

// the time-code columns [k0s, k0s + kt) of the trunk input tile alone (the position part is someone else's): float4 rows
template <int M, int THREADS, class KA = H3KArgs>
__device__ __forceinline__ void build_time_part(_Float16* sXh, _Float16* sXl, const KA& a, long long p0, int tid) {
    constexpr int G = THREADS / M;
    constexpr int CH = 16 / G;
    const int r = build_row<M, THREADS>(tid), q = build_part<M, THREADS>(tid);
    const long long p = p0 + r;
    const bool valid = p < a.n_points;
    const int k0s = (int)a.L.k0s;
    const float* tsrc = a.t_emb + ((valid ? p : a.n_points - 1) / a.pts_per_ray) * a.in_t;
    float4 tv[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        const int c = 4 * (q + j * G);
        tv[j] = (valid && c < a.in_t) ? *reinterpret_cast<const float4*>(tsrc + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) split_store4(sXh, sXl, r * LDH + k0s + 4 * (q + j * G), tv[j]);
}

template <int M, int THREADS>
__device__ __forceinline__ void build_side(_Float16* sXh, _Float16* sXl, const H3KArgs& a, long long p0, int tid) {
    constexpr int G = THREADS / M;
    const int r = build_row<M, THREADS>(tid), q = build_part<M, THREADS>(tid);
    const long long p = p0 + r;
    const bool valid = p < a.n_points;
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
    for (int c = q; c < sk; c += G) {
        float v = 0.f;
        if (valid) {
            if (c < a.in_dir) v = sd[c];
            else if (c < a.in_dir + a.in_a) v = sa[c - a.in_dir];
        }
        split_store(sXh, sXl, r * LDH + c, v);
    }
}

enum { ACT_NONE = 0, ACT_SIGMOID = 1, ACT_FLOW = 2 };

// Narrow heads as one zero-padded 32-row MFMA tile per 32 points.  The workgroup's NW waves split the NPT point
// tiles AND (when NW > NPT) the K range: wave w takes point tile w % NPT and k-steps [16/KS * (w / NPT), ...), the
// partial sums of the upper k ranges travel through `sRed` (one barrier).
// `pre` (use_pre): the wave's first head k-steps, already requested into the weight ring by the tail of the last GEMM
// (gemm_seg `next`) -- four with 32-neuron waves, all eight with 64-neuron waves (a ring slot holds two head k-steps
// there); `wrest` then points at the first k-step not in the ring.  With 32-neuron waves (registers to spare: no spills)
// the three product terms run as three independent accumulator chains; one chain makes every MFMA wait for the one
// before it (24 x the full MFMA latency), which the 64-neuron tilings accept: three chains spill 14 VGPRs there.
// out row = (r&3) + 8*(r>>2) + 4*(lane>>5); only r < 8 (rows < 16) can be live.
template <int NPT, int NW, int MTW>
__device__ __forceinline__ void heads(const _Float16* sXh, const _Float16* sXl, float* sRed, const uint32_t* __restrict__ pk,
                                      uint32_t w_off, uint32_t b_off, int n_rows, unsigned kinds, float flow_scale,
                                      float* sRaw, int slot0, int wave, int lane,
                                      const WRing<MTW>& pre, bool use_pre, const uint4* __restrict__ wrest) {
    constexpr int KS = NW / NPT;                 // k-splits (waves beyond NPT * KS idle)
    constexpr int NK = 16 / KS;                  // k-steps per wave
    static_assert(KS == 1 || KS == 2, "heads: 1 or 2 waves per point tile");
    const int pt = wave % NPT, kh = wave / NPT;
    if (KS == 1 && wave >= NPT) return;
    constexpr int CH = MTW == 1 ? 3 : 1;                 // accumulator chains
    f32x16 acc0[CH];
    H3_HSTAMP(0);
    // the biases of this lane's (up to) eight rows: requested now, used behind the MFMAs and the k-split exchange (round-3 head
    // stamps of the last head call: weights + MFMAs 3.4 k cycles, k-split exchange 1.0 k, bias + activations + record image 2.4 k,
    // final barrier 1.7 k, records 0.8 k; -1.7 % per launch on one box, +-0 on another.  Requesting the second half of the
    // head weights behind the last GEMM as well costs 19 spilled registers: +2.5 %, not kept)
    // (32-neuron waves only: the 64-neuron tilings have no registers to spare)
    constexpr bool EARLY_BIAS = MTW == 1;
    const float* bias = reinterpret_cast<const float*>(pk + b_off);
    [[maybe_unused]] float bv[8];
    if constexpr (EARLY_BIAS) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            bv[r] = row < n_rows ? bias[row] : 0.f;
        }
    }
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[c][r] = 0.f;
    const uint4* w = reinterpret_cast<const uint4*>(pk + w_off) + lane + kh * NK * 2 * 64;
    const _Float16* bh = sXh + (32 * pt + (lane & 31)) * LDH + 8 * (lane >> 5) + kh * NK * 16;
    const _Float16* bl = sXl + (32 * pt + (lane & 31)) * LDH + 8 * (lane >> 5) + kh * NK * 16;
    // weights are requested eight k-steps (8 KiB per wave) at a time, ahead of their MFMAs
#pragma unroll
    for (int half = 0; half < NK / 8; ++half) {
        h8 whv[8], wlv[8];
        bool have = false;
        if (half == 0 && use_pre) {
            have = true;
            if constexpr (MTW == 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { whv[j] = pre.r[j].wh[0]; wlv[j] = pre.r[j].wl[0]; }
#pragma unroll
                for (int j = 4; j < 8; ++j) {
                    whv[j] = __builtin_bit_cast(h8, wrest[((j - 4) * 2 + 0) * 64]);
                    wlv[j] = __builtin_bit_cast(h8, wrest[((j - 4) * 2 + 1) * 64]);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    whv[2 * j] = pre.r[j].wh[0]; wlv[2 * j] = pre.r[j].wl[0];
                    whv[2 * j + 1] = pre.r[j].wh[1]; wlv[2 * j + 1] = pre.r[j].wl[1];
                }
            }
        }
        if (!have) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                whv[j] = __builtin_bit_cast(h8, w[((half * 8 + j) * 2 + 0) * 64]);
                wlv[j] = __builtin_bit_cast(h8, w[((half * 8 + j) * 2 + 1) * 64]);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ks = half * 8 + j;
            const h8 xh = lds_h8(bh + ks * 16);
            const h8 xl = lds_h8(bl + ks * 16);
            acc0[0] = MFMA_H(wlv[j], xh, acc0[0]);
            acc0[CH > 1 ? 1 : 0] = MFMA_H(whv[j], xl, acc0[CH > 1 ? 1 : 0]);
            acc0[CH > 1 ? 2 : 0] = MFMA_H(whv[j], xh, acc0[CH > 1 ? 2 : 0]);
        }
    }
    float part[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        part[r] = acc0[0][r];
        if constexpr (CH > 1) part[r] = (part[r] + acc0[1][r]) + acc0[2][r];
    }
    H3_HSTAMP(1);
    if constexpr (KS == 2) {
        if (kh == 1) {
#pragma unroll
            for (int r = 0; r < 8; ++r) sRed[(pt * 8 + r) * 64 + lane] = part[r];
        }
        __syncthreads();
        if (kh == 1) return;
#pragma unroll
        for (int r = 0; r < 8; ++r) part[r] += sRed[(pt * 8 + r) * 64 + lane];
    }
    H3_HSTAMP(2);
    // the values go to the tile's raw-record image in LDS; the kernel writes whole 64-byte records at its end
    float* rec = sRaw + (32 * pt + (lane & 31)) * NSFF_RAW_STRIDE + slot0;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < n_rows) {
            float v = part[r] + (EARLY_BIAS ? bv[r] : bias[row]);
            const unsigned kind = (kinds >> (2 * row)) & 3u;
            if (kind == ACT_SIGMOID) v = 1.0f / (1.0f + expf(-v));
            else if (kind == ACT_FLOW) v = flow_scale * tanhf(v);
            rec[row] = v;
        }
    }
    H3_HSTAMP(3);
}

// NT = point tiles (of 32) per wave, WM = wave rows: the workgroup has 4*WM waves and 32*NT*WM points.
//   <2,1>: 64 points, 2 workgroups per CU;  <4,1>: 128 points, 1 workgroup per CU, 1 wave per SIMD;
//   <2,2>: 128 points, 8 waves (2 per SIMD): the two wave rows request identical weight lines back to
//          back, so the L1 merges them and the L2 weight stream per FLOP is halved.
// SAVE: training forward -- epilogues also write their activations (fp16) to HBM for the backward pass.
//   <4,1,*,1>: 128 points, EIGHT waves of 32 neurons each (MTW = 1): every weight byte is fetched once per 128
//          points (half the L2 stream of <2,1>) by exactly one wave, and two waves per SIMD hide each other's epilogues.
template <int NT, int WM, bool SAVE = false, int MTW = 2>
__global__ __launch_bounds__(256 * WM * (3 - MTW), (NT == 2 ? 2 : 1)) void nsff_field_kernel_h3(const H3KArgs a) {
    constexpr int M = 32 * NT * WM;
    constexpr int THREADS = 256 * WM * (3 - MTW);
    static_assert(MTW == 2 || WM == 1, "the 32-neuron-per-wave variant has a single row of point tiles");
    __shared__ __attribute__((aligned(16))) _Float16 sX[2 * M * LDH];
    constexpr int NPT = M / 32, NW = THREADS / 64;                       // heads: point tiles, waves
    __shared__ float sRed[NW > NPT ? NPT * 8 * 64 : 1];                  // k-split partial sums of the heads
    // raw records of the tile (16 floats per point): the heads fill them in, the kernel's last act writes them out as
    // whole 64-byte rows (scattered 4-byte stores made the HBM side read-modify-write every record: 5x the bytes)
    __shared__ __attribute__((aligned(16))) float sRaw[M * NSFF_RAW_STRIDE];
    for (int i = threadIdx.x; i < M * NSFF_RAW_STRIDE; i += THREADS) sRaw[i] = 0.f;
    _Float16* sXh = sX;
    _Float16* sXl = sX + M * LDH;
    const int lane = threadIdx.x & 63;
    const int wave_id = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = MTW == 2 ? (wave_id & 3) : (wave_id >> 1);   // 64-neuron block of the packed weight stream
    const int mt0 = MTW == 2 ? 0 : (wave_id & 1);                  // this wave's 32-neuron half of it (MTW = 1)
    const int nb0 = 64 * wave + 32 * mt0;                          // first neuron of this wave
    const int nt0 = MTW == 2 ? (wave_id >> 2) * NT : 0;            // first point tile of this wave
    // Both trunks in one workgroup stream 4.6 MB of (hi + lo) weights -- more than the 4 MB L2 of an XCD, so with tiles
    // in every phase at once 4-5 % of the weight requests miss to the Infinity Cache.  In split mode a workgroup runs ONE
    // trunk of its tile; workgroups are dispatched in index order, so the chip streams the static trunk's 2.2 MB for the
    // first half of the grid and the dynamic trunk's 2.4 MB for the second -- each fits the L2.
    long long tile = blockIdx.x;
    int s_begin = 0, s_end = a.n_steps, piece = 0;               // piece: 0 whole raw record, 1 static slots [0,4), 2 slots [4,16)
    if (a.split_trunks) {
        if (tile < a.grid_tiles) { s_end = a.n_static_steps; piece = 1; }
        else { tile -= a.grid_tiles; s_begin = a.n_static_steps; piece = 2; }
    }
    const long long p0 = tile * M;
    const uint32_t* __restrict__ pk = a.packed;
    const _Float16* sBh = sXh + (32 * nt0 + (lane & 31)) * LDH + 8 * (lane >> 5);
    const _Float16* sBl = sXl + (32 * nt0 + (lane & 31)) * LDH + 8 * (lane >> 5);

    f32x16 acc[MTW][NT];
    WRing<MTW> ring;
    BiasRegs<MTW> br;
    auto seg = [&](uint32_t off, int nks) {
        return reinterpret_cast<const uint4*>(pk + off) + (wave * nks) * 4 * 64 + mt0 * 128 + lane;
    };
    auto fbias = [&](uint32_t off) { return reinterpret_cast<const float*>(pk + off); };

    // raw-position mode: this thread's point, read once (it is re-encoded up to four times: both trunks, layer 0 + skip)
    // (the 128-point fast kernel has no registers to spare for it and re-reads the point in every build instead)
    constexpr bool KEEP_POINT = true;
    float px[3] = {0.f, 0.f, 0.f};
    auto read_point = [&]() {
        const long long bp = p0 + build_row<M, THREADS>(threadIdx.x);
        if (a.xyz != nullptr && bp < a.n_points) { px[0] = a.xyz[bp * 3 + 0]; px[1] = a.xyz[bp * 3 + 1]; px[2] = a.xyz[bp * 3 + 2]; }
    };
    if constexpr (KEEP_POINT) read_point();

    // weights of step 0 start their L2 round trip before the tile's input is even encoded
    // Every workgroup streams the same weights; if all of them walk the program in the same order
    // they hit the same few L2 channels at the same instant.  Half of the workgroups therefore run
    // the dynamic trunk first (the two trunks are independent: separate outputs, the tile is rebuilt).
    const int rot = (!a.split_trunks && a.n_static_steps > 0 && a.n_static_steps < a.n_steps && ((blockIdx.x >> 3) & 1))
                        ? a.n_static_steps : 0;
    auto step_at = [&](int i) { int j = i + rot; if (j >= a.n_steps) j -= a.n_steps; return a.steps[j]; };
    // Training forward: the HBM copy of the tile an epilogue (or an input build) leaves in LDS rides inside the next
    // GEMM that reads the same tile -- one (row pair, 8 points) task per four k-steps, behind the group's weight refills
    // (vmcnt counts loads and stores in one in-order queue: a burst of stores in front of a GEMM holds its refills
    // back until every store is acknowledged).  A tile that is rebuilt or abandoned first is flushed on the spot.
    [[maybe_unused]] _Float16* pend_dst = nullptr;
    [[maybe_unused]] int pend_rows = 0, pend_copied = 0, pend_rblocks = 1, pend_total = 0;
    [[maybe_unused]] long long pend_lo = 0;
    [[maybe_unused]] const long long tile64 = tile * (M / 64);
    [[maybe_unused]] const int ks_end = (M == 128 && tile64 + 1 >= a.n_tiles) ? 4 : M / 16;
    auto pend_set = [&](_Float16* dst, int n_rows, int rows_copied, long long lo_delta) {
        pend_dst = dst; pend_rows = n_rows; pend_copied = rows_copied; pend_rblocks = (rows_copied + 31) >> 5; pend_lo = lo_delta;
        pend_total = fragment_blocks<M>(rows_copied);
    };
    auto pend_flush = [&]() {
        if constexpr (SAVE) {
            if (pend_dst != nullptr) tile_to_fragments<THREADS, M>(sXh, sXl, pend_dst, pend_rows, pend_copied, ks_end, pend_lo);
            pend_dst = nullptr;
        }
    };
    // static sigma reads the last trunk activation, before *_final (nerf.py:169);
    // dynamic rows: rgb(3) sigmoid, sigma raw, fw(3)/bw(3) = flow_scale*tanh (nerf.py:197-208)
    struct HeadSel { uint32_t w_off, b_off; int n_rows, slot0; unsigned kinds; };
    auto head_sel = [&](int head) {
        HeadSel h{a.L.s_sigma_w, a.L.s_sigma_b, 1, 3, (unsigned)ACT_NONE};
        if (head == HEAD_S_RGB) h = HeadSel{a.L.s_rgb_w, a.L.s_rgb_b, 3, 0, 0x15u};
        if (head == HEAD_S_FOLD) h = HeadSel{a.L.s_fold_w, a.L.s_fold_b, 4, 0, 0x15u};
        if (head == HEAD_T) h = HeadSel{a.L.t_head_w, a.L.t_head_b, (int)a.L.t_head_rows, 4, 0x15u | (0xAAAu << 8)};
        if (head == HEAD_T_FOLD) h = HeadSel{a.L.t_fold_w, a.L.t_fold_b, (int)a.L.t_head_rows, 4, 0x15u | (0xAAAu << 8)};
        return h;
    };
    H3_SPAN(0);
    const SpanT span0 = span_begin(a.span);
    const H3Step s0 = step_at(s_begin);
    const uint4* wnext = prefetch_w<MTW>(ring, seg(s0.w_off, s0.nks));
    load_bias<MTW>(br, fbias(s0.bias_off), nb0, lane);
#pragma unroll 1
    for (int i = s_begin; i < s_end; ++i) {
        const H3Step st = step_at(i);
        H3_STAMP(0);
        if (st.pre != PRE_NONE) {
            pend_flush();                          // (a tile nobody multiplied after it was saved: the last one of a trunk)
            __syncthreads();                       // everyone is done reading the previous tile
            if (st.pre == PRE_SIDE) {
                build_side<M, THREADS>(sXh, sXl, a, p0, threadIdx.x);
            } else {
                if constexpr (!KEEP_POINT) read_point();
                build_input<M, THREADS, !SAVE>(sXh, sXl, a, p0, st.pre == PRE_INPUT_T, px, threadIdx.x);
            }
            __syncthreads();
            if constexpr (SAVE) {
                if (st.pre == PRE_SIDE) {
                    if (a.save_side != nullptr) pend_set(a.save_side + tile64 * (64 * a.side_rows), a.side_rows, (int)a.L.side_k, a.lo_delta[2]);
                } else if (a.save_xin != nullptr && st.bias_off != NSFF_NONE && (st.pre == PRE_INPUT_T || a.transient_mode == 0)) {
                    // trunk input of layer 0: columns [0, k0s) xyz embedding, [k0s, k0s + kt) time code; rows past what this
                    // launch encodes stay unwritten (the caller zeroes the buffer then)
                    pend_set(a.save_xin + tile64 * (64 * a.xin_rows), a.xin_rows,
                             (int)a.L.k0s + (st.pre == PRE_INPUT_T ? (int)a.L.kt : 0), a.lo_delta[1]);
                }
            }
        }
        H3_STAMP(1);
        if (st.bias_off != NSFF_NONE) acc_init<NT, MTW>(acc, br);
        // A step with output heads: this wave's share of the head tile is what it multiplies next -- the GEMM's last four
        // k-steps request it into the ring slots they free (gemm_seg `next`), so the head starts with its weights on chip
        // (stamps: heads + records 16.2k -> 11.9k cycles per trunk).  The same hand-over for the NEXT LAYER's weights was
        // measured and dropped: the loads cost the GEMM what they save in front of the barrier (+1.3k / -0.9k cycles).
        const bool has_next = i + 1 < s_end;
        const H3Step nx = step_at(has_next ? i + 1 : i);
        const HeadSel hs = head_sel(st.head);
        constexpr bool HEAD_PREFETCH = NW / NPT == 2;                    // (k-split heads)
        const bool head_next = HEAD_PREFETCH && st.head != HEAD_NONE;
        const uint4* nextp = nullptr;
        if (head_next) nextp = reinterpret_cast<const uint4*>(pk + hs.w_off) + lane + (wave_id / NPT) * 8 * 2 * 64;
        const int nstride = MTW == 1 ? 128 : 256;
        if constexpr (SAVE) {
            // (one instantiation of the GEMM loop; the copy is switched by a wave-uniform flag)
            const bool copy = H3_SAVE_INTERLEAVE && pend_dst != nullptr;
            const uint4* wafter = gemm_seg<NT, MTW>(acc, ring, wnext, b_rows<NT>(sBh, sBl, LDH), st.nks,
                [&](int j) {                     // two blocks per wave behind each group of four k-steps (64 blocks, NW * 4 calls)
#pragma unroll
                    for (int u = 2 * j; u < 2 * j + 2; ++u) {
                        const int blk = wave_id + NW * u;
                        if (copy && blk < pend_total)
                            fragment_block(sXh, sXl, pend_dst, pend_rows, pend_copied, pend_rblocks, blk, ks_end, lane, pend_lo);
                    }
                },
                [&](int j) {
                    if (copy) {
#pragma unroll 1
                        for (int blk = wave_id + NW * 2 * j; blk < pend_total; blk += NW) {
                            fragment_block(sXh, sXl, pend_dst, pend_rows, pend_copied, pend_rblocks, blk, ks_end, lane, pend_lo);
                            H3_PIN();
                        }
                    }
                }, nextp, nstride);
            wnext = wafter;
            if (copy) pend_dst = nullptr;
        } else {
            wnext = gemm_seg<NT, MTW>(acc, ring, wnext, b_rows<NT>(sBh, sBl, LDH), st.nks, NoSide{}, NoSide{}, nextp, nstride);
        }
        H3_STAMP(2);
        if (has_next) {                            // next segment's weights + bias fly during the epilogue
            if (!head_next) wnext = prefetch_w<MTW>(ring, seg(nx.w_off, nx.nks));
            if (nx.bias_off != NSFF_NONE) load_bias<MTW>(br, fbias(nx.bias_off), nb0, lane);
        }
        if (st.post != POST_NONE) {
            pend_flush();                          // (only when the interleave is compiled out: the GEMM above took it along)
            __syncthreads();
            H3_STAMP(3);
            unsigned long long* mk = nullptr;
            if constexpr (SAVE) {
                if (st.save && a.save_masks != nullptr && st.post == POST_RELU)
                    mk = a.save_masks + ((long long)(st.save - 1) * a.n_tiles + tile64) * 256;   // (per-thread slot: acc_store)
            }
            if (st.post == POST_RELU) {
                if (SAVE && mk != nullptr) acc_store<NT, true, MTW, SAVE>(sXh, sXl, acc, nb0, nt0, lane, mk, ks_end > 4);
                else acc_store<NT, true, MTW>(sXh, sXl, acc, nb0, nt0, lane);
            } else acc_store<NT, false, MTW>(sXh, sXl, acc, nb0, nt0, lane);
            H3_STAMP(4);
            __syncthreads();
            H3_STAMP(5);
            if constexpr (SAVE) {
                if (st.save && a.save_acts != nullptr)
                    pend_set(a.save_acts + (long long)(st.save - 1) * a.save_stride + tile64 * (64 * NSFF_W), NSFF_W, NSFF_W, a.lo_delta[0]);
            }
            if (st.head != HEAD_NONE) {
                heads<NPT, NW, MTW>(sXh, sXl, sRed, pk, hs.w_off, hs.b_off, hs.n_rows, hs.kinds, a.flow_scale, sRaw, hs.slot0,
                                           wave_id, lane, ring, head_next, wnext);
                // (a head in the middle of a trunk -- training forward, view directions: the ring carried its weights, so the
                //  next segment's first k-steps are requested here, beside the head's activation arithmetic)
                if (has_next && head_next) wnext = prefetch_w<MTW>(ring, seg(nx.w_off, nx.nks));
            }
        }
    }
    pend_flush();
    { [[maybe_unused]] const int wave = wave_id; H3_HSTAMP(4); }
    __syncthreads();
    { [[maybe_unused]] const int wave = wave_id; H3_HSTAMP(5); }
    for (int i = threadIdx.x; i < M * (NSFF_RAW_STRIDE / 4); i += THREADS) {
        const long long p = p0 + i / (NSFF_RAW_STRIDE / 4);
        const int q4 = i % (NSFF_RAW_STRIDE / 4);                   // 16-byte quarter of the record
        if (p < a.n_points && (piece == 0 || (piece == 1) == (q4 == 0)))
            reinterpret_cast<float4*>(a.raw)[p0 * (NSFF_RAW_STRIDE / 4) + i] = reinterpret_cast<const float4*>(sRaw)[i];
    }
    H3_SPAN(1);
    span_end(a.span, span0);
    { [[maybe_unused]] const int i = 31; H3_STAMP(0); }   // (timing build) end of the workgroup's work: slot 31, stamp 0
}



// Heads of nsff_field_kernel_h3a: the body's HEAD phase (tools/h3asm/gen.py::head_body) multiplies the (folded) head tile with
// the trunk's last activation -- wave w on points 32 w .. 32 w + 31, three accumulator chains, the tile requested in front of
// the trunk's last epilogue -- and leaves the pre-activation sums in the raw-record image; the kernel's records loop adds the
// bias and applies the activation with the hardware exp / rcp (v_exp_f32, v_rcp_f32: ~1e-6 relative; the outputs are
// compared at 1e-4 of their maximum): sigmoid(v) = 1 / (1 + 2^(-v log2 e)), tanh(v) = 1 - 2 / (2^(2 v log2 e) + 1).
__device__ __forceinline__ float fast_sigmoid(float v) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
}
__device__ __forceinline__ float fast_tanh(float v) {
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(2.8853900817779268f * v) + 1.0f);
}
// ---------------------------------------------------------------------------------------------------------------------
// nsff_field_kernel_h3a: the f16x3 inference trunk with a HAND-SCHEDULED body (tools/h3asm/gen.py -> field_h3a_body.inc).
// One wave per SIMD (four waves x 512 registers), 128 points per workgroup as two 64-point halves; a wave owns 64 neurons,
// the current layer's weights of those neurons are resident in its 256 accumulation registers and are multiplied with both
// halves half a layer apart, so that one half's epilogue (ReLU, hi / lo split, LDS stores) and the next layer's weight stream
// run in the MFMA shadow of the other half: 192 MFMAs (6 144 matrix-pipe cycles) between two barriers, every weight byte
// crosses the CU's vector-memory path once per 128 points, every activation fragment is read from LDS once per wave.
// C++ keeps what is not the trunk: the input encoder (build_input), the bias table, the heads and the raw records.
// The body executes a PHASE PROGRAM built on the host (h3a_build_program): one 8-dword descriptor per phase, fetched from
// the kernel-argument segment with scalar loads one phase ahead.
#ifdef H3_TIMING
#include "field_h3a_body_timing.inc"     // (python3 tools/h3asm/gen.py --timing: a stamp per phase, `make timing`)
#elif defined(H3A_BODY_INC)
#include H3A_BODY_INC                    // (timing experiments: make variant NAME=x DEFS='-DH3A_BODY_INC=\"field_h3a_body_x.inc\"')
#else
#include "field_h3a_body.inc"
#endif
// The heads evaluated on a trunk's last activation (HEAD_* of its last step): tile / bias offsets in the packed buffer (words),
// rows, first float of the raw record, activation kinds (two bits per row).
struct H3AHeadSel { uint32_t w_off, b_off; int n_rows, slot0; unsigned kinds; };
__host__ __device__ __forceinline__ H3AHeadSel h3a_head_sel(const NsffLayoutH3& L, int head) {
    if (head == HEAD_S_FOLD) return H3AHeadSel{L.s_fold_w, L.s_fold_b, 4, 0, 0x15u};
    if (head == HEAD_T_FOLD) return H3AHeadSel{L.t_fold_w, L.t_fold_b, (int)L.t_head_rows, 4, 0x15u | (0xAAAu << 8)};
    if (head == HEAD_S_RGB) return H3AHeadSel{L.s_rgb_w, L.s_rgb_b, 3, 0, 0x15u};
    return H3AHeadSel{L.s_sigma_w, L.s_sigma_b, 1, 3, (unsigned)ACT_NONE};
}

constexpr int H3A_MAX_PHASES = 36, H3A_MAX_BIAS = 16;
constexpr uint32_t H3A_TB_ROW = 0x80000000u, H3A_TB_HALF_B = 0x100u;    // bias_off entry: row (entry & 0xff) of the per-ray table, of the ray of half A / B
struct H3APhase { uint32_t d[8]; };     // body, flags, bias table offset, n1, r1 offset / wave stride, r2 offset / wave stride (bytes)
struct H3AArgs {
    H3KArgs k;
    H3APhase ph[2][H3A_MAX_PHASES];     // [static trunk, dynamic trunk]
    __attribute__((aligned(16))) uint32_t bias_off[2][H3A_MAX_BIAS]; // packed word offset of bias table row i, or H3A_TB_ROW | [H3A_TB_HALF_B] | row of H3KArgs::t_bias
    int n_bias[2];
    int head[2];                        // HEAD_* evaluated on the trunk's last activation
    H3AHeadSel hsel[2];                 // ... and what that means (h3a_head_sel of the host): one scalar load in the kernel
    // PERSISTENT launch (p_mode != 0): the grid is one workgroup per CU and a workgroup walks tiles tile0, tile0 + stride, ... of ONE
    // trunk -- everything that does not depend on the tile (the plain bias rows, the head biases, the record image's zeros) is set
    // up once, the next tile's first eight weight slots are requested by the last trunk phase of the current one (B16LP: the phase
    // programs of such a launch went through h3a_make_persistent) and its point by the body's first instructions.  1: one trunk in the launch (whole records); 2: both trunks -- workgroup b runs
    // on XCD b % 8 (observed, not promised: only L2 locality depends on it), XCDs 0..3 take the static trunk, 4..7 the dynamic one,
    // each trunk's 2.3 MB of weights stay in its XCDs' L2s; 3: the dynamic trunk of a launch whose static trunk is another kernel's;
    // 4: both trunks of unequal cost (a view-direction static trunk is 23 % longer, the time code through the matrix pipe makes the
    // dynamic one 7 % longer): trunk by XCD exactly as in mode 2 -- an XCD's 4 MB L2 holds ONE trunk's weights (both are 4.7 MB:
    // the earlier split by workgroup index put both trunks on every XCD and fetched ~2.7 GB per launch, 18x the algorithmic
    // bytes) -- and the LONGER trunk p_long keeps only its tiles [0, p_split) for its own XCDs: tiles [p_split, p_tiles) are a SECOND
    // ROUND of workgroups (grid = 2 x compute units; blockIdx.x >= gridDim.x / 2), which the dispatcher starts where compute
    // units free up first -- on the XCDs of the shorter trunk, whose second-round workgroups take the longer trunk's tail (one L2
    // refill per XCD and launch); second-round workgroups that land on the longer trunk's XCDs have no tile and leave at once.
    // p_split is chosen so that both kinds of XCD finish together.  Every (tile, trunk) pair is computed exactly once whatever
    // the dispatcher does: only L2 locality depends on where a workgroup runs.
    int p_mode, p_split;
    long long p_tiles;                  // 128-point tiles of the launch
    int p_long;                         // mode 4: the longer trunk (0 static, 1 dynamic)
    int sig_ride;                       // static trunk with the view-direction branch: sigma = sum of the 8 partial sums the body's
                                        // sigma ride left at floats 4..11 of the record image + the bias at packed word sig_b_off
    uint32_t sig_b_off;
};
static_assert(sizeof(H3AArgs) <= 4096, "kernel arguments must fit the 4 KiB kernarg segment");

// Weight slots 0..7 (the first segment and the start of the second), requested in front of / inside the encoder: statement K
// loads slot K's 4 KiB of this wave from (K < n1 ? r1 : r2) + 4096 K -- the fields of phase descriptor 0, as the body's refills.
struct H3APre {
    unsigned long long pk = 0;
    unsigned r1 = 0, r2 = 0, n1 = 0, lane16 = 0;
    bool on = false;          // (wave-uniform) false: the slots of this tile were requested by the previous tile's B16LP phase (persistent launch)
    template <int K> __device__ __forceinline__ void slot() const {
        if (!on) return;
        const unsigned off = (K < (int)n1 ? r1 : r2) + 4096u * K;
#define H3A_SLOT_CASE(k) if constexpr (K == k) asm volatile(H3A_PRE_SLOT##k : : [pk] "s"(pk), [off] "s"(off), [lane16] "v"(lane16) : H3A_PRE_SLOT##k##_CLOBBERS)
        H3A_SLOT_CASE(0); H3A_SLOT_CASE(1); H3A_SLOT_CASE(2); H3A_SLOT_CASE(3);
        H3A_SLOT_CASE(4); H3A_SLOT_CASE(5); H3A_SLOT_CASE(6); H3A_SLOT_CASE(7);
#undef H3A_SLOT_CASE
    }
};

// The hand-scheduled kernel's position encoder for the reference's embedding (ten octaves, 63 columns padded to 64): nothing
// is decided at run time.  Two threads per point row; thread part q (wave-uniform: waves 0, 1 / 2, 3) encodes octaves 5 q .. 5 q + 4
// of the three axes -- a sincos at its first and fourth octave, the others by angle doubling (as build_input's OCTAVE path) --
// and stores its 33 / 31 columns as 16-byte LDS stores per plane: q = 0 columns 0..31 and column 32, q = 1 column 33, 34..35,
// 36..39, 40..63 (column 63 = 0).  Rows of points past the end are encoded like any other (their records are never stored).
__device__ __forceinline__ void h3a_split2(float v0, float v1, unsigned& h, unsigned& l) {
    const h2 hh = __builtin_amdgcn_cvt_pkrtz(v0, v1);
    const h2 ll = __builtin_amdgcn_cvt_pkrtz(minus_lo_half(hh, v0), minus_hi_half(hh, v1));
    h = __builtin_bit_cast(unsigned, hh);
    l = __builtin_bit_cast(unsigned, ll);
}
__device__ __forceinline__ void h3a_store8(_Float16* rh, _Float16* rl, int col, const float* w) {
    unsigned h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) h3a_split2(w[2 * i], w[2 * i + 1], h[i], l[i]);
    *reinterpret_cast<u4v*>(rh + col) = u4v{h[0], h[1], h[2], h[3]};
    *reinterpret_cast<u4v*>(rl + col) = u4v{l[0], l[1], l[2], l[3]};
}
// `pre`: the weight pre-issue (H3APre below), one slot = four 16-byte loads per lane at a time: the eight statements are spread over
// the encoder so that the 128 KiB cross the CU's vector-memory path (2 k cycles at 64 B / clock) WHILE it computes -- issued
// back to back they hold every wave at the issue stage for that long.  Slots 0..2 are the caller's: requested behind its own
// loads (the point, the bias rows) and landed with them -- the compiler's wait for the point cannot see the statements' loads
// and waits for everything in flight, so what is requested in front of it shares the point's memory latency.
// EXACT (the training forward): no angle doubling -- one range-reduced sin / cos per column (sincos_cw: ~1e-7 absolute, the accuracy
// class of the library call at a third of its instructions), because the gradients are compared with autograd of the reference
// network, whose ReLU pattern answers a 4-ulp change of the encoding with per-cent changes of single weight gradients.
template <bool EXACT, class Pre, class KA = H3KArgs>
__device__ __forceinline__ void h3a_encode10(_Float16* sXh, _Float16* sXl, const KA& a, const float (&x)[3], int tid, const Pre& pre) {
    const int r = tid & 127;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 7);
    const float fa = a.freqs[5 * q], fb = a.freqs[5 * q + 3];
    float v[30], sn[3], cs[3];
    pre.template slot<3>();
#pragma unroll
    for (int c = 0; c < 3; ++c) sincos_cw(fa * x[c], &sn[c], &cs[c]);
#pragma unroll
    for (int k = 0; k < 5; ++k) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { v[6 * k + c] = sn[c]; v[6 * k + 3 + c] = cs[c]; }
        if (k == 0) pre.template slot<4>();
        if (k == 3) pre.template slot<5>();
        if constexpr (EXACT) {
            if (k < 4) {
                const float fk = a.freqs[5 * q + k + 1];
#pragma unroll
                for (int c = 0; c < 3; ++c) sincos_cw(fk * x[c], &sn[c], &cs[c]);
            }
        } else if (k == 2) {
#pragma unroll
            for (int c = 0; c < 3; ++c) sincos_cw(fb * x[c], &sn[c], &cs[c]);
        } else if (k < 4) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float s2 = 2.f * sn[c] * cs[c], c2 = fmaf(-2.f * sn[c], sn[c], 1.f);
                sn[c] = s2; cs[c] = c2;
            }
        }
    }
    _Float16* rh = sXh + r * LDH;
    _Float16* rl = sXl + r * LDH;
    unsigned h, l;
    if (q == 0) {
        float w[32];
        w[0] = x[0]; w[1] = x[1]; w[2] = x[2];
#pragma unroll
        for (int i = 0; i < 29; ++i) w[3 + i] = v[i];
        h3a_store8(rh, rl, 0, w);
        pre.template slot<6>();
        h3a_store8(rh, rl, 8, w + 8);
        h3a_store8(rh, rl, 16, w + 16);
        pre.template slot<7>();
        h3a_store8(rh, rl, 24, w + 24);
        h3a_split2(v[29], 0.f, h, l);
        *reinterpret_cast<unsigned short*>(rh + 32) = (unsigned short)h;
        *reinterpret_cast<unsigned short*>(rl + 32) = (unsigned short)l;
    } else {
        h3a_split2(v[0], 0.f, h, l);
        *reinterpret_cast<unsigned short*>(rh + 33) = (unsigned short)h;
        *reinterpret_cast<unsigned short*>(rl + 33) = (unsigned short)l;
        h3a_split2(v[1], v[2], h, l);
        *reinterpret_cast<unsigned*>(rh + 34) = h;
        *reinterpret_cast<unsigned*>(rl + 34) = l;
        unsigned hb, lb;
        h3a_split2(v[3], v[4], h, l); h3a_split2(v[5], v[6], hb, lb);
        *reinterpret_cast<u2v*>(rh + 36) = u2v{h, hb};
        *reinterpret_cast<u2v*>(rl + 36) = u2v{l, lb};
        pre.template slot<6>();
        float w[24];
#pragma unroll
        for (int i = 0; i < 23; ++i) w[i] = v[7 + i];
        w[23] = 0.f;
        h3a_store8(rh, rl, 40, w);
        pre.template slot<7>();
        h3a_store8(rh, rl, 48, w + 8);
        h3a_store8(rh, rl, 56, w + 16);
    }
}

// SAVE: the TRAINING forward (nsff_field_kernel_h3a_save) -- the same trunk program through the SAVE build of the body
// (H3A_BODY_SAVE: every layer's activation goes to HBM in the weight-gradient GEMM's fragment order and its ReLU sign words in the
// backward kernel's accumulator order, riding in the phases), an exact sin / cos per embedding column (the gradients are compared
// with autograd of the reference network: see build_input's OCTAVE note), the time code through the matrix pipe (its columns'
// weight gradients need the saved input tile) and the encoded input tile saved in front of the trunk.
// The arguments are read through the kernel-argument segment pointer, made opaque once per tile and once behind the body: the body
// leaves the compiler 40 scalar and 24 (SAVE: 14) vector registers, and what it would keep of the arguments across the body -- or
// hoist out of the tile loop -- it would have to spill; re-read, they are scalar loads that hit the constant cache.
typedef const __attribute__((address_space(4))) H3AArgs H3AKernArgs;
__device__ __forceinline__ H3AKernArgs* h3a_args() {
    H3AKernArgs* p = (H3AKernArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}

template <bool SAVE>
__device__ __forceinline__ void h3a_kernel() {
    constexpr int M = 128, THREADS = 256;
    __shared__ __attribute__((aligned(16))) _Float16 sX[2 * M * LDH];
    __shared__ __attribute__((aligned(16))) float sRaw[M * NSFF_RAW_STRIDE];
    __shared__ __attribute__((aligned(16))) float sBias[(H3A_MAX_BIAS + 1) * NSFF_W];      // (+ one row: the heads' biases [0, 32), sigma's bias [32])
    for (int i = threadIdx.x; i < M * NSFF_RAW_STRIDE; i += THREADS) sRaw[i] = 0.f;
    _Float16* sXh = sX;
    _Float16* sXl = sX + M * LDH;
    long long tile = blockIdx.x, tile_stride = 0x7fffffffffffLL, tile_end = 0x7fffffffffffLL;
    int tr, piece = 0;
    SpanT span0;
  {
    H3AKernArgs& aa = *h3a_args();
    const auto& a = aa.k;
    tr = a.n_static_steps > 0 ? 0 : 1;
    if (aa.p_mode != 0) {
        tile_end = aa.p_tiles;
        tile_stride = gridDim.x;
        if (aa.p_mode == 2) {                       // trunk by XCD, tile j of the trunk's gridDim.x / 2 workgroups
            const int x = blockIdx.x & 7;
            tr = x >> 2; piece = tr + 1;
            tile = (long long)(blockIdx.x >> 3) * 4 + (x & 3);
            tile_stride = gridDim.x >> 1;
        } else if (aa.p_mode == 3) { tr = 1; piece = 2; }
        else if (aa.p_mode == 4) {                  // trunk by XCD + a second round of workgroups for the longer trunk's tail
            const int x = blockIdx.x & 7, round1 = (int)(gridDim.x >> 1);
            const bool second = (int)blockIdx.x >= round1;
            const int b = second ? (int)blockIdx.x - round1 : (int)blockIdx.x;
            tr = x >> 2;
            tile = (long long)(b >> 3) * 4 + (x & 3);
            tile_stride = round1 >> 1;
            if (!second) { if (tr == aa.p_long) tile_end = aa.p_split; }
            else if (tr == aa.p_long) tile = tile_end;          // (nothing to do on the longer trunk's own XCDs)
            else { tr = aa.p_long; tile += aa.p_split; }
            piece = tr + 1;
        }
    } else if (a.split_trunks) {
        if (tile < a.grid_tiles) { tr = 0; piece = 1; }
        else { tile -= a.grid_tiles; tr = 1; piece = 2; }
    }
    span0 = span_begin(a.span);
  }
    // The point of a tile is requested one tile ahead (a persistent workgroup: by the previous tile's body, into registers it
    // leaves to the compiler) -- the first one here.  (rows past the end: the last point)
    float px[3];
    auto request_point = [&px](const auto& a, long long p0, unsigned tid) {
        const long long last = a.n_points - 1;
        const long long bp = p0 + build_row<M, THREADS>(tid) < last ? p0 + build_row<M, THREADS>(tid) : last;
        px[0] = a.xyz[bp * 3 + 0]; px[1] = a.xyz[bp * 3 + 1]; px[2] = a.xyz[bp * 3 + 2];
    };
    request_point(h3a_args()->k, tile * M, threadIdx.x);
    unsigned tid_ = threadIdx.x;            // (the ONE copy of the thread index that crosses the body: an in / out operand of its statement)
    // What every tile's bias-row requests need, read ONCE per workgroup (the compiler parks these scalars in lanes of a register
    // the body leaves alone: a lane read instead of a dependent scalar load per tile): the bias table's entries, the per-ray
    // table's position relative to the packed buffer, its row pitch, the launch's last point, points per ray.
    // (dynamic trunk with the time code folded in: the rows of its input layers are per ray -- every 64-point half of the tile
    // lies inside one ray, the host checked pts_per_ray % 64 == 0 and n_points < 2^31 -- and the body never sees a time-code column;
    // the static trunk of a view-direction model has per-ray rows of its own: static_dir_encoding's [dir | a] part, nsff_side_bias)
    uint32_t boff[H3A_MAX_BIAS];
    uint32_t ray_rows = 0u;     // bit r: table row r is a per-ray row (the only rows a later tile of the workgroup reloads)
    int nb;
    unsigned has_rows;          // (an integer in a scalar register: a parked bool becomes a lane mask in a vector register)
    unsigned last_pt, ppr, row_bytes;
    long long rows_delta;
    {
        H3AKernArgs& aa = *h3a_args();
        const auto& a = aa.k;
        const float* rowtab = tr == 1 ? a.t_bias : a.s_bias;
        has_rows = (unsigned)__builtin_amdgcn_readfirstlane(rowtab != nullptr ? 1 : 0);
        row_bytes = (unsigned)(tr == 1 ? a.tb_rows : a.sb_rows) * (NSFF_W * 4);
        last_pt = (unsigned)(a.n_points - 1);
        ppr = (unsigned)a.pts_per_ray;
        rows_delta = reinterpret_cast<const char*>(rowtab) - reinterpret_cast<const char*>(a.packed);
        nb = aa.n_bias[tr];
#pragma unroll
        for (int r4 = 0; r4 < H3A_MAX_BIAS / 4; ++r4) {
            const u4v q = reinterpret_cast<const __attribute__((address_space(4))) u4v*>(&aa.bias_off[tr][0])[r4];
            boff[4 * r4] = q[0]; boff[4 * r4 + 1] = q[1]; boff[4 * r4 + 2] = q[2]; boff[4 * r4 + 3] = q[3];
        }
#pragma unroll
        for (int r = 0; r < H3A_MAX_BIAS; ++r) boff[r] = r < nb ? boff[r] : 0u;
#pragma unroll
        for (int r = 0; r < H3A_MAX_BIAS; ++r) ray_rows |= (boff[r] >> 31) << r;
    }
#pragma unroll 1
  for (int it = 0; tile < tile_end; ++it, tile += tile_stride) {
    H3AKernArgs& aa = *h3a_args();
    const auto& a = aa.k;
    const uint32_t* __restrict__ pk = a.packed;
    asm volatile("" : "+v"(tid_));          // (opaque per tile: nothing derived from the thread index is carried across the body)
    const int lane = tid_ & 63;
    const int wave_id = __builtin_amdgcn_readfirstlane(tid_ >> 6);
    const long long p0 = tile * M;
#ifdef H3_TIMING
#define H3A_TSTAMP(k) do { if (lane == 0) g_h3_timing[H3A_TBASE + ((blockIdx.x & 255) * 4 + wave_id) * 256 + (k)] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define H3A_TSTAMP(k) do {} while (0)
#endif
    H3A_TSTAMP(58);                         // (timing build: this tile's start; [59] its end -- the last tile's stamps survive)
    // Everything this workgroup reads before its trunk starts is REQUESTED first and waited for once: the point, the bias-table
    // rows (sixteen loads in flight), then -- the pre-issue statement -- weight slots 0..7: 128 KiB that cross the CU's vector-memory
    // path while the encoder below computes.  The compiler's wait for the point sits behind the statement (its first use).
    // bias table of this trunk (fp32 rows of 256): the body initialises its accumulators from it with ds_read_b128.
    const bool tb = tr == 1 && has_rows != 0u;
    float bv[H3A_MAX_BIAS];
    // (opaque copies: what is derived from the parked scalars is derived per tile, not parked as well)
    uint32_t rr = ray_rows;
    asm volatile("" : "+s"(rr));
    {
        // row r's bytes from the packed buffer's start (wave-uniform, branch-free): a plain row is a word offset, a per-ray row
        // lies rows_delta = t_bias - packed further on, at the ray of its half.  Rows past the table's end read the buffer's
        // first words (never stored).
        long long tb_at[2] = {0, 0};
        if (has_rows != 0u) {
            unsigned q0 = (unsigned)p0, lp = last_pt, pp = ppr, rbytes = row_bytes;
            long long rd = rows_delta;
            asm volatile("" : "+s"(lp), "+s"(pp), "+s"(rbytes), "+s"(rd));
            tb_at[0] = rd + (long long)((q0 < lp ? q0 : lp) / pp) * rbytes;
            tb_at[1] = rd + (long long)((q0 + 64u < lp ? q0 + 64u : lp) / pp) * rbytes;
        }
        if (it == 0) {
#pragma unroll
            for (int r = 0; r < H3A_MAX_BIAS; ++r) {
                uint32_t off = boff[r];
                asm volatile("" : "+s"(off));
                // (selects as mask arithmetic: the compiler turns the conditional form into two scalar branches per row)
                const long long per_ray = -(long long)(off >> 31), half_b = -(long long)((off >> 8) & 1u);
                const long long at_ray = tb_at[0] + (half_b & (tb_at[1] - tb_at[0])) + (long long)((off & 0xffu) * (NSFF_W * 4));
                const long long at = (at_ray & per_ray) | ((long long)off * 4 & ~per_ray);
                bv[r] = reinterpret_cast<const float*>(reinterpret_cast<const char*>(pk) + at)[tid_];
            }
        } else {
            // a later tile of a persistent launch: only the per-ray rows change (at most six: one scalar branch per table row)
#pragma unroll
            for (int r = 0; r < H3A_MAX_BIAS; ++r) {
                bv[r] = 0.f;
                if ((rr >> r) & 1u) {
                    uint32_t off = boff[r];
                    asm volatile("" : "+s"(off));
                    const long long at = tb_at[(off >> 8) & 1u] + (long long)((off & 0xffu) * (NSFF_W * 4));
                    bv[r] = reinterpret_cast<const float*>(reinterpret_cast<const char*>(pk) + at)[tid_];
                }
            }
        }
    }
    // (the heads' biases -- 32 floats -- travel with the table: the records loop below reads them from LDS)
    float hbias = 0.f, sig_b = 0.f;
    float act5[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    H3APre pre;
    pre.on = it == 0;
    if (it == 0) {      // (a later tile of a persistent workgroup finds them in LDS / its weight slots resident)
        hbias = reinterpret_cast<const float*>(pk)[aa.hsel[tr].b_off + (tid_ & 31)];
        // what the records loop applies to float 4 q + e of a record (thread tid = 4 q + e < 16 fills entry tid of the table in
        // the spare floats of the bias table's last row): the head row's bias, and its activation as
        // y = fma(1 / (1 + 2^(c x)), ya, yb)  -- sigmoid: c = -log2 e, (ya, yb) = (1, 0); flow: c = 2 log2 e, (-2 s, s) = s tanh(x);
        // none: x itself
        {
            const int row = (int)(tid_ & 15) - aa.hsel[tr].slot0;
            const bool on = row >= 0 && row < aa.hsel[tr].n_rows;
            const unsigned kind = on ? (aa.hsel[tr].kinds >> (2 * (row & 15))) & 3u : (unsigned)ACT_NONE;
            act5[0] = on ? reinterpret_cast<const float*>(pk)[aa.hsel[tr].b_off + (row & 31)] : 0.f;
            act5[1] = kind == ACT_SIGMOID ? -1.4426950408889634f : (kind == ACT_FLOW ? 2.8853900817779268f : 0.f);
            act5[2] = kind == ACT_FLOW ? -2.0f * a.flow_scale : 1.0f;
            act5[3] = kind == ACT_FLOW ? a.flow_scale : 0.f;
            act5[4] = (kind != ACT_SIGMOID && kind != ACT_FLOW) ? 1.f : 0.f;
        }
        sig_b = reinterpret_cast<const float*>(pk)[(tr == 0 && aa.sig_ride != 0) ? aa.sig_b_off : 0u];
        const auto& d0 = aa.ph[tr][0];
        pre.pk = (unsigned long long)(uintptr_t)pk;
        pre.n1 = d0.d[3];
        pre.r1 = d0.d[4] + (unsigned)wave_id * d0.d[5];
        pre.r2 = d0.d[6] + (unsigned)wave_id * d0.d[7];
        pre.lane16 = (unsigned)lane << 4;
    }
    H3A_TSTAMP(52);
    pre.slot<0>(); pre.slot<1>(); pre.slot<2>();        // (a later tile of a persistent workgroup: the previous tile's B16LP phase requested all eight)
    // the point is pinned as landed here (first tile: the one wait for the workgroup's own loads and the three slots behind them)
    asm volatile("" : "+v"(px[0]), "+v"(px[1]), "+v"(px[2]));
    H3A_TSTAMP(53);
    // (the lean encoder builds the position part; the dynamic trunk's time-code columns -- when they go through the matrix pipe:
    //  no folded rows, and every training forward -- are appended by build_time_part)
    const bool lean = a.octave_freqs && a.n_freqs == 10 && (SAVE || !(tr == 1 && !tb));
    if (lean) {
        h3a_encode10<SAVE>(sXh, sXl, a, px, tid_, pre);
        if (SAVE && tr == 1 && !tb) build_time_part<M, THREADS>(sXh, sXl, a, p0, tid_);
    } else {
        pre.slot<3>(); pre.slot<4>(); pre.slot<5>(); pre.slot<6>(); pre.slot<7>();
        build_input<M, THREADS, !SAVE, true>(sXh, sXl, a, p0, tr == 1 && !tb, px, tid_);
    }
    // the bias rows go to LDS behind the encoder (a later tile's per-ray rows were requested in front of it), then the NEXT tile's
    // point is requested: it lands under the body
    if (it == 0) {
#pragma unroll
        for (int r = 0; r < H3A_MAX_BIAS; ++r)
            if (r < nb) sBias[r * NSFF_W + tid_] = bv[r];
        if (tid_ < 32) sBias[H3A_MAX_BIAS * NSFF_W + tid_] = hbias;
        if (tid_ == 32) sBias[H3A_MAX_BIAS * NSFF_W + 32] = sig_b;
        if (tid_ < 16) {
#pragma unroll
            for (int k = 0; k < 5; ++k) sBias[H3A_MAX_BIAS * NSFF_W + 64 + 16 * k + tid_] = act5[k];
        }
    } else {
#pragma unroll
        for (int r = 0; r < H3A_MAX_BIAS; ++r)
            if ((rr >> r) & 1u) sBias[r * NSFF_W + tid_] = bv[r];
    }
    // (the body itself requests the NEXT tile's point into px -- registers the body leaves to the compiler -- from this address;
    //  the last tile requests its own point again)
    [[maybe_unused]] unsigned long long nxa;
    {
        const long long np0 = (tile + tile_stride < tile_end ? tile + tile_stride : tile) * M, last = a.n_points - 1;
        const long long bp = np0 + build_row<M, THREADS>(tid_) < last ? np0 + build_row<M, THREADS>(tid_) : last;
        nxa = (unsigned long long)(uintptr_t)(a.xyz + bp * 3);
    }
    H3A_TSTAMP(57);
    // rows of the time code this thread restores at a skip layer: point row (tid >> 2) of either half, columns [16 q, 16 q + 16)
    const float* tpa = reinterpret_cast<const float*>(pk);
    const float* tpb = tpa;
    if (tr == 1 && !tb) {
        const int row = tid_ >> 2, q = tid_ & 3;
        const long long last = a.n_points - 1;
        const long long pa = p0 + row < last ? p0 + row : last, pb = p0 + 64 + row < last ? p0 + 64 + row : last;
        if (16 * q < a.in_t) {
            tpa = a.t_emb + (pa / a.pts_per_ray) * a.in_t + 16 * q;
            tpb = a.t_emb + (pb / a.pts_per_ray) * a.in_t + 16 * q;
        }
    }
    __syncthreads();
    [[maybe_unused]] unsigned long long sv_act = 0ull, sv_mask = 0ull;
    [[maybe_unused]] unsigned sv_astride = 0u, sv_mstride = 0u;
    if constexpr (SAVE) {
        // the encoded trunk input (the dynamic trunk's tile carries the time code; a static trunk saves it only when it is the
        // launch's only trunk -- as the eight-wave training forward does), then the destinations of the body's rides: slot 0 of this
        // trunk at this workgroup's first 64-point tile (both of its tiles exist: the host checked that the tile count is even)
        const long long tile64 = tile * 2;
        if (a.save_xin != nullptr && (tr == 1 || a.transient_mode == 0))
            tile_to_fragments<THREADS, M>(sXh, sXl, a.save_xin + tile64 * (64 * a.xin_rows), a.xin_rows,
                                          (int)a.L.k0s + (tr == 1 ? (int)a.L.kt : 0), M / 16);
        const long long slot0 = tr == 0 ? 0 : a.D + 1;
        sv_act = (unsigned long long)(uintptr_t)(a.save_acts + (slot0 * a.n_tiles + tile64) * (64 * NSFF_W));
        sv_mask = (unsigned long long)(uintptr_t)(a.save_masks + (slot0 * a.n_tiles + tile64) * 256 + 64 * wave_id);
        sv_astride = (unsigned)(a.n_tiles * (64 * NSFF_W * 2));
        sv_mstride = (unsigned)(a.n_tiles * (256 * 8));
    }
    {
        const unsigned long long phases = (unsigned long long)(uintptr_t)__builtin_amdgcn_kernarg_segment_ptr() +
                                          offsetof(H3AArgs, ph) + (size_t)tr * sizeof(aa.ph[0]);
        const unsigned long long pkb = (unsigned long long)(uintptr_t)pk;
        const unsigned lds = (unsigned)(uintptr_t)sX, biaslds = (unsigned)(uintptr_t)sBias;
        const unsigned in_t = (tr == 1 && !tb) ? (unsigned)a.in_t : 0u;
        const unsigned tpa0 = (unsigned)((uintptr_t)tpa), tpa1 = (unsigned)((uintptr_t)tpa >> 32);
        const unsigned tpb0 = (unsigned)((uintptr_t)tpb), tpb1 = (unsigned)((uintptr_t)tpb >> 32);
#ifdef H3_TIMING
        // 256 dwords per (workgroup & 255, wave): [0] kernel entry, [1] input built, [52..] C++ stamps, [62] body left, [63] records
        // stored, [64 + 6 i ..] the body's records: dispatcher visit i and the five stamps of the phase before it
        unsigned* tdbg = g_h3_timing + H3A_TBASE + ((blockIdx.x & 255) * 4 + wave_id) * 256;
        if (lane == 0) { tdbg[0] = (unsigned)span0.t; tdbg[1] = (unsigned)__builtin_amdgcn_s_memtime(); }
        const unsigned long long dbg = (unsigned long long)(uintptr_t)(tdbg + 64);
#endif
        if constexpr (SAVE) {
            asm volatile(H3A_BODY_SAVE
                         : [tid] "+v"(tid_)
                         : [pk] "s"(pkb), [phases] "s"(phases), [lds] "s"(lds), [biaslds] "s"(biaslds), [rawlds] "s"((unsigned)(uintptr_t)sRaw),
                           [wave] "s"(wave_id), [in_t] "s"(in_t), [tpa0] "v"(tpa0), [tpa1] "v"(tpa1), [tpb0] "v"(tpb0), [tpb1] "v"(tpb1),
                           [act] "s"(sv_act), [mask] "s"(sv_mask), [astride] "s"(sv_astride), [mstride] "s"(sv_mstride)
                         : H3A_SAVE_CLOBBERS);
        } else {
        asm volatile(H3A_BODY
                     : [nx0] "=&v"(px[0]), [nx1] "=&v"(px[1]), [nx2] "=&v"(px[2]), [tid] "+v"(tid_)
                     : [nxa] "v"(nxa), [pk] "s"(pkb),
#ifdef H3_TIMING
                       [dbg] "s"(dbg),
#endif
                       [phases] "s"(phases), [lds] "s"(lds), [biaslds] "s"(biaslds), [rawlds] "s"((unsigned)(uintptr_t)sRaw),
                       [wave] "s"(wave_id), [in_t] "s"(in_t), [tpa0] "v"(tpa0), [tpa1] "v"(tpa1), [tpb0] "v"(tpb0), [tpb1] "v"(tpb1)
                     : H3A_CLOBBERS);
        }
    }
#ifdef H3_TIMING
    if ((tid_ & 63) == 0) g_h3_timing[H3A_TBASE + ((blockIdx.x & 255) * 4 + (tid_ >> 6)) * 256 + 62] = (unsigned)__builtin_amdgcn_s_memtime();
#endif
    // The body's HEAD phase left the heads' pre-activation sums in the raw-record image; bias and activation are applied where
    // the records leave: a thread always handles the same 16-byte quarter of a record (256 threads, 4 quarters per point).
    __syncthreads();
   {
    H3AKernArgs& aa = *h3a_args();
    const auto& a = aa.k;
    const unsigned tix = tid_;          // (opaque: an output of the body's statement)
    [[maybe_unused]] const int lane = tix & 63;
    [[maybe_unused]] const int wave_id = __builtin_amdgcn_readfirstlane(tix >> 6);
    const bool sig_ride = tr == 0 && aa.sig_ride != 0;
    const long long p0 = tile * M;
    H3A_TSTAMP(54);
    H3A_TSTAMP(55);
    H3A_TSTAMP(56);
    {
        // per thread, for its four record floats: bias and activation constants from the table the first tile left in LDS
        const int q4 = (int)tix & 3;
        const float4* act = reinterpret_cast<const float4*>(sBias + H3A_MAX_BIAS * NSFF_W + 64);
        const float4 hb4 = act[q4], hc4 = act[4 + q4], ya4 = act[8 + q4], yb4 = act[12 + q4], pl4 = act[16 + q4];
        const float hb[4] = {hb4.x, hb4.y, hb4.z, hb4.w}, hc[4] = {hc4.x, hc4.y, hc4.z, hc4.w};
        const float ya[4] = {ya4.x, ya4.y, ya4.z, ya4.w}, yb[4] = {yb4.x, yb4.y, yb4.z, yb4.w};
        const bool plain[4] = {pl4.x != 0.f, pl4.y != 0.f, pl4.z != 0.f, pl4.w != 0.f};
        for (int i = (int)tix; i < M * (NSFF_RAW_STRIDE / 4); i += THREADS) {
            const long long p = p0 + i / (NSFF_RAW_STRIDE / 4);
            if (p < a.n_points && (piece == 0 || (piece == 1) == (q4 == 0))) {
                float4 v = reinterpret_cast<const float4*>(sRaw)[i];
                float* ve = reinterpret_cast<float*>(&v);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = ve[e] + hb[e];
                    const float y = fmaf(__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(hc[e] * x)), ya[e], yb[e]);
                    ve[e] = plain[e] ? x : y;
                }
                if (sig_ride) {
                    // sigma of a view-direction static trunk: the 8 partial sums of the body's sigma ride (wave, lane half) in a
                    // fixed order + the bias; the floats that carried them leave as zeros (a static-only launch stores whole records)
                    if (q4 == 0) {
                        const float4 s1 = reinterpret_cast<const float4*>(sRaw)[i + 1], s2 = reinterpret_cast<const float4*>(sRaw)[i + 2];
                        v.w = ((((s1.x + s1.y) + (s1.z + s1.w)) + ((s2.x + s2.y) + (s2.z + s2.w)))) + sBias[H3A_MAX_BIAS * NSFF_W + 32];
                    } else if (q4 != 3) {
                        v = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
                reinterpret_cast<float4*>(a.raw)[p0 * (NSFF_RAW_STRIDE / 4) + i] = v;       // (non-temporal: measured +-0, round 6)
            }
        }
    }
    H3A_TSTAMP(59);
   }
    if constexpr (SAVE) break;      // (the training forward is never launched in the persistent form: no loop for the compiler to keep state around)
  }     // (tiles of this workgroup)
    span_end(h3a_args()->k.span, span0, tid_);
#ifdef H3_TIMING
    if ((tid_ & 63) == 0) g_h3_timing[H3A_TBASE + ((blockIdx.x & 255) * 4 + (tid_ >> 6)) * 256 + 63] = (unsigned)__builtin_amdgcn_s_memtime();
#endif
}

__global__ __launch_bounds__(256, 1) void nsff_field_kernel_h3a(const H3AArgs) { h3a_kernel<false>(); }
__global__ __launch_bounds__(256, 1) void nsff_field_kernel_h3a_save(const H3AArgs) { h3a_kernel<true>(); }

// Phase program of one trunk = steps [s0, s1) of the step program (see tools/h3asm/check.py::build_program, the reference
// implementation of this function, which the simulator runs).  Returns false when the trunk's structure is not one the body
// executes -- the caller then launches the compiler-scheduled kernel instead.
// fold_t (dynamic trunk, H3KArgs::t_bias given): the time-code part of every input segment is not executed -- its product is in
// the per-ray rows of the bias table (H3A_TB_ROW entries: one row per half for the layer's first segment) -- so an input
// segment runs its position part only (k0s / 16 of its k-steps; the waves' blocks of the packed segment keep their stride).
// side_fold (static trunk of a view-direction model, H3KArgs::s_bias given; the step program of h3_step_program(side_fold)):
// static_dir_encoding is the trunk's last 256-wide segment, its bias rows are per ray like the folded time code's, and the
// sigma head of the step before it (static_sigma reads the last TRUNK layer, nerf.py:169) is not a HEAD phase but the SIGMA RIDE
// of that layer's epilogues (tools/h3asm/gen.py: B16RS / A16RS); *sig_ride tells the kernel's records loop to add the partial sums up.
// save: the training forward's program (nsff_field_kernel_h3a_save): a SAVE_LAST phase in front of the heads copies the trunk's
// last activation to its slot (every other activation rides in the phase that multiplies it).
static bool h3a_build_program(const H3KArgs& k, int s0, int s1, bool dynamic, bool fold_t, H3APhase* ph, uint32_t* bias_off,
                              int& n_bias, int& head, int* n_phases = nullptr, bool side_fold = false, int* sig_ride = nullptr,
                              bool save = false) {
    struct Seg { uint32_t off, stride; int nks, bias, bias_b; bool relu, rebuild; };
    Seg segs[MAX_STEPS];
    int n = 0;
    n_bias = 0;
    head = HEAD_NONE;
    if (sig_ride) *sig_ride = 0;
    if (side_fold && (dynamic || fold_t || s1 - s0 < 3)) return false;
    for (int i = s0; i < s1; ++i) {
        const H3Step& st = k.steps[i];
        if (st.nks != 4 && st.nks != 8 && st.nks != 16) return false;
        if (st.post != POST_RELU && st.post != POST_NONE) return false;
        if (st.pre == PRE_SIDE) return false;         // (st.save only names an activation slot: this path is never a saving launch)
        if (st.head != HEAD_NONE) {
            if (side_fold && i == s1 - 2) {
                if (st.head != HEAD_S_SIGMA) return false;          // (the sigma ride, see below)
            } else {
                if (i != s1 - 1) return false;
                if (side_fold ? st.head != HEAD_S_RGB : (st.head != HEAD_S_FOLD && st.head != HEAD_T_FOLD && st.head != HEAD_S_SIGMA)) return false;
                head = st.head;
            }
        }
        Seg& g = segs[n++];
        g.off = st.w_off * 4u; g.nks = st.nks; g.relu = st.post == POST_RELU;
        g.stride = (uint32_t)st.nks * 4096u;
        g.rebuild = i > s0 && st.pre != PRE_NONE;
        g.bias = g.bias_b = -1;
        if (st.bias_off != NSFF_NONE) {
            if (n_bias >= H3A_MAX_BIAS) return false;
            g.bias = g.bias_b = n_bias;
            bias_off[n_bias++] = st.bias_off;
        }
        if (i == s0 && st.pre == PRE_NONE) return false;
    }
    if (n == 0 || head == HEAD_NONE) return false;
    if (segs[0].nks == 16 || !segs[0].relu || segs[0].bias < 0) return false;
    if (fold_t) {
        if (!dynamic) return false;
        int row = 0;
        for (int t = 0; t < n; ++t) {
            if (segs[t].nks == 16) continue;
            if (segs[t].nks * 16 != (int)(k.L.k0s + k.L.kt)) return false;
            segs[t].nks = (int)k.L.k0s / 16;
            Seg& first = segs[t == 0 ? 0 : t - 1];          // (a skip layer's 256-wide segment precedes its input segment and carries the bias)
            if (first.bias < 0 || first.bias != first.bias_b || n_bias >= H3A_MAX_BIAS) return false;
            bias_off[first.bias] = H3A_TB_ROW | (uint32_t)row;
            first.bias_b = n_bias;
            bias_off[n_bias++] = H3A_TB_ROW | H3A_TB_HALF_B | (uint32_t)row;
            ++row;
        }
        if (row != k.tb_rows) return false;
    }
    int sig_row = -1;
    if (side_fold) {
        // the last segment = static_dir_encoding on the last trunk layer: 256-wide, ReLU, rows of the per-ray table; the segment
        // in front of it = the last trunk layer, whose epilogues carry the sigma ride: a plain 256-wide ReLU layer
        Seg& dirl = segs[n - 1];
        const Seg& last = segs[n - 2];
        if (k.sb_rows != 1 || dirl.nks != 16 || !dirl.relu || dirl.bias < 0 || dirl.rebuild) return false;
        if (last.nks != 16 || !last.relu) return false;
        if (n_bias + 2 > H3A_MAX_BIAS) return false;
        bias_off[dirl.bias] = H3A_TB_ROW | 0u;
        dirl.bias_b = n_bias;
        bias_off[n_bias++] = H3A_TB_ROW | H3A_TB_HALF_B | 0u;
        sig_row = n_bias;
        bias_off[n_bias++] = k.L.s_sigma_f32;            // (a plain row: the sigma weights, read by the ride as a bias-table row)
    }
    int np = 0;
    auto put = [&](uint32_t body, uint32_t flags, int bias, int n1, const Seg& r1, const Seg& r2) {
        if (np >= H3A_MAX_PHASES) return false;
        H3APhase& p = ph[np++];
        p.d[0] = body; p.d[1] = flags; p.d[2] = 1024u * (uint32_t)(bias < 0 ? 0 : bias); p.d[3] = (uint32_t)n1;
        p.d[4] = r1.off; p.d[5] = r1.stride; p.d[6] = r2.off; p.d[7] = r2.stride;
        return true;
    };
    const uint32_t rb = H3A_F_REBUILD | ((dynamic && !fold_t) ? H3A_F_REBUILD_T : 0u);
    // descriptor 0 (the body's prologue): acc_A := row `bias`, acc_B := the row (flags >> 16) bytes behind it
    if (!put(H3A_BODY_END, H3A_F_INIT | ((1024u * (uint32_t)(segs[0].bias_b - segs[0].bias)) << 16), segs[0].bias, segs[0].nks, segs[0],
             n > 1 ? segs[1] : segs[0])) return false;
    bool pending_b = false;
    for (int t = 0; t < n; ++t) {
        const Seg& g = segs[t];
        const Seg* nxt = t + 1 < n ? &segs[t + 1] : nullptr;
        const Seg* nx2 = t + 2 < n ? &segs[t + 2] : nullptr;
        const uint32_t init_next = (nxt && nxt->bias >= 0) ? H3A_F_INIT : 0u;
        const int nbias = (nxt && nxt->bias >= 0) ? nxt->bias : 0;
        // weight-slot refills of this segment's B phase: slot j < n1 <- the next segment, j >= n1 <- the one after it
        const Seg& r1 = nxt ? *nxt : g;
        const Seg& r2 = (nxt && nxt->nks < 16 && nx2) ? *nx2 : r1;
        const int n1 = nxt ? nxt->nks : 16;
        bool ok = true;
        if (g.nks == 16) {
            if (t == 0 || !pending_b) return false;
            const bool sig_a = side_fold && t == n - 1, sig_b = side_fold && t == n - 2;    // (the ride on half B's / half A's epilogue)
            ok = ok && put(sig_a ? H3A_BODY_A16RS : H3A_BODY_A16R, g.bias >= 0 ? H3A_F_INIT : 0u, g.bias_b, 16, g, g);   // (an A phase's tail initialises acc_B)
            if (g.relu) {
                if (nxt && nxt->nks != 16) return false;      // (B16R requests from ONE stream: a 16-k-step segment follows)
                if (sig_b)      // ... and loads the sigma weights from table row (flags >> 16)
                    ok = ok && put(H3A_BODY_B16RS, init_next | ((1024u * (uint32_t)sig_row) << 16), nbias, n1, r1, r2);
                else
                    ok = ok && put(nxt ? H3A_BODY_B16R : H3A_BODY_B16L, init_next, nbias, n1, r1, r2);   // (the last segment requests nothing)
                pending_b = true;
            } else {
                if (!nxt || !nxt->rebuild || nxt->bias >= 0 || nxt->nks == 16) return false;
                ok = ok && put(H3A_BODY_B16X, rb, 0, n1, r1, r2);
                pending_b = false;
            }
        } else {
            if (pending_b || !g.relu) return false;
            if (t > 0 && (!g.rebuild || g.bias >= 0)) return false;
            if (nxt && nxt->nks != 16) return false;          // (its B phase requests from ONE stream)
            if (t == 0)     // slots 0..7 were requested in front of the encoder (H3A_PRE, descriptor 0), 8..15 ride in this A phase
                ok = ok && put(g.nks == 4 ? H3A_BODY_A4F : H3A_BODY_A8F, 0u, 0, g.nks, g, n > 1 ? segs[1] : g);
            else
                ok = ok && put(g.nks == 4 ? H3A_BODY_A4 : H3A_BODY_A8, rb, 0, 16, g, g);
            // the B phase carries the epilogue of half A and ends with the bias rows of the next segment in acc_A
            ok = ok && put(g.nks == 4 ? H3A_BODY_B4 : H3A_BODY_B8, init_next, nbias, n1, r1, r2);
            pending_b = true;
        }
        if (!ok) return false;
    }
    if (!pending_b) return false;
    // the last epilogue requests the head tile (stream r1, no wave stride; n1 = its rows), the HEAD phase leaves the rows'
    // pre-activation sums at float slot0 of the raw records (bias field = 4 slot0)
    const H3AHeadSel hd = h3a_head_sel(k.L, head);
    Seg tile = segs[0];
    tile.off = hd.w_off * 4u; tile.stride = 0u;
    if (save && fold_t) return false;                     // (the training forward keeps the time code on the matrix pipe: its columns' weight gradients need the saved tile)
    bool done = put(H3A_BODY_EPI_B, 0u, 0, hd.n_rows, tile, tile) && (!save || put(H3A_BODY_SAVE_LAST, 0u, 0, hd.n_rows, tile, tile)) &&
                put(H3A_BODY_HEAD, 0u, 0, hd.n_rows, tile, tile);
    if (done) ph[np - 1].d[2] = 4u * (uint32_t)hd.slot0;
    done = done && put(H3A_BODY_END, 0u, 0, 16, segs[0], segs[0]) && put(H3A_BODY_END, 0u, 0, 16, segs[0], segs[0]);
    if (n_phases) *n_phases = np;
    if (sig_ride) *sig_ride = side_fold ? 1 : 0;
    return done;
}

// ---------------------------------------------------------------------------------
struct PackSegH3 {
    const float* src;
    uint32_t dst;       // word offset
    int32_t kind;       // 0 flat fp32 copy, 1 tiled hi/lo segment, 2 head rows
    int32_t ld;
    int32_t kpad;       // tiled: padded K; flat: element count; head: K (=256)
    int32_t n0, s0, p1, n1, s1;   // column map (tiled)
    int32_t row0, nrows;          // head: destination row range
};
constexpr int PACK_BATCH = 56;          // one launch per model (3.1 KB of kernel arguments)
struct PackArgsH3 { PackSegH3 seg[PACK_BATCH]; uint32_t* dst; };

__device__ __forceinline__ float seg_value(const PackSegH3& s, int n, int c) {
    if (c < s.n0) return s.src[(long long)n * s.ld + s.s0 + c];
    if (c >= s.p1 && c < s.p1 + s.n1) return s.src[(long long)n * s.ld + s.s1 + (c - s.p1)];
    return 0.f;
}

__global__ void nsff_pack_kernel_h3(const PackArgsH3 a) {
    const PackSegH3& s = a.seg[blockIdx.y];
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (s.kind == 0) {
        if (idx < s.kpad) reinterpret_cast<float*>(a.dst + s.dst)[idx] = s.src[idx];
        return;
    }
    const int nks = s.kpad / 16;
    int lane, part, n, ks;
    if (s.kind == 1) {
        if (idx >= 4 * nks * 4 * 64) return;          // chunks: [wave][ks][mt][part][lane]
        lane = idx & 63; part = (idx >> 6) & 1;
        const int mt = (idx >> 7) & 1;
        const int wk = idx >> 8;
        ks = wk % nks;
        n = 64 * (wk / nks) + 32 * mt + (lane & 31);
    } else {
        if (idx >= nks * 2 * 64) return;              // chunks: [ks][part][lane]
        lane = idx & 63; part = (idx >> 6) & 1; ks = idx >> 7;
        n = (lane & 31) - s.row0;
        if (n < 0 || n >= s.nrows) return;            // other rows stay zero (buffer is memset first)
    }
    h8 out;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int c = ks * 16 + 8 * (lane >> 5) + t;
        const float x = s.kind == 1 ? seg_value(s, n, c) : s.src[(long long)n * s.ld + c];
        const _Float16 hi = (_Float16)x;
        out[t] = part == 0 ? hi : (_Float16)(x - (float)hi);
    }
    reinterpret_cast<h8*>(a.dst + s.dst)[idx] = out;
}

// Folded head rows (see NsffLayoutH3): out_w[r][i] = sum_o W_head[r][o] * W_final[o][i],  out_b[r] = sum_o W_head[r][o] *
// b_final[o] + b_head[r]  for the rows of the heads that read *_final; `plain` rows (static sigma) are copied.  One
// workgroup per row, one thread per input column, double accumulation (once per weight update: the cost is nothing).
struct FoldArgs {
    const float* w_head[4]; const float* b_head[4];     // folded heads: (nrows, ld_head) weights (first 256 columns), (nrows) biases
    int row0[4], nrows[4], n_heads, ld_head;
    const float* w_final; const float* b_final;         // (256, 256), (256)
    const float* w_plain; const float* b_plain; int plain_row;   // a (1, 256) head that reads h itself, or plain_row < 0
    float* out_w; float* out_b;                         // (32, 256) fp32 scratch, 32 fp32 biases
};
__global__ __launch_bounds__(256) void nsff_fold_kernel_h3(const FoldArgs a) {
    const int r = blockIdx.x, i = threadIdx.x;
    float w = 0.f, b = 0.f;
    if (r == a.plain_row) {
        w = a.w_plain[i]; b = a.b_plain[0];
    } else {
        for (int j = 0; j < a.n_heads; ++j) {
            if (r < a.row0[j] || r >= a.row0[j] + a.nrows[j]) continue;
            const float* wh = a.w_head[j] + (long long)(r - a.row0[j]) * a.ld_head;
            double sw = 0.0, sb = 0.0;
            for (int o = 0; o < NSFF_W; ++o) {
                sw += (double)wh[o] * (double)a.w_final[(long long)o * NSFF_W + i];
                sb += (double)wh[o] * (double)a.b_final[o];
            }
            w = (float)sw; b = (float)(sb + (double)a.b_head[j][r - a.row0[j]]);
        }
    }
    a.out_w[(long long)r * NSFF_W + i] = w;
    if (i == 0) a.out_b[r] = b;
}

}  // namespace

#ifdef H3_TIMING
extern "C" int nsff_debug_read_timing(unsigned* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_h3_timing), sizeof(unsigned) * n) == hipSuccess ? 0 : -4;
}
extern "C" int nsff_debug_span(unsigned long long* host, int reset) {       // host[2] = {first, last} tick of the launches since reset
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_h3_span), 16) != hipSuccess) return -4;
    if (reset) { const unsigned long long init[2] = {~0ull, 0ull}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_h3_span), init, 16) != hipSuccess) return -4; }
    return 0;
}
#endif

int nsff_h3_packed_bytes(const NsffModelDesc* desc, size_t* bytes) {
    NsffLayoutH3 L;
    const int rc = nsff_make_layout_h3(*desc, L);
    if (rc) return rc;
    *bytes = (size_t)L.total * 4;
    return NSFF_OK;
}

int nsff_h3_pack_weights(const NsffModelDesc* desc, const float* const* params, void* packed, bool fold, hipStream_t st) {
    NsffLayoutH3 L;
    const int rc = nsff_make_layout_h3(*desc, L);
    if (rc) return rc;
    const NsffModelDesc& d = *desc;
    std::vector<PackSegH3> segs;
    int pi = 0;
    auto tiled = [&](const float* src, uint32_t dst, int ld, int kpad, int n0, int s0, int p1, int n1, int s1) {
        segs.push_back(PackSegH3{src, dst, 1, ld, kpad, n0, s0, p1, n1, s1, 0, 0});
    };
    auto flat = [&](const float* src, uint32_t dst, int count) {
        segs.push_back(PackSegH3{src, dst, 0, 0, count, 0, 0, 0, 0, 0, 0, 0});
    };
    auto head = [&](const float* src, uint32_t dst, int row0, int nrows) {
        segs.push_back(PackSegH3{src, dst, 2, NSFF_W, NSFF_W, 0, 0, 0, 0, 0, row0, nrows});
    };
    auto trunk = [&](const NsffTrunkLayoutH3& T, int in_t) {
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
    { const float* w = params[pi++]; const float* b = params[pi++]; head(w, L.s_sigma_w, 0, 1); flat(b, L.s_sigma_b, 1);
      flat(w, L.s_sigma_f32, NSFF_W); }
    { const float* w = params[pi++]; const float* b = params[pi++]; head(w, L.s_rgb_w, 0, 3); flat(b, L.s_rgb_b, 3); }
    if (d.has_transient) {
        trunk(L.tr, d.in_t);
        const float* ws = params[pi++]; const float* bs = params[pi++];
        const float* wc = params[pi++]; const float* bc = params[pi++];
        head(wc, L.t_head_w, 0, 3); flat(bc, L.t_head_b, 3);
        head(ws, L.t_head_w, 3, 1); flat(bs, L.t_head_b + 3, 1);
        if (d.has_flow) {
            const float* wf = params[pi++]; const float* bf = params[pi++];
            const float* wb = params[pi++]; const float* bb = params[pi++];
            head(wf, L.t_head_w, 4, 3); flat(bf, L.t_head_b + 4, 3);
            head(wb, L.t_head_w, 7, 3); flat(bb, L.t_head_b + 7, 3);
        }
    }
    for (int i = 0; i < pi; ++i) if (!params[i]) return NSFF_ERR_NULL;
    hipError_t e = hipMemsetAsync(packed, 0, (size_t)L.total * 4, st);
    if (e != hipSuccess) return nsff_hip_fail(e);
    for (size_t base = 0; base < segs.size(); base += PACK_BATCH) {
        PackArgsH3 pa{};
        pa.dst = reinterpret_cast<uint32_t*>(packed);
        const int n = (int)std::min<size_t>(PACK_BATCH, segs.size() - base);
        int max_threads = 0;
        for (int i = 0; i < n; ++i) {
            pa.seg[i] = segs[base + i];
            const PackSegH3& s = pa.seg[i];
            const int thr = s.kind == 0 ? s.kpad : (s.kind == 1 ? 64 * s.kpad : 8 * s.kpad);
            max_threads = std::max(max_threads, thr);
        }
        hipLaunchKernelGGL(nsff_pack_kernel_h3, dim3((max_threads + 255) / 256, n), dim3(256), 0, st, pa);
    }
    return fold ? nsff_h3_fold_heads(desc, params, packed, st) : nsff_launch_status();
}

// Folded head rows (NsffLayoutH3): products in fp32 scratch inside the packed buffer, then one head tile per trunk.
// Separate from the pack so that a caller that re-packs after every optimizer step (training forwards execute the *_final
// layers: the backward pass needs their output) does not pay for rows nobody reads.
// the head sets of a model as FoldArgs (parameter order of nsff_pack_weights): fs static (rgb folded, sigma plain), ft dynamic
// (all rows folded), fd static_dir_encoding (use_viewdir); returns the number of parameters read
static int fold_args(const NsffModelDesc& d, const float* const* params, FoldArgs& fs, FoldArgs& ft, FoldArgs& fd) {
    // static trunk [w, b] x D, static final, [dir], static sigma, static rgb, dynamic trunk [w, b] x D, dynamic final,
    // dynamic sigma, dynamic rgb, [flow fw, flow bw]
    int pi = 2 * d.D;
    fs = FoldArgs{}; ft = FoldArgs{}; fd = FoldArgs{};
    fs.ld_head = ft.ld_head = NSFF_W;
    fs.w_final = params[pi]; fs.b_final = params[pi + 1]; pi += 2;
    if (d.use_viewdir) {                                   // static_dir_encoding: 256 rows, its first 256 columns read *_final
        fd.w_final = fs.w_final; fd.b_final = fs.b_final; fd.plain_row = -1;
        fd.w_head[0] = params[pi]; fd.b_head[0] = params[pi + 1]; fd.row0[0] = 0; fd.nrows[0] = NSFF_W; fd.n_heads = 1;
        fd.ld_head = NSFF_W + d.in_dir + d.in_a;
        pi += 2;
    }
    fs.w_plain = params[pi]; fs.b_plain = params[pi + 1]; fs.plain_row = 3; pi += 2;
    fs.w_head[0] = params[pi]; fs.b_head[0] = params[pi + 1]; fs.row0[0] = 0; fs.nrows[0] = 3; fs.n_heads = 1; pi += 2;
    if (d.has_transient) {
        pi += 2 * d.D;
        ft.w_final = params[pi]; ft.b_final = params[pi + 1]; ft.plain_row = -1; pi += 2;
        ft.w_head[1] = params[pi]; ft.b_head[1] = params[pi + 1]; ft.row0[1] = 3; ft.nrows[1] = 1; pi += 2;    // sigma
        ft.w_head[0] = params[pi]; ft.b_head[0] = params[pi + 1]; ft.row0[0] = 0; ft.nrows[0] = 3; pi += 2;    // rgb
        ft.n_heads = 2;
        if (d.has_flow) {
            ft.w_head[2] = params[pi]; ft.b_head[2] = params[pi + 1]; ft.row0[2] = 4; ft.nrows[2] = 3; pi += 2;
            ft.w_head[3] = params[pi]; ft.b_head[3] = params[pi + 1]; ft.row0[3] = 7; ft.nrows[3] = 3; pi += 2;
            ft.n_heads = 4;
        }
    }
    return pi;
}

// fp32 rows for the exact-fp32 kernel's VALU heads: s_w (4, 256) / s_b (4) unless use_viewdir, t_w (rows, 256) / t_b (rows)
int nsff_fold_rows_f32(const NsffModelDesc* desc, const float* const* params, float* s_w, float* s_b, float* t_w, float* t_b,
                       hipStream_t st) {
    const NsffModelDesc& d = *desc;
    FoldArgs fs, ft, fd;
    const int pi = fold_args(d, params, fs, ft, fd);
    for (int i = 0; i < pi; ++i) if (!params[i]) return NSFF_ERR_NULL;
    if (!d.use_viewdir) {
        fs.out_w = s_w; fs.out_b = s_b;
        hipLaunchKernelGGL(nsff_fold_kernel_h3, dim3(4), dim3(256), 0, st, fs);
    }
    if (d.has_transient) {
        ft.out_w = t_w; ft.out_b = t_b;
        hipLaunchKernelGGL(nsff_fold_kernel_h3, dim3(d.has_flow ? 10 : 4), dim3(256), 0, st, ft);
    }
    return nsff_launch_status();
}

int nsff_h3_fold_heads(const NsffModelDesc* desc, const float* const* params, void* packed, hipStream_t st) {
    NsffLayoutH3 L;
    const int rc = nsff_make_layout_h3(*desc, L);
    if (rc) return rc;
    const NsffModelDesc& d = *desc;
    FoldArgs fs, ft, fd;
    const int pi = fold_args(d, params, fs, ft, fd);
    for (int i = 0; i < pi; ++i) if (!params[i]) return NSFF_ERR_NULL;
    uint32_t* pw = reinterpret_cast<uint32_t*>(packed);
    PackArgsH3 pf{};
    pf.dst = pw;
    int nf = 0;
    auto fold = [&](FoldArgs& f, uint32_t scratch, uint32_t tile, uint32_t bias) {
        f.out_w = reinterpret_cast<float*>(pw + scratch); f.out_b = reinterpret_cast<float*>(pw + bias);
        hipLaunchKernelGGL(nsff_fold_kernel_h3, dim3(32), dim3(256), 0, st, f);
        pf.seg[nf++] = PackSegH3{f.out_w, tile, 2, NSFF_W, NSFF_W, 0, 0, 0, 0, 0, 0, 32};
    };
    if (!d.use_viewdir) fold(fs, L.fold_f32, L.s_fold_w, L.s_fold_b);
    if (d.has_transient) fold(ft, L.fold_f32 + 32 * NSFF_W, L.t_fold_w, L.t_fold_b);
    int max_threads = 8 * NSFF_W;
    if (d.use_viewdir) {                                   // a whole 256 x 256 segment, tiled like any trunk layer
        fd.out_w = reinterpret_cast<float*>(pw + L.dir_fold_f32); fd.out_b = reinterpret_cast<float*>(pw + L.dir_b_fold);
        hipLaunchKernelGGL(nsff_fold_kernel_h3, dim3(NSFF_W), dim3(256), 0, st, fd);
        pf.seg[nf++] = PackSegH3{fd.out_w, L.dir_h_fold, 1, NSFF_W, NSFF_W, NSFF_W, 0, 0, 0, 0, 0, 0};
        max_threads = 64 * NSFF_W;
    }
    if (nf > 0) hipLaunchKernelGGL(nsff_pack_kernel_h3, dim3((max_threads + 255) / 256, nf), dim3(256), 0, st, pf);
    return nsff_launch_status();
}

// Step program of a launch (reference nerf.py:162-208): fills k.steps / k.n_steps / k.n_static_steps from the layout k.L.
// fold: an inference launch (nothing saved for a backward pass).
// side_fold: a view-direction static trunk whose [dir | a] columns arrive as per-ray bias rows (H3KArgs::s_bias): the program
// the hand-scheduled kernel runs -- static_dir_encoding as ONE folded 256-wide ReLU segment, no side-input tile.
static int h3_step_program(const NsffModelDesc& d, int static_mode, int transient_mode, bool fold, H3KArgs& k, bool side_fold = false) {
    int n = 0;
    auto push = [&](uint32_t w, uint32_t b, uint32_t kcols, int pre, int post, int head, int slot = -1) {
        H3Step& s = k.steps[n++];
        s.w_off = w; s.bias_off = b; s.nks = (uint16_t)(kcols / 16);
        s.pre = (uint8_t)pre; s.post = (uint8_t)post; s.head = (uint8_t)head;
        s.save = (uint8_t)(slot + 1);
    };
    // activation slots of the training forward: trunk layer l -> slot0 + l; slot0 + D was *_final's (folded into the heads since round
    // 5: never written) -- the static trunk's slot D now is static_dir_encoding's (round 6: consecutive with the trunk's, so that the
    // SAVE build's running slot pointers reach it without a jump; it was 2 D + 2)
    auto trunk = [&](const NsffTrunkLayoutH3& T, int pre_kind, int last_head, int slot0) {
        for (int l = 0; l < d.D; ++l) {
            const int head = (l == d.D - 1) ? last_head : HEAD_NONE;
            if (l == 0) {
                push(T.seg_x[0], T.bias[0], T.k0, pre_kind, POST_RELU, HEAD_NONE, slot0);
            } else if ((nsff_skip_layers(&d) >> l) & 1u) {
                push(T.seg_h[l], T.bias[l], NSFF_W, PRE_NONE, POST_NONE, HEAD_NONE);
                push(T.seg_x[l], NSFF_NONE, T.k0, pre_kind, POST_RELU, head, slot0 + l);
            } else {
                push(T.seg_h[l], T.bias[l], NSFF_W, PRE_NONE, POST_RELU, head, slot0 + l);
            }
        }
    };
    // Inference launches (nothing saved for a backward pass) never execute the activation-free *_xyz_encoding_final
    // layers: the heads that read them are evaluated on the last trunk activation with pre-multiplied rows (NsffLayoutH3).
    if (static_mode == 2 && fold && !d.use_viewdir) {
        trunk(k.L.st, PRE_INPUT, HEAD_S_FOLD, 0);
    } else if (static_mode == 2 && fold && side_fold) {
        trunk(k.L.st, PRE_INPUT, HEAD_S_SIGMA, 0);
        push(k.L.dir_h_fold, k.L.dir_b_fold, NSFF_W, PRE_NONE, POST_RELU, HEAD_S_RGB, d.D);
    } else if (static_mode == 2 && fold) {               // view directions: *_final folded into static_dir_encoding
        trunk(k.L.st, PRE_INPUT, HEAD_S_SIGMA, 0);
        push(k.L.dir_h_fold, k.L.dir_b_fold, NSFF_W, PRE_NONE, POST_NONE, HEAD_NONE);
        push(k.L.dir_x, NSFF_NONE, k.L.side_k, PRE_SIDE, POST_RELU, HEAD_S_RGB, d.D);
    } else if (static_mode) {
        trunk(k.L.st, PRE_INPUT, HEAD_S_SIGMA, 0);
        if (static_mode == 2) {
            push(k.L.st.final_w, k.L.st.final_b, NSFF_W, PRE_NONE, POST_LINEAR, d.use_viewdir ? HEAD_NONE : HEAD_S_RGB, d.D);
            if (d.use_viewdir) {
                push(k.L.dir_h, k.L.dir_b, NSFF_W, PRE_NONE, POST_NONE, HEAD_NONE);
                push(k.L.dir_x, NSFF_NONE, k.L.side_k, PRE_SIDE, POST_RELU, HEAD_S_RGB, d.D);
            }
        }
    }
    k.n_static_steps = n;
    if (transient_mode && fold) {
        trunk(k.L.tr, PRE_INPUT_T, HEAD_T_FOLD, d.D + 1);
    } else if (transient_mode) {
        trunk(k.L.tr, PRE_INPUT_T, HEAD_NONE, d.D + 1);
        push(k.L.tr.final_w, k.L.tr.final_b, NSFF_W, PRE_NONE, POST_LINEAR, HEAD_T, 2 * d.D + 1);
    }
    if (n > MAX_STEPS) return NSFF_ERR_INVALID;
    k.n_steps = n;
    return NSFF_OK;
}

// Host-only export of what a launch would execute (no GPU work): the step program and, when the hand-scheduled body covers the
// launch's trunks, its phase programs -- tests pin h3a_build_program to the builder the simulator runs (tools/h3asm/check.py).
// steps: [n][4] = {w_off (words), bias_off (words, NSFF_NONE = accumulate), nks | pre << 8 | post << 16 | head << 24, 0};
// phases_static / phases_dynamic: [H3A_MAX_PHASES][8] descriptors; n_phases[2] = descriptors written (0 = trunk absent or
// not covered).
// The phase program of a PERSISTENT workgroup (tools/h3asm/check.py::make_persistent is the simulated reference): the last
// segment's B phase B16L becomes B16LP with descriptor 0's stream fields -- behind its k-steps 1..8 it requests weight slots 0..7
// of the trunk's first segments, which the workgroup's next tile finds resident (as a first tile finds what the pre-issue
// statements requested).  false: the trunk does not end with a 256-wide segment (a skip layer last) -- one workgroup per tile then.
static bool h3a_make_persistent(H3APhase* ph) {
    int at = -1;
    for (int i = 1; i < H3A_MAX_PHASES && !(i > 1 && ph[i].d[0] == H3A_BODY_END); ++i)
        if (ph[i].d[0] == H3A_BODY_B16L) { if (at >= 0) return false; at = i; }
    if (at < 0) return false;
    ph[at].d[0] = H3A_BODY_B16LP;
    for (int j = 3; j < 8; ++j) ph[at].d[j] = ph[0].d[j];
    return true;
}

extern "C" int nsff_field_phase_program(const NsffModelDesc* desc, int static_mode, int transient_mode, int fold_t, uint32_t* steps,
                                int* n_steps, int* n_static_steps, uint32_t* phases_static, uint32_t* phases_dynamic, int* n_phases) {
    if (!desc || !steps || !n_steps || !n_static_steps || !phases_static || !phases_dynamic || !n_phases) return NSFF_ERR_NULL;
    H3KArgs k{};
    int rc = nsff_make_layout_h3(*desc, k.L);
    if (rc) return rc;
    // fold_t bit 1: the launch is given NsffFieldArgs::s_bias (a view-direction static trunk with per-ray [dir | a] rows)
    const bool side_fold = (fold_t & 2) != 0 && static_mode == 2 && desc->use_viewdir;
    const bool persist = (fold_t & 4) != 0;       // bit 2: the programs of a persistent launch
    fold_t &= 1;
    rc = h3_step_program(*desc, static_mode, transient_mode, true, k, side_fold);
    if (rc) return rc;
    for (int i = 0; i < k.n_steps; ++i) {
        const H3Step& st = k.steps[i];
        steps[4 * i + 0] = st.w_off; steps[4 * i + 1] = st.bias_off;
        steps[4 * i + 2] = (uint32_t)st.nks | ((uint32_t)st.pre << 8) | ((uint32_t)st.post << 16) | ((uint32_t)st.head << 24);
        steps[4 * i + 3] = 0;
    }
    *n_steps = k.n_steps; *n_static_steps = k.n_static_steps;
    n_phases[0] = n_phases[1] = 0;
    H3APhase ph[H3A_MAX_PHASES];
    uint32_t boff[H3A_MAX_BIAS];
    int nb = 0, head = 0;
    if (k.n_static_steps > 0 && (side_fold || !(static_mode == 2 && desc->use_viewdir))) {
        for (auto& p : ph) for (auto& x : p.d) x = 0;
        int np = 0;
        k.sb_rows = side_fold ? 1 : 0;
        if (h3a_build_program(k, 0, k.n_static_steps, false, false, ph, boff, nb, head, &np, side_fold) && (!persist || h3a_make_persistent(ph))) {
            n_phases[0] = np;
            for (int i = 0; i < n_phases[0]; ++i) for (int j = 0; j < 8; ++j) phases_static[8 * i + j] = ph[i].d[j];
        }
    }
    if (k.n_steps > k.n_static_steps) {
        for (auto& p : ph) for (auto& x : p.d) x = 0;
        int np = 0;
        k.tb_rows = fold_t ? nsff_time_bias_rows(desc) : 0;
        if (h3a_build_program(k, k.n_static_steps, k.n_steps, true, fold_t != 0, ph, boff, nb, head, &np) && (!persist || h3a_make_persistent(ph))) {
            n_phases[1] = np;
            for (int i = 0; i < n_phases[1]; ++i) for (int j = 0; j < 8; ++j) phases_dynamic[8 * i + j] = ph[i].d[j];
        }
    }
    return NSFF_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// nsff_time_bias: per-ray rows  b_l + W_l[:, in_xyz : in_xyz + in_t] t  of the dynamic trunk's input layers (see the header).
// One thread per neuron, sixteen rays per workgroup; weights and time codes are staged in LDS.  ~75 MFLOP per C2 call -- its cost is the launch.
namespace {
struct TimeBiasJobK { const float* w[NSFF_MAX_LAYERS]; const float* b[NSFF_MAX_LAYERS]; int ld[NSFF_MAX_LAYERS];
                      const float* t_rows; float* out; int rows, in_xyz, in_t;
                      const float* table; const long long* ts; long long n_table, max_t; int delta; float* rows_out; };
struct TimeBiasArgs { TimeBiasJobK job[NSFF_MAX_TIME_BIAS_JOBS]; long long n_rays; };
constexpr int TB_RAYS = 16;
__global__ __launch_bounds__(256) void nsff_time_bias_kernel(const TimeBiasArgs a) {
    const TimeBiasJobK& j = a.job[blockIdx.z];
    const int i = blockIdx.y, n = threadIdx.x;
    if (i >= j.rows) return;
    const long long r0 = (long long)blockIdx.x * TB_RAYS;
    __shared__ __attribute__((aligned(16))) float st[TB_RAYS][64];     // the workgroup's time codes, zero-padded (in_t <= 64: the layout check)
    // the layer's time-code columns go through LDS sixteen at a time: consecutive lanes read consecutive floats of a row (a row's
    // slice starts at an odd offset of a (256, ld) matrix: one lane per row would touch 64 cache lines per load), the tile is
    // stored column-major with a one-float pad (conflict-free both ways).  Every global load of the workgroup is requested up
    // front -- the kernel is one memory latency long, not one per stage.
    __shared__ float sw[16][NSFF_W + 1];
    const float* __restrict__ wl = j.w[i] + j.in_xyz;
    const int ld = j.ld[i], in_t = j.in_t;
    float tv[TB_RAYS / 4], wv[4][16];
#pragma unroll
    for (int k = 0; k < TB_RAYS / 4; ++k) {
        const int e = threadIdx.x + 256 * k, r = e >> 6, c = e & 63;
        const long long ray = r0 + r < a.n_rays ? r0 + r : a.n_rays - 1;
        const float* row = j.t_rows != nullptr ? j.t_rows + ray * in_t : nullptr;
        if (row == nullptr) {                        // index mode: the row of the (clamped) neighbouring frame, rendering.py:153,218,224
            long long t = j.ts[ray] + j.delta;
            if (j.delta > 0) t = t < j.max_t ? t : j.max_t;
            if (j.delta < 0) t = t > 0 ? t : 0;
            t = t < 0 ? 0 : (t > j.n_table - 1 ? j.n_table - 1 : t);
            row = j.table + t * in_t;
        }
        tv[k] = c < in_t ? row[c] : 0.f;
        if (j.rows_out != nullptr && i == 0 && c < in_t && r0 + r < a.n_rays) j.rows_out[(r0 + r) * in_t + c] = tv[k];
    }
    const float b = j.b[i][n];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int e = threadIdx.x + 256 * k, row = e >> 4, c = 16 * q + (e & 15);
            wv[q][k] = c < in_t ? wl[row * ld + c] : 0.f;
        }
#pragma unroll
    for (int k = 0; k < TB_RAYS / 4; ++k) { const int e = threadIdx.x + 256 * k; st[e >> 6][e & 63] = tv[k]; }
    float acc[TB_RAYS];
#pragma unroll
    for (int r = 0; r < TB_RAYS; ++r) acc[r] = b;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (16 * q >= in_t) break;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) { const int e = threadIdx.x + 256 * k; sw[e & 15][e >> 4] = wv[q][k]; }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 16; u += 4) {               // (columns in ascending order: one fixed summation order per output)
            const float w0 = sw[u][n], w1 = sw[u + 1][n], w2 = sw[u + 2][n], w3 = sw[u + 3][n];
#pragma unroll
            for (int r = 0; r < TB_RAYS; ++r) {         // (the time codes are wave-uniform: one 16-byte broadcast read per four columns)
                const float4 t4 = *reinterpret_cast<const float4*>(&st[r][16 * q + u]);
                acc[r] = fmaf(w3, t4.w, fmaf(w2, t4.z, fmaf(w1, t4.y, fmaf(w0, t4.x, acc[r]))));
            }
        }
    }
#pragma unroll
    for (int r = 0; r < TB_RAYS; ++r)
        if (r0 + r < a.n_rays) j.out[((r0 + r) * j.rows + i) * NSFF_W + n] = acc[r];
}
}  // namespace

extern "C" int nsff_time_bias_rows(const NsffModelDesc* desc) {
    if (!desc || !desc->has_transient || desc->in_t < 1) return 0;
    return 1 + __builtin_popcount(nsff_skip_layers(desc));
}

extern "C" int nsff_time_bias(const NsffTimeBiasJob* jobs, int32_t n_jobs, int64_t n_rays, void* stream) {
    if (!jobs) return NSFF_ERR_NULL;
    if (n_jobs < 1 || n_jobs > NSFF_MAX_TIME_BIAS_JOBS || n_rays < 0) return NSFF_ERR_INVALID;
    if (n_rays == 0) return NSFF_OK;
    TimeBiasArgs a{};
    a.n_rays = n_rays;
    int max_rows = 0;
    for (int q = 0; q < n_jobs; ++q) {
        const NsffTimeBiasJob& jb = jobs[q];
        if (!jb.desc || !jb.out) return NSFF_ERR_NULL;
        if (!jb.t_rows && (!jb.table || !jb.ts || jb.n_table < 1 || jb.delta < -1 || jb.delta > 1)) return NSFF_ERR_NULL;
        const NsffModelDesc& d = *jb.desc;
        NsffLayoutH3 L;
        const int rc = nsff_make_layout_h3(d, L);
        if (rc) return rc;
        TimeBiasJobK& k = a.job[q];
        k.rows = nsff_time_bias_rows(&d);
        if (k.rows < 1) return NSFF_ERR_INVALID;
        // (the kernel stages 64 time-code columns per ray as float4s: wider or odd-width time codes are not its job -- such
        //  models keep their time-code columns on the matrix pipe, the field dispatcher never asks for rows here)
        if (d.in_t > 64 || (d.in_t & 3) != 0) return NSFF_ERR_INVALID;
        k.in_xyz = d.in_xyz; k.in_t = d.in_t; k.t_rows = jb.t_rows; k.out = jb.out;
        k.table = jb.table; k.ts = reinterpret_cast<const long long*>(jb.ts); k.n_table = jb.n_table; k.max_t = jb.max_t;
        k.delta = jb.delta; k.rows_out = jb.t_rows ? nullptr : jb.rows_out;
        const uint32_t skips = nsff_skip_layers(&d);
        int i = 0;
        for (int l = 0; l < d.D; ++l) {
            if (l != 0 && !((skips >> l) & 1u)) continue;
            k.w[i] = jb.w[i]; k.b[i] = jb.b[i];
            if (!k.w[i] || !k.b[i]) return NSFF_ERR_NULL;
            k.ld[i] = d.in_xyz + d.in_t + (l ? NSFF_W : 0);
            ++i;
        }
        max_rows = std::max(max_rows, k.rows);
    }
    const long long gx = (n_rays + TB_RAYS - 1) / TB_RAYS;
    if (gx > 0x7fffffffLL) return NSFF_ERR_INVALID;
    hipLaunchKernelGGL(nsff_time_bias_kernel, dim3((unsigned)gx, (unsigned)max_rows, (unsigned)n_jobs), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), a);
    return nsff_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------
// nsff_side_bias: per-ray rows  b_fold + W_dir[:, 256 : 256 + in_dir + in_a] [dir | a]  of static_dir_encoding (see the header).
// One thread per neuron, sixteen rays per workgroup; the rays' side inputs and sixteen weight columns at a time are staged in
// LDS (consecutive lanes read consecutive floats of a weight row; the tile is stored column-major with a one-float pad).
namespace {
struct SideBiasArgs { const float* w; const float* b; const float* dir_rows; const float* a_rows; float* out;
                      int in_dir, in_a, ld; long long n_rays; };
__global__ __launch_bounds__(256) void nsff_side_bias_kernel(const SideBiasArgs a) {
    const int n = threadIdx.x;
    const long long r0 = (long long)blockIdx.x * TB_RAYS;
    const int in = a.in_dir + a.in_a;
    __shared__ __attribute__((aligned(16))) float st[TB_RAYS][NSFF_W];      // [dir | a] of the workgroup's rays, zero-padded
    __shared__ float sw[16][NSFF_W + 1];
    for (int e = threadIdx.x; e < TB_RAYS * NSFF_W; e += 256) {
        const int r = e >> 8, c = e & 255;
        const long long ray = r0 + r < a.n_rays ? r0 + r : a.n_rays - 1;
        float v = 0.f;
        if (c < a.in_dir) v = a.dir_rows[ray * a.in_dir + c];
        else if (c < in) v = a.a_rows[ray * a.in_a + (c - a.in_dir)];
        st[r][c] = v;
    }
    float acc[TB_RAYS];
    const float b = a.b[n];
#pragma unroll
    for (int r = 0; r < TB_RAYS; ++r) acc[r] = b;
    const float* __restrict__ wl = a.w + NSFF_W;          // the [dir | a] columns follow the 256 columns that read *_final
    for (int c0 = 0; c0 < in; c0 += 16) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int e = threadIdx.x + 256 * k, row = e >> 4, c = c0 + (e & 15);
            sw[e & 15][row] = c < in ? wl[(long long)row * a.ld + c] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 16; u += 4) {                   // (columns in ascending order: one fixed summation order per output)
            const float w0 = sw[u][n], w1 = sw[u + 1][n], w2 = sw[u + 2][n], w3 = sw[u + 3][n];
#pragma unroll
            for (int r = 0; r < TB_RAYS; ++r) {
                const float4 t4 = *reinterpret_cast<const float4*>(&st[r][c0 + u]);
                acc[r] = fmaf(w3, t4.w, fmaf(w2, t4.z, fmaf(w1, t4.y, fmaf(w0, t4.x, acc[r]))));
            }
        }
    }
#pragma unroll
    for (int r = 0; r < TB_RAYS; ++r)
        if (r0 + r < a.n_rays) a.out[(r0 + r) * NSFF_W + n] = acc[r];
}
}  // namespace

extern "C" int nsff_side_bias(const NsffModelDesc* desc, const void* packed_f16x3, const float* w_dir, const float* dir_rows,
                              const float* a_rows, int64_t n_rays, float* out, void* stream) {
    if (!desc || !packed_f16x3 || !w_dir || !dir_rows || !out) return NSFF_ERR_NULL;
    const NsffModelDesc& d = *desc;
    NsffLayoutH3 L;
    const int rc = nsff_make_layout_h3(d, L);
    if (rc) return rc;
    if (!d.use_viewdir || d.in_dir < 1 || n_rays < 0) return NSFF_ERR_INVALID;
    if (d.in_a > 0 && !a_rows) return NSFF_ERR_NULL;
    if (n_rays == 0) return NSFF_OK;
    SideBiasArgs a{};
    a.w = w_dir; a.b = reinterpret_cast<const float*>(packed_f16x3) + L.dir_b_fold;
    a.dir_rows = dir_rows; a.a_rows = a_rows; a.out = out;
    a.in_dir = d.in_dir; a.in_a = d.in_a; a.ld = NSFF_W + d.in_dir + d.in_a; a.n_rays = n_rays;
    const long long gx = (n_rays + TB_RAYS - 1) / TB_RAYS;
    if (gx > 0x7fffffffLL) return NSFF_ERR_INVALID;
    hipLaunchKernelGGL(nsff_side_bias_kernel, dim3((unsigned)gx), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    return nsff_launch_status();
}

// which kernel the last f16 / f16x3 launch of this process took (nsff_last_field_kernel: tests assert that large inference
// launches really run the hand-scheduled body instead of silently falling back)
int g_nsff_last_h3_kernel = 0;
// compute units of the CURRENT device (per device id: a process may drive several devices, and a persistent grid / trunk-by-XCD
// split sized for another part would be wrong for this one -- results would stay correct, occupancy not)
static int h3a_device_cus() {
    static int cus[64];                 // 0 = not asked yet, -1 = the query failed
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (cus[dev] == 0) {
        int n = 0;
        cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : -1;
    }
    return cus[dev] > 0 ? cus[dev] : 0;
}

int g_nsff_last_h3_grid = 0;          // workgroups of that launch when it was a hand-scheduled inference launch (nsff_last_field_grid)

// The [dir | a] input tile of static_dir_encoding as the weight-gradient GEMM of those columns reads it (NsffFieldArgs::save_side:
// fp16, per 64-point tile [16-point group][32-row block][lane = row + 32 (8-point group)][8 points]; rows [0, in_dir) the direction
// embedding, [in_dir, in_dir + in_a) the appearance code, zeros behind) -- from the per-RAY rows (rendering.py:153-172 repeat them over
// a ray's samples).  The hand-scheduled training forward never sees these columns (they arrive as bias rows): one 16-byte store per
// thread and block here instead.
namespace {
struct SideTileArgs { const float* dir_emb; const float* a_emb; _Float16* out; long long n_points; int pts_per_ray, in_dir, in_a, side_rows; };
__global__ __launch_bounds__(256) void nsff_side_tile_kernel(const SideTileArgs a) {
    const long long tile = blockIdx.x, p0 = tile * 64;
    const int lane = threadIdx.x & 63, rblocks = a.side_rows >> 5;
    _Float16* dst = a.out + tile * (64LL * a.side_rows);
    for (int blk = threadIdx.x >> 6; blk < 4 * rblocks; blk += 4) {
        const int ks = blk / rblocks, rb = blk % rblocks;
        const int row = 32 * rb + (lane & 31), pt0 = 16 * ks + 8 * (lane >> 5);
        h8 out;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const long long p = p0 + pt0 + t;
            float v = 0.f;
            if (p < a.n_points) {
                const long long ray = p / a.pts_per_ray;
                if (row < a.in_dir) v = a.dir_emb[ray * a.in_dir + row];
                else if (row < a.in_dir + a.in_a) v = a.a_emb[ray * a.in_a + (row - a.in_dir)];
            }
            out[t] = (_Float16)v;
        }
        *reinterpret_cast<h8*>(dst + (((long long)ks * rblocks + rb) * 64 + lane) * 8) = out;
    }
}
}  // namespace

int nsff_h3_field_query(const NsffModelDesc* desc, const void* packed, const NsffFieldArgs* args,
                        int points_per_block, hipStream_t st, unsigned long long* span) {
    const NsffModelDesc& d = *desc;
    const NsffFieldArgs& g = *args;
    g_nsff_last_h3_grid = 0;
    H3KArgs k{};
    const int rc = nsff_make_layout_h3(d, k.L);
    if (rc) return rc;
    if (g.xyz && 3 + 6 * g.n_freqs != d.in_xyz) return NSFF_ERR_INVALID;
    k.packed = reinterpret_cast<const uint32_t*>(packed);
    k.span = span;
    k.xyz = g.xyz; k.x_emb = g.x_emb; k.dir_emb = g.dir_emb; k.a_emb = g.a_emb; k.t_emb = g.t_emb;
    k.raw = g.raw; k.n_points = g.n_points; k.pts_per_ray = g.pts_per_ray;
    k.save_acts = reinterpret_cast<_Float16*>(g.save_acts);
    k.save_xin = reinterpret_cast<_Float16*>(g.save_xin);
    k.save_masks = reinterpret_cast<unsigned long long*>(g.save_masks);
    k.save_side = reinterpret_cast<_Float16*>(g.save_side);
    k.n_tiles = (g.n_points + 63) / 64;
    k.save_stride = k.n_tiles * 64 * NSFF_W;
    if ((g.save_acts || g.save_xin || g.save_masks || g.save_side) && !g.xyz) return NSFF_ERR_INVALID;
    const void* const lo_of[3] = {g.save_acts, g.save_xin, g.save_side};
    bool save_lo = false;
    for (int i = 0; i < 3; ++i) {
        if (g.save_lo_delta[i] < 0 || (g.save_lo_delta[i] & 7) || (g.save_lo_delta[i] != 0 && !lo_of[i])) return NSFF_ERR_INVALID;
        k.lo_delta[i] = g.save_lo_delta[i];
        save_lo = save_lo || g.save_lo_delta[i] != 0;
    }
    k.xin_rows = (k.L.k0s + k.L.kt) <= 128 ? 128 : 256;          // row geometry of save_xin / save_side (nsff_train_dims)
    k.side_rows = k.L.side_k <= 128 ? 128 : 256;
    k.static_mode = g.static_mode; k.transient_mode = g.transient_mode;
    k.D = d.D; k.skip = d.skip;   // (the step program below carries the skip layers)
    k.in_xyz = d.in_xyz; k.in_dir = d.in_dir; k.in_a = d.in_a; k.in_t = d.in_t;
    k.use_viewdir = d.use_viewdir; k.flow_scale = d.flow_scale;
    k.n_freqs = g.n_freqs;
    for (int i = 0; i < NSFF_MAX_FREQS; ++i) k.freqs[i] = g.freqs[i];
    k.octave_freqs = (g.xyz != nullptr && g.n_freqs >= 1 && g.n_freqs <= 10 && k.L.k0s == 64) ? 1 : 0;
    for (int i = 0; i + 1 < g.n_freqs; ++i)
        if (g.freqs[i + 1] != 2.0f * g.freqs[i]) k.octave_freqs = 0;
    k.ld_emb = g.ld_emb; k.off_xyz = g.off_xyz; k.off_dir = g.off_dir; k.off_a = g.off_a; k.off_t = g.off_t;

    // Every f16x3 launch -- inference and training forward alike -- runs the FOLDED step program: the activation-free
    // *_xyz_encoding_final layers are never executed, the heads that read them are evaluated with pre-multiplied rows.  A
    // training forward saves the trunk layers' activations only; the backward pass (field_bwd.hip) is folded the same way.
    const bool fold = true;
    {
        const int prc = h3_step_program(d, g.static_mode, g.transient_mode, fold, k);
        if (prc != NSFF_OK) return prc;
    }
    const int n = k.n_steps;

    // one trunk per workgroup when the launch evaluates both (see the kernel): grid = 2 x tiles
    const bool both = n > k.n_static_steps && k.n_static_steps > 0;
    auto launch = [&](auto kernel, int tile_points, int threads) -> int {
        const long long tiles = (g.n_points + tile_points - 1) / tile_points;
        if (tiles * 2 > 0x7fffffffLL) return NSFF_ERR_INVALID;
        k.grid_tiles = tiles;
        k.split_trunks = both ? 1 : 0;
        hipLaunchKernelGGL(kernel, dim3((unsigned)(both ? 2 * tiles : tiles)), dim3(threads), 0, st, k);
        return NSFF_OK;
    };
    int lrc;
    const bool saves = k.save_acts || k.save_xin || k.save_masks || k.save_side;
    if (saves && points_per_block == 64) { // training forward, 64-point tiling (A/B against the default below)
        lrc = launch(nsff_field_kernel_h3<2, 1, true>, 64, 256);
        g_nsff_last_h3_kernel = NSFF_KERNEL_H3_SAVE;
    } else if (saves) {
        // Training forward, 128-point tiles.  Default: the hand-scheduled body in its SAVE build (nsff_field_kernel_h3a_save) for
        // launches it covers -- raw positions, a 64-column position embedding, time codes of at most 64 columns in float4 rows,
        // no view-direction branch (its [dir | a] input tile is the eight-wave kernel's), activation and sign-word buffers both
        // given, an EVEN number of 64-point tiles (a workgroup saves both of its tiles) below 4 GiB per slot; otherwise -- and with
        // points_per_block == 131 -- eight waves of 32 neurons, compiler-scheduled.
        H3AArgs ka{};
        ka.n_bias[0] = ka.n_bias[1] = 0; ka.head[0] = ka.head[1] = HEAD_NONE;
        // A view-direction static trunk (round 6; the reference's documented training configuration, README.md:226-233): covered when
        // the caller supplied its per-ray rows (nsff_side_bias) and no 64-point half straddles two rays -- the launch then runs the
        // side-fold step program of the inference launches (static_dir_encoding as one folded 256-wide segment with per-ray bias
        // rows, sigma as a ride of the last trunk layer's epilogues) in its SAVE build; the [dir | a] input tile the weight-gradient
        // GEMM of those columns reads (save_side) is written by a small launch of its own from the per-ray rows.
        const bool viewdir_static = g.static_mode == 2 && d.use_viewdir;
        H3KArgs ks = k;
        bool side = false;
        // (remainder planes for the three-product backward, NsffFieldArgs::save_lo_delta: the eight-wave kernel writes them)
        bool asm_body = points_per_block != 131 && !save_lo && g.xyz != nullptr && k.L.k0s == 64 &&
                        k.save_acts != nullptr && k.save_masks != nullptr && (k.n_tiles & 1) == 0 &&
                        k.n_tiles * (64LL * NSFF_W * 2) < 0x100000000LL && g.n_points <= 0x7fffffffLL;
        if (asm_body && viewdir_static)
            asm_body = g.s_bias != nullptr && g.s_bias_rows == 1 && g.pts_per_ray > 0 && g.pts_per_ray % 64 == 0 && g.dir_emb != nullptr &&
                       (d.in_a == 0 || g.a_emb != nullptr) && h3_step_program(d, g.static_mode, g.transient_mode, true, ks, true) == NSFF_OK;
        else if (asm_body)
            asm_body = k.save_side == nullptr;
        if (asm_body && g.transient_mode)
            asm_body = k.L.kt == 64 && (d.in_t & 3) == 0 && ((uintptr_t)g.t_emb & 15) == 0;
        if (asm_body) {
            if (viewdir_static) { ks.s_bias = g.s_bias; ks.sb_rows = 1; side = true; }
            ka.k = ks;
            if (ks.n_static_steps > 0)
                asm_body = h3a_build_program(ks, 0, ks.n_static_steps, false, false, ka.ph[0], ka.bias_off[0], ka.n_bias[0], ka.head[0],
                                             nullptr, side, side ? &ka.sig_ride : nullptr, true);
            if (asm_body && ks.n_steps > ks.n_static_steps)
                asm_body = h3a_build_program(ks, ks.n_static_steps, ks.n_steps, true, false, ka.ph[1], ka.bias_off[1], ka.n_bias[1], ka.head[1],
                                             nullptr, false, nullptr, true);
        }
        if (asm_body) {
            const long long tiles = (g.n_points + 127) / 128;
            if (tiles * 2 > 0x7fffffffLL) return NSFF_ERR_INVALID;
            if (side) {
                ka.sig_b_off = k.L.s_sigma_b;
                if (k.save_side != nullptr) {
                    SideTileArgs sa{g.dir_emb, g.a_emb, k.save_side, g.n_points, g.pts_per_ray, d.in_dir, d.in_a, k.side_rows};
                    hipLaunchKernelGGL(nsff_side_tile_kernel, dim3((unsigned)k.n_tiles), dim3(256), 0, st, sa);
                }
            }
            ka.hsel[0] = h3a_head_sel(k.L, ka.head[0]); ka.hsel[1] = h3a_head_sel(k.L, ka.head[1]);
            ka.k.grid_tiles = tiles;
            ka.k.split_trunks = both ? 1 : 0;
            hipLaunchKernelGGL(nsff_field_kernel_h3a_save, dim3((unsigned)(both ? 2 * tiles : tiles)), dim3(256), 0, st, ka);
            lrc = NSFF_OK;
            g_nsff_last_h3_kernel = NSFF_KERNEL_H3A_SAVE;
        } else {                                  // eight waves of 32 neurons
            lrc = launch(nsff_field_kernel_h3<4, 1, true, 1>, 128, 512);
            g_nsff_last_h3_kernel = NSFF_KERNEL_H3_SAVE;
        }
    } else if (points_per_block == 64) {
        lrc = launch(nsff_field_kernel_h3<2, 1>, 64, 256);
        g_nsff_last_h3_kernel = NSFF_KERNEL_H3_64;
    } else {
        // 128 points per workgroup.  Default: the hand-scheduled body (nsff_field_kernel_h3a) whenever the launch's trunks have a
        // structure it executes (raw positions, a 64-column position embedding, a time code of at most 64 columns in float4
        // rows, no view-direction branch in this launch); otherwise -- and with points_per_block == 131 -- the
        // compiler-scheduled eight-wave form.
        // A launch whose STATIC trunk has the view-direction branch (not covered) next to a dynamic trunk is issued as two
        // launches: the static workgroups on the eight-wave kernel, the dynamic ones on the hand-scheduled kernel -- each
        // writes its own part of the raw records (piece 1 / piece 2), as the workgroups of one split launch do.
        bool static_uncovered = g.static_mode == 2 && d.use_viewdir;
        H3AArgs ka{};
        ka.n_bias[0] = ka.n_bias[1] = 0; ka.head[0] = ka.head[1] = HEAD_NONE;
        const bool asm_inputs = points_per_block != 131 && g.xyz != nullptr && k.L.k0s == 64;
        // A view-direction static trunk is covered when the caller supplied its per-ray rows (nsff_side_bias) and no 64-point half
        // straddles two rays: the launch then runs the side-fold step program (static_dir_encoding as one folded 256-wide segment
        // with per-ray bias rows, sigma as a ride of the last trunk layer's epilogues) -- `ks` replaces `k` for this launch.
        H3KArgs ks = k;
        bool side = false;
        if (static_uncovered && asm_inputs && g.s_bias != nullptr && g.s_bias_rows == 1 && g.pts_per_ray > 0 &&
            g.pts_per_ray % 64 == 0 && g.n_points <= 0x7fffffffLL &&
            h3_step_program(d, g.static_mode, g.transient_mode, true, ks, true) == NSFF_OK) {
            ks.s_bias = g.s_bias; ks.sb_rows = 1;
            side = h3a_build_program(ks, 0, ks.n_static_steps, false, false, ka.ph[0], ka.bias_off[0], ka.n_bias[0], ka.head[0], nullptr,
                                     true, &ka.sig_ride);
        }
        if (side) { static_uncovered = false; ka.sig_b_off = k.L.s_sigma_b; }
        else { ks = k; ka.sig_ride = 0; ka.n_bias[0] = 0; ka.head[0] = HEAD_NONE; }
        const int ns = ks.n_steps;
        bool asm_body = asm_inputs && !(static_uncovered && !g.transient_mode);
        if (asm_body && g.transient_mode)
            asm_body = k.L.kt == 64 && (d.in_t & 3) == 0 && ((uintptr_t)g.t_emb & 15) == 0;
        if (asm_body) {
            ka.k = ks;
            if (ks.n_static_steps > 0 && !static_uncovered && !side)
                asm_body = h3a_build_program(ks, 0, ks.n_static_steps, false, false, ka.ph[0], ka.bias_off[0], ka.n_bias[0], ka.head[0]);
            if (asm_body && ns > ks.n_static_steps) {
                // the time code as per-ray bias rows (nsff_time_bias): whenever the caller supplied them and a 64-point half
                // never straddles two rays; otherwise the body multiplies the time-code columns like any other input
                bool fold_t = g.t_bias != nullptr && g.pts_per_ray > 0 && g.pts_per_ray % 64 == 0 && g.n_points <= 0x7fffffffLL &&
                              g.t_bias_rows == nsff_time_bias_rows(desc);
                if (fold_t) {
                    ka.k.t_bias = g.t_bias; ka.k.tb_rows = g.t_bias_rows;
                    fold_t = h3a_build_program(ka.k, ks.n_static_steps, ns, true, true, ka.ph[1], ka.bias_off[1], ka.n_bias[1], ka.head[1]);
                }
                if (!fold_t) {
                    ka.k.t_bias = nullptr; ka.k.tb_rows = 0;
                    asm_body = h3a_build_program(ks, ks.n_static_steps, ns, true, false, ka.ph[1], ka.bias_off[1], ka.n_bias[1], ka.head[1]);
                }
            }
        }
        if (side && !asm_body) {
            // (the dynamic trunk of this launch is not covered: back to the plain step program, static trunk on the eight-wave kernel)
            static_uncovered = true; side = false;
            asm_body = false;
        }
        const long long tiles = (g.n_points + 127) / 128;
        if (tiles * 2 > 0x7fffffffLL) return NSFF_ERR_INVALID;
        ka.hsel[0] = h3a_head_sel(k.L, ka.head[0]); ka.hsel[1] = h3a_head_sel(k.L, ka.head[1]);
        const int which = side ? NSFF_KERNEL_H3A_SIDE : (ka.k.t_bias ? NSFF_KERNEL_H3A_TBIAS : NSFF_KERNEL_H3A);
        // Persistent form (H3AArgs::p_mode): one workgroup per CU walking its tiles -- for launches that give every workgroup at
        // least one tile.  Each workgroup keeps ONE trunk: trunks of equal cost split the chip by XCD, unequal ones (the
        // view-direction static trunk is 23 % longer than the dynamic one) by their share of the matrix steps.
        // NSFF_NO_PERSIST=1: one workgroup per tile (A/B).
        const int n_cus = h3a_device_cus();
        const bool no_persist = g.launch_form == 1 || getenv("NSFF_NO_PERSIST") != nullptr;       // (the variable is read per launch: tests flip it)
        const bool can_persist = !no_persist && n_cus >= 8 && n_cus % 8 == 0;
        auto cost = [&](const H3APhase* ph) {
            int c = 0;
            for (int i = 1; i < H3A_MAX_PHASES && !(i > 1 && ph[i].d[0] == H3A_BODY_END); ++i) {
                const uint32_t b = ph[i].d[0];
                c += (b == H3A_BODY_A16R || b == H3A_BODY_A16RS) ? 16 : ((b == H3A_BODY_A4 || b == H3A_BODY_A4F) ? 4 : ((b == H3A_BODY_A8 || b == H3A_BODY_A8F) ? 8 : 0));
            }
            return c;
        };
        if (asm_body && static_uncovered) {
            // (1) static trunk: the first half of a split launch's grid = static workgroups only
            k.grid_tiles = tiles;
            k.split_trunks = 1;
            hipLaunchKernelGGL((nsff_field_kernel_h3<4, 1, false, 1>), dim3((unsigned)tiles), dim3(512), 0, st, k);
            // (2) dynamic trunk: a split launch whose static half is empty (grid_tiles = 0: every workgroup is a dynamic one)
            ka.k.grid_tiles = 0;
            ka.k.split_trunks = 1;
            unsigned grid = (unsigned)tiles;
            if (can_persist && tiles >= n_cus && h3a_make_persistent(ka.ph[1])) { ka.p_mode = 3; ka.p_tiles = tiles; grid = (unsigned)n_cus; }
            hipLaunchKernelGGL(nsff_field_kernel_h3a, dim3(grid), dim3(256), 0, st, ka);
            lrc = NSFF_OK;
            g_nsff_last_h3_kernel = which;
            g_nsff_last_h3_grid = (int)grid;
        } else if (asm_body) {
            const bool both2 = ns > ks.n_static_steps && ks.n_static_steps > 0;
            ka.k.grid_tiles = tiles;
            ka.k.split_trunks = both2 ? 1 : 0;
            unsigned grid = (unsigned)(both2 ? 2 * tiles : tiles);
            if (can_persist && !both2 && tiles >= n_cus && h3a_make_persistent(ka.ph[ks.n_static_steps > 0 ? 0 : 1])) {
                ka.p_mode = 1; ka.p_tiles = tiles; grid = (unsigned)n_cus;
            }
            if (can_persist && both2 && tiles >= n_cus / 2) {
                const int cs = cost(ka.ph[0]), cd = cost(ka.ph[1]);
                const bool equal = 25 * std::abs(cs - cd) <= std::max(cs, cd);
                // Unequal trunks (mode 4, see H3AArgs): the longer trunk (cost cl per tile) hands its last `steal` tiles to the
                // second round, run by the shorter trunk's XCDs behind their own tiles:  (tiles - steal) cl = tiles cs' + steal cl
                const int cl = std::max(cs, cd), csh = std::min(cs, cd);
                const long long steal = equal ? 0 : (tiles * (cl - csh) + cl) / (2LL * cl);
                H3APhase keep[H3A_MAX_PHASES];
                for (int i = 0; i < H3A_MAX_PHASES; ++i) keep[i] = ka.ph[0][i];
                if (tiles - steal >= n_cus / 2 && h3a_make_persistent(ka.ph[0])) {       // (every first-round workgroup needs a tile)
                    if (h3a_make_persistent(ka.ph[1])) {
                        ka.p_tiles = tiles; grid = (unsigned)n_cus;
                        if (steal == 0) ka.p_mode = 2;
                        else { ka.p_mode = 4; ka.p_long = cs > cd ? 0 : 1; ka.p_split = (int)(tiles - steal); grid = 2u * (unsigned)n_cus; }
                    } else for (int i = 0; i < H3A_MAX_PHASES; ++i) ka.ph[0][i] = keep[i];
                }
            }
            hipLaunchKernelGGL(nsff_field_kernel_h3a, dim3(grid), dim3(256), 0, st, ka);
            lrc = NSFF_OK;
            g_nsff_last_h3_kernel = which;
            g_nsff_last_h3_grid = (int)grid;
        } else {                                  // eight waves of 32 neurons (half the weight stream of the 64-point tiling)
            lrc = launch(nsff_field_kernel_h3<4, 1, false, 1>, 128, 512);
            g_nsff_last_h3_kernel = NSFF_KERNEL_H3_8WAVE;
        }
    }
    if (lrc != NSFF_OK) return lrc;
    return nsff_launch_status();
}
