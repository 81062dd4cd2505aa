// Register-resident fp16-split field kernel for gfx950 ("RA" = register activations).
//
// Same function as field.hip / field_h3.hip: PosEmbedding + NeRF.forward of the reference
// (models/nerf.py:17-30,118-213) for a batch of points, arithmetic = f16x3 (every fp32 product as
// three f16 MFMAs with fp32 accumulation, see field_h3.hip).  What changes is the dataflow:
//
//   * a wave owns 32 points and ALL 256 neurons of every layer.  The 32x32 accumulator of
//     output tile mt, after bias+ReLU and the hi/lo split, IS the B operand of k-steps 2mt, 2mt+1
//     of the next layer (the weight columns are permuted to make this true, nsff_layout_ra.h),
//     so activations never leave the register file: no LDS round trip, no inter-wave barrier
//     on the data path, no bank conflicts, and the epilogue VALU work of one tile pair overlaps
//     the MFMAs of the next pair inside the same wave;
//   * the only thing that moves is the weight stream: 4 KiB chunks, global -> LDS by
//     direct-to-LDS DMA (global_load_lds_dwordx4) into a 128 KiB ring shared by the 4 waves of
//     the workgroup (one workgroup per CU, 128 points per weight byte fetched from L2 -- half
//     the L2 traffic of the LDS-activation kernels, which were L2-stream bound);
//   * one s_barrier per 16 KiB group keeps producer and consumers in step; DMA completion is
//     tracked with counted s_waitcnt vmcnt (never 0), so eight groups stay in flight.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <vector>
#include "nsff_layout_ra.h"
#include "nsff_common.h"
#include "nsff_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __fp16 h2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int NTHREADS = 256;
constexpr int RING_GROUPS = 8;
constexpr int GROUP_BYTES = 4 * RA_CHUNK_BYTES;
constexpr int RING_BYTES = RING_GROUPS * GROUP_BYTES;        // 131072
constexpr int MAX_BIAS_FLOATS = 20 * NSFF_W + 96;
constexpr int LDS_BYTES = RING_BYTES + MAX_BIAS_FLOATS * 4;  // 151,936 <= 163,840
constexpr int MAX_STEPS = 28;

enum { STEP_LAYER = 0, STEP_HEAD = 1 };
enum { ACT_NONE = 0, ACT_SIGMOID = 1, ACT_FLOW = 2 };

struct RAStep {
    uint32_t chunk0;
    uint32_t kinds;       // heads: 2-bit ACT_* per row
    uint16_t bias_off4;   // bias table offset in units of 4 floats
    uint8_t kind;         // STEP_*
    uint8_t nkh;          // 0 or 16
    uint8_t xs;           // RA_XS_*
    uint8_t relu;
    uint8_t n_rows;       // heads: live rows
    uint8_t slot0;        // heads: first raw-record slot
};
static_assert(sizeof(RAStep) == 16, "RAStep layout");

// Step program of one (static_mode, transient_mode) combination.  The nine programs of a model are
// written into the packed buffer at pack time and read by the kernel through the scalar cache
// (a by-value kernel argument indexed at run time would be spilled to scratch by hipcc).
struct RAProgram {
    int n_steps;
    int trunk_on[2];              // [static, dynamic]
    int trunk_sigma_head[2];      // head evaluated on the last trunk activation (static sigma)
    int trunk_full[2];            // *_final (+ dir) + main head
    int trunk_dir[2];             // static_dir_encoding present
    int pad[7];
    RAStep steps[MAX_STEPS];      // consumed strictly in order by the kernel's fixed trunk structure
};
static_assert(sizeof(RAProgram) == 64 + 16 * MAX_STEPS, "RAProgram layout");

struct RAKArgs {
    const RAProgram* __restrict__ prog;
    const char* stream;         // packed weight stream (chunks)
    const float* bias;          // bias table (n_bias_floats)
    int n_bias_floats;
    const float* xyz;
    const float* x_emb;
    const float* dir_emb;
    const float* a_emb;
    const float* t_emb;
    float* raw;
    long long n_points;
    int pts_per_ray;
    int in_xyz, in_dir, in_a, in_t;
    float flow_scale;
    int n_freqs;
    float freqs[NSFF_MAX_FREQS];
    int ld_emb, off_xyz, off_dir, off_a, off_t;
};

#define MFMA_H(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

struct Acts { h8 hi[16], lo[16]; };     // a 256-wide activation as B operands (this wave's 32 points)
struct X4 { h8 hi[4], lo[4]; };         // four k-steps (64 columns) of network / side input
struct Chunk { uint4 p[4]; };           // one 4 KiB chunk as seen by a lane

__device__ __forceinline__ h8 as_h8(const uint4& v) { return __builtin_bit_cast(h8, v); }

// direct-to-LDS DMA of 16 bytes per lane: LDS[lds_dst + 16*lane] <- *gsrc  (lds_dst wave-uniform)
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

struct Ctx {
    const char* src_lane;     // stream + wave*1024 + lane*16
    unsigned lds_wave;        // LDS byte address of ring + wave*1024 (wave-uniform)
    const char* ring_lane;    // generic pointer to ring + lane*16 (for ds_read)
    int prod_step, prod_chunk, prod_group;   // producer cursor
    int cons_group;
};

__device__ __forceinline__ uint32_t step_chunks(const RAStep& s) {
    return s.kind == STEP_HEAD ? 8u : 4u * (s.nkh + (s.xs == RA_XS_NONE ? 0u : (s.xs == RA_XS_EMB ? 4u : 8u)));
}

// Issue the DMA of the next 16 KiB group of the stream (always 4 instructions per wave so that
// the vmcnt arithmetic is uniform; past the end of the program the last group is re-loaded).
__device__ __forceinline__ void issue_group(Ctx& cx, const RAKArgs& a) {
    const RAStep st = a.prog->steps[__builtin_amdgcn_readfirstlane(cx.prod_step)];
    const uint32_t src = st.chunk0 + (uint32_t)cx.prod_chunk;
    const unsigned dst = cx.lds_wave + (unsigned)(cx.prod_group & (RING_GROUPS - 1)) * GROUP_BYTES;
    const char* g = cx.src_lane + (size_t)src * RA_CHUNK_BYTES;
#pragma unroll
    for (int c = 0; c < 4; ++c)
        dma16(g + c * RA_CHUNK_BYTES, __builtin_amdgcn_readfirstlane(dst + c * RA_CHUNK_BYTES));
    cx.prod_group++;
    const int n = (int)step_chunks(st);
    cx.prod_chunk += 4;
    if (cx.prod_chunk >= n) {
        if (cx.prod_step + 1 < a.prog->n_steps) { cx.prod_step++; cx.prod_chunk = 0; }
        else cx.prod_chunk = n - 4;
    }
}

// Start consuming the next group: my DMA pieces of the group AFTER it have landed, everyone's have
// (barrier), everyone is done with the previous group -> its ring slot is refilled.
// Returns this group's LDS pointer (lane-adjusted); *next = the following group's.
__device__ __forceinline__ const uint4* group_begin(Ctx& cx, const RAKArgs& a, const uint4** next) {
    asm volatile("s_waitcnt vmcnt(20)" ::: "memory");        // (RING_GROUPS-3) groups x 4 may stay in flight
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue_group(cx, a);
    const int g = cx.cons_group++;
    *next = reinterpret_cast<const uint4*>(cx.ring_lane + ((g + 1) & (RING_GROUPS - 1)) * GROUP_BYTES);
    return reinterpret_cast<const uint4*>(cx.ring_lane + (g & (RING_GROUPS - 1)) * GROUP_BYTES);
}

__device__ __forceinline__ Chunk read_chunk(const uint4* gp, int c) {
    Chunk k;
    k.p[0] = gp[c * 256 + 0]; k.p[1] = gp[c * 256 + 64]; k.p[2] = gp[c * 256 + 128]; k.p[3] = gp[c * 256 + 192];
    return k;
}

// layer chunk = [mt2][part]: p0 = hi(mt 0), p1 = lo(mt 0), p2 = hi(mt 1), p3 = lo(mt 1)
__device__ __forceinline__ void mma_layer_chunk(f32x16& a0, f32x16& a1, const Chunk& k, const h8& bh, const h8& bl) {
    a0 = MFMA_H(as_h8(k.p[1]), bh, a0);
    a1 = MFMA_H(as_h8(k.p[3]), bh, a1);
    a0 = MFMA_H(as_h8(k.p[0]), bl, a0);
    a1 = MFMA_H(as_h8(k.p[2]), bl, a1);
    a0 = MFMA_H(as_h8(k.p[0]), bh, a0);
    a1 = MFMA_H(as_h8(k.p[2]), bh, a1);
}

#ifndef RA_PF
#define RA_PF 1        // LDS chunk prefetch distance (1 or 2 chunks)
#endif
#ifndef RA_DEFER
#define RA_DEFER 0     // 1: a pair's bias/ReLU/split runs inside the next pair's first group (overlaps its MFMAs)
#endif
// Run one 4-chunk group.  c0 (and c1 when RA_PF == 2) hold the first chunk(s) of this group on entry and of
// the NEXT group on exit (the barrier of group g guarantees that group g+1 has landed as well).
#if RA_PF == 2
#define RA_GROUP(cx, a, c0, c1, BODY)                                  \
    do {                                                             \
        const uint4* gn_;                                            \
        const uint4* gp_ = group_begin(cx, a, &gn_);                 \
        const Chunk c2_ = read_chunk(gp_, 2);                        \
        { const Chunk& K = c0; constexpr int CI = 0; BODY }          \
        const Chunk c3_ = read_chunk(gp_, 3);                        \
        { const Chunk& K = c1; constexpr int CI = 1; BODY }          \
        c0 = read_chunk(gn_, 0);                                     \
        { const Chunk& K = c2_; constexpr int CI = 2; BODY }         \
        c1 = read_chunk(gn_, 1);                                     \
        { const Chunk& K = c3_; constexpr int CI = 3; BODY }         \
    } while (0)
#else
#define RA_GROUP(cx, a, c0, c1, BODY)                                  \
    do {                                                             \
        const uint4* gn_;                                            \
        const uint4* gp_ = group_begin(cx, a, &gn_);                 \
        const Chunk c1_ = read_chunk(gp_, 1);                        \
        { const Chunk& K = c0; constexpr int CI = 0; BODY }          \
        const Chunk c2_ = read_chunk(gp_, 2);                        \
        { const Chunk& K = c1_; constexpr int CI = 1; BODY }         \
        const Chunk c3_ = read_chunk(gp_, 3);                        \
        { const Chunk& K = c2_; constexpr int CI = 2; BODY }         \
        c0 = read_chunk(gn_, 0);                                     \
        { const Chunk& K = c3_; constexpr int CI = 3; BODY }         \
    } while (0)
#endif

__device__ __forceinline__ void acc_from_bias(f32x16& acc, const float* sBias, int mt, int lane) {
    // rows (r&3) + 8*(r>>2) + 4h + 32*mt: four consecutive floats per register quad
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 b = *reinterpret_cast<const float4*>(sBias + 32 * mt + 8 * q + 4 * (lane >> 5));
        acc[4 * q + 0] = b.x; acc[4 * q + 1] = b.y; acc[4 * q + 2] = b.z; acc[4 * q + 3] = b.w;
    }
}

// accumulator tile -> the two B-operand k-steps it feeds in the next layer (hi and lo halfs)
__device__ __forceinline__ void split_tile(const f32x16& acc, float floor_, h8& hi0, h8& lo0, h8& hi1, h8& lo1) {
#pragma unroll
    for (int w = 0; w < 2; ++w) {
        h8 hv, lv;
#pragma unroll
        for (int t = 0; t < 8; t += 2) {
            const float v0 = fmaxf(acc[8 * w + t], floor_), v1 = fmaxf(acc[8 * w + t + 1], floor_);
            const h2 h = __builtin_amdgcn_cvt_pkrtz(v0, v1);
            const h2 l = __builtin_amdgcn_cvt_pkrtz(v0 - (float)h[0], v1 - (float)h[1]);
            hv[t] = (_Float16)h[0]; hv[t + 1] = (_Float16)h[1];
            lv[t] = (_Float16)l[0]; lv[t + 1] = (_Float16)l[1];
        }
        if (w == 0) { hi0 = hv; lo0 = lv; } else { hi1 = hv; lo1 = lv; }
    }
}

__device__ __forceinline__ void split_pair(const f32x16 (&acc)[2], float floor_, Acts& out, int mp) {
    // static indices only (mp comes from an unrolled loop)
    split_tile(acc[0], floor_, out.hi[4 * mp + 0], out.lo[4 * mp + 0], out.hi[4 * mp + 1], out.lo[4 * mp + 1]);
    split_tile(acc[1], floor_, out.hi[4 * mp + 2], out.lo[4 * mp + 2], out.hi[4 * mp + 3], out.lo[4 * mp + 3]);
}

// A layer whose input includes the 256-wide hidden activation `in` (hidden, skip, *_final, dir).
// nx in {0,4,8}: extra input-segment k-steps taken from xin.
__device__ __forceinline__ void layer_h(const Acts& in, Acts& out, const X4& xa, const X4& xb, int nx, float floor_,
                                        const float* sBias, Ctx& cx, const RAKArgs& a, Chunk& c0, Chunk& c1, int lane) {
    f32x16 acc[2], prev[2];
#pragma unroll
    for (int mp = 0; mp < 4; ++mp) {
        acc_from_bias(acc[0], sBias, 2 * mp, lane);
        acc_from_bias(acc[1], sBias, 2 * mp + 1, lane);
#pragma unroll
        for (int gh = 0; gh < 4; ++gh) {
            RA_GROUP(cx, a, c0, c1, {
                if (RA_DEFER && gh == 0 && CI == 0 && mp > 0) split_pair(prev, floor_, out, mp - 1);
                mma_layer_chunk(acc[0], acc[1], K, in.hi[4 * gh + CI], in.lo[4 * gh + CI]);
            });
        }
        if (nx >= 4) {
            RA_GROUP(cx, a, c0, c1, { mma_layer_chunk(acc[0], acc[1], K, xa.hi[CI], xa.lo[CI]); });
        }
        if (nx == 8) {
            RA_GROUP(cx, a, c0, c1, { mma_layer_chunk(acc[0], acc[1], K, xb.hi[CI], xb.lo[CI]); });
        }
        if (RA_DEFER) { prev[0] = acc[0]; prev[1] = acc[1]; }
        else split_pair(acc, floor_, out, mp);
    }
    if (RA_DEFER) split_pair(prev, floor_, out, 3);
}

// First layer of a trunk: input segment only (nx = 4 or 8).
__device__ __forceinline__ void layer_x(Acts& out, const X4& xa, const X4& xb, int nx, float floor_, const float* sBias,
                                        Ctx& cx, const RAKArgs& a, Chunk& c0, Chunk& c1, int lane) {
#pragma unroll
    for (int mp = 0; mp < 4; ++mp) {
        f32x16 acc[2];
        acc_from_bias(acc[0], sBias, 2 * mp, lane);
        acc_from_bias(acc[1], sBias, 2 * mp + 1, lane);
        RA_GROUP(cx, a, c0, c1, { mma_layer_chunk(acc[0], acc[1], K, xa.hi[CI], xa.lo[CI]); });
        if (nx == 8) {
            RA_GROUP(cx, a, c0, c1, { mma_layer_chunk(acc[0], acc[1], K, xb.hi[CI], xb.lo[CI]); });
        }
        split_pair(acc, floor_, out, mp);
    }
}

// Narrow heads: one zero-padded 32-row tile, 8 chunks of two k-steps ([ksub][part]).
__device__ __forceinline__ void head(const Acts& in, const RAStep& st, const float* sBias, const RAKArgs& a,
                                     long long p, Ctx& cx, const RAKArgs&, Chunk& c0, Chunk& c1, int lane) {
    f32x16 e, o;
#pragma unroll
    for (int r = 0; r < 16; ++r) { e[r] = 0.f; o[r] = 0.f; }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        RA_GROUP(cx, a, c0, c1, {
            const int s0 = 2 * (4 * g + CI);
            e = MFMA_H(as_h8(K.p[1]), in.hi[s0], e);
            o = MFMA_H(as_h8(K.p[3]), in.hi[s0 + 1], o);
            e = MFMA_H(as_h8(K.p[0]), in.lo[s0], e);
            o = MFMA_H(as_h8(K.p[2]), in.lo[s0 + 1], o);
            e = MFMA_H(as_h8(K.p[0]), in.hi[s0], e);
            o = MFMA_H(as_h8(K.p[2]), in.hi[s0 + 1], o);
        });
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < st.n_rows) {
            float v = e[r] + o[r] + sBias[row];
            const unsigned kind = (st.kinds >> (2 * row)) & 3u;
            if (kind == ACT_SIGMOID) v = 1.0f / (1.0f + expf(-v));
            else if (kind == ACT_FLOW) v = a.flow_scale * tanhf(v);
            if (p < a.n_points) a.raw[p * NSFF_RAW_STRIDE + st.slot0 + row] = v;
        }
    }
}

__device__ __forceinline__ void split_pack8(const float (&v)[8], h8& hi, h8& lo) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const _Float16 h = (_Float16)v[t];
        hi[t] = h;
        lo[t] = (_Float16)(v[t] - (float)h);
    }
}

// xyz embedding of this lane's point in the lane-half column order of ra_col_emb
__device__ __forceinline__ void build_emb(X4& x, const RAKArgs& a, long long p, int lane) {
    const int h = lane >> 5;
    const bool valid = p < a.n_points;
    float v[32];
    if (a.xyz != nullptr) {
        float c[3] = {0.f, 0.f, 0.f};
        if (valid) { c[0] = a.xyz[p * 3 + 0]; c[1] = a.xyz[p * 3 + 1]; c[2] = a.xyz[p * 3 + 2]; }
#pragma unroll
        for (int fi = 0; fi < 5; ++fi) {
            const int f = 2 * fi + h;
            const float fr = h ? a.freqs[2 * fi + 1] : a.freqs[2 * fi];
            const bool live = f < a.n_freqs;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                float s, co;
                sincosf(fr * c[d], &s, &co);
                v[6 * fi + d] = live ? s : 0.f;
                v[6 * fi + 3 + d] = live ? co : 0.f;
            }
        }
        v[30] = h ? c[2] : c[0];
        v[31] = h ? 0.f : c[1];
    } else {
        const float* src = a.x_emb + p * a.ld_emb + a.off_xyz;
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const int col = ra_col_emb(q >> 3, h, q & 7, (a.in_xyz - 3) / 6);
            v[q] = (valid && col >= 0) ? src[col] : 0.f;
        }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        float w[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) w[t] = v[8 * s + t];
        split_pack8(w, x.hi[s], x.lo[s]);
    }
}

// natural-order columns [16s + 8h, +8) of a per-point row, into k-step `dst` of X8
__device__ __forceinline__ void load_natural(h8& hi, h8& lo, const float* row, int s, int h, int n_cols) {
    float w[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int c = 16 * s + 8 * h + t;
        w[t] = (row != nullptr && c < n_cols) ? row[c] : 0.f;
    }
    split_pack8(w, hi, lo);
}

__global__ __launch_bounds__(NTHREADS, 1) void nsff_field_kernel_ra(const RAKArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* sBiasAll = reinterpret_cast<float*>(lds + RING_BYTES);
    const long long p = (long long)blockIdx.x * 128 + 32 * wave + (lane & 31);

    Ctx cx;
    cx.src_lane = a.stream + wave * 1024 + lane * 16;
    cx.lds_wave = (unsigned)(size_t)lds + (unsigned)wave * 1024u;
    cx.lds_wave = __builtin_amdgcn_readfirstlane(cx.lds_wave);
    cx.ring_lane = lds + lane * 16;
    cx.prod_step = 0; cx.prod_chunk = 0; cx.prod_group = 0; cx.cons_group = 0;

    // the weight stream starts flowing before anything else
#pragma unroll 1
    for (int g = 0; g < RING_GROUPS - 1; ++g) issue_group(cx, a);
    for (int i = threadIdx.x; i < a.n_bias_floats; i += NTHREADS) sBiasAll[i] = a.bias[i];

    X4 emb;                 // xyz embedding of this lane's point: kept for the whole tile
    build_emb(emb, a, p, lane);
    const long long pc = p < a.n_points ? p : a.n_points - 1;
    const long long ray = pc / a.pts_per_ray;

    // groups 0 and 1 have landed (bias table visible after the same barrier)
    asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    __syncthreads();
    Chunk c0 = read_chunk(reinterpret_cast<const uint4*>(cx.ring_lane), 0);
    Chunk c1 = read_chunk(reinterpret_cast<const uint4*>(cx.ring_lane), RA_PF == 2 ? 1 : 0);

    // Fixed structure per trunk (reference nerf.py:162-208, D = 8): layer 0 -> A, layers 1..7 ping-pong
    // A->B / B->A (ending in B), [sigma head on B], *_final B->A, [dir A->B], main head.  Only bank A is
    // live across the loop back-edges, so the two 128-register banks never coexist with a third.
    int ci = 0;
    auto next = [&]() { return a.prog->steps[__builtin_amdgcn_readfirstlane(ci++)]; };
    auto bias_of = [&](const RAStep& st) { return sBiasAll + 4 * st.bias_off4; };
    auto floor_of = [&](const RAStep& st) { return st.relu ? 0.f : -INFINITY; };
#pragma unroll 1
    for (int tk = 0; tk < 2; ++tk) {
        if (!a.prog->trunk_on[tk]) continue;
        X4 x2;                       // second input half: time code (dynamic trunk) or unused
        auto load_x2 = [&]() {
            const float* row = a.xyz != nullptr ? a.t_emb + ray * a.in_t : a.x_emb + pc * a.ld_emb + a.off_t;
#pragma unroll
            for (int s = 0; s < 4; ++s) load_natural(x2.hi[s], x2.lo[s], tk == 1 ? row : nullptr, s, lane >> 5, a.in_t);
        };
        Acts A, B;
        {
            const RAStep st = next();
            load_x2();
            layer_x(A, emb, x2, ra_nx(st.xs), floor_of(st), bias_of(st), cx, a, c0, c1, lane);
        }
#pragma unroll 1
        for (int l = 1; l < 7; l += 2) {
            {
                const RAStep st = next();
                const int nx = ra_nx(st.xs);
                if (nx == 8) load_x2();
                layer_h(A, B, emb, x2, nx, floor_of(st), bias_of(st), cx, a, c0, c1, lane);
            }
            {
                const RAStep st = next();
                const int nx = ra_nx(st.xs);
                if (nx == 8) load_x2();
                layer_h(B, A, emb, x2, nx, floor_of(st), bias_of(st), cx, a, c0, c1, lane);
            }
        }
        {
            const RAStep st = next();
            const int nx = ra_nx(st.xs);
            if (nx == 8) load_x2();
            layer_h(A, B, emb, x2, nx, floor_of(st), bias_of(st), cx, a, c0, c1, lane);
        }
        if (a.prog->trunk_sigma_head[tk]) {
            const RAStep st = next();
            head(B, st, bias_of(st), a, p, cx, a, c0, c1, lane);
        }
        if (a.prog->trunk_full[tk]) {
            {
                const RAStep st = next();
                layer_h(B, A, emb, x2, 0, floor_of(st), bias_of(st), cx, a, c0, c1, lane);
            }
            if (a.prog->trunk_dir[tk]) {
                // static_dir_encoding: [feat | dir | a]  (nerf.py:183-185); side columns in natural order
                const float* rd = a.xyz != nullptr ? a.dir_emb + ray * a.in_dir : a.x_emb + pc * a.ld_emb + a.off_dir;
                const float* ra = a.in_a > 0 ? (a.xyz != nullptr ? a.a_emb + ray * a.in_a : a.x_emb + pc * a.ld_emb + a.off_a)
                                             : nullptr;
                X4 sa, sb;
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    float w[8];
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const int c = 16 * s + 8 * (lane >> 5) + t;
                        w[t] = c < a.in_dir ? rd[c] : (c < a.in_dir + a.in_a ? ra[c - a.in_dir] : 0.f);
                    }
                    if (s < 4) split_pack8(w, sa.hi[s], sa.lo[s]); else split_pack8(w, sb.hi[s - 4], sb.lo[s - 4]);
                }
                const RAStep st = next();
                layer_h(A, B, sa, sb, 8, floor_of(st), bias_of(st), cx, a, c0, c1, lane);
                const RAStep sh = next();
                head(B, sh, bias_of(sh), a, p, cx, a, c0, c1, lane);
            } else {
                const RAStep sh = next();
                head(A, sh, bias_of(sh), a, p, cx, a, c0, c1, lane);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // no DMA may outlive the workgroup's LDS
}

// ---------------------------------------------------------------------------------
struct RAPackSeg {
    const float* src;
    int32_t kind;        // 0 flat fp32 (bias), 1 layer, 2 head rows
    int32_t ld;
    uint32_t dst;        // layer/head: chunk index; flat: float offset in the bias table
    int32_t count;       // flat: floats
    int32_t nkh, xs;
    int32_t h_col0;      // source column of hidden column 0
    int32_t in_xyz, in_t, side_n, n_freqs;
    int32_t row0, nrows; // head
};
constexpr int PACK_BATCH = 10;
struct RAPackArgs { RAPackSeg seg[PACK_BATCH]; char* stream; float* bias; };

__global__ void nsff_pack_kernel_ra(const RAPackArgs a) {
    const RAPackSeg& s = a.seg[blockIdx.y];
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (s.kind == 0) {
        if (idx < s.count) a.bias[s.dst + idx] = s.src[idx];
        return;
    }
    const int lane = idx & 63, h = lane >> 5;
    h8 out;
    if (s.kind == 1) {
        const int nx = ra_nx(s.xs), per_pair = s.nkh + nx;
        if (idx >= 4 * per_pair * 256) return;              // 256 uint4 per chunk
        const int piece = (idx >> 6) & 3, chunk = idx >> 8;
        const int mp = chunk / per_pair, sidx = chunk % per_pair;
        const int mt2 = piece >> 1, part = piece & 1;
        const int n = 32 * (2 * mp + mt2) + (lane & 31);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            int col;
            if (sidx < s.nkh) col = s.h_col0 + ra_col_hidden(sidx, h, t);
            else {
                const int sx = sidx - s.nkh;
                if (s.xs == RA_XS_SIDE) { const int c = 16 * sx + 8 * h + t; col = c < s.side_n ? NSFF_W + c : -1; }
                else if (sx < 4) col = ra_col_emb(sx, h, t, s.n_freqs);
                else { const int c = 16 * (sx - 4) + 8 * h + t; col = c < s.in_t ? s.in_xyz + c : -1; }
            }
            const float x = col >= 0 ? s.src[(long long)n * s.ld + col] : 0.f;
            const _Float16 hi = (_Float16)x;
            out[t] = part == 0 ? hi : (_Float16)(x - (float)hi);
        }
        reinterpret_cast<h8*>(a.stream + (size_t)s.dst * RA_CHUNK_BYTES)[idx] = out;
    } else {
        if (idx >= 8 * 256) return;
        const int piece = (idx >> 6) & 3, q = idx >> 8;
        const int ksub = piece >> 1, part = piece & 1;
        const int row = (lane & 31) - s.row0;
        if (row < 0 || row >= s.nrows) return;               // other rows stay zero (memset)
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float x = s.src[(long long)row * s.ld + ra_col_hidden(2 * q + ksub, h, t)];
            const _Float16 hi = (_Float16)x;
            out[t] = part == 0 ? hi : (_Float16)(x - (float)hi);
        }
        reinterpret_cast<h8*>(a.stream + (size_t)s.dst * RA_CHUNK_BYTES)[idx] = out;
    }
}

bool build_program(const NsffModelDesc& d, const RALayout& L, int static_mode, int transient_mode, RAProgram& P);

}  // namespace

int nsff_ra_packed_bytes(const NsffModelDesc* desc, size_t* bytes) {
    RALayout L;
    const int rc = nsff_make_layout_ra(*desc, L);
    if (rc) return rc;
    *bytes = L.total_bytes;
    return NSFF_OK;
}

int nsff_ra_pack_weights(const NsffModelDesc* desc, const float* const* params, void* packed, hipStream_t st) {
    RALayout L;
    const int rc = nsff_make_layout_ra(*desc, L);
    if (rc) return rc;
    const NsffModelDesc& d = *desc;
    const int n_freqs = (d.in_xyz - 3) / 6;
    std::vector<RAPackSeg> segs;
    int pi = 0;
    auto layer = [&](const float* w, const float* b, const RALayerDesc& r, int ld, int h_col0, int in_t, int side_n) {
        segs.push_back(RAPackSeg{w, 1, ld, r.chunk0, 0, r.nkh, r.xs, h_col0, d.in_xyz, in_t, side_n, n_freqs, 0, 0});
        segs.push_back(RAPackSeg{b, 0, 0, (uint32_t)r.bias_slot * NSFF_W, NSFF_W, 0, 0, 0, 0, 0, 0, 0, 0, 0});
    };
    auto headw = [&](const float* w, uint32_t chunk0, int row0, int nrows) {
        segs.push_back(RAPackSeg{w, 2, NSFF_W, chunk0, 0, 0, 0, 0, 0, 0, 0, 0, row0, nrows});
    };
    auto headb = [&](const float* b, int head_idx, int row0, int nrows) {
        segs.push_back(RAPackSeg{b, 0, 0, L.head_bias0 + 32u * head_idx + row0, nrows, 0, 0, 0, 0, 0, 0, 0, 0, 0});
    };
    auto trunk = [&](const RATrunkLayout& T, int in_t) {
        const int in = d.in_xyz + in_t;
        for (int l = 0; l < d.D; ++l) {
            const float* w = params[pi++]; const float* b = params[pi++];
            if (l == 0) layer(w, b, T.layer[l], in, 0, in_t, 0);
            else if (l == d.skip) layer(w, b, T.layer[l], in + NSFF_W, in, in_t, 0);
            else layer(w, b, T.layer[l], NSFF_W, 0, 0, 0);
        }
        const float* w = params[pi++]; const float* b = params[pi++];
        layer(w, b, T.final_, NSFF_W, 0, 0, 0);
    };
    trunk(L.st, 0);
    if (d.use_viewdir) {
        const float* w = params[pi++]; const float* b = params[pi++];
        layer(w, b, L.dir, NSFF_W + d.in_dir + d.in_a, 0, 0, d.in_dir + d.in_a);
    }
    { const float* w = params[pi++]; const float* b = params[pi++]; headw(w, L.head_s_sigma, 0, 1); headb(b, 0, 0, 1); }
    { const float* w = params[pi++]; const float* b = params[pi++]; headw(w, L.head_s_rgb, 0, 3); headb(b, 1, 0, 3); }
    if (d.has_transient) {
        trunk(L.tr, d.in_t);
        const float* ws = params[pi++]; const float* bs = params[pi++];
        const float* wc = params[pi++]; const float* bc = params[pi++];
        headw(wc, L.head_t, 0, 3); headb(bc, 2, 0, 3);
        headw(ws, L.head_t, 3, 1); headb(bs, 2, 3, 1);
        if (d.has_flow) {
            const float* wf = params[pi++]; const float* bf = params[pi++];
            const float* wb = params[pi++]; const float* bb = params[pi++];
            headw(wf, L.head_t, 4, 3); headb(bf, 2, 4, 3);
            headw(wb, L.head_t, 7, 3); headb(bb, 2, 7, 3);
        }
    }
    for (int i = 0; i < pi; ++i) if (!params[i]) return NSFF_ERR_NULL;
    if (d.D != 8) return NSFF_ERR_INVALID;
    hipError_t e = hipMemsetAsync(packed, 0, L.total_bytes, st);
    if (e != hipSuccess) return nsff_hip_fail(e);
    {
        std::vector<RAProgram> progs(9);
        for (int sm = 0; sm < 3; ++sm)
            for (int tm = 0; tm < 3; ++tm)
                if (!build_program(d, L, sm, tm, progs[sm * 3 + tm])) return NSFF_ERR_INVALID;
        e = hipMemcpyAsync(reinterpret_cast<char*>(packed) + L.prog_offset_bytes, progs.data(),
                           progs.size() * sizeof(RAProgram), hipMemcpyHostToDevice, st);
        if (e != hipSuccess) return nsff_hip_fail(e);
        e = hipStreamSynchronize(st);          // the staging vector dies at the end of this scope
        if (e != hipSuccess) return nsff_hip_fail(e);
    }
    for (size_t base = 0; base < segs.size(); base += PACK_BATCH) {
        RAPackArgs pa{};
        pa.stream = reinterpret_cast<char*>(packed);
        pa.bias = reinterpret_cast<float*>(reinterpret_cast<char*>(packed) + L.bias_offset_bytes);
        const int n = (int)std::min<size_t>(PACK_BATCH, segs.size() - base);
        int max_threads = 0;
        for (int i = 0; i < n; ++i) {
            pa.seg[i] = segs[base + i];
            const RAPackSeg& s = pa.seg[i];
            const int thr = s.kind == 0 ? s.count : (s.kind == 1 ? 4 * (s.nkh + ra_nx(s.xs)) * 256 : 8 * 256);
            max_threads = std::max(max_threads, thr);
        }
        hipLaunchKernelGGL(nsff_pack_kernel_ra, dim3((max_threads + 255) / 256, n), dim3(256), 0, st, pa);
    }
    return nsff_launch_status();
}

namespace {
// program of one mode; returns false if it does not fit
bool build_program(const NsffModelDesc& d, const RALayout& L, int static_mode, int transient_mode, RAProgram& P) {
    P = RAProgram{};
    int n = 0;
    auto push_layer = [&](const RALayerDesc& r, int relu) {
        if (n >= MAX_STEPS) { ++n; return; }
        RAStep& s = P.steps[n++];
        s.chunk0 = r.chunk0; s.bias_off4 = (uint16_t)(r.bias_slot * (NSFF_W / 4)); s.kind = STEP_LAYER;
        s.nkh = r.nkh; s.xs = r.xs; s.relu = (uint8_t)relu;
    };
    auto push_head = [&](uint32_t chunk0, int head_idx, int n_rows, int slot0, unsigned kinds) {
        if (n >= MAX_STEPS) { ++n; return; }
        RAStep& s = P.steps[n++];
        s.chunk0 = chunk0; s.bias_off4 = (uint16_t)((L.head_bias0 + 32 * head_idx) / 4); s.kind = STEP_HEAD;
        s.n_rows = (uint8_t)n_rows; s.slot0 = (uint8_t)slot0; s.kinds = kinds;
    };
    if (static_mode) {
        for (int l = 0; l < d.D; ++l) push_layer(L.st.layer[l], 1);
        push_head(L.head_s_sigma, 0, 1, 3, ACT_NONE);                      // before *_final (nerf.py:169)
        if (static_mode == 2) {
            push_layer(L.st.final_, 0);
            if (d.use_viewdir) push_layer(L.dir, 1);
            push_head(L.head_s_rgb, 1, 3, 0, 0x15u);
        }
    }
    if (transient_mode && d.has_transient) {
        for (int l = 0; l < d.D; ++l) push_layer(L.tr.layer[l], 1);
        push_layer(L.tr.final_, 0);
        push_head(L.head_t, 2, (int)L.t_head_rows, 4, 0x15u | (0xAAAu << 8));
    }
    if (n > MAX_STEPS) return false;
    P.n_steps = n;
    P.trunk_on[0] = static_mode != 0; P.trunk_on[1] = transient_mode != 0 && d.has_transient;
    P.trunk_sigma_head[0] = 1; P.trunk_sigma_head[1] = 0;
    P.trunk_full[0] = static_mode == 2; P.trunk_full[1] = 1;
    P.trunk_dir[0] = d.use_viewdir; P.trunk_dir[1] = 0;
    return true;
}
}  // namespace

int nsff_ra_field_query(const NsffModelDesc* desc, const void* packed, const NsffFieldArgs* args, hipStream_t stq) {
    const NsffModelDesc& d = *desc;
    const NsffFieldArgs& g = *args;
    RALayout L;
    const int rc = nsff_make_layout_ra(d, L);
    if (rc) return rc;
    if (L.n_bias_floats > MAX_BIAS_FLOATS) return NSFF_ERR_INVALID;
    if (d.D != 8) return NSFF_ERR_INVALID;          // the register ping-pong is laid out for the reference depth
    if (g.xyz && 3 + 6 * g.n_freqs != d.in_xyz) return NSFF_ERR_INVALID;
    RAKArgs k{};
    k.prog = reinterpret_cast<const RAProgram*>(reinterpret_cast<const char*>(packed) + L.prog_offset_bytes) +
             (g.static_mode * 3 + g.transient_mode);
    k.stream = reinterpret_cast<const char*>(packed);
    k.bias = reinterpret_cast<const float*>(k.stream + L.bias_offset_bytes);
    k.n_bias_floats = (int)L.n_bias_floats;
    k.xyz = g.xyz; k.x_emb = g.x_emb; k.dir_emb = g.dir_emb; k.a_emb = g.a_emb; k.t_emb = g.t_emb;
    k.raw = g.raw; k.n_points = g.n_points; k.pts_per_ray = g.pts_per_ray;
    k.in_xyz = d.in_xyz; k.in_dir = d.in_dir; k.in_a = d.in_a; k.in_t = d.in_t;
    k.flow_scale = d.flow_scale; k.n_freqs = g.n_freqs;
    for (int i = 0; i < NSFF_MAX_FREQS; ++i) k.freqs[i] = g.freqs[i];
    k.ld_emb = g.ld_emb; k.off_xyz = g.off_xyz; k.off_dir = g.off_dir; k.off_a = g.off_a; k.off_t = g.off_t;
    const long long tiles = (g.n_points + 127) / 128;
    if (tiles > 0x7fffffffLL) return NSFF_ERR_INVALID;
    hipLaunchKernelGGL(nsff_field_kernel_ra, dim3((unsigned)tiles), dim3(NTHREADS), 0, stq, k);
    return nsff_launch_status();
}
