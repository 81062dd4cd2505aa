// Weight stream of the register-resident ("RA") fp16-split field kernel.
//
// In this kernel a wave owns 32 points and ALL 256 neurons of a layer; activations stay in
// VGPRs from the input encoding to the heads and only the weights move: they are streamed
// global -> LDS by direct-to-LDS DMA in 4 KiB chunks through a ring shared by the four waves
// of a workgroup (128 points per weight byte fetched from L2).
//
//   layer chunk (pair mp, k-step s):  [mt2 0..1][part hi,lo][lane 0..63][8 halfs]
//        value = part( W[32*(2*mp+mt2) + (lane&31)][ col(s, lane>>5, t) ] ),  t = 0..7
//   stream of a layer: for mp 0..3: hidden-segment k-steps 0..15 (if any), then input-segment
//        k-steps 0..nx-1  -> 4*(nkh+nx) chunks; a pair's two 32-neuron accumulators finish together.
//   head chunk q (0..7):  [ksub 0..1][part][lane][8 halfs], k-step s = 2q+ksub, row = lane&31.
//
// Column maps (which input column a k-step element contracts) are chosen so that the MFMA
// accumulator layout of layer l IS the B-operand layout of layer l+1 (no data movement):
//   hidden:  col(s,h,t) = 32*(s>>1) + (r&3) + 8*(r>>2) + 4*h,   r = 8*(s&1) + t
//   xyz embedding (4 k-steps, q = 8s+t): lane-half h holds frequencies f = 2*(q/6)+h (q < 30,
//        e = q%6: sin xyz, cos xyz -> source column 3+6f+e) and raw x,y (h=0) / z,0 (h=1) at q = 30,31,
//        so every lane evaluates whole sin/cos pairs;
//   time code / view-dir side input: natural order col = 16s + 8h + t.
// Biases stay fp32 (copied to LDS once per workgroup).  Offsets are in 4 KiB chunks / floats.
#pragma once
#include <stdint.h>
#include "../../include/nsff_render.h"

#define NSFF_W 256
#define NSFF_NONE 0xFFFFFFFFu
#define RA_CHUNK_BYTES 4096

enum { RA_XS_NONE = 0, RA_XS_EMB = 1, RA_XS_EMB_T = 2, RA_XS_SIDE = 3 };

struct RALayerDesc {          // one Linear layer in the stream
    uint32_t chunk0;          // first chunk
    uint16_t bias_slot;       // index into the bias table (256 floats each)
    uint8_t nkh;              // hidden-segment k-steps: 0 or 16
    uint8_t xs;               // RA_XS_*
};

struct RATrunkLayout {
    RALayerDesc layer[NSFF_MAX_LAYERS];
    RALayerDesc final_;
};

struct RALayout {
    RATrunkLayout st, tr;
    RALayerDesc dir;                       // static_dir_encoding (use_viewdir)
    uint32_t head_s_sigma, head_s_rgb, head_t;   // chunk0 of the 8-chunk head tiles
    uint32_t head_bias0;                   // float offset of 3 x 32 head biases in the bias table
    uint32_t t_head_rows;
    uint32_t n_chunks;                     // stream length in chunks
    uint32_t n_bias_floats;                // bias table length (floats)
    uint32_t bias_offset_bytes;            // byte offset of the bias table inside the packed buffer
    uint32_t prog_offset_bytes;            // nine step programs (field_ra.hip RAProgram), 8 KiB reserved
    uint32_t total_bytes;
};

#if defined(__HIPCC__) || defined(__HIP__)
#define RA_HD __host__ __device__
#else
#define RA_HD
#endif

RA_HD static inline int ra_nx(int xs) { return xs == RA_XS_NONE ? 0 : (xs == RA_XS_EMB ? 4 : 8); }
static inline uint32_t ra_layer_chunks(const RALayerDesc& d) { return 4u * (d.nkh + ra_nx(d.xs)); }

RA_HD static inline int ra_col_hidden(int s, int h, int t) {
    const int r = 8 * (s & 1) + t;
    return 32 * (s >> 1) + (r & 3) + 8 * (r >> 2) + 4 * h;
}
// source column of the xyz embedding (or -1 = zero) for k-step s (0..3), lane-half h, element t
RA_HD static inline int ra_col_emb(int s, int h, int t, int n_freqs) {
    const int q = 8 * s + t;
    if (q < 30) {
        const int f = 2 * (q / 6) + h;
        return f < n_freqs ? 3 + 6 * f + (q % 6) : -1;
    }
    if (q == 30) return h == 0 ? 0 : 2;
    return h == 0 ? 1 : -1;
}

static inline int nsff_make_layout_ra(const NsffModelDesc& d, RALayout& L) {
    if (d.W != NSFF_W || d.D < 2 || d.D > NSFF_MAX_LAYERS) return NSFF_ERR_INVALID;
    if (d.skip < 1 || d.skip >= d.D) return NSFF_ERR_INVALID;
    if (d.in_xyz < 3 || d.in_xyz > 63 || (d.in_xyz - 3) % 6 != 0) return NSFF_ERR_INVALID;
    if (d.in_t < 0 || d.in_t > 64 || d.in_a < 0 || d.in_dir < 0) return NSFF_ERR_INVALID;
    if (d.use_viewdir && d.in_dir + d.in_a > 128) return NSFF_ERR_INVALID;
    if (d.has_transient && d.in_t < 1) return NSFF_ERR_INVALID;
    if (d.has_flow && !d.has_transient) return NSFF_ERR_INVALID;
    uint32_t chunk = 0, slot = 0;
    auto layer = [&](RALayerDesc& r, int nkh, int xs) {
        r.chunk0 = chunk; r.bias_slot = (uint16_t)slot++; r.nkh = (uint8_t)nkh; r.xs = (uint8_t)xs;
        chunk += ra_layer_chunks(r);
    };
    auto trunk = [&](RATrunkLayout& T, int xs) {
        for (int l = 0; l < NSFF_MAX_LAYERS; ++l) T.layer[l] = RALayerDesc{NSFF_NONE, 0, 0, 0};
        for (int l = 0; l < d.D; ++l) layer(T.layer[l], l == 0 ? 0 : 16, (l == 0 || l == d.skip) ? xs : RA_XS_NONE);
        layer(T.final_, 16, RA_XS_NONE);
    };
    trunk(L.st, RA_XS_EMB);
    L.dir = RALayerDesc{NSFF_NONE, 0, 0, 0};
    if (d.use_viewdir) layer(L.dir, 16, RA_XS_SIDE);
    L.head_s_sigma = chunk; chunk += 8;
    L.head_s_rgb = chunk; chunk += 8;
    L.head_t = NSFF_NONE; L.t_head_rows = 0;
    if (d.has_transient) {
        trunk(L.tr, RA_XS_EMB_T);
        L.head_t = chunk; chunk += 8;
        L.t_head_rows = d.has_flow ? 10 : 4;
    } else {
        L.tr = RATrunkLayout{};
    }
    L.n_chunks = chunk;
    L.head_bias0 = slot * NSFF_W;
    L.n_bias_floats = slot * NSFF_W + 3 * 32;
    L.bias_offset_bytes = chunk * RA_CHUNK_BYTES;
    L.prog_offset_bytes = L.bias_offset_bytes + ((L.n_bias_floats * 4 + 4095u) & ~4095u);
    L.total_bytes = L.prog_offset_bytes + 8192u + 16384u;
    return NSFF_OK;
}
