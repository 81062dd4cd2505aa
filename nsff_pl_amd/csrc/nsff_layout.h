// Packed-weight layout shared by the pack kernel, the field kernel and the host API.
//
// A Linear layer y = x W^T + b with W (256, K) is cut into K-segments (the plain
// hidden input, or the re-read network input of the first / skip / view-dir layer).
// Every segment is stored as MFMA B-operand tiles for v_mfma_f32_32x32x2_f32:
//
//     seg[ntile 0..7][kb 0..Kpad/8-1][lane 0..63][4]   (fp32)
//     value = W[ntile*32 + (lane&31)][col(kb*8 + (lane>>5)*4 + t)],  t = 0..3
//
// so that one wave reads its B fragments for four consecutive k-steps with a single
// coalesced 1 KiB global_load_dwordx4, and k-step t of block kb contracts input
// columns {kb*8+t, kb*8+4+t}.  col() maps padded segment columns to source columns
// (zero for padding).  Offsets below are in floats from the start of the buffer.
#pragma once
#include <stdint.h>
#include "../../include/nsff_render.h"

#define NSFF_W 256
#define NSFF_NONE 0xFFFFFFFFu

struct NsffTrunkLayout {
    uint32_t k0;                       // padded width of the network-input segment
    uint32_t seg_x[NSFF_MAX_LAYERS];   // input segment (layer 0 and the skip layer)
    uint32_t seg_h[NSFF_MAX_LAYERS];   // hidden segment (layers >= 1)
    uint32_t bias[NSFF_MAX_LAYERS];
    uint32_t final_w, final_b;         // *_xyz_encoding_final
};

struct NsffLayout {
    NsffTrunkLayout st, tr;
    uint32_t k0s;          // ceil8(in_xyz): xyz-embedding columns [0,in_xyz), zero pad
    uint32_t kt;           // ceil8(in_t):   transient code columns [k0s, k0s+in_t)
    uint32_t side_k;       // ceil8(in_dir+in_a) or 0 (static_dir_encoding side input)
    uint32_t dir_h, dir_x, dir_b;
    uint32_t s_sigma_w, s_sigma_b;     // (1,256) + 1
    uint32_t s_rgb_w, s_rgb_b;         // (3,256) + 3
    uint32_t t_head_w, t_head_b;       // rows: rgb(3) sigma(1) [fw(3) bw(3)]
    uint32_t t_head_rows;              // 0, 4 or 10
    // folded heads (see nsff_layout_h3.h): rows pre-multiplied with the activation-free *_final layers, applied to the
    // last trunk activation -- static (no view directions): rgb(3) folded + sigma(1); dynamic: all t_head_rows
    uint32_t s_fold_w, s_fold_b, t_fold_w, t_fold_b;
    uint32_t total;                    // floats
};

static inline uint32_t nsff_ceil8(uint32_t v) { return (v + 7u) & ~7u; }

// Returns 0 on success.  Host only.
static inline int nsff_make_layout(const NsffModelDesc& d, NsffLayout& L) {
    if (d.W != NSFF_W || d.D < 2 || d.D > NSFF_MAX_LAYERS) return NSFF_ERR_INVALID;
    const uint32_t skips = nsff_skip_layers(&d);
    if (skips & ~(((1u << d.D) - 1u) & ~1u)) return NSFF_ERR_INVALID;       // skip layers are among 1..D-1
    if (d.in_xyz < 1 || d.in_xyz > 248) return NSFF_ERR_INVALID;
    if (d.in_t < 0 || d.in_a < 0 || d.in_dir < 0) return NSFF_ERR_INVALID;
    if (d.has_transient && d.in_t < 1) return NSFF_ERR_INVALID;
    if (d.has_flow && !d.has_transient) return NSFF_ERR_INVALID;
    uint32_t off = 0;
    auto take = [&](uint32_t n) { uint32_t o = off; off += (n + 3u) & ~3u; return o; };
    L.k0s = nsff_ceil8((uint32_t)d.in_xyz);
    L.kt = d.has_transient ? nsff_ceil8((uint32_t)d.in_t) : 0;
    if (L.k0s + L.kt > NSFF_W) return NSFF_ERR_INVALID;
    L.side_k = d.use_viewdir ? nsff_ceil8((uint32_t)(d.in_dir + d.in_a)) : 0;
    if (L.side_k > NSFF_W) return NSFF_ERR_INVALID;
    auto trunk = [&](NsffTrunkLayout& T, uint32_t k0) {
        T.k0 = k0;
        for (int l = 0; l < NSFF_MAX_LAYERS; ++l) T.seg_x[l] = T.seg_h[l] = T.bias[l] = NSFF_NONE;
        for (int l = 0; l < d.D; ++l) {
            if (l == 0 || ((skips >> l) & 1u)) T.seg_x[l] = take(NSFF_W * k0);
            if (l > 0) T.seg_h[l] = take(NSFF_W * NSFF_W);
            T.bias[l] = take(NSFF_W);
        }
        T.final_w = take(NSFF_W * NSFF_W);
        T.final_b = take(NSFF_W);
    };
    trunk(L.st, L.k0s);
    L.dir_h = L.dir_x = L.dir_b = NSFF_NONE;
    if (d.use_viewdir) {
        L.dir_h = take(NSFF_W * NSFF_W);
        L.dir_x = take(NSFF_W * L.side_k);
        L.dir_b = take(NSFF_W);
    }
    L.s_sigma_w = take(NSFF_W); L.s_sigma_b = take(1);
    L.s_rgb_w = take(3 * NSFF_W); L.s_rgb_b = take(3);
    L.t_head_rows = 0; L.t_head_w = L.t_head_b = NSFF_NONE;
    if (d.has_transient) {
        trunk(L.tr, L.k0s + L.kt);
        L.t_head_rows = d.has_flow ? 10 : 4;
        L.t_head_w = take(L.t_head_rows * NSFF_W);
        L.t_head_b = take(L.t_head_rows);
    } else {
        L.tr = NsffTrunkLayout{};
    }
    L.s_fold_w = take(4 * NSFF_W); L.s_fold_b = take(4);
    L.t_fold_w = L.t_fold_b = NSFF_NONE;
    if (d.has_transient) { L.t_fold_w = take(L.t_head_rows * NSFF_W); L.t_fold_b = take(L.t_head_rows); }
    L.total = off;
    return NSFF_OK;
}
