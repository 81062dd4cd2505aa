// Packed-weight layout of the fp16-split ("f16x3") field kernel.
//
// Every fp32 weight w is stored as two halfs  hi = f16(w), lo = f16(w - hi)  (22 significant
// bits; gfx950's f16 MFMA takes subnormal inputs exactly, so no pre-scaling is needed).
// A Linear segment W (256, K), K padded to a multiple of 64, is stored as A-operand tiles of
// v_mfma_f32_32x32x16_f16 in the order one wave streams them:
//
//     seg[wave 0..3][ks 0..K/16-1][mt 0..1][part hi,lo][lane 0..63][8 halfs]      (16 B per lane)
//     value = part(W[64*wave + 32*mt + (lane&31)][col(16*ks + 8*(lane>>5) + t)]),  t = 0..7
//
// i.e. 4 KiB contiguous per (wave, k-step).  Narrow heads are one zero-padded 32-row tile
//     head[ks][part][lane][8 halfs],  row = lane&31.
// Offsets are in 4-byte words from the start of the buffer; biases stay fp32.
#pragma once
#include <stdint.h>
#include "../../include/nsff_render.h"

#define NSFF_W 256
#define NSFF_NONE 0xFFFFFFFFu
#define NSFF_H3_HEAD_WORDS (32 * NSFF_W)      /* 32 rows x 256 k x (hi+lo) halfs = 32 KiB */

struct NsffTrunkLayoutH3 {
    uint32_t k0;
    uint32_t seg_x[NSFF_MAX_LAYERS];
    uint32_t seg_h[NSFF_MAX_LAYERS];
    uint32_t bias[NSFF_MAX_LAYERS];
    uint32_t final_w, final_b;
};

struct NsffLayoutH3 {
    NsffTrunkLayoutH3 st, tr;
    uint32_t k0s, kt, side_k;          // ceil64 of in_xyz, in_t, in_dir+in_a
    uint32_t dir_h, dir_x, dir_b;
    uint32_t s_sigma_w, s_sigma_b;     // head tile + 32 fp32 biases
    uint32_t s_rgb_w, s_rgb_b;
    uint32_t t_head_w, t_head_b;       // rows: rgb(3) sigma(1) [fw(3) bw(3)]
    uint32_t t_head_rows;
    // Inference-only folded heads: *_xyz_encoding_final is a Linear WITHOUT activation (nerf.py:170,195), so the heads that
    // read it are linear maps of the last trunk activation h:  W_head (W_final h + b_final) + b_head = (W_head W_final) h +
    // (W_head b_final + b_head).  One 32-row tile per trunk holds the products (static: rgb rows 0-2 folded + sigma row 3,
    // which reads h anyway; dynamic: all t_head_rows folded); the 256x256 *_final layer is then never executed.
    uint32_t s_fold_w, s_fold_b, t_fold_w, t_fold_b;
    uint32_t fold_f32;                 // scratch: 2 x (32 x 256) fp32 products the tiles are packed from
    // view-direction models: static_dir_encoding reads [*_final | dir | a]; its *_final part folds the same way,
    // (W_dir[:, :256] W_final) h + (W_dir[:, :256] b_final + b_dir), a full 256 x 256 segment
    uint32_t dir_h_fold, dir_b_fold, dir_fold_f32;
    // fp32 copy of static_sigma's weight row (256): the hand-scheduled kernel's sigma ride (a view-direction static trunk
    // evaluates sigma in the epilogue of its last trunk layer, one layer before the trunk's end) reads it as a bias-table row
    uint32_t s_sigma_f32;
    uint32_t total;                    // words
};

// every K-segment is zero-padded to a multiple of 64 columns (= 4 k-steps of the weight ring)
static inline uint32_t nsff_ceil64(uint32_t v) { return (v + 63u) & ~63u; }

static inline int nsff_make_layout_h3(const NsffModelDesc& d, NsffLayoutH3& L) {
    if (d.W != NSFF_W || d.D < 2 || d.D > NSFF_MAX_LAYERS) return NSFF_ERR_INVALID;
    const uint32_t skips = nsff_skip_layers(&d);
    if (skips & ~(((1u << d.D) - 1u) & ~1u)) return NSFF_ERR_INVALID;       // skip layers are among 1..D-1
    if (d.in_xyz < 1 || d.in_xyz > 192) return NSFF_ERR_INVALID;
    if (d.in_t < 0 || d.in_a < 0 || d.in_dir < 0) return NSFF_ERR_INVALID;
    if (d.has_transient && d.in_t < 1) return NSFF_ERR_INVALID;
    if (d.has_flow && !d.has_transient) return NSFF_ERR_INVALID;
    uint32_t off = 0;
    auto take = [&](uint32_t n) { uint32_t o = off; off += (n + 3u) & ~3u; return o; };
    L.k0s = nsff_ceil64((uint32_t)d.in_xyz);
    L.kt = d.has_transient ? nsff_ceil64((uint32_t)d.in_t) : 0;
    if (L.k0s + L.kt > NSFF_W) return NSFF_ERR_INVALID;
    L.side_k = d.use_viewdir ? nsff_ceil64((uint32_t)(d.in_dir + d.in_a)) : 0;
    if (L.side_k > NSFF_W) return NSFF_ERR_INVALID;
    auto trunk = [&](NsffTrunkLayoutH3& T, uint32_t k0) {
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
    L.s_sigma_w = take(NSFF_H3_HEAD_WORDS); L.s_sigma_b = take(32);
    L.s_rgb_w = take(NSFF_H3_HEAD_WORDS); L.s_rgb_b = take(32);
    L.t_head_rows = 0; L.t_head_w = L.t_head_b = NSFF_NONE;
    if (d.has_transient) {
        trunk(L.tr, L.k0s + L.kt);
        L.t_head_rows = d.has_flow ? 10 : 4;
        L.t_head_w = take(NSFF_H3_HEAD_WORDS);
        L.t_head_b = take(32);
    } else {
        L.tr = NsffTrunkLayoutH3{};
    }
    L.s_fold_w = take(NSFF_H3_HEAD_WORDS); L.s_fold_b = take(32);
    L.t_fold_w = take(NSFF_H3_HEAD_WORDS); L.t_fold_b = take(32);
    L.fold_f32 = take(2 * 32 * NSFF_W);
    L.dir_h_fold = L.dir_b_fold = L.dir_fold_f32 = NSFF_NONE;
    if (d.use_viewdir) {
        L.dir_h_fold = take(NSFF_W * NSFF_W);
        L.dir_b_fold = take(NSFF_W);
        L.dir_fold_f32 = take(NSFF_W * NSFF_W);
    }
    L.s_sigma_f32 = take(NSFF_W);
    L.total = off;
    return NSFF_OK;
}
