/*
 * nsff_render.h -- C-ABI of the MI355X (gfx950) NSFF ray renderer.
 *
 * Drop-in boundary for ONE hot path of kwea123/nsff_pl: models/rendering.py
 * (render_rays, inference, render_transient_warping, sample_pdf) and
 * models/nerf.py (PosEmbedding, NeRF).  The reference has no FFI of its own
 * (pure Python on torch); these entry points are what a maintainer would bind
 * with ctypes to replace the torch ops of that path.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 (int32 where said);
 *     the library never allocates, frees or copies user-visible memory;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it and
 *     the call returns without synchronising;
 *   - return value: 0 = ok, <0 = NSFF_ERR_* (no exceptions cross the ABI);
 *   - not thread-safe per stream; re-entrant across streams (no global state).
 */
#ifndef NSFF_RENDER_H
#define NSFF_RENDER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NSFF_OK               0
#define NSFF_ERR_INVALID     -1   /* bad shape / flag / unsupported architecture */
#define NSFF_ERR_NULL        -2   /* required pointer missing                      */
#define NSFF_ERR_ALIGN       -3   /* pointer not 16-byte aligned where required    */
#define NSFF_ERR_HIP         -4   /* a HIP runtime call failed (see nsff_last_hip_error) */

#define NSFF_ABI_VERSION      32
#define NSFF_RAW_STRIDE      16   /* floats per point in a raw field record        */
#define NSFF_MAX_FREQS       24
#define NSFF_MAX_LAYERS       8

/* Raw field record (what NeRF.forward returns per point, reference nerf.py:187-213):
 *   [0..2] static rgb (sigmoid)  [3] static sigma (raw)
 *   [4..6] transient rgb         [7] transient sigma (raw)
 *   [8..10] flow_fw = flow_scale*tanh(.)   [11..13] flow_bw   [14..15] unused
 * nsff_field_query writes whole 64-byte records: slots a launch does not evaluate are written as 0. */

/* ---- model description: mirrors NeRF.__init__ (reference nerf.py:34-40) ---- */
typedef struct NsffModelDesc {
    int32_t D;              /* trunk depth, 2..8 (reference: 8)                     */
    int32_t W;              /* width, must be 256                                  */
    int32_t skip;           /* the layer (0-based, 1..D-1) that re-reads the input (reference: 4); 0 = none or see skip_mask */
    int32_t in_xyz;         /* 63                                                  */
    int32_t in_dir;         /* 27                                                  */
    int32_t in_a;           /* appearance code width or 0                          */
    int32_t in_t;           /* transient code width or 0                           */
    int32_t use_viewdir;    /* static_dir_encoding present                         */
    int32_t has_transient;  /* encode_transient                                    */
    int32_t has_flow;       /* transient_flow_fw / _bw heads present               */
    float   flow_scale;     /* 0.2                                                 */
    int32_t skip_mask;      /* several skip layers (the reference's `skips` list, nerf.py:34-40,163-167): bit l set = layer l
                               reads [input | previous layer]; 0 = just `skip`.  Any subset of layers 1..D-1, in the
                               inference kernels and in the training kernels alike                         */
} NsffModelDesc;

/* effective set of skip layers of a description */
static inline uint32_t nsff_skip_layers(const NsffModelDesc* d) {
    return d->skip_mask ? (uint32_t)d->skip_mask : (d->skip > 0 ? 1u << d->skip : 0u);
}

/* Arithmetic of the field kernel's dense layers:
 *   NSFF_PREC_F32   exact fp32 MFMA (v_mfma_f32_32x32x2_f32)
 *   NSFF_PREC_F16X3 fp32 operands split into two halfs, three f16 MFMAs per product,
 *                   fp32 accumulate (v_mfma_f32_32x32x16_f16), activations staged in LDS;
 *                   agrees with F32 to fp32 rounding level (same 1e-4 parity tests)
 * (Value 3 was the single-product fast mode of earlier ABI versions: not parity-grade, removed -- NSFF_ERR_INVALID now.)
 */
#define NSFF_PREC_F32        0
#define NSFF_PREC_F16X3      1

/* Size in bytes of the packed-weight buffer for `desc` at `precision`. */
int nsff_packed_bytes(const NsffModelDesc* desc, int precision, size_t* bytes);

/* Number of parameter tensors nsff_pack_weights expects, in this order
 * (PyTorch Linear layout, weight (out,in) row-major then bias):
 *   static_xyz_encoding_{1..D}.0, static_xyz_encoding_final,
 *   [static_dir_encoding.0], static_sigma, static_rgb.0,
 *   [transient_xyz_encoding_{1..D}.0, transient_xyz_encoding_final,
 *    transient_sigma, transient_rgb.0, [transient_flow_fw.0, transient_flow_bw.0]] */
int nsff_param_count(const NsffModelDesc* desc);

/* Repack the module parameters into MFMA B-operand tiles (DESIGN.md "weight pack").
 * `params`: HOST array of nsff_param_count() device pointers. */
int nsff_pack_weights(const NsffModelDesc* desc, int precision, const float* const* params,
                      void* packed, void* stream);

/* Finer-grained form for callers that re-pack after every optimizer step.  Inference launches (every precision; the
 * exact-fp32 kernel folds everything but the static trunk of view-direction models) read "folded" head rows -- the heads that consume the activation-free *_xyz_encoding_final layers (nerf.py:170,195),
 * pre-multiplied with them, so that those two 256x256 layers are never executed.  nsff_pack_weights builds them;
 * nsff_pack_weights_ex(..., NSFF_PACK_SKIP_FOLD, ...) does not (enough for training forwards, i.e. nsff_field_query
 * with save_* buffers, which execute the layers because the backward pass needs their output), and nsff_fold_heads
 * adds them to such a buffer later.  An inference launch on a buffer without them computes garbage heads.          */
#define NSFF_PACK_SKIP_FOLD 1
int nsff_pack_weights_ex(const NsffModelDesc* desc, int precision, const float* const* params,
                         void* packed, int32_t flags, void* stream);
int nsff_fold_heads(const NsffModelDesc* desc, int precision, const float* const* params, void* packed, void* stream);

/* ---- a1: PosEmbedding.forward (reference nerf.py:17-30) ---- */
int nsff_posenc(const float* x, int64_t n_rows, const float* freqs_host, int n_freqs,
                float* out, void* stream);

/* ---- a2: the two neighbouring-frame gathers of the warp path, embedding_t(clamp(ts + 1, max=max_t)) and
 * embedding_t(clamp(ts - 1, min=0)) (reference rendering.py:218,224), from the table of an nn.Embedding:
 * table (n_table, width) fp32, ts (n) int64 on the device; next / prev (n, width), either may be NULL. */
int nsff_time_rows(const float* table, int64_t n_table, int32_t width, const int64_t* ts, int64_t n, int64_t max_t,
                   float* next, float* prev, void* stream);

/* N1: gradient of those gathers and of embedding_t(ts) itself (rendering.py:162) w.r.t. the table: d_table (n_table, width),
 * every row written, = the rows of g_cur / g_next / g_prev (each (n, width) or NULL) summed by their (clamped) frame index.
 * Replaces three embedding backwards (torch: atomics on ~30 distinct rows) with one deterministic launch. */
int nsff_time_rows_backward(const float* g_cur, const float* g_next, const float* g_prev, const int64_t* ts, int64_t n,
                            int64_t max_t, int64_t n_table, int32_t width, float* d_table, void* stream);

/* ---- a3/a5: fused field query = encode -> trunk(s) -> heads (NeRF.forward) ---- */
typedef struct NsffFieldArgs {
    int64_t n_points;        /* P                                                   */
    int32_t precision;       /* NSFF_PREC_*: must match the packed buffer           */
    int32_t tile_points;     /* F16X3 only: 0 = library default (128-point tiles; 64 points for inference launches below
                                32768 points), 64 = 64-point tiles, 130 = 128-point tiles: the hand-scheduled body
                                (nsff_field_kernel_h3a: one wave per SIMD, resident weights, two 64-point halves) for
                                inference launches it covers, else eight waves of 32 neurons; 131 = 128-point tiles,
                                always the compiler-scheduled eight-wave kernel                                       */
    int32_t pts_per_ray;     /* ray index of point p is p / pts_per_ray             */
    int32_t static_mode;     /* 0 skip, 1 sigma only, 2 rgb+sigma                   */
    int32_t transient_mode;  /* 0 skip, 1 sigma only, 2 rgb+sigma(+flow heads)      */
    int32_t flow_heads;      /* flow heads the caller consumes (0,1,2); the kernel always
                                evaluates every head the model has, this only feeds the
                                algorithmic-FLOP accounting of nsff_prof_collect          */
    /* input A: raw positions, encoded in-kernel (render path) */
    const float* xyz;        /* (P,3) or NULL                                       */
    int32_t n_freqs;         /* of the xyz embedding                                */
    float   freqs[NSFF_MAX_FREQS];
    const float* dir_emb;    /* (n_rays, in_dir) per-ray, needed iff use_viewdir    */
    const float* a_emb;      /* (n_rays, in_a)   per-ray, needed iff in_a>0 && use_viewdir */
    const float* t_emb;      /* (n_rays, in_t)   per-ray, needed iff transient_mode */
    /* input B: rows already embedded by the caller (NeRF.forward API) */
    const float* x_emb;      /* (P, ld_emb) or NULL                                 */
    int32_t ld_emb;
    int32_t off_xyz, off_dir, off_a, off_t;  /* column offsets in x_emb, -1 = absent */
    float*  raw;             /* (P, NSFF_RAW_STRIDE) output                         */
    /* training forward (F16X3, input A): also keep what nsff_field_backward / nsff_weight_grad need.  T = ceil(P/64) point
     * tiles; any of them may be NULL.  NS = 2*D+2 activation slots.  XR / t_row0 / SR: nsff_train_dims.
     *   save_acts : fp16 (NS, T, 4, 256, 16): slot l = ReLU output of static layer l (l < D), slots D+1 .. 2*D the same for
     *               the transient trunk, slot D = static_dir_encoding output (use_viewdir; ABI 30 -- it was slot 2*D+2); slot 2*D+1
     *               (and slot D without view directions: the *_xyz_encoding_final outputs of earlier ABI versions) is never
     *               written: *_final is folded into the heads in training as in inference (slots of a trunk that is not
     *               evaluated stay unwritten too).  Inside a tile: [16-point
     *               group][neuron][point] = the fragment order of the weight-gradient GEMM (K = points);
     *   save_xin  : fp16 (T, 4, XR, 16) trunk input, rows [0,in_xyz) xyz embedding, rows [t_row0, t_row0+in_t) time code
     *               (t_row0 = ceil64(in_xyz); XR = 128 or 256); rows from ceil64 of what the launch encodes up stay unwritten;
     *   save_masks: uint64 (NS, T, 256) ReLU sign bits of every trunk activation, in accumulator order;
     *   save_side : fp16 (T, 4, SR, 16) [dir | a] input of static_dir_encoding (use_viewdir, static_mode 2), rows
     *               [0, in_dir + in_a); rows from ceil64(in_dir + in_a) up stay unwritten (SR = 128 or 256).          */
    void*   save_acts;
    void*   save_xin;
    void*   save_masks;
    void*   save_side;
    /* inference (F16X3, input A), optional: the time code's part of the dynamic trunk's input layers, computed once per ray by
     * nsff_time_bias from the same t_emb rows: (n_rays, t_bias_rows, 256) fp32.  With it -- and pts_per_ray a multiple of 64 --
     * the hand-scheduled kernel multiplies no time-code column (8 of a dynamic trunk's 128 k-steps); results agree with the
     * plain launch to fp32 rounding.  t_emb must be given either way (it is what every other kernel variant reads). */
    const float* t_bias;
    int32_t t_bias_rows;     /* nsff_time_bias_rows(desc), or 0                     */
    int32_t reserved0;
    /* inference (F16X3, input A, use_viewdir models, static_mode 2), optional: what [dir | a] contributes to
     * static_dir_encoding, computed once per ray by nsff_side_bias from the same dir_emb / a_emb rows: (n_rays, 1, 256) fp32.
     * With it -- and pts_per_ray a multiple of 64 -- the static trunk of a view-direction model runs on the hand-scheduled
     * kernel: static_dir_encoding becomes one more 256-wide layer with per-ray bias rows, sigma is accumulated in the epilogue
     * of the last trunk layer.  dir_emb / a_emb must be given either way (every other kernel variant reads them). */
    const float* s_bias;
    int32_t s_bias_rows;     /* 1, or 0                                             */
    /* 0: the library's choice -- large hand-scheduled launches are PERSISTENT (one workgroup per compute unit walking its tiles,
     * nsff_last_field_grid); 1: one workgroup per 128-point tile (bit-identical records).  A persistent launch holds every
     * compute unit until it ends: a caller whose collective kernel runs beside the render stream (the overlapped pixel
     * all-gather of a multi-GPU evaluation, nsff_pl_amd/dist.py::all_gather_pixels_async) asks for 1, so that the collective gets
     * a compute unit at the next tile boundary and a render workgroup never waits behind it for a whole launch. */
    int32_t launch_form;
    /* Training forward for the THREE-PRODUCT backward (nsff_field_backward / nsff_weight_grad* with grad_x3): every saved tile is
     * written twice -- the fp16 value `v = fp16(x)` as before and, save_lo_delta[i] ELEMENTS behind it in the same buffer
     * (i = 0 save_acts, 1 save_xin, 2 save_side; 0 = not wanted), the remainder `fp16(x - v)`: the backward's GEMMs then multiply
     * hi + lo operands as the forward does.  Eight-wave SAVE kernel only (the launch takes it when any delta is set). */
    int64_t save_lo_delta[3];
} NsffFieldArgs;

int nsff_field_query(const NsffModelDesc* desc, const void* packed,
                     const NsffFieldArgs* args, void* stream);

/* ---- a2/a3: what the time code contributes to the dynamic trunk, once per RAY instead of once per point.  Every sample of a
 * ray shares its time code (reference rendering.py:153,168,221,227 repeat it), so for layer 0 and the skip layers -- the layers
 * whose input holds [xyz | t] (nerf.py:163-167) -- the product of the time-code columns is a per-ray vector:
 *     out[ray][i][n] = b_l[n] + sum_j W_l[n][in_xyz + j] * t_rows[ray][j],   l = 0, then the skip layers in ascending order
 * (i = 0 .. nsff_time_bias_rows(desc) - 1), in fp32.  nsff_field_query takes the result as NsffFieldArgs::t_bias.
 * Up to NSFF_MAX_TIME_BIAS_JOBS (model, time rows) pairs per launch: a render_rays call needs the coarse model at t and the fine
 * model at t, t + 1, t - 1.  w[i] / b[i]: weight (256, in_xyz + in_t [+ 256 for a skip layer]) and bias (256) of
 * transient_xyz_encoding_{l_i + 1}, the parameters themselves (PyTorch Linear layout, no pack needed).
 * Time codes wider than 64 columns or with in_t % 4 != 0 are refused (NSFF_ERR_INVALID): the kernel stages 64 columns per ray
 * as float4s, and such models keep their time-code columns on the matrix pipe (nsff_field_query ignores t_bias for them). */
#define NSFF_MAX_TIME_BIAS_JOBS 4
typedef struct NsffTimeBiasJob {
    const NsffModelDesc* desc;
    const float* w[NSFF_MAX_LAYERS];
    const float* b[NSFF_MAX_LAYERS];
    const float* t_rows;     /* (n_rays, in_t), or NULL: index mode (below)         */
    float* out;              /* (n_rays, nsff_time_bias_rows(desc), 256)            */
    /* index mode (t_rows == NULL): the job gathers its rows itself, t_rows[r] = table[clamp(ts[r] + delta)] -- the reference's
     * embedding_t(ts), embedding_t(clamp(ts + 1, max=max_t)), embedding_t(clamp(ts - 1, min=0)) (rendering.py:153,218,224) for
     * delta = 0, +1, -1 (the index is also kept inside the table) -- and leaves them in rows_out (n_rays, in_t) when that is
     * given: one launch instead of a gather, the neighbour-row kernel and this one. */
    const float* table;      /* (n_table, in_t) time-code table                     */
    const int64_t* ts;       /* (n_rays) frame indices                              */
    int64_t n_table, max_t;
    int32_t delta, pad_;
    float* rows_out;         /* (n_rays, in_t) or NULL                              */
} NsffTimeBiasJob;
int nsff_time_bias_rows(const NsffModelDesc* desc);      /* 0: the model has no dynamic trunk */
int nsff_time_bias(const NsffTimeBiasJob* jobs, int32_t n_jobs, int64_t n_rays, void* stream);

/* ---- a4: the random draws of a render_rays call (reference rendering.py:321 perturb, 207 / 213 noise per pass, 128 the two
 * warps, 338-340 the inverse-CDF draws) in ONE launch, bit-identical to torch.rand / torch.randn of the same generator state:
 * job j fills out[0, numel) with what a torch kernel of `grid` blocks of 256 threads -- ATen's launch geometry for numel
 * elements: min(#CU * (max threads per CU / 256), ceil(numel / 256)) -- started at Philox offset `offset` (a multiple of 4) of
 * `seed` would write; the caller advances the generator by ((numel - 1) / (1024 grid) + 1) * 4 per job, as torch does.
 * kind 0: uniform [0, 1) (torch.rand), 1: standard normal (torch.randn).  */
#define NSFF_MAX_RNG_JOBS 12
typedef struct NsffRngJob {
    float*   out;
    int64_t  numel;
    uint64_t offset;
    int32_t  kind;
    uint32_t grid;
} NsffRngJob;
int nsff_rng_draws(const NsffRngJob* jobs, int32_t n_jobs, uint64_t seed, void* stream);
/* The same launch also makes the coarse depths (reference rendering.py:314-324, 332; nsff_coarse_samples is the stand-alone
 * form) from the stratified-sampling draw, where that draw is made: job `job` is the uniform (n_rays, n_samples) draw of
 * rendering.py:321 (its `out` may be NULL: nobody else reads it), zs / xyz as nsff_coarse_samples writes them (perturb > 0). */
typedef struct NsffRngCoarse {
    const float* rays;       /* (n_rays, 6)                       */
    const float* z_lin;      /* (n_samples)                       */
    float*       zs;         /* (n_rays, n_samples)               */
    float*       xyz;        /* (n_rays, n_samples, 3)            */
    int32_t      n_samples;
    float        perturb;
    int32_t      job;
    int32_t      pad_;
} NsffRngCoarse;
int nsff_rng_draws_coarse(const NsffRngJob* jobs, int32_t n_jobs, uint64_t seed, const NsffRngCoarse* coarse, void* stream);

/* ---- a3: the same for a view-direction model's per-ray inputs (reference nerf.py:183-185, rendering.py:153-172 repeat the
 * direction embedding and the appearance code over a ray's samples):
 *     out[ray][n] = b_fold[n] + sum_j W_dir[n][256 + j] * [dir_rows[ray] | a_rows[ray]][j]
 * with W_dir = static_dir_encoding's weight (256, 256 + in_dir + in_a), the parameter itself, and b_fold = W_dir[:, :256] b_final
 * + b_dir from the F16X3 packed buffer of an inference pack (nsff_pack_weights / nsff_fold_heads).  fp32, columns in ascending
 * order.  a_rows may be NULL iff in_a == 0.  nsff_field_query takes the result as NsffFieldArgs::s_bias. */
int nsff_side_bias(const NsffModelDesc* desc, const void* packed_f16x3, const float* w_dir, const float* dir_rows,
                   const float* a_rows, int64_t n_rays, float* out, void* stream);

/* ---- N1: backward of the field query (training).  Mixed precision: fp16 MFMA operands, fp32 accumulation;
 * every point's gradient row is normalised by its own power of two (block floating point), weight-gradient
 * operands are expressed on one global power-of-two scale G = 2^floor(log2(4096 / *gmax)).
 *
 * nsff_bwd_packed_bytes / nsff_pack_weights_bwd: the TRANSPOSED fp16 weight tiles the data-gradient chain
 * streams (same parameter order as nsff_pack_weights).
 * nsff_field_backward: d_raw (P,16) -> d(trunk input) and the pre-activation gradients of every layer:
 *   dpre : fp16 (NS, T, 4, 256, 16)  slot t*(D+1)+l = trunk t (0 static, 1 transient) layer l < D (slot l = D, *_final, is
 *          never written: the layer is folded into the heads), slot D = static_dir_encoding (use_viewdir); values = true
 *          gradient * G, fragment order as save_acts;
 *   dhead: fp16 (2, T, 4, 32, 16) head pre-activation gradients * G, rows 0..15: static rgb(3) sigma(1);
 *          transient rgb(3) sigma(1) fw(3) bw(3); rows 16..31: the fp16 rounding remainder of rows 0..15 (the head
 *          weight / bias gradients are the sum of both halves);
 *   d_xin: fp32 (P, XR) true gradient w.r.t. the transient trunk input (rows as save_xin), or NULL;
 *   d_side: fp32 (P, SR) true gradient w.r.t. the [dir | a] input of static_dir_encoding (rows as save_side), or NULL.
 * nsff_train_dims: XR (xin_rows), t_row0 and SR (side_rows) of a model -- 128 unless ceil64(in_xyz) + ceil64(in_t) > 128
 * (resp. in_dir + in_a > 128), then 256.  Any skip list the forward takes is differentiated (nerf.py:34-40,163-167). */
int nsff_train_dims(const NsffModelDesc* desc, int32_t* xin_rows, int32_t* t_row0, int32_t* side_rows);
int nsff_bwd_packed_bytes(const NsffModelDesc* desc, size_t* bytes);
int nsff_pack_weights_bwd(const NsffModelDesc* desc, const float* const* params, void* packed, void* stream);
/* ... given the F16X3 forward pack of the SAME weights (nsff_pack_weights with folded heads), whose fp32 scratch already holds the
 * folded products W_head W_final the transposed tiles need (else they are multiplied here); fwd_packed may be NULL. */
int nsff_pack_weights_bwd_ex(const NsffModelDesc* desc, const float* const* params, void* packed, const void* fwd_packed, void* stream);

typedef struct NsffFieldBwdArgs {
    int64_t n_points;
    int32_t static_mode;        /* 0 skip, 2 rgb+sigma (as in the forward call)            */
    int32_t transient_mode;     /* 0 skip, 2                                                */
    const float* d_raw;         /* (P, NSFF_RAW_STRIDE) gradient w.r.t. the raw record      */
    const float* raw;           /* (P, NSFF_RAW_STRIDE) forward outputs (activation derivatives) */
    const float* gmax;          /* device float[16]: max |d_raw| per record column (nsff_absmax_raw): each trunk's fragments are on
                                 * the power-of-two scale of its largest column, every head row of dhead on its own        */
    const void*  masks;         /* save_masks of the forward call                            */
    void*  dpre;                /* OUT */
    void*  dhead;               /* OUT */
    float* d_xin;               /* OUT or NULL                                               */
    float* d_side;              /* OUT or NULL (use_viewdir models, static_mode 2)           */
    int64_t dpre_lo_delta;      /* 0: one fp16 product per multiply-accumulate (the measured path).  Otherwise THREE, as the forward
                                 * multiplies: gradient tiles and weights as fp16 value + fp16 remainder, every pre-activation-gradient
                                 * tile written twice -- dpre as above and its remainder this many ELEMENTS behind it (multiple of 8) --
                                 * for nsff_weight_grad* jobs with a_lo_delta set.  The forward of such a step saves its activations
                                 * the same way (NsffFieldArgs::save_lo_delta).  Gradients then match the reference's fp32 ones to
                                 * ~1e-5 instead of ~1e-3; the chain runs at about a third of the speed. */
} NsffFieldBwdArgs;
int nsff_field_backward(const NsffModelDesc* desc, const void* packed_bwd, const NsffFieldBwdArgs* args, void* stream);
/* Which kernel the last nsff_field_backward launch took: 2 = both (a view-direction model's launch of both trunks: the static trunk
 * on nsff_field_bwd_kernel, the dynamic one on the hand-scheduled body, as two launches); 1 = nsff_field_bwd_kernel_h3b, the hand-scheduled body (128-point
 * workgroups, one wave per SIMD, resident transposed weights, epilogues / fragment copies / refills riding in the other half's MFMA
 * gaps: tools/h3asm/gen_bwd.py) -- launches with an even number of 64-point tiles whose trunks it executes (no view-direction static
 * trunk, at most one skip layer with a trunk-input gradient, none at the last layer); 0 = the kernel nsff_field_bwd_kernel -- compiler-scheduled,
 * 64-point workgroups; NSFF_BWD_KERNEL=c forces it.  Both leave bit-identical dpre / dhead / d_xin. */
int nsff_last_bwd_kernel(void);
/* Workgroups of the last hand-scheduled data-gradient launch (0: none since the last nsff_field_backward).  Launches that give
 * every compute unit at least two (tile, trunk) items run PERSISTENT: one workgroup per compute unit, further items from a device
 * counter, the next item's records brought into LDS behind the current item's body; NSFF_BWD_PERSIST=0: one workgroup per item. */
int nsff_last_bwd_grid(void);
/* Host-only (no GPU): the hand-scheduled body's phase program for one trunk of `desc` (dynamic: the transient trunk, with or without
 * the trunk-input gradient) at n_tiles 64-point tiles -> out[max_phases][8] uint32 descriptors; seg_offsets (or NULL): byte offsets of
 * the trunk's step segments in the transposed pack.  Returns the number of descriptors, 0 when the body does not cover the trunk. */
int nsff_field_bwd_phase_program(const NsffModelDesc* desc, int32_t dynamic, int32_t want_xin, int64_t n_tiles, uint32_t* out,
                                 int32_t max_phases, uint32_t* seg_offsets, int32_t max_segs);

/* d_xin (P, xin_rows) of nsff_field_backward -> gradient w.r.t. the points (derivative of PosEmbedding, reference
 * nerf.py:17-30) and w.r.t. the per-ray time codes (sum over the ray's pts_per_ray consecutive points, the repeat of
 * rendering.py:168).  d_xyz (P,3) / d_t (n_rays, in_t): either may be NULL.  freqs_host: HOST array.                */
int nsff_field_input_backward(const float* d_xin, int32_t xin_rows, int32_t t_row0, const float* xyz, int64_t n_rays,
                              int32_t pts_per_ray, const float* freqs_host, int32_t n_freqs, int32_t in_t, float* d_xyz,
                              float* d_t, void* stream);

/* Batched weight-gradient GEMMs, K = points:  out_j = (1/G) * A_j^T . B_j over all point tiles, G as above.
 * A_j: fp16 fragment-major (T,4,a_rows,16), a_rows in {256, 32};  B_j: (T,4,b_rows,16), b_rows in {256, 128}.
 * Split-K: every job is cut into about n_splits tile ranges (rounded per job shape so that the workgroups of equally
 * shaped jobs fill whole rounds of the 256 CUs; heads: 8x as many) whose partial sums go to `scratch`
 * (nsff_weight_grad_scratch floats) and are summed, scaled and written by a second launch to
 *   out + out_off[j]  : fp32 a_rows_j x b_rows_j (row-major),     bias + 256*j : fp32 row sums of A_j (bias gradients).
 * `jobs` is a HOST array.                                                                                      */
typedef struct NsffWgradJob {
    const void* a;  const void* b;
    int32_t a_rows, b_rows;
    int64_t out_off;            /* floats */
    int32_t trunk, pad_;        /* 0 static / 1 dynamic: which of the two scales (gmax[trunk]) the job's A operand is on */
    int64_t a_lo_delta, b_lo_delta;   /* 0 / 0: one fp16 product per multiply-accumulate.  Three products (the launches behind a
                                 * nsff_field_backward with dpre_lo_delta): ELEMENTS from a / b to their remainder planes (dpre's twin,
                                 * the forward's save_lo_delta twins); every job of a call or none; a_lo_delta stays 0 for the 32-row
                                 * head jobs, whose A carries its remainder rows itself */
} NsffWgradJob;
int64_t nsff_weight_grad_scratch(const NsffWgradJob* jobs, int32_t n_jobs, int64_t n_tiles, int32_t n_splits);
int nsff_weight_grad(const NsffWgradJob* jobs, int32_t n_jobs, int64_t n_tiles, int32_t n_splits,
                     float* scratch, float* out, float* bias, const float* gmax, void* stream);

/* The same GEMMs, but the second launch ACCUMULATES straight into the parameters' gradient memory (what autograd's
 * AccumulateGrad does for torch.nn.Linear's grad_weight / grad_bias): for every map entry
 *   grad_base[dst] += (1/G) * (S(job_a, e_a) + S(job_b, e_b)),      S(j, e) = sum over the splits of job j's partials,
 * e < a_rows*b_rows: element e of out_j (row-major);  e >= a_rows*b_rows: row sum (bias gradient) e - a_rows*b_rows;
 * job_b < 0: no second term (it carries the fp16-remainder rows of the head gradients).  Every dst must appear at most
 * once (one owner per gradient element: the result is deterministic).  `map` is a DEVICE array, 16-byte aligned;
 * jobs[].out_off is ignored.                                                                                       */
typedef struct NsffGradMapEntry {
    int32_t dst;                /* floats from grad_base */
    int32_t e_a, e_b;
    int16_t job_a, job_b;
} NsffGradMapEntry;
int nsff_weight_grad_accumulate(const NsffWgradJob* jobs, int32_t n_jobs, int64_t n_tiles, int32_t n_splits, float* scratch,
                                const NsffGradMapEntry* map, int64_t n_map, float* grad_base, const float* gmax, void* stream);

/* nsff_weight_grad_accumulate with a second destination: map entries with dst < 0 STORE their value at aux[-(dst + 1)]
 * -- dense sums of the jobs the FOLDED parameters are made of: *_xyz_encoding_final is never executed as a layer (the heads that
 * read it are evaluated as (W_head W_final) h), so its gradient and the heads' are small matrix products of the folded heads'
 * gradient G = sum_p dpre_p (x) h_p:  dW_head = G W_final^T + gb (x) b_final,  dW_final = W_head^T G,  db_final = W_head^T gb. */
int nsff_weight_grad_accumulate_aux(const NsffWgradJob* jobs, int32_t n_jobs, int64_t n_tiles, int32_t n_splits, float* scratch,
                                    const NsffGradMapEntry* map, int64_t n_map, float* grad_base, float* aux, const float* gmax,
                                    void* stream);

/* One launch that turns the folded heads' gradient into the gradients of *_xyz_encoding_final and of the heads that read it
 * and ADDS them to the parameters' gradient memory:  g (32, 256) / gb (32): the dense sum / row sums of the heads' job (rows r
 * and 16 + r = fp16 value + rounding remainder of folded row r < n_rows <= 16);
 *   d_w_head[r] += G[r] W_final^T + gb[r] b_final,  *d_b_head[r] += gb[r],  d_w_final += W_head^T G,  d_b_final += W_head^T gb. */
typedef struct NsffFoldGradArgs {
    int32_t n_rows, pad_;
    const float* g; const float* gb;
    const float* w_final; const float* b_final;        /* (256, 256), (256) */
    const float* w_head[16];                           /* row r of the heads' weights: 256 floats */
    float* d_w_head[16]; float* d_b_head[16];
    float* d_w_final; float* d_b_final;
} NsffFoldGradArgs;
int nsff_fold_grads(const NsffFoldGradArgs* args, void* stream);

/* The same algebra for a folded layer of ANY height (n_rows <= 256) whose weights are one strided matrix -- the view-direction
 * layer static_dir_encoding (reference models/nerf.py:83-91,183-186: it reads [*_final | dir | a]; its first 256 columns are the
 * folded part, ld_head = 256 + in_dir + in_a) -- and for callers that want the results as tensors instead of accumulated
 * (accumulate = 0: every output element is STORED):
 *   g (n_rows, 256) [+ g2: a second summand, the fp16 rounding-remainder rows of a heads' job, or NULL], gb (n_rows) [+ gb2]
 *   d_w_head[r * ld_dhead + o] (+)= sum_i G[r][i] W_final[o][i] + gb[r] b_final[o]      d_b_head[r] (+)= gb[r]
 *   d_w_final[o][i] (+)= sum_r W_head[r * ld_head + o] G[r][i]                          d_b_final[o] (+)= sum_r W_head[r * ld_head + o] gb[r]
 * Two launches (rows of the head layer / neurons of *_final); fp32 FMAs in a fixed order: deterministic.  This is what ran
 * through torch.addmm / `@` (rocBLAS) for view-direction models and for gradients returned as tensors until round 5.       */
typedef struct NsffFoldDenseArgs {
    int32_t n_rows, accumulate;
    int32_t ld_head, ld_dhead;
    const float* g; const float* g2; const float* gb; const float* gb2;
    const float* w_head; const float* w_final; const float* b_final;
    float* d_w_head; float* d_b_head; float* d_w_final; float* d_b_final;
} NsffFoldDenseArgs;
int nsff_fold_grads_dense(const NsffFoldDenseArgs* args, void* stream);

/* out[0] = max |x[i]| (0 for n == 0): the device scalar `gmax` of nsff_field_backward / nsff_weight_grad without a
 * host round trip (replaces d_raw.abs().max()).  x 16-byte aligned.                                                */
int nsff_absmax(const float* x, int64_t n, float* out, void* stream);
/* out16[c] = max |d_raw[:, c]| for the 16 floats of a record: the `gmax` vector of nsff_field_backward / nsff_weight_grad*.  A trunk's
 * fragments (dpre) are on the scale of its largest column (static: 0..3, dynamic: 4..13), every head row of dhead on its own: in a
 * real NSFF step the columns' gradients are 10^6-10^7 apart (the 2D flow terms of the loss are in pixels); on a common scale the
 * smaller ones' fp16 fragments fall into and below the subnormal range -- a bit or two, then zero (round 6, golden g20:
 * static weight gradients 40-70 % too small at 512 rays, transient_sigma.weight 25 %). */
int nsff_absmax_raw(const float* d_raw, int64_t n_points, float* out16, void* stream);

/* ---- N1: the optimizer step: torch.optim.Adam(lr, betas, eps, weight_decay) as the reference builds it
 * (utils/__init__.py:45-47 get_optimizer, used by train.py:140-146), amsgrad off, on flat fp32 buffers of n elements
 * (n % 4 == 0, 16-byte aligned):  g' = g + weight_decay * p;  m = lerp(m, g', 1 - beta1);  v = beta2 * v + (1 - beta2) * g'^2;
 * p -= lr / (1 - beta1^t) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps).
 * state: DEVICE float[4]; state[0] = number of steps taken so far (zero it once; the call increments it), state[1..2]
 * scratch.  lr: DEVICE scalar.  Two launches, no host round trip: capturable into a hipGraph.                      */
int nsff_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float* state,
                   const float* lr, double beta1, double beta2, double eps, double weight_decay, void* stream);

/* The same step with torch.optim.Adam's treatment of parameters that receive no gradient (`if p.grad is None: continue`,
 * what train.py's optimizer does for a head the step never used): the flat buffers are n_seg consecutive parameter tensors,
 * tensor k = elements [seg_start[k], seg_start[k+1]) (DEVICE int64[n_seg + 1], ascending, seg_start[0] = 0); a tensor whose
 * gradient slice is identically zero this step keeps its values AND its moments (no weight decay, no moment decay);
 * seg_used: DEVICE int32[n_seg] scratch (receives the per-tensor flags).  One memset + three launches, capturable.
 * seg_start == NULL (with seg_used == NULL): every element is updated, exactly nsff_adam_step.                          */
int nsff_adam_step_segments(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float* state,
                            const float* lr, double beta1, double beta2, double eps, double weight_decay,
                            const int64_t* seg_start, int n_seg, int32_t* seg_used, void* stream);

/* ---- N1: the training objective NeRFWLoss (reference losses.py:8-28, 31-171) on the render dict, NSFF train-mode
 * configuration (flows + disocclusion present, topk == 1, no per-ray weights, thickness == 1), every term reduced to
 * its scalar.  mode 1: terms[11] = col_l, disp_l, entropy_l, cross_entropy_l, flow_fw_l, flow_bw_l, pho_l, cyc_l,
 * reg_temp_sm_l, reg_min_l, reg_sp_sm_l (six launches; `stats`, `per_ray` and `coef` receive what mode 2 re-uses).
 * mode 2: g_* = gradient of sum_k term_w[k] * term_k w.r.t. the tensor of the same name (two launches).
 * Reduction of a term (losses.py:162-169): its per-ray values v_n (times weights[n] when given) over its population -- every
 * ray, or for the two flow terms the M rays whose projection is valid -- are reduced to the mean of the K LARGEST,
 * K = int(topk * M) for topk < 1, else K = M (the plain mean; 0 for an empty population).  `thickness` > 1 dilates the
 * detached transient weights of cross_entropy_l with a 1 x thickness box filter, zero padded (losses.py:91-95).
 * n_rays <= 4096.  hyper (device): lambda_geo_d, lambda_geo_f, cross-entropy weight, lambda_reg, lambda_ent.      */
typedef struct NsffLossArgs {
    int64_t n_rays; int32_t n_samples; int32_t n_keep;   /* n_keep = int(n_samples * z_far): samples the regularisers see */
    int32_t n_frames; int32_t max_t;
    double  topk;                                        /* --topk (opt.py:80): >= 1 = plain means                 */
    int32_t thickness; int32_t pad_;                     /* --thickness (opt.py:49), >= 1                          */
    /* render dict */
    const float* rgb_fine; const float* rgb_coarse;      /* (N,3); coarse pair may be NULL                         */
    const float* depth_fine; const float* depth_coarse;  /* (N)                                                    */
    const float* t_weights; const float* s_weights;      /* (N,S) transient_weights_fine, static_weights_fine      */
    const float* xyz_fw; const float* xyz_bw;            /* (N,3)                                                  */
    const float* rgb_fw; const float* rgb_bw;            /* (N,3)                                                  */
    const float* disocc_fw; const float* disocc_bw;      /* (N)                                                    */
    const float* disoccs_fw; const float* disoccs_bw;    /* (N,S)                                                  */
    const float* xyzs_fw_bw; const float* xyzs_bw_fw;    /* (N,S,3)                                                */
    const float* xyzs_fine; const float* xyzs_fw; const float* xyzs_bw;   /* (N,S,3)                               */
    /* targets / camera buffers (train.py:136-138) */
    const float* rgbs; const float* disps;               /* (N,3), (N)                                             */
    const int64_t* ts; const int64_t* cam_ids;           /* (N); cam_ids may be NULL (= 0)                         */
    const float* uv_fw; const float* uv_bw;              /* (N,2)                                                  */
    const float* Ks; const float* Ps;                    /* (n_cam,3,3), (n_cam,n_frames,3,4)                      */
    const float* hyper;                                  /* device, 5 floats                                       */
    float* stats;                                        /* device, 24 floats: written by mode 1, read by mode 2   */
    float* terms;                                        /* OUT mode 1: 11 floats                                  */
    const float* term_w;                                 /* mode 2: 11 upstream scalars (device)                   */
    const float* weights;                                /* (N) per-ray loss weights (hard sampling) or NULL       */
    float* per_ray;                                      /* device (11, N): weighted per-ray values, mode 1 OUT; < 0 marks a
                                                            ray outside a flow term's population                    */
    float* coef;                                         /* device (11, N): d term_k / d v_n = weights[n] * [n selected] / K,
                                                            mode 1 OUT, mode 2 IN                                   */
    float* g_rgb_fine; float* g_rgb_coarse; float* g_depth_fine; float* g_depth_coarse;
    float* g_t_weights; float* g_s_weights; float* g_xyz_fw; float* g_xyz_bw; float* g_rgb_fw; float* g_rgb_bw;
    float* g_xyzs_fw_bw; float* g_xyzs_bw_fw; float* g_xyzs_fw; float* g_xyzs_bw;
} NsffLossArgs;
int nsff_nerfw_loss(const NsffLossArgs* args, int mode, void* stream);

/* ---- a4: coarse sample placement (reference rendering.py:314-324,332) ----
 * zs[n][i] = z_lin[i]                                   (perturb == 0)
 *          = lower + (upper-lower) * perturb * rnd[n][i] (perturb > 0)
 * xyz[n][i] = o[n] + d[n]*zs[n][i]                                             */
int nsff_coarse_samples(const float* rays, int64_t n_rays, const float* z_lin, int32_t n_samples,
                        float perturb, const float* perturb_rand,
                        float* zs, float* xyz, void* stream);

/* ---- a9: sample_pdf x{1,2} + merge + sort (reference rendering.py:10-49,335-348,359) ----
 * weights_* are the (n_rays, n_samples) coarse weights (the [1:-1] slice is taken here).
 * u_* : (n_importance) when u_per_ray==0 (deterministic linspace), else (n_rays, n_importance).
 * zs_fine: (n_rays, n_samples + k*n_importance) sorted ascending, k = 1 + (weights_transient!=NULL)
 * xyz_fine = o + d*zs_fine.  samples_static/transient (n_rays,n_importance) optional. */
int nsff_fine_samples(const float* rays, int64_t n_rays, const float* z_lin, const float* zs_coarse,
                      int32_t n_samples, int32_t n_importance,
                      const float* weights_static, const float* weights_transient,
                      const float* u_static, const float* u_transient, int32_t u_per_ray,
                      float* samples_static, float* samples_transient,
                      float* zs_fine, float* xyz_fine, void* stream);

/* standalone sample_pdf(bins, weights, N_importance) with explicit u (reference rendering.py:10-49) */
int nsff_sample_pdf(const float* bins, const float* weights, int64_t n_rays, int32_t n_bins_minus1,
                    const float* u, int32_t n_importance, int32_t u_per_ray, float eps,
                    float* samples, void* stream);

/* ---- a8 (first half): xyzs_fw = xyz + flow_fw, xyzs_bw = xyz + flow_bw with the
 * z > z_far flow zeroing (reference rendering.py:187-188,218,224) ---- */
int nsff_warp_points(const float* raw, const float* xyz, const float* zs, int64_t n_points,
                     float z_far, float* xyz_fw, float* xyz_bw, void* stream);

/* ---- N3 (step before the path): NDC camera rays of a pinhole frame, generated on the device
 * (reference datasets/ray_utils.py:7-106 via datasets/monocular.py:268-276).
 * K4_host = {fx, fy, cx, cy}, c2w_host = row-major (3,4), both HOST pointers; pixel p = row*W + col;
 * rays: (n_pixels, 6) = NDC origin | NDC direction for pixels [first_pixel, first_pixel+n_pixels). */
int nsff_frame_rays(const float* K4_host, const float* c2w_host, int32_t H, int32_t W, float near, float shift_near,
                    int64_t first_pixel, int64_t n_pixels, float* rays, void* stream);

/* ---- a6: eval-only frustum visibility (reference rendering.py:190-200 with datasets/ray_utils.py:127-151,154-181) ---- */
typedef struct NsffFrustumArgs {
    const float*   w2c;         /* device (n_cams * n_frames, 12): row-major first three rows of inverse([c2w; 0 0 0 1]),
                                   camera i of frame f at row i * n_frames + f (dataset.poses order, rendering.py:199)  */
    const int64_t* ts;          /* device: the frame is ts[0] (read on the device: no host synchronisation); a frame outside
                                   [0, n_frames) is treated as seen by no camera (count 0), never read out of bounds      */
    float   K4[4];              /* fx, fy, cx, cy of dataset.Ks[0]                                                     */
    int32_t n_cams, n_frames, H, W;
} NsffFrustumArgs;

/* ---- a7/a8: sigma->alpha compositing and every per-ray / per-sample output ---- */
typedef struct NsffCompositeArgs {
    int64_t n_rays;
    int32_t n_samples;          /* S of this pass                                   */
    int32_t has_transient;      /* output_transient                                 */
    int32_t has_rgb;            /* 0 for the sigma-only coarse test-time pass       */
    int32_t flow_mode;          /* 0 none, 1 per-ray/per-sample flow outputs, 2 + warped renders */
    int32_t want_disocc;        /* 'disocc' in output_transient_flow (flow_mode 2)   */
    float   noise_std;
    float   z_far;              /* 0.95                                             */
    const float* raw;           /* (N*S,16) field records at xyz                    */
    const float* raw_fw;        /* (N*S,16) records at xyz+flow_fw, t+1 (flow_mode 2)*/
    const float* raw_bw;        /* (N*S,16) records at xyz+flow_bw, t-1             */
    const float* zs;            /* (N,S)                                            */
    const float* xyz;           /* (N,S,3)                                          */
    const float* xyz_fw;        /* (N,S,3) (flow_mode 2)                            */
    const float* xyz_bw;
    const float* noise_static;  /* (N,S) standard normal draws or NULL (=> 0)       */
    const float* noise_transient;
    const float* noise_fw;
    const float* noise_bw;
    const float* visibility;    /* (N*S) or NULL; ==0 => raw transient sigma := -10 (rendering.py:200) */
    /* per-sample outputs (NULL = not wanted) */
    float* static_rgbs;  float* transient_rgbs;        /* (N,S,3) */
    float* flows_fw;     float* flows_bw;              /* (N,S,3) zeroed beyond z_far */
    float* static_sigmas; float* transient_sigmas;     /* (N,S) softplus'd */
    float* static_alphas; float* transient_alphas;     /* (N,S) */
    float* static_weights; float* transient_weights; float* weights;   /* (N,S) */
    float* xyzs_fw_bw;   float* xyzs_bw_fw;            /* (N,S,3) */
    float* disoccs_fw;   float* disoccs_bw;            /* (N,S) */
    /* per-ray outputs */
    float* depth;  float* rgb;  float* transient_alpha;  float* transient_rgb;
    float* static_only_rgb;  float* static_only_depth;
    float* xyz_exp;  float* flow_fw_exp;  float* flow_bw_exp;  float* xyz_fw_exp;  float* xyz_bw_exp;
    float* rgb_fw;   float* rgb_bw;
    float* disocc_fw;  float* disocc_bw;
    NsffFrustumArgs vis;        /* a6 evaluated INSIDE this kernel: vis.w2c != NULL => the raw transient sigma of every sample
                                   no training camera of frame ts[0] sees becomes -10 (`visibility` above is the same mask
                                   handed in as an array) */
} NsffCompositeArgs;

int nsff_composite(const NsffCompositeArgs* args, void* stream);

/* The a6 mask as a stage of its own (tests, callers that want the mask): vis_count[p] = number of the frame's training
 * cameras whose image contains NDC point xyz[p] (float, like the reference's `visibilities`). */
int nsff_frustum_visibility(const NsffFrustumArgs* vis, const float* xyz, int64_t n_points, float* vis_count,
                            void* stream);

/* ---- N2: time interpolation of two test-time renders (reference models/rendering.py:365-460 with
 * models/softsplat.py:6-44,303-326 'average' splatting).  The S sample planes of a frame are splatted by ONE
 * launch (the reference: 2*S cupy launches with host round trips) into a per-pixel, per-plane accumulator
 *     accum[(pixel*S + s)*8 + c],  c = r*1, g*1, b*1, a*1 (bilinear-weighted sums), 4 = sum of weights, 5..7 unused
 * with hardware fp32 atomic adds; nsff_mpi_composite normalises (zeros -> 1) and composites front to back. ---- */
typedef struct NsffSplatArgs {
    int32_t H, W, n_planes;     /* n_planes = S = samples per ray (xyzs_fine.shape[1])                      */
    float   K4[4];              /* fx, fy, cx, cy (ndc2world, datasets/ray_utils.py:127-151)                 */
    float   P[12];              /* row-major (3,4) K @ w2c with rows 1,2 of w2c negated (rendering.py:390-394) */
    float   scale;              /* dt for the forward splat of frame t, 1-dt for the backward splat of t+1  */
    const float* xyz;           /* (H*W, S, 3) NDC sample points (xyzs_fine)                                 */
    const float* flow;          /* (H*W, S, 3) transient_flows_fw of t  /  transient_flows_bw of t+1         */
    const float* rgb;           /* (H*W, S, 3) transient_rgbs_fine                                           */
    const float* alpha;         /* (H*W, S)    transient_alphas_fine                                         */
    float*       accum;         /* (H*W, S, 8) OUT; cleared by this call                                     */
    void*        work;          /* device workspace of the binned far path (32-byte aligned) or NULL: samples that land
                                   farther than 4 pixels from their own pixel are then added with device-scope atomics */
    int64_t      work_bytes;    /* nsff_splat_work_bytes(H, W, S) holds every far sample twice (a sample makes one record
                                   per 32 x 8 output block it touches, 1.16 on average); what does not fit takes the
                                   atomic route                                                                       */
} NsffSplatArgs;
int64_t nsff_splat_work_bytes(int32_t H, int32_t W, int32_t n_planes);
int nsff_splat_planes(const NsffSplatArgs* args, void* stream);

typedef struct NsffMpiArgs {
    int32_t H, W, n_planes;
    float   dt;
    const float* accum_fw;      /* nsff_splat_planes of (t, flows_fw, dt)                                    */
    const float* accum_bw;      /* nsff_splat_planes of (t+1, flows_bw, 1-dt)                                */
    const float* static_rgb;    /* (H*W, S, 3) static_rgbs_fine of t                                         */
    const float* static_alpha;  /* (H*W, S)    static_alphas_fine of t                                       */
    const float* zs;            /* (H*W, S)    zs_fine of t                                                  */
    float* rgb;                 /* (H*W, 3) OUT                                                              */
    float* depth;               /* (H*W)    OUT (NDC)                                                        */
} NsffMpiArgs;
int nsff_mpi_composite(const NsffMpiArgs* args, void* stream);

/* ---- N1: backward of nsff_composite for the training configurations (has_rgb = 1, flow_mode 0 or 2): gradients
 * w.r.t. the raw records of the main pass and of the two warped re-queries, and w.r.t. the (far-masked) per-sample
 * flows that enter the per-ray flow expectations.  g_* = gradient of the output of that name (NULL = zero):
 * per sample (n_rays, S): static_sigmas, transient_sigmas, static_weights, transient_weights, weights;
 * per ray: depth (n), rgb (n,3), transient_alpha (n), transient_rgb (n,3), so_rgb = _static_rgb (n,3), so_depth =
 * _static_depth (n), xyz_exp = xyz_fine (n,3), flow_fw_exp / flow_bw_exp = transient_flow_fw / _bw (n,3), rgb_fw, rgb_bw.
 * The per-sample passthrough outputs (rgbs, flows, warped points) are plain views of the inputs and stay with the
 * caller.  scratch: (n_rays, S, 4) floats.  d_raw* : (P, NSFF_RAW_STRIDE), slots 0-7 written, the rest zeroed. ---- */
typedef struct NsffCompositeBwdArgs {
    int64_t n_rays;
    int32_t n_samples, has_transient, flow_mode;
    float   noise_std;
    const float* raw;  const float* raw_fw;  const float* raw_bw;
    const float* zs;   const float* xyz;     const float* f_fw;  const float* f_bw;
    const float* noise_static;  const float* noise_transient;  const float* noise_fw;  const float* noise_bw;
    const float* g_static_sigmas;  const float* g_transient_sigmas;
    const float* g_static_weights; const float* g_transient_weights;  const float* g_weights;
    const float* g_depth;  const float* g_rgb;  const float* g_transient_alpha;  const float* g_transient_rgb;
    const float* g_so_rgb; const float* g_so_depth;
    const float* g_xyz_exp;  const float* g_flow_fw_exp;  const float* g_flow_bw_exp;
    const float* g_rgb_fw;   const float* g_rgb_bw;
    float* scratch;
    float* d_raw;  float* d_raw_fw;  float* d_raw_bw;  float* d_f_fw;  float* d_f_bw;
} NsffCompositeBwdArgs;
int nsff_composite_backward(const NsffCompositeBwdArgs* args, void* stream);

/* ---- N1: gradient of the scene-flow glue of `inference` (reference models/rendering.py:187-188 flows zeroed beyond
 * z = 0.95, :218/:224 warped points xyz + flow, :226-232 cycle points xyz_fw + flow_bw(xyz_fw)) w.r.t. the field record it
 * reads.  For every point p (depth zs[p]):
 *     out[p][col_a .. col_a+2] (+)= m * sum_k g_a[k][p][0..2],   out[p][col_b .. col_b+2] (+)= m * sum_k g_b[k][p][0..2],
 * m = zs[p] > z_far ? 0 : 1; accumulate = 0 also writes zeros to the other 10 columns of the (P,16) record (a whole-record
 * store), accumulate = 1 adds into the named columns only.  g_a / g_b: up to four (P,3) cotangents each (NULL = absent) -- the
 * consumers of one flow: the compositing node, the warped query, the loss.  col_a / col_b < 0 disables a group.  One launch
 * replaces autograd's where / slice / zeros / add chain (~12 small kernels per flow direction). */
typedef struct {
    int64_t n_points;
    const float* zs;            /* (P) sample depths                                          */
    float z_far;                /* 0.95 (rendering.py:187)                                    */
    int32_t accumulate;
    int32_t col_a, col_b;       /* first column of each 3-wide group in the 16-float record   */
    int32_t pad_;
    const float* g_a[4];
    const float* g_b[4];
    float* out;                 /* (P,16), 16-byte aligned                                    */
} NsffFlowGradArgs;
int nsff_flow_grad(const NsffFlowGradArgs* args, void* stream);

/* ---- profiling hooks used by bench.py (HIP events around field-query launches) ---- */
int nsff_prof_enable(int on);
/* Synchronises the recorded events; returns launches, summed milliseconds and summed
 * algorithmic FLOPs (2*MACs of the reference's Linear layers, unpadded K) since the last reset, and the FLOPs the
 * kernels executed for them: inference launches evaluate the heads that read the activation-free
 * *_xyz_encoding_final layers (nerf.py:170,195) with pre-multiplied rows and skip those 256x256 layers. */
int nsff_prof_collect(int64_t* launches, double* total_ms, double* total_flops, double* executed_flops);
/* The same, plus the shader clock the f16 / f16x3 field kernels ran at: while profiling is on a sample of every launch's
 * workgroups measure their own lifetime with s_memtime (shader-clock ticks) and s_memrealtime (the constant-rate wall clock,
 * hipDeviceAttributeWallClockRate); this call returns the summed shader ticks and the summed wall time of those lifetimes in
 * milliseconds: clock [GHz] = shader_ticks / ticks_ms * 1e-6 (0 / 0 when nothing was measured, e.g. exact-fp32 launches). */
int nsff_prof_collect_clock(int64_t* launches, double* total_ms, double* total_flops, double* executed_flops,
                            double* shader_ticks, double* ticks_ms);

/* Which kernel the LAST nsff_field_query of this process launched (diagnostics; tests assert that large inference launches run
 * the hand-scheduled body and not a fallback): */
#define NSFF_KERNEL_F32       1   /* nsff_field_kernel: exact fp32 MFMA                                               */
#define NSFF_KERNEL_H3_64     2   /* f16x3, 64-point tiles (launches below 32768 points, tile_points = 64)           */
#define NSFF_KERNEL_H3_8WAVE  3   /* f16x3, 128-point tiles, eight waves of 32 neurons, compiler-scheduled           */
#define NSFF_KERNEL_H3A       4   /* f16x3, 128-point tiles, hand-scheduled body (nsff_field_kernel_h3a)             */
#define NSFF_KERNEL_H3_SAVE   5   /* f16x3 training forward (keeps activations)                                      */
/*      (6: the single-product fast mode of earlier ABI versions, removed) */
#define NSFF_KERNEL_H3A_TBIAS 7   /* NSFF_KERNEL_H3A with the time code folded into per-ray bias rows (NsffFieldArgs::t_bias) */
#define NSFF_KERNEL_H3A_SAVE  9   /* f16x3 training forward on the hand-scheduled body (nsff_field_kernel_h3a_save)          */
#define NSFF_KERNEL_H3A_SIDE  8   /* NSFF_KERNEL_H3A[_TBIAS] whose static trunk has the view-direction branch (NsffFieldArgs::s_bias) */
int         nsff_last_field_kernel(void);
/* Workgroups of that launch when it ran the hand-scheduled inference kernel (0 otherwise).  Large launches of that kernel are
 * PERSISTENT: one workgroup per compute unit, each walking tiles  first, first + stride, ...  of one trunk (both trunks: the
 * workgroups of XCDs 0..3 take the static trunk, 4..7 the dynamic one -- taken when both trunks cost the same number of matrix
 * steps); the plain bias rows are loaded once per workgroup, the next tile's first eight weight slots are requested by the last
 * trunk phase of the current one and its point by the body's first instructions.  Results are bit-identical to the one-workgroup-per-tile form (environment
 * NSFF_NO_PERSIST=1, read per launch, selects that form for A/B measurements). */
int         nsff_last_field_grid(void);
/* Host-only (no GPU work): the f16x3 step program of an inference launch with these modes -- steps[n][4] = {weight segment
 * offset (words), bias offset (words; 0xFFFFFFFF = accumulate), nks | pre << 8 | post << 16 | head << 24, 0} -- and the phase
 * programs (8-dword descriptors, 36 at most per trunk) the hand-scheduled kernel would execute for its static / dynamic trunk;
 * n_phases[t] = 0 when trunk t is absent or not covered by that kernel.  Used by the tests that pin the host-side program
 * builder to the one the simulator runs (tools/h3asm/check.py).  steps: room for 28 x 4, phases_*: room for 36 x 8.
 * fold_t != 0: the dynamic trunk's program of a launch that was given NsffFieldArgs::t_bias (bias fields of its descriptors
 * then index a table whose per-ray rows follow the plain ones: half A's row replaces the layer's bias row, half B's is appended).
 * fold_t bit 1: the static trunk's program of a view-direction launch given NsffFieldArgs::s_bias; bit 2: the programs of a
 * PERSISTENT launch (nsff_last_field_grid): the last segment's B phase carries descriptor 0's stream fields and requests the
 * first segments' weight slots 0..7 for the workgroup's next tile (n_phases[t] = 0 for a trunk that ends with a skip layer). */
int         nsff_field_phase_program(const NsffModelDesc* desc, int static_mode, int transient_mode, int fold_t, uint32_t* steps,
                             int* n_steps, int* n_static_steps, uint32_t* phases_static, uint32_t* phases_dynamic, int* n_phases);

int         nsff_abi_version(void);
const char* nsff_last_hip_error(void);

#ifdef __cplusplus
}
#endif
#endif /* NSFF_RENDER_H */
