"""Native field node of the backward graph (SURVEY.md 8f, row N1 -- second stage).

The field network (PosEmbedding + NeRF.forward, reference models/nerf.py:17-30,118-213) inside the backward graph of
:mod:`nsff_pl_amd.autograd` is this ``autograd.Function``:

* forward  = the gfx950 f16x3 field kernel in its *training* variant (``nsff_field_kernel_h3<2,1,true>``),
  which also keeps every post-activation tensor of the trunks (fp16) and the encoded trunk input in HBM
  -- 288 GB per GPU make keeping ~4.6 KB per point-trunk cheaper than any recomputation;
* backward = two hand-written gfx950 kernels (csrc/field_bwd.hip), fp16 MFMA operands with fp32 accumulation:
  ``nsff_field_backward`` runs the whole data-gradient chain of a 64-point tile in one workgroup (head
  derivatives, dX = dY.W with the transposed weights streamed from L2, ReLU masks from the forward's sign
  bits) and ``nsff_weight_grad`` the batched split-K weight-gradient GEMMs (dW = dY^T.X, K = all points) over
  the fragment-major tiles both passes left in HBM.  Every point's gradient row carries its own power-of-two
  scale (block floating point); the positional-encoding derivative and the per-ray reduction of the time-code
  gradient are one more launch on the (P, xin_rows) result (``nsff_field_input_backward``).

View-direction / appearance models (``static_dir_encoding``, reference nerf.py:83-91,183-185) are covered: the
training forward also keeps that layer's input rows and activation, K1 differentiates it and returns the
gradient of the per-ray appearance code.  Every architecture the inference kernels take trains as well: any list
of skip layers (nerf.py:34-40,163-167; the lowest one's input share stays in LDS, further ones add theirs to d_xin
through memory), position / time embeddings up to ceil64(in_xyz) + ceil64(in_t) <= 256 columns and [dir | a] inputs
up to 256 (``_lib.train_dims``: the saved input tiles have 128 or 256 rows).  What is refused, by name
(``why_unsupported``): widths other than 256, and the gradient with respect to the points of a STATIC trunk (never
needed by ``render_rays``: only warped dynamic points carry gradients) -- there is no torch fallback in the product.
"""
import os

import numpy as np

import torch

from . import _lib, config



def enabled():
    return os.environ.get("NSFF_NATIVE_BACKWARD", "1") != "0"


def why_unsupported(model):
    """None, or what csrc/field_bwd.hip and the training forward of csrc/field_h3.hip are not built for."""
    if model.W != 256:
        return f"W={model.W} (the field kernels tile W=256 trunks)"
    skips = sorted(set(int(v) for v in model.skips))
    if not 2 <= model.D <= 8 or any(not 1 <= v < model.D for v in skips):
        return f"D={model.D}, skips={list(model.skips)} (need 2 <= D <= 8 and skip layers among 1..D-1)"
    pad = lambda n: (int(n) + 63) // 64 * 64
    if pad(model.in_channels_xyz) + (pad(model.in_channels_t) if model.encode_transient else 0) > 256:
        return (f"in_channels_xyz={model.in_channels_xyz}, in_channels_t={model.in_channels_t}: the trunk input is padded to "
                "64-column segments and must fit 256 columns (inference has the same limit)")
    if model.use_viewdir and model.in_channels_dir + model.in_channels_a > 256:
        return f"in_channels_dir + in_channels_a = {model.in_channels_dir + model.in_channels_a} > 256"
    return None


def _kernel_handles(model):
    return why_unsupported(model) is None


def supported(model, xyz):
    return enabled() and xyz.is_cuda and xyz.dtype == torch.float32 and _kernel_handles(model)


def n_slots(model):
    """Activation / pre-activation-gradient slots: trunk t layer l -> t*(D+1)+l; slot D of the STATIC trunk: static_dir_encoding
    (view-direction models).  (Slot l = D of a trunk was *_xyz_encoding_final's in earlier versions; the layer is folded into the
    heads that read it, forward and backward, so the dynamic trunk's slot D is never written.)"""
    return 2 * model.D + 2


def _lin(m):
    return m[0] if isinstance(m, torch.nn.Sequential) else m


def _planes(x3, *shape, device):
    """An fp16 buffer of the backward pass; x3 (config.set_grad_precision("f16x3")): with its remainder plane directly behind it --
    the caller gets plane 0 (`_lib.lo_delta` finds the twin)."""
    if not x3:
        return torch.empty(*shape, device=device, dtype=torch.float16), None
    both = torch.empty(2, *shape, device=device, dtype=torch.float16)
    return both[0], both


def alloc_saves(model, n_points, device, transient, static=True, x3=False):
    """Buffers the training forward fills for the backward kernels (layouts: include/nsff_render.h, NsffFieldArgs):
    (acts, xin, masks, side) -- side is None unless the launch evaluates static_dir_encoding."""
    tiles = (n_points + 63) // 64
    xin_rows, t_row0, side_rows = _lib.train_dims(model)
    acts, _ = _planes(x3, n_slots(model), tiles, 64 * 256, device=device)
    xin, xin2 = _planes(x3, tiles, 64 * xin_rows, device=device)
    masks = torch.empty(n_slots(model), tiles, 256, device=device, dtype=torch.int64)
    written = t_row0 + ((model.in_channels_t + 63) // 64 * 64 if transient else 0)
    if written < xin_rows:
        (xin2 if x3 else xin).zero_()        # rows the launch does not encode are never written
    side = None
    if model.use_viewdir and static:
        side, side2 = _planes(x3, tiles, 64 * side_rows, device=device)
        if (model.in_channels_dir + model.in_channels_a + 63) // 64 * 64 < side_rows:
            (side2 if x3 else side).zero_()
    return acts, xin, masks, side


def forward_can_save(model, static_mode, transient_mode):
    """True when render_rays' own forward launch can already be the training forward (so that backward does not
    re-run it): f16x3 arithmetic selected, no view directions, full (rgb+sigma) modes."""
    return (enabled() and _kernel_handles(model) and config.precision_code(model) == config.PRECISIONS["f16x3"]
            and static_mode in (0, 2) and transient_mode in (0, 2) and (static_mode or transient_mode))


class _FieldFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cfg, xyz, t_rows, dir_rows, a_rows, *params):
        model, freqs, s = cfg["model"], cfg["freqs"], cfg["pts_per_ray"]
        static, transient = cfg["static"], cfg["transient"]
        P = xyz.shape[0]
        ctx.cfg, ctx.P = cfg, P
        x3 = ctx.x3 = bool(cfg.get("x3"))        # (decided where the saves were / are allocated: field())
        if cfg.get("saved") is not None:         # render_rays' own launch was the training forward: nothing to redo
            raw, acts, xin, masks, xyz_c, side = cfg["saved"]
            ctx.has_side = side is not None
            ctx.save_for_backward(raw, acts, xin, masks, xyz_c, *([side] if side is not None else []), *params)
            return raw.detach().view_as(raw)
        dev = xyz.device
        raw = torch.empty(P, _lib.RAW_STRIDE, device=dev)      # (the kernel writes whole records, zeros in unevaluated slots)
        acts, xin, masks, side = alloc_saves(model, P, dev, transient, static, x3)
        xyz_c = xyz.detach().contiguous()
        use_side = model.use_viewdir and static
        _lib.field_query(model, raw, P, s, 2 if static else 0, 2 if transient else 0,
                         2 if (transient and model.output_flow) else 0, xyz=xyz_c, freqs=freqs,
                         t_emb=None if t_rows is None else t_rows.detach().contiguous(),
                         dir_emb=dir_rows.detach().contiguous() if use_side else None,
                         a_emb=a_rows.detach().contiguous() if (use_side and model.in_channels_a > 0) else None,
                         save_acts=acts, save_xin=xin, save_masks=masks, save_side=side,
                         precision=config.PRECISIONS["f16x3"], save_lo=x3)
        ctx.has_side = side is not None
        ctx.save_for_backward(raw, acts, xin, masks, xyz_c, *([side] if side is not None else []), *params)
        return raw

    @staticmethod
    def backward(ctx, d_raw):
        cfg = ctx.cfg
        if cfg["static"] and ctx.needs_input_grad[1]:
            raise NotImplementedError("field(): gradient w.r.t. the points of a STATIC trunk is not built (render_rays never "
                                      "asks for it: only the warped points of the dynamic trunk carry gradients)")
        model, freqs, s = cfg["model"], cfg["freqs"], cfg["pts_per_ray"]
        raw, acts, xin, masks, xyz = ctx.saved_tensors[:5]
        side = ctx.saved_tensors[5] if ctx.has_side else None
        params = ctx.saved_tensors[6 if ctx.has_side else 5:]
        P, D = ctx.P, model.D
        xin_rows, t_row0, side_rows = _lib.train_dims(model)
        tiles, dev = acts.shape[1], d_raw.device
        if P == 0:                               # an empty batch contributes nothing
            return (None, None if not ctx.needs_input_grad[1] else torch.zeros_like(xyz), None, None, None) + (None,) * len(params)
        static, transient = cfg["static"], cfg["transient"]
        viewdir = bool(model.use_viewdir and static)
        S_DIR = D
        n_xyz, n_t = model.in_channels_xyz, (model.in_channels_t if transient else 0)
        d_raw = d_raw.contiguous()
        gmax = _lib.absmax(d_raw)                 # [static, dynamic]: one power-of-two scale per trunk (see nsff_absmax_raw)
        x3 = ctx.x3
        dpre, _ = _planes(x3, n_slots(model), tiles, 64 * 256, device=dev)
        dhead = torch.empty(2, tiles, 64 * 32, device=dev, dtype=torch.float16)
        want_in = transient and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        d_xin = torch.empty(P, xin_rows, device=dev) if want_in else None
        want_a = viewdir and model.in_channels_a > 0 and ctx.needs_input_grad[4]
        d_side = torch.empty(P, side_rows, device=dev) if want_a else None
        _lib.field_backward(model, P, static, transient, d_raw, raw, gmax, masks, dpre, dhead, d_xin, d_side, x3=x3)

        # ---- weight-gradient GEMMs: one batched launch per output shape ----
        meta = _wgrad_jobs(model, static, transient)             # (kind, trunk, layer)
        jobs, sizes = [], []
        for kind, t, l in meta:
            base = t * (D + 1)
            if kind == "x":
                a_, b_, rows = dpre[base + l], xin, (256, xin_rows)
            elif kind == "h":
                a_, b_, rows = dpre[base + l], acts[base + l - 1], (256, 256)
            elif kind == "dir_h":                                 # (folded with *_final: reads the last trunk activation)
                a_, b_, rows = dpre[S_DIR], acts[D - 1], (256, 256)
            elif kind == "dir_x":
                a_, b_, rows = dpre[S_DIR], side, (256, side_rows)
            elif l == 1:                                          # static sigma reads the trunk
                a_, b_, rows = dhead[0], acts[base + D - 1], (32, 256)
            else:                                                 # the (folded) heads read the last trunk activation as well
                a_, b_, rows = dhead[t], acts[S_DIR if (t == 0 and viewdir) else base + D - 1], (32, 256)
            jobs.append([a_.data_ptr(), b_.data_ptr(), rows[0], rows[1], 0, t])      # (t: the trunk whose scale gmax[t] dpre / dhead are on)
            if x3:                               # the operands' remainder planes (dhead carries its own: rows 16..31)
                b_src = xin if kind == "x" else (side if kind == "dir_x" else acts)
                jobs[-1] += [_lib.lo_delta(dpre) if rows[0] == 256 else 0, _lib.lo_delta(b_src)]
            sizes.append(rows[0] * rows[1])
        # requested split-K factor; the library rounds it to whole rounds of the 256 CUs (16 -> one round of 14 splits x 18 jobs:
        # 7.52 ms per C2 step against 7.63 at 32 = two rounds, 7.79 at 48: fewer partial sums for the accumulate pass to read)
        n_splits = max(1, min(int(os.environ.get("NSFF_WGRAD_SPLITS", "16")), tiles // 4))
        off = 0
        for j, sz in zip(jobs, sizes):
            j[4] = off
            off += sz
        # The weight-gradient GEMMs depend only on this node's K1 output and nothing downstream needs them before the
        # optimizer: they run on a side stream (HBM-bound, they overlap with the next node's store-bound K1 and the
        # small torch kernels) and a callback at the end of the backward pass adds them to .grad.
        overlap = _overlap_enabled()
        main = torch.cuda.current_stream()
        # inside a hipGraph capture the fork / join costs more than the concurrency returns (13.98 vs 12.56 ms per
        # step): there only the accumulation is deferred and fused, on the capture stream itself
        wstream = _side_stream(dev) if (overlap and not torch.cuda.is_current_stream_capturing()) else main
        if wstream is not main:
            wstream.wait_stream(main)
        plist = _lib.param_list(model)
        grad_map = _grad_map(model, static, transient, meta, jobs, sizes, plist) if overlap else None
        with torch.cuda.stream(wstream):
            if grad_map is not None:
                # deferred mode with every .grad in place: the reduction of the split-K partials accumulates straight
                # into the parameters' gradient memory (no per-parameter tensors, adds or cats) -- except the FOLDED
                # parameters (*_final, the heads / the view-direction layer that read it): the same launch leaves their jobs'
                # dense sums in `aux`, one more launch per trunk (nsff_fold_grads) turns them into gradients in place
                table, base, aux_at, aux_total = grad_map
                aux = torch.empty(aux_total, device=dev)
                keep = _lib.weight_grad_accumulate([tuple(j) for j in jobs], tiles, n_splits, table, base, gmax, aux=aux)
                dense = lambda i: aux[aux_at[i]:aux_at[i] + sizes[i]].view(jobs[i][2], jobs[i][3])
                dense_bias = lambda i: aux[aux_at[i] + sizes[i]:aux_at[i] + sizes[i] + 256]
                _fold_in_place(model, static, transient, meta, plist, dense, dense_bias)
                keep = (keep, aux)
                grads = None
            else:
                out = torch.empty(off, device=dev)
                bias = torch.empty(len(jobs), 256, device=dev)
                _lib.weight_grad([tuple(j) for j in jobs], tiles, n_splits, out, bias, gmax)  # summed and unscaled in there
                keep = (out, bias)
                result = lambda i: out[jobs[i][4]:jobs[i][4] + sizes[i]].view(jobs[i][2], jobs[i][3])
                grads = _assemble(model, static, transient, meta, plist, result, lambda i: bias[i], torch.cat)
                for i, g_ in _folded_grads(model, static, transient, meta, plist, result, lambda i: bias[i]).items():
                    grads[i] = g_

        d_xyz = d_t = d_a = None
        if d_xin is not None:              # derivative of the positional encoding + per-ray sum of the time-code rows
            d_xyz, d_t = _lib.field_input_backward(d_xin, t_row0, xyz, s, freqs, n_t, ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        if d_side is not None:               # per-ray appearance code: sum over the ray's points (rendering.py:168,172)
            c0 = model.in_channels_dir
            d_a = d_side[:, c0:c0 + model.in_channels_a].reshape(P // s, s, -1).sum(1)
        if overlap:
            # keep what the side stream still reads alive until the join, then hand the gradients over there.
            # Every node queues the (idempotent) flush: a callback queued by an earlier backward pass that died
            # half-way is dropped by the engine, so "already queued" cannot be remembered across passes.
            _PENDING.append((_graph_task_id(), list(plist), grads, (dpre, dhead, acts, xin, side, keep, gmax)))
            torch.autograd.Variable._execution_engine.queue_callback(_flush_weight_grads)
            return (None, d_xyz, d_t, None, d_a) + (None,) * len(params)
        return (None, d_xyz, d_t, None, d_a) + tuple(grads)


def job_shape(model, kind):
    """(a_rows, b_rows) of a weight-gradient GEMM of `_wgrad_jobs`."""
    xin_rows, _, side_rows = _lib.train_dims(model)
    return {"x": (256, xin_rows), "h": (256, 256), "dir_h": (256, 256), "dir_x": (256, side_rows), "head": (32, 256)}[kind]


def _skips(model):
    return sorted(set(int(v) for v in model.skips))


def _wgrad_jobs(model, static, transient):
    """The weight-gradient GEMMs of one node as (kind, trunk, layer) tags: 'x' = trunk-input part of layer l (layer 0 and
    every skip layer), 'h' = hidden part of layer l, 'dir_h' / 'dir_x' = the two parts of static_dir_encoding, 'head' = the
    output heads (layer 1: static_sigma of a view-direction model -- its rgb head reads static_dir_encoding, sigma the trunk;
    without view directions both read the last trunk activation and share ONE job).  *_xyz_encoding_final has no job: it is
    folded into the heads / into 'dir_h', whose jobs therefore multiply with the LAST TRUNK activation (_folded_grads)."""
    D, skips = model.D, _skips(model)
    viewdir = bool(model.use_viewdir and static)
    meta = []
    for t in ([0] if static else []) + ([1] if transient else []):
        for l in range(D):
            if l == 0:
                meta.append(("x", t, 0))
            else:
                meta.append(("h", t, l))
                if l in skips:
                    meta.append(("x", t, l))
        if t == 0 and viewdir:            # static_dir_encoding: [*_final | dir | a] -> 256, static_rgb reads it
            meta += [("dir_h", 0, 0), ("dir_x", 0, 0)]
        meta.append(("head", t, 0))
        if t == 0 and viewdir:
            meta.append(("head", 0, 1))
    return meta


def _assemble(model, static, transient, meta, plist, result, bias_of, cat):
    """Gradients of every parameter of `model` (order of `plist`) from the weight-gradient jobs of one node.
    result(i): the (a_rows, b_rows) matrix of job i; bias_of(i): its 256 row sums; `meta[i]` = (kind, trunk, layer).
    Works on tensors and on the index objects of :func:`_grad_map` alike (only slicing, `+` and `cat` are used)."""
    D, skips = model.D, _skips(model)
    viewdir = bool(model.use_viewdir and static)
    n_xyz, n_t = model.in_channels_xyz, (model.in_channels_t if transient else 0)
    _, t_row0, _ = _lib.train_dims(model)
    index = {id(q): i for i, q in enumerate(plist)}
    grads = [None] * len(plist)
    res = {tag: i for i, tag in enumerate(meta)}

    def put(layer, w, b):
        grads[index[id(layer.weight)]] = w
        grads[index[id(layer.bias)]] = b

    def xcols(m, in_t):                       # (256, xin_rows) in trunk-input rows -> (256, in_dim) in Linear columns
        return m[:, :n_xyz] if in_t == 0 else cat([m[:, :n_xyz], m[:, t_row0:t_row0 + in_t]], 1)
    for t in ([0] if static else []) + ([1] if transient else []):
        prefix, in_t = ("static", 0) if t == 0 else ("transient", n_t)
        for l in range(D):
            layer = _lin(getattr(model, f"{prefix}_xyz_encoding_{l + 1}"))
            if l == 0:
                i = res[("x", t, 0)]
                put(layer, xcols(result(i), in_t), bias_of(i))
            elif l in skips:
                i = res[("h", t, l)]
                put(layer, cat([xcols(result(res[("x", t, l)]), in_t), result(i)], 1), bias_of(i))
            else:
                i = res[("h", t, l)]
                put(layer, result(i), bias_of(i))
        # (*_final, the heads that read it and static_dir_encoding come from _folded_grads: products, not slices)
        i = res[("head", t, 0)]
        hw, hb = result(i), bias_of(i)
        hw, hb = hw[0:16] + hw[16:32], hb[0:16] + hb[16:32]      # fp16 value + rounding remainder rows
        if t == 0 and viewdir:            # static_rgb reads static_dir_encoding, static_sigma the trunk: plain slices
            put(_lin(model.static_rgb), hw[0:3], hb[0:3])
            i2 = res[("head", 0, 1)]
            put(_lin(model.static_sigma), result(i2)[3:4] + result(i2)[19:20], bias_of(i2)[3:4] + bias_of(i2)[19:20])
        elif t == 0:                      # static_sigma reads the last trunk activation like the folded rgb rows: row 3 of their job
            put(_lin(model.static_sigma), hw[3:4], hb[3:4])
    return grads


def _fold_job_indices(model, static, transient, meta):
    """Jobs whose dense sums _folded_grads needs."""
    viewdir = bool(model.use_viewdir and static)
    res = {tag: i for i, tag in enumerate(meta)}
    sel = []
    for t in ([0] if static else []) + ([1] if transient else []):
        if t == 0 and viewdir:
            sel += [res[("dir_h", 0, 0)], res[("dir_x", 0, 0)]]
        else:
            sel.append(res[("head", t, 0)])
    return sel


def _fold_heads(model, t):
    """[(head module, first row, end row)] of the folded heads of trunk t (rows of the heads' job), in row order."""
    if t == 0:
        return [(model.static_rgb, 0, 3)]
    heads = [(model.transient_rgb, 0, 3), (model.transient_sigma, 3, 4)]
    if model.output_flow:
        heads += [(model.transient_flow_fw, 4, 7), (model.transient_flow_bw, 7, 10)]
    return heads


def _fold_in_place(model, static, transient, meta, plist, dense, dense_bias):
    """The folded parameters' gradients ADDED to their .grad (every one exists: the in-place path): one nsff_fold_grads launch
    per trunk (the heads' rows are addressed one by one: they live in four modules); the view-direction static trunk -- its
    folded layer is static_dir_encoding, 256 rows of one strided matrix -- through nsff_fold_grads_dense, accumulating."""
    viewdir = bool(model.use_viewdir and static)
    res = {tag: i for i, tag in enumerate(meta)}
    tensor_trunks = set()
    for t in ([0] if static else []) + ([1] if transient else []):
        prefix = "static" if t == 0 else "transient"
        fin = _lin(getattr(model, f"{prefix}_xyz_encoding_final"))
        if t == 0 and viewdir:
            layer = _lin(model.static_dir_encoding)
            involved = [fin.weight, fin.bias, layer.weight, layer.bias]
            if not all(q.requires_grad and q.grad is not None for q in involved):
                tensor_trunks.add(t)
                continue
            ih, ix = res[("dir_h", 0, 0)], res[("dir_x", 0, 0)]
            n_side = model.in_channels_dir + model.in_channels_a
            _lib.fold_grads_dense(dense(ih), dense_bias(ih), layer.weight.detach()[:, :256], fin.weight.detach(), fin.bias.detach(),
                                  layer.weight.grad[:, :256], layer.bias.grad, fin.weight.grad, fin.bias.grad, accumulate=True)
            layer.weight.grad[:, 256:].add_(dense(ix)[:, :n_side])      # (the [dir | a] columns: an ordinary job's slice)
            continue
        heads = _fold_heads(model, t)
        involved = [fin.weight, fin.bias] + [q for m, _, _ in heads for q in (_lin(m).weight, _lin(m).bias)]
        if not all(q.requires_grad and q.grad is not None for q in involved):
            tensor_trunks.add(t)
            continue
        i = res[("head", t, 0)]
        rows = []
        for m, a, b in heads:
            lin = _lin(m)
            for r in range(b - a):
                rows.append((lin.weight.detach()[r], lin.weight.grad[r], lin.bias.grad[r:r + 1]))
        _lib.fold_grads(dense(i), dense_bias(i), fin.weight.detach(), fin.bias.detach(), rows, fin.weight.grad, fin.bias.grad)
    if tensor_trunks:       # (some parameter of the fold is frozen or has no .grad yet: results as tensors, added where a .grad exists)
        with torch.no_grad():
            for i, g_ in _folded_grads(model, static, transient, meta, plist, dense, dense_bias, only=tensor_trunks).items():
                if plist[i].requires_grad:
                    plist[i].grad.add_(g_)


def _folded_grads(model, static, transient, meta, plist, result, bias_of, only=None):
    """{index in plist: gradient tensor} of the parameters the fold touches.  *_xyz_encoding_final is a Linear without activation
    (reference nerf.py:170,195) that the kernels never execute: a head that reads it is evaluated as (W_head W_final) h + (W_head
    b_final + b_head) on the last trunk activation h, and the backward kernels differentiate that folded map.  With
    G = sum_p dpre_p (x) h_p (the folded head's weight gradient, a job of the weight-gradient GEMM) and gb = sum_p dpre_p:
        dW_head = G W_final^T + gb (x) b_final      db_head = gb
        dW_final = W_head^T G                       db_final = W_head^T gb
    -- exactly what autograd of the two layers gives (tests/torch_path.py::folded_grads_reference is this algebra in torch; a CPU
    test holds it to float64 autograd, a GPU test holds this function to it).  The view-direction layer static_dir_encoding
    (which reads [*_final | dir | a]) takes W_head's place for the static trunk of such a model; its [dir | a] columns are an
    ordinary job.  The products run in nsff_fold_grads_dense (two launches per trunk, fp32, deterministic).
    result(i) / bias_of(i): dense (a_rows, b_rows) sum and row sums of job i (fp32 GPU tensors)."""
    viewdir = bool(model.use_viewdir and static)
    index = {id(q): i for i, q in enumerate(plist)}
    res = {tag: i for i, tag in enumerate(meta)}
    out = {}
    for t in ([0] if static else []) + ([1] if transient else []):
        if only is not None and t not in only:
            continue
        prefix = "static" if t == 0 else "transient"
        fin = _lin(getattr(model, f"{prefix}_xyz_encoding_final"))
        w_f, b_f = fin.weight.detach().contiguous(), fin.bias.detach().contiguous()
        d_wf, d_bf = torch.empty_like(w_f), torch.empty_like(b_f)
        if t == 0 and viewdir:
            ih, ix = res[("dir_h", 0, 0)], res[("dir_x", 0, 0)]
            layer = _lin(model.static_dir_encoding)
            n_side = model.in_channels_dir + model.in_channels_a
            d_w = torch.empty_like(layer.weight)
            d_b = torch.empty_like(layer.bias)
            _lib.fold_grads_dense(result(ih).contiguous(), bias_of(ih).contiguous(), layer.weight.detach()[:, :256], w_f, b_f,
                                  d_w[:, :256], d_b, d_wf, d_bf, accumulate=False)
            d_w[:, 256:] = result(ix)[:, :n_side]
            out[index[id(layer.weight)]], out[index[id(layer.bias)]] = d_w, d_b
            out[index[id(fin.weight)]], out[index[id(fin.bias)]] = d_wf, d_bf
            continue
        i = res[("head", t, 0)]
        hw, hb = result(i), bias_of(i)                                            # rows r and 16 + r: fp16 value + rounding remainder
        heads = _fold_heads(model, t)
        n_rows = heads[-1][2]                                                      # (the folded rows are rows 0 .. R - 1 of the job)
        w_h = torch.cat([_lin(m).weight.detach() for m, _, _ in heads], 0)        # (R, 256)
        d_heads, d_hb = torch.empty_like(w_h), torch.empty(n_rows, device=w_h.device, dtype=w_h.dtype)
        _lib.fold_grads_dense(hw[0:n_rows].contiguous(), hb[0:n_rows].contiguous(), w_h, w_f, b_f, d_heads, d_hb, d_wf, d_bf,
                              accumulate=False, g2=hw[16:16 + n_rows].contiguous(), gb2=hb[16:16 + n_rows].contiguous())
        k = 0
        for m, a, b in heads:
            lin = _lin(m)
            out[index[id(lin.weight)]] = d_heads[k:k + (b - a)]
            out[index[id(lin.bias)]] = d_hb[a:b]
            k += b - a
        out[index[id(fin.weight)]], out[index[id(fin.bias)]] = d_wf, d_bf
    return out


class _Idx:
    """Stand-in for a gradient matrix that remembers where each element comes from: (job, element) of the primary
    source and, after a `+`, of a second one.  Slicing and concatenation behave like the tensor's."""

    def __init__(self, ja, ea, jb=None, eb=None):
        self.ja, self.ea = ja, ea
        self.jb = np.full_like(ja, -1) if jb is None else jb
        self.eb = np.zeros_like(ea) if eb is None else eb

    def __getitem__(self, key):
        return _Idx(self.ja[key], self.ea[key], self.jb[key], self.eb[key])

    def __add__(self, other):
        assert (self.jb < 0).all() and (other.jb < 0).all() and self.ja.shape == other.ja.shape
        return _Idx(self.ja, self.ea, other.ja, other.ea)

    @staticmethod
    def cat(items, dim):
        return _Idx(*[np.concatenate([getattr(it, f) for it in items], dim) for f in ("ja", "ea", "jb", "eb")])


_GRAD_MAPS = {}


def _grad_map(model, static, transient, meta, jobs, sizes, plist):
    """(device (n,4) int32 NsffGradMapEntry rows, base address) telling nsff_weight_grad_accumulate where every
    gradient element of this node lives, or None when some parameter has no fp32 contiguous .grad of its own shape yet
    (first backward of a caller that does not pre-allocate: the node then hands tensors to the flush instead)."""
    if os.environ.get("NSFF_WGRAD_INPLACE", "1") == "0":          # debug switch (A/B against the tensor hand-over)
        return None
    live = [(i, p) for i, p in enumerate(plist) if p.requires_grad]
    for _, p in live:
        g = p.grad
        if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.shape != p.shape or not g.is_cuda:
            return None
    if not live:
        return None
    ptrs = tuple(p.grad.data_ptr() for _, p in live)
    shape_key = tuple((j[2], j[3]) for j in jobs)
    key = (id(model), bool(static), bool(transient), tuple(meta), shape_key, ptrs)
    hit = _GRAD_MAPS.get(key)
    if hit is not None:
        return hit
    base = min(ptrs)
    if (max(ptrs) - base) // 4 + max(p.numel() for _, p in live) >= 2 ** 31:
        return None
    if torch.cuda.is_current_stream_capturing():      # the map is built (host work + an upload) by the eager warm-up
        return None

    def result(i):
        rows, cols = jobs[i][2], jobs[i][3]
        return _Idx(np.full((rows, cols), i, np.int16), np.arange(rows * cols, dtype=np.int32).reshape(rows, cols))

    def bias_of(i):
        return _Idx(np.full(256, i, np.int16), sizes[i] + np.arange(256, dtype=np.int32))
    srcs = _assemble(model, static, transient, meta, plist, result, bias_of, _Idx.cat)
    rows = []
    for i, p in live:
        src = srcs[i]
        if src is None:
            continue
        assert src.ja.shape == tuple(p.shape), (src.ja.shape, tuple(p.shape))
        n = p.numel()
        ent = np.empty((n, 4), np.int32)
        ent[:, 0] = (p.grad.data_ptr() - base) // 4 + np.arange(n, dtype=np.int64)
        ent[:, 1] = src.ea.reshape(-1)
        ent[:, 2] = src.eb.reshape(-1)
        pair = np.stack([src.ja.reshape(-1).astype(np.int16), src.jb.reshape(-1).astype(np.int16)], 1)   # little-endian
        ent[:, 3] = np.ascontiguousarray(pair).view(np.int32).reshape(-1)
        rows.append(ent)
    # the jobs the folded parameters are made of: their dense sums (matrix, then the 256 row sums) go to `aux` (dst < 0)
    aux_at, aux_total = {}, 0
    for j in _fold_job_indices(model, static, transient, meta):
        n = sizes[j] + 256
        ent = np.empty((n, 4), np.int32)
        ent[:, 0] = -(1 + aux_total + np.arange(n, dtype=np.int64))
        ent[:, 1] = np.arange(n, dtype=np.int32)
        ent[:, 2] = 0
        pair = np.stack([np.full(n, j, np.int16), np.full(n, -1, np.int16)], 1)
        ent[:, 3] = np.ascontiguousarray(pair).view(np.int32).reshape(-1)
        rows.append(ent)
        aux_at[j] = aux_total
        aux_total += n
    table = torch.from_numpy(np.concatenate(rows, 0)).to(plist[0].device)
    if len(_GRAD_MAPS) > 64:
        _GRAD_MAPS.clear()
    _GRAD_MAPS[key] = (table, base, aux_at, max(aux_total, 1))
    return _GRAD_MAPS[key]


_PENDING = []            # (parameters, gradients, buffers to keep alive) of the field nodes of the running backward pass
_SIDE = {}
_DEFER = [False]


class deferred_weight_grads:
    """Context manager (used by ``NSFFTrainer.step``): inside it the weight-gradient GEMMs of every field node run
    on a side stream and are ADDED TO ``.grad`` by an end-of-backward callback instead of being returned through
    autograd -- so ``torch.autograd.grad(loss, params)`` and post-accumulate hooks do not see them.  Outside (the
    default) the node returns its gradients like any other autograd function.  ``NSFF_WGRAD_OVERLAP=1`` / ``0``
    forces the behaviour on / off everywhere."""

    def __enter__(self):
        self.old, _DEFER[0] = _DEFER[0], True

    def __exit__(self, *exc):
        _DEFER[0] = self.old
        return False


def _overlap_enabled():
    env = os.environ.get("NSFF_WGRAD_OVERLAP")
    return _DEFER[0] if env is None else env != "0"


def _graph_task_id():
    """Identity of the running backward pass (-1 outside one): what a pending entry belongs to."""
    return torch._C._current_graph_task_id()


def drop_stale_pending():
    """Release what a backward pass that raised half-way left behind (the engine drops that pass's queued end-of-pass
    callbacks).  Never needed for correctness -- :func:`_flush_weight_grads` only delivers the entries of the pass it
    runs in and discards everything else -- it just frees the buffers early (``NSFFTrainer.step`` calls it).  Not called
    from the field's forward: a forward may legitimately run INSIDE a backward pass (activation recomputation, a hook
    that renders) and must not discard the gradients that pass has already queued."""
    tid = _graph_task_id()
    _PENDING[:] = [e for e in _PENDING if e[0] == tid and tid >= 0]


def _side_stream(device):
    if device not in _SIDE:                  # created by the first (eager) backward, i.e. before any graph capture
        _SIDE[device] = torch.cuda.Stream(device=device)
    return _SIDE[device]


def _flush_weight_grads():
    """End-of-backward callback: join the side stream and add the weight gradients of every field node to .grad
    (one fused add per node instead of one per parameter).  Gradients therefore reach the parameters through
    ``loss.backward()``; ``torch.autograd.grad(..., parameters)`` needs NSFF_WGRAD_OVERLAP=0."""
    tid = _graph_task_id()
    items = [e[1:] for e in _PENDING if e[0] == tid]
    # only THIS pass's entries leave the list: an enclosing backward that is still running (re-entrant backward, checkpoint
    # recompute, a hook that renders and differentiates) keeps its own; entries of passes that died are drop_stale_pending's
    _PENDING[:] = [e for e in _PENDING if e[0] != tid]
    for dev, side in _SIDE.items():           # always: the optimizer step must not race the side-stream accumulation
        torch.cuda.current_stream(dev).wait_stream(side)
    if not items:
        return
    with torch.no_grad():
        for plist, grads, _keep in items:
            if grads is None:                # already accumulated in place by nsff_weight_grad_accumulate
                continue
            have, new = [], []
            for p, g in zip(plist, grads):
                if g is None or not p.requires_grad:
                    continue
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    have.append(p.grad)
                    new.append(g)
            if have:
                torch._foreach_add_(have, new)


def field(model, xyz, freqs, t_rows, pts_per_ray, static, transient, saved=None, dir_rows=None, a_rows=None):
    """Differentiable field query on raw points: returns the (P,16) raw record (layout of include/nsff_render.h).
    dir_rows / a_rows: per-ray view-direction embedding and appearance code (use_viewdir models, static pass).
    saved: (raw, acts, xin, masks, xyz, side[, x3]) of an earlier training-forward launch on exactly these inputs, or None (x3:
    the launch wrote the remainder planes of the three-product backward, config.set_grad_precision)."""
    x3 = config.grad_x3()
    if saved is not None:
        x3 = bool(saved[6]) if len(saved) > 6 else False
        saved = tuple(saved[:6])
    cfg = dict(model=model, freqs=[float(f) for f in freqs], pts_per_ray=int(pts_per_ray), static=bool(static),
               transient=bool(transient), saved=saved, x3=x3)
    return _FieldFn.apply(cfg, xyz, t_rows, dir_rows, a_rows, *_lib.param_list(model))
