"""``render_rays`` / ``sample_pdf`` with the reference's signatures, on the gfx950 kernels.

Drop-in for ``models/rendering.py:10-49`` (sample_pdf) and ``:52-362`` (render_rays) of
kwea123/nsff_pl: same positional/keyword arguments, same result keys, shapes and dtypes,
same torch RNG draw order.  This module contains no arithmetic of the path itself: it
resolves flags, draws the random tensors with torch exactly where the reference does,
allocates the result tensors and enqueues the C-ABI stages of ``include/nsff_render.h``
on the current stream:

    coarse_samples -> field_query(coarse) -> composite -> fine_samples
                   -> field_query(fine) [-> warp_points -> field_query x2] -> composite

The eval-time frustum-visibility mask (rendering.py:190-200) is part of the compositing kernel;
:mod:`nsff_pl_amd.ray_geometry` only prepares its camera table (once per dataset, no host sync per call).
"""
import os

import torch

from . import _lib
from . import autograd
from . import config
from . import field_grad
from . import ray_geometry

Z_FAR = 0.95  # flows are zeroed beyond this depth (reference rendering.py:316)

def _new(ref, *shape):
    return torch.empty(*shape, device=ref.device, dtype=torch.float32)


def sample_pdf(bins, weights, N_importance, det=False, eps=1e-5):
    """Inverse-CDF sampling (reference rendering.py:10-49).

    bins: (N_rays, M+1), weights: (N_rays, M) -> (N_rays, N_importance).
    ``det=False`` draws ``torch.rand(N_rays, N_importance)`` like the reference.
    """
    _lib.require_gpu_tensor(bins, "sample_pdf bins")
    bins = bins.contiguous().float()
    weights = weights.contiguous().float()
    n_rays = weights.shape[0]
    if det:
        u = torch.linspace(0, 1, N_importance, device=bins.device)
    else:
        u = torch.rand(n_rays, N_importance, device=bins.device)
    out = _new(bins, n_rays, N_importance)
    if n_rays:
        _lib.sample_pdf(bins, weights, u.contiguous(), 0 if det else 1, eps, out)
    return out


_TORCH_RAND, _TORCH_RANDN = torch.rand, torch.randn


class _Draws:
    """The torch.rand / torch.randn draws of one render_rays call, in the reference's order (rendering.py:321 perturb; per model
    pass 207, 213 the density noise; 338-340 the inverse-CDF draws between the passes; 128 the two warps' noise).  Their shapes
    follow from the call's arguments, so ONE launch makes them all when the call starts (_lib.fused_draws: torch's own generator,
    bit for bit, the generator left where the separate calls leave it); draws whose values nobody reads (noise with
    noise_std = 0: the reference multiplies them by zero) only advance the generator.  While a hipGraph is captured, with
    NSFF_TORCH_RNG=1, when torch.rand / torch.randn are not torch's own (the parity tests replay recorded draws through them),
    or when the launch's first-use self-check against torch fails on this device (_lib.fused_draws_match_torch: a torch /
    hiprand upgrade that changed the generator's geometry) every draw is the torch call it mirrors."""

    def __init__(self, device, n_rays, N_samples, N_importance, perturb, noise_std, test_time, models, kwargs, coarse=None):
        """coarse = (rays, z_lin, zs, xyz): with perturb > 0 the launch that draws also computes the coarse depths from the
        stratified-sampling draw (rendering.py:314-324, 332) -- ``coarse_done`` tells the caller that nsff_coarse_samples is not
        needed; the draw itself is then not kept (nothing else reads it)."""
        self.device = device
        self.coarse_done = False
        self.fused = (device.type == "cuda" and torch.rand is _TORCH_RAND and torch.randn is _TORCH_RANDN
                      and not os.environ.get("NSFF_TORCH_RNG") and not torch.cuda.is_current_stream_capturing()
                      and _lib.fused_draws_match_torch(device))      # (checked once per device against torch itself)
        self.at = 0
        if not self.fused:
            return
        first = models["coarse"] if N_importance > 0 else models["fine"]
        transient = bool(kwargs.get("output_transient", True) and first.encode_transient)
        flows = kwargs.get("output_transient_flow", []) if transient else []
        noisy = float(noise_std) != 0
        plan, need = [], []

        def add(kind, need_values, *shape):
            plan.append((kind, shape))
            need.append(bool(need_values))
        hook = None
        if perturb > 0:
            add("rand", coarse is None, n_rays, N_samples)
            if coarse is not None and n_rays:
                hook = (coarse[0], coarse[1], float(perturb), coarse[2], coarse[3])
        S = N_samples
        if N_importance > 0:
            add("randn", noisy, n_rays, S)
            if transient:
                add("randn", noisy, n_rays, S)
            if perturb != 0:
                add("rand", True, n_rays, N_importance)
                if transient:
                    add("rand", True, n_rays, N_importance)
            S = N_samples + (2 if transient else 1) * N_importance
        add("randn", noisy, n_rays, S)
        if transient:
            add("randn", noisy, n_rays, S)
            if flows and not test_time:
                add("randn", noisy, n_rays, S)
                add("randn", noisy, n_rays, S)
        self.plan = plan
        self.values = _lib.fused_draws(plan, device, need, coarse=hook)
        self.coarse_done = hook is not None

    def take(self, kind, *shape):
        if not self.fused:
            return (torch.rand if kind == "rand" else torch.randn)(*shape, device=self.device)
        if self.at >= len(self.plan) or self.plan[self.at] != (kind, shape):
            raise RuntimeError(f"draw {self.at}: {kind}{shape} is not the planned {self.plan[self.at:self.at + 1]}")
        self.at += 1
        return self.values[self.at - 1]


class _Pass:
    """Inputs shared by the coarse and the fine pass of one render_rays call."""
    __slots__ = ("embeddings", "rays", "ts", "max_t", "noise_std", "test_time", "kwargs",
                 "freqs_xyz", "dir_embedded", "n_rays", "rec", "tbias", "neighbour_rows", "draws")


def _embed_rows(embeddings, key, idx):
    return embeddings[key](idx).detach().contiguous().float()


_LINSPACE = {}


def _unit_linspace(n, device):
    """torch.linspace(0, 1, n) on `device`, built once per (n, device): the reference re-creates it in every call."""
    key = (int(n), str(device))
    if key not in _LINSPACE:
        if len(_LINSPACE) > 64:
            _LINSPACE.clear()
        _LINSPACE[key] = torch.linspace(0, 1, n, device=device)
    return _LINSPACE[key]


def _neighbour_time_rows(embeddings, ts, max_t):
    """(E_t[clamp(ts + 1, max=max_t)], E_t[clamp(ts - 1, min=0)]) -- reference rendering.py:218,224.  A plain nn.Embedding
    table is read by one kernel; any other module the caller passed as embeddings['t'] is simply called twice."""
    m = embeddings['t']
    if (isinstance(m, torch.nn.Embedding) and m.padding_idx is None and m.max_norm is None and m.weight.is_cuda
            and m.weight.dtype == torch.float32 and m.weight.is_contiguous() and ts.is_cuda and ts.dtype == torch.int64):
        return _lib.time_rows(m.weight.detach(), ts.contiguous(), max_t)
    return (_embed_rows(embeddings, 't', torch.clamp(ts + 1, max=max_t)),
            _embed_rows(embeddings, 't', torch.clamp(ts - 1, min=0)))


def _time_codes(ctx, models, kwargs, N_samples, N_importance, output_transient, flows):
    """The (n_rays, in_t) time codes of this call -- kwargs['t_embedded'] or embeddings['t'](ts) (rendering.py:153) -- and, for the
    launches that can use them, the time code's part of the dynamic trunk as per-RAY bias rows.

    The time code enters the dynamic trunk at layer 0 and at the skip layers, and every sample of a ray shares it
    (rendering.py:153,168,221,227 repeat the row): its product with those layers' time-code columns is computed once per RAY --
    one launch for every (model, time rows) pair of this call -- and handed to the field launches as bias rows
    (`_lib.time_bias`); the hand-scheduled f16x3 kernel then multiplies no time-code column.  Fills ctx.tbias[(typ, which)],
    which in 't' | 'fw' | 'bw', for the launches that can use it (inference in f16x3, 128-point tiles, samples per ray a
    multiple of 64); the others run as before.  With a plain nn.Embedding table and integer frame indices the SAME launch also
    gathers the rows -- E_t[ts] and the neighbour rows E_t[clamp(ts +- 1)] of the re-queries (rendering.py:218,224) -- instead
    of a gather, a neighbour-row kernel and the bias kernel."""
    ctx.tbias, ctx.neighbour_rows = {}, None
    if not output_transient:
        return None
    n_rays, ts, m = ctx.n_rays, ctx.ts, ctx.embeddings.get('t')
    override = kwargs.get('t_embedded') if 't_embedded' in kwargs else None
    plan = []
    if not (ctx.rec is not None or n_rays == 0 or config.get_precision() != "f16x3" or config.get_tile_points() not in (0, 130)):
        passes = [('coarse', N_samples), ('fine', N_samples + 2 * N_importance)] if N_importance > 0 else [('fine', N_samples)]
        for typ, S in passes:
            model = models[typ]
            if not model.encode_transient or S % 64 or (n_rays * S < 32768 and config.get_tile_points() == 0):
                continue
            if model.in_channels_t > 64 or model.in_channels_t % 4:   # (nsff_time_bias stages 64 columns as float4s; refused there)
                continue
            if typ == 'fine' and flows and not ctx.test_time and hasattr(model, "transient_flow_fw"):
                plan += [(typ, 'fw', model, 1), (typ, 'bw', model, -1)]      # (fw, bw adjacent: one buffer, see time_bias)
            plan.append((typ, 't', model, 0))
    plain = (override is None and isinstance(m, torch.nn.Embedding) and m.padding_idx is None and m.max_norm is None
             and m.weight.is_cuda and m.weight.dtype == torch.float32 and m.weight.is_contiguous()
             and torch.is_tensor(ts) and ts.is_cuda and ts.dtype == torch.int64
             and ts.dim() == 1 and ts.shape[0] == n_rays)      # (a malformed ts takes the gather below and the reference's shape error)
    if (plan and plain and all(mod.in_channels_t == m.weight.shape[1] for _, _, mod, _ in plan)
            and not os.environ.get('NSFF_NO_TIME_INDEX')):          # (`NSFF_NO_TIME_INDEX=1`: separate gather / neighbour-row launches, A/B)
        outs, rows = _lib.time_bias([(mod, d) for _, _, mod, d in plan], index=(m.weight.detach(), ts.contiguous(), ctx.max_t))
        for (typ, which, _, _), out in zip(plan, outs):
            ctx.tbias[(typ, which)] = out
        if 1 in rows:
            ctx.neighbour_rows = (rows[1], rows[-1])
        return rows[0]
    t_embedded = override if override is not None else ctx.embeddings['t'](ts)
    t_embedded = t_embedded.detach().contiguous().float()
    plan = [p for p in plan if t_embedded.shape == (n_rays, p[2].in_channels_t)]
    if plan:
        if any(w == 'fw' for _, w, _, _ in plan):
            ctx.neighbour_rows = _neighbour_time_rows(ctx.embeddings, ts, ctx.max_t)
        src = {0: t_embedded, 1: ctx.neighbour_rows[0] if ctx.neighbour_rows else None, -1: ctx.neighbour_rows[1] if ctx.neighbour_rows else None}
        for (typ, which, _, _), out in zip(plan, _lib.time_bias([(mod, src[d]) for _, _, mod, d in plan])):
            ctx.tbias[(typ, which)] = out
    return t_embedded


def _inference(results, ctx, model, xyz, zs, output_transient, output_transient_flow,
               t_embedded, a_embedded):
    """One model pass: field query, optional flow-warp re-queries, compositing.

    Mirrors the nested ``inference`` of the reference (rendering.py:83-300), including the
    order of its ``torch.randn`` draws.
    """
    typ = model.typ
    n_rays, S = zs.shape
    test_time = ctx.test_time
    results[f'zs_{typ}'] = zs
    results[f'xyzs_{typ}'] = xyz
    P = n_rays * S
    sigma_only = typ == 'coarse' and test_time
    want_flow = bool(output_transient_flow) and output_transient and not sigma_only
    warps = want_flow and not test_time
    disocc = warps and 'disocc' in output_transient_flow
    if want_flow and not hasattr(model, "transient_flow_fw"):
        raise AttributeError(f"{typ} model has no flow heads")

    raw = _new(zs, P, _lib.RAW_STRIDE)
    side = dict(dir_emb=ctx.dir_embedded if model.use_viewdir and not sigma_only else None,
                a_emb=a_embedded if (model.use_viewdir and model.in_channels_a > 0 and not sigma_only) else None)
    # View directions and appearance codes are per RAY (rendering.py:153-172 repeat them over the samples): their part of
    # static_dir_encoding is computed once per ray (one small launch) and handed to the field launch as bias rows -- the static
    # trunk of a view-direction model then runs on the hand-scheduled f16x3 kernel (inference launches with 128-point tiles and
    # samples per ray a multiple of 64; the others read dir_emb / a_emb as before).
    # (`NSFF_NO_SIDE_BIAS=1`: the earlier form for A/B -- static workgroups on the eight-wave kernel as a launch of their own.)
    # Round 6: training forwards too -- the launch that keeps the activations for the backward pass runs the same side-fold program in
    # the SAVE build of the body (the reference's documented training configuration, README.md:226-233, is a view-direction model);
    # a training forward takes 128-point tiles at any size.
    big_enough = ctx.rec is not None or config.get_tile_points() == 130 or (config.get_tile_points() == 0 and P >= 32768)
    if (side["dir_emb"] is not None and P and S % 64 == 0 and config.get_precision() == "f16x3" and config.get_tile_points() in (0, 130)
            and big_enough and not os.environ.get('NSFF_NO_SIDE_BIAS')):
        side["s_bias"] = _lib.side_bias(model, side["dir_emb"], side["a_emb"])

    def query(tag, raw_out, pts, static_mode, transient_mode, flow_heads, t_rows, which='t', **extra):
        """One field launch.  When gradients will be taken (ctx.rec) and the configuration allows it, this launch
        already is the training forward: it keeps the activations the backward kernels need."""
        saves = {}
        if ctx.rec is not None and field_grad.forward_can_save(model, static_mode, transient_mode):
            x3 = config.grad_x3()           # (three-product backward: the launch writes the remainder planes too)
            acts, xin, masks, side_rows = field_grad.alloc_saves(model, P, zs.device, bool(transient_mode), bool(static_mode), x3)
            saves = dict(save_acts=acts, save_xin=xin, save_masks=masks, save_side=side_rows, save_lo=x3)
            ctx.rec.setdefault("saved", {})[tag] = (raw_out, acts, xin, masks, pts.view(-1, 3), side_rows, x3)
        t_bias = ctx.tbias.get((typ, which)) if (transient_mode and not saves) else None
        _lib.field_query(model, raw_out, P, S, static_mode=static_mode, transient_mode=transient_mode,
                         flow_heads=flow_heads, xyz=pts, freqs=ctx.freqs_xyz, t_emb=t_rows, t_bias=t_bias, **saves, **extra)
    if P:
        query(typ, raw, xyz, 1 if sigma_only else 2, 0 if not output_transient else (1 if sigma_only else 2),
              2 if want_flow else 0, t_embedded if output_transient else None, **side)

    vis = None            # a6: evaluated inside the compositing kernel for the frame ts[0] (read on the device)
    if test_time and output_transient and 'dataset' in ctx.kwargs and P:
        vis = ray_geometry.frustum_args(ctx.kwargs['dataset'], ctx.ts, zs.device)

    # RNG draws, in the reference's order (rendering.py:207, 213, then 128 for fw and bw)
    nstd = float(ctx.noise_std)
    noise_s = ctx.draws.take("randn", n_rays, S)
    noise_t = ctx.draws.take("randn", n_rays, S) if output_transient else None
    if ctx.rec is not None and nstd != 0:        # the draws are needed again when gradients are taken
        ctx.rec[f"{typ}_static"], ctx.rec[f"{typ}_transient"] = noise_s, noise_t

    args = dict(n_rays=n_rays, n_samples=S, has_transient=int(output_transient),
                has_rgb=int(not sigma_only), flow_mode=0, want_disocc=int(disocc),
                noise_std=nstd, z_far=Z_FAR, raw=raw, zs=zs, xyz=xyz, vis=vis,
                noise_static=noise_s if nstd != 0 else None,
                noise_transient=noise_t if (nstd != 0 and output_transient) else None)

    def out(key, *shape):
        t = _new(zs, *shape)
        results[key] = t
        return t

    if not sigma_only:
        args['static_rgbs'] = out(f'static_rgbs_{typ}', n_rays, S, 3)
        if output_transient:
            args['transient_rgbs'] = out(f'transient_rgbs_{typ}', n_rays, S, 3)
            if want_flow:
                args['flows_fw'] = out('transient_flows_fw', n_rays, S, 3)
                args['flows_bw'] = out('transient_flows_bw', n_rays, S, 3)
    args['static_sigmas'] = out(f'static_sigmas_{typ}', n_rays, S)
    if output_transient:
        args['transient_sigmas'] = out(f'transient_sigmas_{typ}', n_rays, S)

    if warps:
        # points pushed along their own scene flow, re-queried one frame later / earlier
        ts = ctx.ts
        # the two re-queries are ONE launch over 2 n_rays rays when nothing is kept for a backward pass (same kernel, same
        # per-point arithmetic; one launch boundary less): the warped points, their records, the t +- 1 rows and their bias rows
        # are the two halves of one buffer each
        xyz_w = _new(zs, 2, n_rays, S, 3)
        xyz_fw, xyz_bw = xyz_w[0], xyz_w[1]
        results['xyzs_fw'] = xyz_fw
        raw_w = _new(zs, 2 * P, _lib.RAW_STRIDE)
        raw_fw, raw_bw = raw_w[:P], raw_w[P:]
        tp1, tm1 = ctx.neighbour_rows if ctx.neighbour_rows is not None else _neighbour_time_rows(ctx.embeddings, ts, ctx.max_t)
        tb_fw, tb_bw = ctx.tbias.get((typ, 'fw')), ctx.tbias.get((typ, 'bw'))
        # (halves of ONE allocation: the same storage AND adjacent -- two separate allocations may be adjacent by accident)
        def halves(a, b):
            return (a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()
                    and a.data_ptr() + a.numel() * 4 == b.data_ptr())
        merged = (ctx.rec is None and P > 0 and not os.environ.get('NSFF_NO_MERGED_REQUERY') and halves(tp1, tm1)
                  and (tb_fw is None) == (tb_bw is None) and (tb_fw is None or halves(tb_fw, tb_bw)))
        if P:
            _lib.warp_points(raw, xyz, zs, Z_FAR, xyz_fw, xyz_bw)
            if merged:
                ctx.tbias[(typ, 'fwbw')] = None if tb_fw is None else torch.as_strided(tb_fw, (2 * n_rays,) + tuple(tb_fw.shape[1:]),
                                                                                         tb_fw.stride())
                _lib.field_query(model, raw_w, 2 * P, S, static_mode=0, transient_mode=2, flow_heads=1, xyz=xyz_w,
                                 freqs=ctx.freqs_xyz, t_emb=torch.as_strided(tp1, (2 * n_rays, tp1.shape[1]), tp1.stride()),
                                 t_bias=ctx.tbias[(typ, 'fwbw')])
            else:
                query(f"{typ}_warp_fw", raw_fw, xyz_fw, 0, 2, 1, tp1, which='fw')
        noise_fw = ctx.draws.take("randn", n_rays, S)
        out('rgb_fw', n_rays, 3)
        results['xyzs_bw'] = xyz_bw
        if P and not merged:
            query(f"{typ}_warp_bw", raw_bw, xyz_bw, 0, 2, 1, tm1, which='bw')
        noise_bw = ctx.draws.take("randn", n_rays, S)
        if ctx.rec is not None and nstd != 0:
            ctx.rec[f"{typ}_warp_fw"], ctx.rec[f"{typ}_warp_bw"] = noise_fw, noise_bw
        out('rgb_bw', n_rays, 3)
        args.update(raw_fw=raw_fw, raw_bw=raw_bw, xyz_fw=xyz_fw, xyz_bw=xyz_bw,
                    noise_fw=noise_fw if nstd != 0 else None,
                    noise_bw=noise_bw if nstd != 0 else None,
                    rgb_fw=results['rgb_fw'], rgb_bw=results['rgb_bw'],
                    xyzs_fw_bw=out('xyzs_fw_bw', n_rays, S, 3),
                    xyzs_bw_fw=out('xyzs_bw_fw', n_rays, S, 3))

    if output_transient:
        args['static_weights'] = out(f'static_weights_{typ}', n_rays, S)
        args['transient_weights'] = out(f'transient_weights_{typ}', n_rays, S)
        args['weights'] = out(f'weights_{typ}', n_rays, S)
    else:
        args['static_weights'] = out(f'static_weights_{typ}', n_rays, S)
    if test_time and output_transient:
        args['static_alphas'] = out(f'static_alphas_{typ}', n_rays, S)
        args['transient_alphas'] = out(f'transient_alphas_{typ}', n_rays, S)

    if not sigma_only:
        args['depth'] = out(f'depth_{typ}', n_rays)
        args['rgb'] = out(f'rgb_{typ}', n_rays, 3)
        if output_transient:
            args['transient_alpha'] = out(f'transient_alpha_{typ}', n_rays)
            args['transient_rgb'] = out(f'transient_rgb_{typ}', n_rays, 3)
            args['static_only_rgb'] = out(f'_static_rgb_{typ}', n_rays, 3)
            args['static_only_depth'] = out(f'_static_depth_{typ}', n_rays)
            if want_flow:
                args['flow_mode'] = 2 if warps else 1
                args['xyz_exp'] = out('xyz_fine', n_rays, 3)
                args['flow_fw_exp'] = out('transient_flow_fw', n_rays, 3)
                args['xyz_fw_exp'] = out('xyz_fw', n_rays, 3)
                args['flow_bw_exp'] = out('transient_flow_bw', n_rays, 3)
                args['xyz_bw_exp'] = out('xyz_bw', n_rays, 3)
                if disocc:
                    args['disocc_fw'] = out('disocc_fw', n_rays, 1)
                    args['disoccs_fw'] = out('disoccs_fw', n_rays, S, 1)
                    args['disocc_bw'] = out('disocc_bw', n_rays, 1)
                    args['disoccs_bw'] = out('disoccs_bw', n_rays, S, 1)
    if n_rays:
        _lib.composite(**args)


def render_rays(models,
                embeddings,
                rays,
                ts,
                max_t,
                N_samples=64,
                perturb=0,
                noise_std=0,
                N_importance=0,
                chunk=1024 * 32,
                test_time=False,
                **kwargs):
    """Render rays through the NSFF fields (reference rendering.py:52-362).

    models: {'fine': NeRF[, 'coarse': NeRF]}; embeddings: {'xyz','dir'[, 't', 'a']};
    rays: (N_rays, 6) origins+directions (NDC); ts: (N_rays,) int64 or None; max_t: int.
    Recognised kwargs: output_transient, output_transient_flow, view_dir, t_embedded,
    a_embedded, dataset (eval visibility); others (epoch, K, ...) are ignored like the
    reference does.  ``chunk`` is accepted and ignored: the fused field kernel tiles the
    points itself, so there is no inner point-chunk loop to size.
    Results are fresh contiguous fp32 GPU tensors computed by the HIP kernels.  With autograd enabled,
    ``test_time=False`` and parameters that require grad, the results carry a graph to the model /
    embedding parameters (see :mod:`nsff_pl_amd.autograd`).
    """
    return _render_rays(models, embeddings, rays, ts, max_t, N_samples, perturb, noise_std, N_importance, chunk, test_time, kwargs)


def _render_rays(models, embeddings, rays, ts, max_t, N_samples, perturb, noise_std, N_importance, chunk, test_time, kwargs,
                 fine_points=None):
    """The body of :func:`render_rays`.  ``fine_points`` (None in every product call) is an injection point between the fine
    sampling stage and the fine field pass: a callable ``(rays, zs_fine, xyz_fine) -> (zs_fine, xyz_fine)``.  The parity tests use
    it (tests/common.py::render_rays_at) to evaluate the fine pass at the reference's depths -- the inverse-CDF draw is
    ill-conditioned in near-empty bins (tests/parity.py), so per-sample fine keys of two correct fp32 implementations are only
    comparable at identical depths; the arithmetic of that substitution lives in the tests, not here."""
    _lib.require_gpu_tensor(rays, "rays")
    _lib.load()
    # Gradients (training): forward values still come from the kernels below; the autograd graph of native
    # backward nodes is attached afterwards at the recorded depths / draws (nsff_pl_amd.autograd).
    want_grad = (torch.is_grad_enabled() and not test_time and
                 bool(autograd.grad_parameters(models, embeddings)))
    rec = {} if want_grad else None
    with torch.cuda.device(rays.device), torch.no_grad():
        results = {}
        rays = rays.contiguous().float()
        n_rays = rays.shape[0]
        embedding_xyz, embedding_dir = embeddings['xyz'], embeddings['dir']

        ctx = _Pass()
        ctx.embeddings, ctx.rays, ctx.ts, ctx.max_t = embeddings, rays, ts, max_t
        ctx.noise_std, ctx.test_time, ctx.kwargs, ctx.n_rays = noise_std, test_time, kwargs, n_rays
        ctx.freqs_xyz = [float(f) for f in embedding_xyz.freqs]
        ctx.rec = rec
        ctx.tbias, ctx.neighbour_rows = {}, None
        ctx.dir_embedded = None
        if any(m.use_viewdir for m in models.values()):
            view_dir = kwargs.get('view_dir', rays[:, 3:6])
            ctx.dir_embedded = embedding_dir(view_dir.contiguous().float())

        # coarse depths: one linspace shared by all rays, optional stratified jitter
        z_lin = _unit_linspace(N_samples, rays.device)
        zs = _new(rays, n_rays, N_samples)
        xyz_coarse = _new(rays, n_rays, N_samples, 3)
        ctx.draws = _Draws(rays.device, n_rays, N_samples, N_importance, perturb, noise_std, test_time, models, kwargs,
                           coarse=(rays, z_lin, zs, xyz_coarse))
        perturb_rand = ctx.draws.take("rand", n_rays, N_samples) if perturb > 0 else None
        if n_rays and not ctx.draws.coarse_done:
            _lib.coarse_samples(rays, z_lin, perturb, perturb_rand, zs, xyz_coarse)

        t_embedded = None
        if N_importance > 0:  # coarse to fine
            model = models['coarse']
            output_transient = bool(kwargs.get('output_transient', True) and model.encode_transient)
            t_embedded = _time_codes(ctx, models, kwargs, N_samples, N_importance, output_transient,
                                     kwargs.get('output_transient_flow', []))
            _inference(results, ctx, model, xyz_coarse, zs, output_transient, [], t_embedded, None)

            det = perturb == 0
            if det:
                u_s = _unit_linspace(N_importance, rays.device)
                u_t = u_s
            else:
                u_s = ctx.draws.take("rand", n_rays, N_importance)
                u_t = ctx.draws.take("rand", n_rays, N_importance) if output_transient else None
            S_fine = N_samples + (2 if output_transient else 1) * N_importance
            zs_static = _new(rays, n_rays, N_importance) if test_time else None
            zs_transient = _new(rays, n_rays, N_importance) if (test_time and output_transient) else None
            zs_fine = _new(rays, n_rays, S_fine)
            xyz_fine = _new(rays, n_rays, S_fine, 3)
            if n_rays:
                _lib.fine_samples(rays, z_lin, zs, N_importance,
                                  results['static_weights_coarse'],
                                  results['transient_weights_coarse'] if output_transient else None,
                                  u_s, u_t if output_transient else None, 0 if det else 1,
                                  zs_static, zs_transient, zs_fine, xyz_fine)
            if test_time:
                results['static_zs_fine'] = zs_static
                if output_transient:
                    results['transient_zs_fine'] = zs_transient
            if fine_points is not None:
                zs_fine, xyz_fine = fine_points(rays, zs_fine, xyz_fine)
                if tuple(zs_fine.shape) != (n_rays, S_fine) or tuple(xyz_fine.shape) != (n_rays, S_fine, 3):
                    raise ValueError(f"fine_points: expected shapes {(n_rays, S_fine)} / {(n_rays, S_fine, 3)}")
            zs, xyz = zs_fine, xyz_fine
        else:
            xyz = xyz_coarse

        model = models['fine']
        a_embedded = None
        if model.encode_appearance:
            a_embedded = kwargs['a_embedded'] if 'a_embedded' in kwargs else embeddings['a'](ts)
            a_embedded = a_embedded.detach().contiguous().float()
        if N_importance == 0:
            output_transient = bool(kwargs.get('output_transient', True) and model.encode_transient)
            t_embedded = _time_codes(ctx, models, kwargs, N_samples, 0, output_transient, kwargs.get('output_transient_flow', []))
        output_transient_flow = [] if not output_transient else kwargs.get('output_transient_flow', [])
        _inference(results, ctx, model, xyz, zs, output_transient, output_transient_flow,
                   t_embedded, a_embedded)
    if rec is None:
        return results
    rec.update(N_importance=N_importance, noise_std=float(noise_std), output_transient=output_transient,
               flows=list(output_transient_flow), zs_coarse=results.get('zs_coarse', results.get('zs_fine')),
               zs_fine=results['zs_fine'], view_dir=kwargs.get('view_dir', rays[:, 3:6]), dir_embedded=ctx.dir_embedded,
               t_embedded_override=kwargs.get('t_embedded'), a_embedded_override=kwargs.get('a_embedded'))
    return autograd.attach(results, models, embeddings, rays, ts, max_t, rec)


from .interpolation import interpolate  # noqa: E402,F401  (`from models.rendering import render_rays, interpolate`, eval.py:11)
