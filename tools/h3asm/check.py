#!/usr/bin/env python
"""Run the generated trunk body (gen.py) in the functional simulator (isa.py) on one 128-point tile and compare the activation
tile it leaves in LDS with a numpy evaluation of the same layers in the same f16x3 arithmetic.

    python tools/h3asm/check.py [static|dynamic|noskip|twoskips] ...

What this proves before any GPU time is spent: register allocation, every s_waitcnt count (a register written by an
outstanding load may not be touched), barrier placement (cross-wave LDS race detector), the weight-slot refill order, the
bias-table initialisation, the rebuild of the skip layer's input tile, the phase program.  What it cannot prove: the
hardware's own semantics (MFMA operand layouts, v_permlane32_swap, exec-masked loads) -- those are taken from the kernels
that are already parity-green on the MI355X (csrc/field_h3.hip) -- and timing.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen
from isa import Sim, SimError, halfs_of, f16_rtz_pack

PK_BASE = 0x1_0000_0000
PH_BASE = 0x2_0000_0000
T_BASE = 0x3_0000_0000
ACT_BASE = 0x4_0000_0000                   # SAVE build: activation slots (slot, tile64, 4 ks, 256 rows, 16 points) fp16
MASK_BASE = 0x5_0000_0000                  # SAVE build: sign words (slot, tile64, 256 threads) uint64
N_TILES64 = 6                              # the simulated workgroup is 128-point tile 1 of 3: 64-point tiles 2 and 3
LDS_X = 0
LDS_RAW = 2 * gen.PLANE_B                  # raw-record image: 128 points x 16 floats
LDS_BIAS = LDS_RAW + 8192                  # behind the two planes and the raw-record image


def split_rtz(x):
    """fp32 array -> (hi, lo) fp16 arrays, both rounded toward zero (v_cvt_pkrtz), lo = rtz(x - hi)"""
    z = np.zeros_like(x, np.float32)
    hi = halfs_of(f16_rtz_pack(x.astype(np.float32), z))[0]
    lo = halfs_of(f16_rtz_pack((x.astype(np.float32) - hi).astype(np.float32), z))[0]
    return hi.astype(np.float16), lo.astype(np.float16)


def pack_segment(W):
    """W (256, K) fp32, K = 16 nks -> u32 stream [wave][ks][mt][part][lane][8 halfs], hi = fp16(x) to nearest, lo = fp16(x - hi)"""
    K = W.shape[1]
    nks = K // 16
    hi = W.astype(np.float16)
    lo = (W - hi.astype(np.float32)).astype(np.float16)
    out = np.zeros((4, nks, 2, 2, 64, 8), np.float16)
    lane = np.arange(64)
    for wv in range(4):
        for ks in range(nks):
            for mt in range(2):
                n = 64 * wv + 32 * mt + (lane & 31)
                for t in range(8):
                    c = 16 * ks + 8 * (lane >> 5) + t
                    out[wv, ks, mt, 0, :, t] = hi[n, c]
                    out[wv, ks, mt, 1, :, t] = lo[n, c]
    return out.reshape(-1).view(np.uint32), hi, lo


def pack_head(W):
    """W (32, 256) fp32 -> u32 stream [ks][hi, lo][lane][8 halfs], row = lane & 31, column 16 ks + 8 (lane >> 5) + t"""
    hi = W.astype(np.float16)
    lo = (W - hi.astype(np.float32)).astype(np.float16)
    out = np.zeros((16, 2, 64, 8), np.float16)
    lane = np.arange(64)
    for ks in range(16):
        for t in range(8):
            c = 16 * ks + 8 * (lane >> 5) + t
            out[ks, 0, :, t] = hi[lane & 31, c]
            out[ks, 1, :, t] = lo[lane & 31, c]
    return out.reshape(-1).view(np.uint32), hi, lo


def make_persistent(ph):
    """The phase program of a PERSISTENT workgroup (field_h3.hip: h3a_make_persistent does the same): the last segment's B phase
    B16L becomes B16LP with descriptor 0's stream fields -- it requests the first segments' weight slots 0..7 for the next tile."""
    at = [i for i in range(1, len(ph)) if ph[i][0] == gen.BODY["B16L"]]
    if len(at) != 1:
        return False
    ph[at[0]][0] = gen.BODY["B16LP"]
    ph[at[0]][3:8] = ph[0][3:8]
    return True


def build_program(segs, in_t, head=None, sig_row=None, save=False, persist=False):
    """Phase descriptors for a trunk (the host-side builder in csrc/field_h3a.hip does the same).
    segs: list of dict(nks, off, bias (index or None), post ('relu' | 'none'), rebuild (bool)[, wstride (bytes between the
    waves' blocks of the packed segment, default nks * 4096), bias_b (table row of half B, default = bias)]).
    head: dict(off (bytes of the head tile), n_rows, slot0) -- the last epilogue requests the tile, the HEAD phase leaves the
    pre-activation sums of rows 0 .. n_rows - 1 at floats slot0 .. of every point's raw record.
    sig_row: bias-table row of the sigma weights -- the sigma ride (gen.SIG) runs on the epilogues of the LAST BUT ONE segment
    (a view-direction static trunk: static_sigma reads the last trunk layer, static_dir_encoding follows it)."""
    B = gen.BODY
    ph = []

    def desc(body, flags=0, bias=0, n1=16, r1=None, r2=None):
        r1 = r1 if r1 is not None else segs[0]
        r2 = r2 if r2 is not None else r1
        return [body, flags, 1024 * bias, n1, r1["off"], r1.get("wstride", r1["nks"] * 4096), r2["off"], r2.get("wstride", r2["nks"] * 4096)]

    def refill_fields(t):
        nxt = segs[t + 1] if t + 1 < len(segs) else None
        nx2 = segs[t + 2] if t + 2 < len(segs) else None
        if nxt is None:
            return dict(n1=16, r1=segs[t], r2=segs[t])
        return dict(n1=nxt["nks"], r1=nxt, r2=nx2 if (nxt["nks"] < 16 and nx2 is not None) else nxt)
    rb = (1 << gen.F_REBUILD) | ((1 << gen.F_REBUILD_T) if in_t > 0 else 0)
    s0 = segs[0]
    assert s0["nks"] in (4, 8) and s0["post"] == "relu" and s0["bias"] is not None
    first = dict(n1=s0["nks"], r1=s0, r2=segs[1] if len(segs) > 1 else s0)
    # descriptor 0 initialises both halves: acc_A from row `bias`, acc_B from the row (flags >> 16) bytes behind it
    ph.append(desc(0, (1 << gen.F_INIT) | ((1024 * (s0.get("bias_b", s0["bias"]) - s0["bias"])) << 16), s0["bias"], **first))
    pending_b = False
    for t, sg in enumerate(segs):
        nxt = segs[t + 1] if t + 1 < len(segs) else None
        init_next = (1 << gen.F_INIT) if (nxt is not None and nxt["bias"] is not None) else 0
        nbias = nxt["bias"] if (nxt is not None and nxt["bias"] is not None) else 0
        if sg["nks"] == 16:
            assert t > 0 and pending_b, "a 16-wide segment rides the epilogue of the one before it"
            sig_a = sig_row is not None and t == len(segs) - 1      # half B's epilogue of the layer sigma reads rides here
            sig_b = sig_row is not None and t == len(segs) - 2      # half A's
            # (an A phase's tail initialises acc_B: the segment's row of half B -- they differ when the time code is folded in)
            ph.append(desc(B["A16RS" if sig_a else "A16R"], (1 << gen.F_INIT) if sg["bias"] is not None else 0, sg.get("bias_b", sg["bias"]) or 0))
            if sg["post"] == "relu":
                if sig_b:
                    assert nxt is not None and nxt["nks"] == 16 and sg["post"] == "relu"
                    ph.append(desc(B["B16RS"], init_next | ((1024 * sig_row) << 16), nbias, **refill_fields(t)))
                else:
                    ph.append(desc(B["B16R"] if nxt is not None else B["B16L"], init_next, nbias, **refill_fields(t)))
                pending_b = True
            else:
                assert nxt is not None and nxt["rebuild"] and nxt["bias"] is None
                ph.append(desc(B["B16X"], rb, 0, **refill_fields(t)))
                pending_b = False
        else:
            assert not pending_b and sg["post"] == "relu"
            if t == 0:          # the first segment: slots 0..7 were requested by the pre-issue statement, 8..15 ride in its A phase
                ph.append(desc(B["A4F"] if sg["nks"] == 4 else B["A8F"], 0, 0, **first))
            else:
                ph.append(desc(B["A4"] if sg["nks"] == 4 else B["A8"], rb if sg["rebuild"] else 0))
            ph.append(desc(B["B4"] if sg["nks"] == 4 else B["B8"], init_next, nbias, **refill_fields(t)))
            pending_b = True
    assert pending_b
    hd = head or dict(off=segs[0]["off"], n_rows=0, slot0=0)
    ph.append([B["EPI_B"], 0, 0, hd["n_rows"], hd["off"], 0, hd["off"], 0])
    if save:                # the last activation goes to its slot before the heads read it
        ph.append([B["SAVE_LAST"], 0, 0, hd["n_rows"], hd["off"], 0, hd["off"], 0])
    ph.append([B["HEAD"], 0, 4 * hd["slot0"], hd["n_rows"], hd["off"], 0, hd["off"], 0])
    ph.append(desc(B["END"]))
    ph.append(desc(B["END"]))
    ph = np.array(ph, np.uint32)
    if persist:
        assert make_persistent(ph)
    return ph


def make_case(kind, seed=0):
    rng = np.random.RandomState(seed)
    # *_tb: the dynamic trunk with the time code folded into per-ray bias rows (every 64-point half inside one ray): the body runs
    # the position part of the input segments only (4 of their 8 k-steps), the table holds b + W_t t(ray of the half)
    tb = kind.endswith("_tb")
    in_t = 48 if kind in ("dynamic", "dynamic_tb", "twoskips_tb") else 0
    k0 = 128 if in_t else 64
    D = 8
    viewdir = kind == "viewdir"
    skips = {"static": [4], "dynamic": [4], "noskip": [], "twoskips": [2, 5], "dynamic_tb": [4], "twoskips_tb": [2, 5], "viewdir": [4]}[kind]
    layers = []
    segs = []
    off = 4096                                  # (packed buffer: keep offset 0 unused)
    bufs = []

    def add_seg(W, bias, post, rebuild):
        nonlocal off
        stream, hi, lo = pack_segment(W)
        bidx = None
        if bias is not None:
            bidx = len([s_ for s_ in segs if s_["bias"] is not None])
        segs.append(dict(nks=W.shape[1] // 16, off=off, bias=bidx, post=post, rebuild=rebuild, hi=hi, lo=lo, b=bias))
        if tb and W.shape[1] == k0:
            segs[-1].update(nks=4, wstride=(k0 // 16) * 4096)
        bufs.append((off, stream))
        off += stream.size * 4
    scale = 2.5 / np.sqrt(256.0)
    for l in range(D):
        b = (rng.randn(256) * 0.1).astype(np.float32)
        if l == 0:
            add_seg((rng.randn(256, k0) * 2.5 / np.sqrt(k0)).astype(np.float32), b, "relu", False)
        elif l in skips:
            add_seg((rng.randn(256, 256) * scale).astype(np.float32), b, "none", False)
            add_seg((rng.randn(256, k0) * 2.5 / np.sqrt(k0)).astype(np.float32), None, "relu", True)
        else:
            add_seg((rng.randn(256, 256) * scale).astype(np.float32), b, "relu", False)
    sig_w = None
    if viewdir:
        # static_dir_encoding behind the trunk: a 256-wide relu layer whose bias rows are per RAY (b + W[:, 256:] [dir | a]):
        # one row for half A, another for half B; the sigma weights are one more row of the table
        add_seg((rng.randn(256, 256) * scale).astype(np.float32), (rng.randn(256) * 0.1).astype(np.float32), "relu", False)
        sig_w = (rng.randn(256) * 0.2).astype(np.float32)
    # the heads: 10 rows (dynamic) / 4 rows (static) / 3 rows (view directions: rgb only) of a 32-row tile, at slot 4 / 0 of the raw records
    n_rows, slot0 = (10, 4) if in_t else ((3, 0) if viewdir else (4, 0))
    Wh_ = np.zeros((32, 256), np.float32)
    Wh_[:n_rows] = (rng.randn(n_rows, 256) * 0.3).astype(np.float32)
    hstream, hhi, hlo = pack_head(Wh_)
    head = dict(off=off, n_rows=n_rows, slot0=slot0, hi=hhi, lo=hlo)
    bufs.append((off, hstream))
    off += hstream.size * 4
    pk = np.zeros(off // 4 + 16, np.uint32)
    for o, st in bufs:
        pk[o // 4:o // 4 + st.size] = st
    # input tile: xyz part (64 columns) random, time part = per-ray rows of a small table
    x_xyz = (rng.randn(128, 64) * 0.7).astype(np.float32)
    x_xyz[:, 63] = 0
    n_rays = 5
    t_table = (rng.randn(n_rays, max(in_t, 4)) * 0.5).astype(np.float32)
    ray_of = (np.arange(128) * n_rays) // 128 if not tb else np.arange(128) // 64 * 3
    x_in = np.zeros((128, k0), np.float32)
    x_in[:, :64] = x_xyz
    if in_t:
        x_in[:, 64:64 + in_t] = t_table[ray_of, :in_t]
    rows = {sg["bias"]: sg["b"] for sg in segs if sg["bias"] is not None}
    if tb:
        # rows of the segments that carry the bias of a layer with an input part: the layer's first segment (layer 0: the input
        # segment itself, a skip layer: the 256-wide segment in front of its input segment)
        th, tl = split_rtz(x_in[:, 64:])
        nrow = len(rows)
        for i, sg in enumerate(segs):
            if sg["hi"].shape[1] != k0:
                continue
            tgt = segs[i] if i == 0 else segs[i - 1]
            assert tgt["bias"] is not None
            Wh, Wl = sg["hi"][:, 64:].astype(np.float64), sg["lo"][:, 64:].astype(np.float64)
            part = th.astype(np.float64) @ Wh.T + tl.astype(np.float64) @ Wh.T + th.astype(np.float64) @ Wl.T   # (128, 256)
            rows[tgt["bias"]] = (tgt["b"] + part[0]).astype(np.float32)
            tgt["bias_b"] = nrow
            rows[nrow] = (tgt["b"] + part[64]).astype(np.float32)
            nrow += 1
    sig_row = None
    if viewdir:
        last = segs[-1]
        last["bias_b"] = len(rows)
        rows[last["bias_b"]] = (last["b"] + (rng.randn(256) * 0.3).astype(np.float32)).astype(np.float32)    # half B lies in another ray
        sig_row = len(rows)
        rows[sig_row] = sig_w
    return dict(kind=kind, in_t=in_t, k0=k0, segs=segs, pk=pk, x_in=x_in, t_table=t_table, ray_of=ray_of, rows=rows, tb=tb, head=head,
                sig_row=sig_row, sig_w=sig_w)


PRE_RELU = []


def reference(case, keep=None):
    """the same layers in numpy: products Wh.xh + Wh.xl + Wl.xh accumulated in float64, fp32 after every layer
    keep: list that receives the fp32 post-ReLU values of every relu segment"""
    xin_h, xin_l = split_rtz(case["x_in"])
    xh_, xl_ = xin_h.astype(np.float64), xin_l.astype(np.float64)
    acc = None
    for sg in case["segs"]:
        if sg.get("bias_b") is not None and not case["tb"] and sg["b"] is not None:       # per-half rows (view directions)
            sg = dict(sg, b=np.where(np.arange(128)[:, None] < 64, case["rows"][sg["bias"]][None, :], case["rows"][sg["bias_b"]][None, :]))
        Wh, Wl = sg["hi"].astype(np.float64), sg["lo"].astype(np.float64)
        if sg["rebuild"]:
            xh_, xl_ = xin_h.astype(np.float64), xin_l.astype(np.float64)
        K = Wh.shape[1]
        prod = xh_[:, :K] @ Wh.T + xl_[:, :K] @ Wh.T + xh_[:, :K] @ Wl.T          # (128, 256)
        acc = (prod + (np.atleast_2d(sg["b"]) if sg["b"] is not None else acc)).astype(np.float32).astype(np.float64) \
            if sg["b"] is not None else (acc + prod).astype(np.float32).astype(np.float64)
        if sg["post"] == "relu":
            v = np.maximum(acc, 0).astype(np.float32)
            if keep is not None:
                keep.append(v)
                PRE_RELU.append(acc.astype(np.float32))
            h, l = split_rtz(v)
            xh_, xl_ = h.astype(np.float64), l.astype(np.float64)
    return (xh_ + xl_).astype(np.float32), xh_, xl_


def head_reference(case, xh_, xl_):
    """(chain Wl.xh + chain Wh.xl) + chain Wh.xh, each chain accumulated in fp32 order of the k-steps (float64 per MFMA)"""
    hd = case["head"]
    Wh, Wl = hd["hi"].astype(np.float64), hd["lo"].astype(np.float64)
    c = [np.zeros((128, 32), np.float32) for _ in range(3)]
    for ks in range(16):
        k = slice(16 * ks, 16 * ks + 16)
        c[0] = (c[0] + xh_[:, k] @ Wl[:, k].T).astype(np.float32)
        c[1] = (c[1] + xl_[:, k] @ Wh[:, k].T).astype(np.float32)
        c[2] = (c[2] + xh_[:, k] @ Wh[:, k].T).astype(np.float32)
    return ((c[0] + c[1]).astype(np.float32) + c[2]).astype(np.float32)[:, :hd["n_rows"]]


def run_case(kind, seed=0, verbose=True):
    save = kind.endswith("_save")
    persist = kind.endswith("_persist")
    case = make_case(kind[:-5] if save else (kind[:-8] if persist else kind), seed)
    pre, prog, _ = gen.build(save=save)
    sim = Sim(pre + prog)                      # the two asm statements back to back (the encoder between them is C++)
    sim.add_buffer(PK_BASE, case["pk"])
    body_in_t = 0 if case["tb"] else case["in_t"]       # (the body sees no time-code columns when they are folded into the table)
    phases = build_program(case["segs"], body_in_t, case["head"], case["sig_row"], save=save, persist=persist)
    n_relu = sum(1 for sg in case["segs"] if sg["post"] == "relu")
    if save:
        # slots 0 .. n_relu - 1 of this trunk, two spare slots in front (the trunk's first slot is not slot 0 of the buffers)
        acts = np.full((n_relu + 2) * N_TILES64 * 64 * 256 // 2, 0xFFFFFFFF, np.uint32)
        masks = np.full((n_relu + 2) * N_TILES64 * 256 * 2, 0xFFFFFFFF, np.uint32)
        sim.add_buffer(ACT_BASE, acts)
        sim.add_buffer(MASK_BASE, masks)
        sim.mem_written = {ACT_BASE: np.zeros(acts.size, bool), MASK_BASE: np.zeros(masks.size, bool)}
    sim.add_buffer(PH_BASE, phases.reshape(-1))
    sim.add_buffer(T_BASE, case["t_table"].reshape(-1).view(np.uint32))
    # LDS: input tile as the encoder leaves it (hi / lo planes), bias table
    lds_h = sim.lds.view(np.float16)
    xh_, xl_ = split_rtz(case["x_in"])
    for r in range(128):
        base = (LDS_X + r * gen.LDH_B) // 2
        lds_h[base:base + case["k0"]] = xh_[r]
        lds_h[base + gen.PLANE_B // 2:base + gen.PLANE_B // 2 + case["k0"]] = xl_[r]
    lds_f = sim.lds.view(np.float32)
    for row, vec in case["rows"].items():
        o = (LDS_BIAS + 1024 * row) // 4
        lds_f[o:o + 256] = vec
    stride_t = case["t_table"].shape[1] * 4
    I_S, I_V = gen.IN_S, gen.IN_V
    for w in sim.waves:
        tid = 64 * w.id + np.arange(64)
        for name, val in (("pk", PK_BASE), ("phases", PH_BASE)):
            w.s[I_S[name].i], w.s[I_S[name].i + 1] = val & 0xFFFFFFFF, val >> 32
        for name, val in (("lds", LDS_X), ("biaslds", LDS_BIAS), ("wave", w.id), ("in_t", body_in_t), ("rawlds", LDS_RAW), ("n1", phases[0][3]),
                          ("r1", phases[0][4]), ("r1w", phases[0][5]), ("r2", phases[0][6]), ("r2w", phases[0][7])):
            w.s[I_S[name].i] = int(val)
        if save:            # the first slot of this trunk = slot 2; half A = 64-point tile 2
            act0 = ACT_BASE + (2 * N_TILES64 + 2) * 64 * 256 * 2
            msk0 = MASK_BASE + ((2 * N_TILES64 + 2) * 256 + 64 * w.id) * 8
            for name, val in (("act", act0), ("mask", msk0)):
                w.s[I_S[name].i], w.s[I_S[name].i + 1] = val & 0xFFFFFFFF, val >> 32
            w.s[I_S["astride"].i], w.s[I_S["mstride"].i] = N_TILES64 * 64 * 256 * 2, N_TILES64 * 256 * 8
        w.v[I_V["tid"].i] = tid
        row, q = tid >> 2, tid & 3
        for names, rows in ((("tpa0", "tpa1"), row), (("tpb0", "tpb1"), row + 64)):
            addr = T_BASE + case["ray_of"][rows].astype(np.int64) * stride_t + 64 * q
            addr = np.where(16 * q < max(body_in_t, 1), addr, T_BASE)        # (lanes that load nothing: any valid address)
            w.v[I_V[names[0]].i] = (addr & 0xFFFFFFFF).astype(np.uint32)
            w.v[I_V[names[1]].i] = (addr >> 32).astype(np.uint32)
    t0 = time.time()
    sim.run()
    dt = time.time() - t0
    # result: the last activation in the two planes
    got = np.zeros((128, 256), np.float32)
    for r in range(128):
        base = (LDS_X + r * gen.LDH_B) // 2
        got[r] = lds_h[base:base + 256].astype(np.float32) + lds_h[base + gen.PLANE_B // 2:base + gen.PLANE_B // 2 + 256].astype(np.float32)
    acts_relu = []
    del PRE_RELU[:]
    want, fxh, fxl = reference(case, acts_relu)
    pre_relu = list(PRE_RELU)
    err = float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-30))
    # the heads: pre-activation sums in the raw-record image, everything else untouched (zero)
    hd = case["head"]
    raw = sim.lds.view(np.float32)[LDS_RAW // 4:LDS_RAW // 4 + 128 * 16].reshape(128, 16).copy()
    hwant = head_reference(case, fxh, fxl)
    herr = float(np.abs(raw[:, hd["slot0"]:hd["slot0"] + hd["n_rows"]] - hwant).max() / max(np.abs(hwant).max(), 1e-30))
    raw[:, hd["slot0"]:hd["slot0"] + hd["n_rows"]] = 0
    if case["sig_row"] is not None:
        # the sigma ride: floats 4 + 2 wave + (lane >> 5) of a record = sum over that lane's 32 neurons of w_sigma relu(layer D-1)
        v = acts_relu[-2].astype(np.float64)
        for wv in range(4):
            for hh in range(2):
                n = np.array([64 * wv + 32 * mt + 8 * q + 4 * hh + e for mt in range(2) for q in range(4) for e in range(4)])
                part = v[:, n] @ case["sig_w"][n].astype(np.float64)
                got_p = raw[:, 4 + 2 * wv + hh]
                serr = float(np.abs(got_p - part).max() / max(np.abs(part).max(), 1e-30))
                assert serr < 2e-6, (wv, hh, serr)
                err = max(err, serr)
        total = raw[:, 4:12].sum(1)
        assert np.abs(total - v @ case["sig_w"].astype(np.float64)).max() <= 1e-5 * np.abs(v @ case["sig_w"]).max()
        raw[:, 4:12] = 0
    assert not raw.any(), "the HEAD phase wrote outside its slots"
    assert herr < 2e-6, herr
    err = max(err, herr)
    if save:
        # every ReLU layer's output: fp16(hi + lo) in fragment order [16-point group][neuron][point], sign words in accumulator order
        # a tile = [16-point group ks][32-neuron block rb][lane = (neuron & 31) + 32 (8-point group)][8 points]: 1 KiB blocks, each one
        # MFMA operand fragment of the weight-gradient GEMM in lane order
        a16 = acts.view(np.float16).reshape(n_relu + 2, N_TILES64, 4, 8, 2, 32, 8)
        m64 = masks.view(np.uint64).reshape(n_relu + 2, N_TILES64, 256)
        assert len(acts_relu) == n_relu
        for l, v in enumerate(acts_relu):
            h_, l_ = split_rtz(v)
            want16 = (h_.astype(np.float32) + l_.astype(np.float32)).astype(np.float16)       # (128, 256)
            for hb in range(2):
                got16 = a16[2 + l, 2 + hb].transpose(0, 2, 4, 1, 3).reshape(64, 256)          # [16 ks + 8 g + t][32 rb + r]
                # (the reference accumulates in float64, the MFMAs in fp32: a value whose fp32 bits differ in the last place may
                #  round to the neighbouring half -- one fp16 ulp, on a fraction of a per cent of the values; values near zero differ by the
                #  layer's absolute accumulation noise)
                w16 = want16[64 * hb:64 * hb + 64].astype(np.float32)
                dif = np.abs(got16.astype(np.float32) - w16)
                assert (dif <= np.abs(w16) * 2.0 ** -9 + 2e-6 * np.abs(w16).max()).all() and (dif > 0).mean() < 0.01, (l, hb, dif.max(), (dif > 0).mean())
                for wv in range(4):
                    for lane in range(64):
                        word = int(m64[2 + l, 2 + hb, 64 * wv + lane])
                        for mt in range(2):
                            for nt in range(2):
                                pt = 64 * hb + 32 * nt + (lane & 31)
                                for q in range(4):
                                    for e_ in range(4):
                                        n = 64 * wv + 32 * mt + 8 * q + 4 * (lane >> 5) + e_
                                        bit = (word >> (32 * mt + 16 * nt + 4 * q + e_)) & 1
                                        assert bit == int(v[pt, n] > 0) or abs(pre_relu[l][pt, n]) < 1e-5, (l, hb, wv, lane, mt, nt, q, e_)
        # nothing else was written: the spare slots and the other tiles keep their fill
        keep = np.ones(a16.shape[:2], bool); keep[2:2 + n_relu, 2:4] = False
        assert (a16.view(np.uint16)[keep] == 0xFFFF).all() and (m64[keep] == np.uint64(0xFFFFFFFFFFFFFFFF)).all()
        assert sim.mem_written[ACT_BASE].reshape(n_relu + 2, N_TILES64, -1)[2:, 2:4].all()
    if persist:
        # the next tile's first segments are resident in weight slots 0..7, exactly where the pre-issue statement puts a first tile's
        n1, r1, r1w, r2, r2w = (int(v) for v in phases[0][3:8])
        lane = np.arange(64)
        for w in sim.waves:
            for k in range(8):
                base = ((r1 + w.id * r1w) if k < n1 else (r2 + w.id * r2w)) + 4096 * k
                for c in range(4):
                    for e_ in range(4):
                        want_w = case["pk"][(base + 1024 * c + 16 * lane) // 4 + e_]
                        assert np.array_equal(w.a[16 * k + 4 * c + e_], want_w), (w.id, k, c, e_)
    n_mf = sim.waves[0].n_mfma
    want_mf = sum(sg["nks"] for sg in case["segs"]) * 24 + 48
    if verbose:
        print(f"{kind:9s} phases {len(phases) - 2:2d}  MFMAs/wave {n_mf} (expected {want_mf})  instructions/wave {sim.waves[0].n_inst}  "
              f"max-norm rel err {err:.2e}  |want| max {np.abs(want).max():.3g}  ({dt:.1f} s)")
    assert n_mf == want_mf
    assert err < 2e-6, err
    return err


if __name__ == "__main__":
    kinds = sys.argv[1:] or ["static", "dynamic", "noskip", "twoskips", "dynamic_tb", "twoskips_tb", "viewdir",
                             "static_save", "dynamic_save", "twoskips_save", "viewdir_save", "static_persist", "dynamic_tb_persist",
                             "viewdir_persist"]
    for k in kinds:
        try:
            run_case(k)
        except SimError as e:
            sys.exit(f"{k}: SIMULATION ERROR: {e}")
    print("simulation OK")
