#!/usr/bin/env python
"""Run the generated backward body (gen_bwd.py) in the functional simulator (isa.py) on one 128-point tile of one trunk and compare
everything it leaves behind -- the pre-activation-gradient fragments of every layer in HBM, d(trunk input), the last gradient tile
in LDS -- with a numpy evaluation of the same chain in the same arithmetic (csrc/field_bwd.hip::nsff_field_bwd_kernel restated:
one f16 product per MAC accumulated in fp32 k-step by k-step, ReLU mask from sign words, clamp, fp16 to nearest, the two per-point
power-of-two factors of the fragment copy): BIT FOR BIT.

    python tools/h3asm/check_bwd.py [static|dynamic|noskip|short|ragged|overflow] ...

What this proves before any GPU time is spent: register allocation, every s_waitcnt count, barrier placement (cross-wave LDS race
detector), the weight-slot refill order, the running slot pointers, the exec-masked d_xin stores, the phase program
(build_program below is the reference of csrc/field_bwd_h3b.hip::h3b_build_program; a CPU test compares the two).
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_bwd as gb
from isa import Sim, SimError

PK_BASE = 0x1_0000_0000
PH_BASE = 0x2_0000_0000
ACT_BASE = 0x4_0000_0000                   # fragment slots (slot, tile64, 4 ks, 8 row blocks, 64 lanes, 8 points) fp16
MASK_BASE = 0x5_0000_0000                  # sign words (slot, tile64, 256 threads) uint64
DX_BASE = 0x6_0000_0000                    # d_xin (points, ld) fp32
N_TILES64 = 6                              # the simulated workgroup is 128-point tile 1 of 3: 64-point tiles 2 and 3
LDS_X, LDS_STASH = 0, gb.TILE_B
LDS_INV = 2 * gb.TILE_B                    # 128 floats
LDS_REL = LDS_INV + 512                    # rel1: 128 halfs, rel2: 128 halfs behind


def pack_segment(Wt):
    """Wt (256 rows, K) fp16 -> u32 stream [wave][ks][mt][lane][8 halfs]: value = Wt[64 wave + 32 mt + (lane & 31)][16 ks + 8 (lane >> 5) + t]
    (csrc/field_bwd.hip: the transposed pack)"""
    K = Wt.shape[1]
    nks = K // 16
    out = np.zeros((4, nks, 2, 64, 8), np.float16)
    lane = np.arange(64)
    for wv in range(4):
        for ks in range(nks):
            for mt in range(2):
                row = 64 * wv + 32 * mt + (lane & 31)
                for t in range(8):
                    out[wv, ks, mt, :, t] = Wt[row, 16 * ks + 8 * (lane >> 5) + t]
    return out.reshape(-1).view(np.uint32)


def build_program(steps, dynamic, skip, slot_bytes=(N_TILES64 * 64 * 256 * 2, N_TILES64 * 256 * 8)):
    """Phase descriptors of one trunk.  steps: [dict(off, nks, stash)] -- the head step (nks 4), the D - 1 layer steps (16), then
    (dynamic) the x0 step and (skip) the skip layer's input step; stash: the step's output tile is the one the skip step reads.
    -> uint32 (n, 8) or None when the body does not cover the structure (the host then launches the compiler-scheduled kernel)."""
    B = gb.BODY
    n_tail = (2 if skip else 1) if dynamic else 0
    layers = steps[1:len(steps) - n_tail]
    if steps[0]["nks"] != 4 or not layers or any(s["nks"] != 16 for s in steps[1:]) or steps[0].get("stash"):
        return None
    if dynamic and skip and sum(1 for s in steps if s.get("stash")) != 1:
        return None
    ph = []

    def desc(body, nxt=None):
        ph.append([body, 0, nxt["off"] if nxt else 0, nxt["nks"] * 2048 if nxt else 0, 0, 0, 0, 0])
    # descriptor 0 is not a phase: the trunk's constants (first 16-k-step segment, head segment, bytes per fragment / sign-word slot)
    ph.append([0, 0, steps[1]["off"], 16 * 2048, steps[0]["off"], 4 * 2048, slot_bytes[0], slot_bytes[1]])
    desc(B["AH"], steps[1])               # (its gaps carry the loads of weight slots 4..15)
    desc(B["BH"], steps[1])
    for j, st in enumerate(layers):
        i = 1 + j
        prev_stash = steps[i - 1].get("stash", False)
        desc(B["A16F"] if j == 0 else (B["A16S"] if prev_stash else B["A16"]))
        if j == 0 and prev_stash:
            return None
        last = j == len(layers) - 1
        if last and not dynamic:
            if st.get("stash"):
                return None
            desc(B["B16L"])
        else:
            if last and st.get("stash"):
                return None               # (its half B epilogue would ride in AX)
            desc(B["B16S"] if st.get("stash") else B["B16"], steps[i + 1])
    if not dynamic:
        desc(B["EPI_B"])
    else:
        desc(B["AX"])
        if skip:
            desc(B["BX"], steps[-1]); desc(B["AXS"]); desc(B["BXS"])
        else:
            desc(B["BXD"])
        desc(B["EPI_DXB"])
    desc(B["END"]); desc(B["END"])
    return np.array(ph, np.uint32)


def make_case(kind, seed=0):
    rng = np.random.RandomState(seed)
    dynamic = kind in ("dynamic", "noskip", "ragged", "short", "overflow")
    D = {"static": 8, "dynamic": 8, "noskip": 5, "short": 2, "ragged": 4, "overflow": 4}[kind]
    skip_l = {"static": None, "dynamic": 4, "noskip": None, "short": None, "ragged": 1, "overflow": 2}[kind]     # the skip LAYER whose tile is stashed
    if not dynamic:
        skip_l = None
    xin_rows = 128
    ld = xin_rows
    n_valid = 100 if kind == "ragged" else 128
    steps, bufs = [], []
    off = 4096

    def add(Wt, **kw):
        nonlocal off
        st = pack_segment(Wt)
        steps.append(dict(off=off, nks=Wt.shape[1] // 16, Wt=Wt, **kw))
        bufs.append((off, st))
        off += st.size * 4
    # head: K = 64 (16 head rows used)
    Wh = np.zeros((256, 64), np.float16)
    Wh[:, :16] = (rng.randn(256, 16) * 0.3).astype(np.float16)
    add(Wh, stash=(skip_l == D - 1), out_slot=D - 1)
    for l in range(D - 1, 0, -1):           # layer l: tile of slot l -> slot l - 1
        # ("overflow": weights large enough that pre-activation gradients leave the fp16 range: the conversion must clamp to +-65504)
        add((rng.randn(256, 256) * ((40.0 if kind == "overflow" else 1.6) / 16)).astype(np.float16), stash=(skip_l == l - 1), out_slot=l - 1)
    if dynamic:
        Wx = np.zeros((256, 256), np.float16)
        Wx[:xin_rows] = (rng.randn(xin_rows, 256) * 0.1).astype(np.float16)
        add(Wx, stash=False, out_slot=None)
        if skip_l is not None:
            Wx2 = np.zeros((256, 256), np.float16)
            Wx2[:xin_rows] = (rng.randn(xin_rows, 256) * 0.1).astype(np.float16)
            add(Wx2, stash=False, out_slot=None)
    pk = np.zeros(off // 4 + 16, np.uint32)
    for o, st in bufs:
        pk[o // 4:o // 4 + st.size] = st
    # head tile: 16 columns of block-scaled head derivatives (|max| of a row in [1024, 2048)), zeros behind
    hv = rng.randn(128, 16).astype(np.float32) * (10.0 ** rng.uniform(-3, 0, (128, 1))).astype(np.float32)
    amax = np.abs(hv).max(1)
    s = np.array([2.0 ** (11 - np.frexp(a)[1]) if a > 0 else 1.0 for a in amax], np.float32)
    T0 = np.zeros((128, 256), np.float16)
    T0[:, :16] = (hv * s[:, None]).astype(np.float16)
    inv = (1.0 / s).astype(np.float32)
    # the global-scale factors rel = G / s_p <= 1 as two fp16 factors (field_bwd.hip's head stage); some reach far down
    e = rng.randint(0, 30, 128)
    e[::7] = rng.randint(14, 40, e[::7].shape)
    rel = 2.0 ** (-e.astype(np.float64))
    f1 = np.maximum(rel, 2.0 ** -14)
    rel1, rel2 = f1.astype(np.float16), (rel / f1).astype(np.float16)
    n_slots = D + 2
    masks = rng.randint(0, 2 ** 32, (n_slots, N_TILES64, 256, 2)).astype(np.uint32)          # [.., 0] = word mt 0, [.., 1] = word mt 1
    return dict(kind=kind, D=D, dynamic=dynamic, skip=skip_l is not None, steps=steps, pk=pk, T0=T0, inv=inv, rel1=rel1, rel2=rel2,
                masks=masks, n_slots=n_slots, xin_rows=xin_rows, ld=ld, n_valid=n_valid)


def gemm(Wt, T, acc=None):
    """acc[pt][row] (+)= sum_k T[pt][k] Wt[row][k], fp32 accumulation k-step by k-step (16 columns: one MFMA; float64 inside)"""
    out = np.zeros((T.shape[0], 256), np.float32) if acc is None else acc.copy()
    for ks in range(Wt.shape[1] // 16):
        k = slice(16 * ks, 16 * ks + 16)
        out = (out.astype(np.float64) + T[:, k].astype(np.float64) @ Wt[:, k].astype(np.float64).T).astype(np.float32)
    return out


def mask_of(case, slot):
    """keep[pt][row] (bool) from the sign words of `slot` (tiles 2 and 3): thread 64 wave + lane of the point's 64-point tile, bit
    16 nt + 4 q + e of word mt for row 64 wave + 32 mt + 8 q + 4 (lane >> 5) + e, point 32 nt + (lane & 31)"""
    keep = np.zeros((128, 256), bool)
    for pt in range(128):
        hb, p64 = pt // 64, pt % 64
        nt, l31 = p64 // 32, p64 % 32
        for row in range(256):
            wv, r = row // 64, row % 64
            mt, r = r // 32, r % 32
            q, r = r // 8, r % 8
            hh, e_ = r // 4, r % 4
            word = int(case["masks"][slot, 2 + hb, 64 * wv + l31 + 32 * hh, mt])
            keep[pt, row] = (word >> (16 * nt + 4 * q + e_)) & 1
    return keep


def fragments(T, rel1, rel2):
    """(64 points, 256 rows) fp16 tile -> the fragment slot of its 64-point tile: [ks][rb][lane][t] = ((T * rel1) * rel2)[16 ks + 8 (lane >> 5) + t][32 rb + (lane & 31)]"""
    with np.errstate(under="ignore"):
        v = ((T.astype(np.float32) * rel1[:, None].astype(np.float32)).astype(np.float16).astype(np.float32) *
             rel2[:, None].astype(np.float32)).astype(np.float16)
    out = np.zeros((4, 8, 64, 8), np.float16)
    lane = np.arange(64)
    for ks in range(4):
        for rb in range(8):
            for t in range(8):
                out[ks, rb, :, t] = v[16 * ks + 8 * (lane >> 5) + t, 32 * rb + (lane & 31)]
    return out


def reference(case):
    """-> ({slot: (128, 256) fp16 tile}, d_xin (128, 256) fp32 or None, the last tile)"""
    T = case["T0"]
    tiles = {}
    stash = None
    acc = None
    dxin = None
    for st in case["steps"]:
        if st["out_slot"] is not None:
            a = gemm(st["Wt"], T[:, :st["Wt"].shape[1]])
            a = np.where(mask_of(case, st["out_slot"]), a, np.float32(0))
            with np.errstate(over="ignore"):
                case["n_clamped"] = case.get("n_clamped", 0) + int((np.abs(a) > 65504.0).sum())
                T = np.clip(a, -65504.0, 65504.0).astype(np.float16)
            tiles[st["out_slot"]] = T
            if st["stash"]:
                stash = T
        elif acc is None:
            acc = gemm(st["Wt"], T)
            if not case["skip"]:
                dxin = acc * case["inv"][:, None]
        else:
            acc = gemm(st["Wt"], stash, acc)
            dxin = acc * case["inv"][:, None]
    return tiles, dxin, T


def run_case(kind, seed=0, verbose=True):
    case = make_case(kind, seed)
    pre, prog, _ = gb.build()
    sim = Sim(pre + prog)                      # the two asm statements back to back (the head stage between them is C++)
    sim.add_buffer(PK_BASE, case["pk"])
    phases = build_program(case["steps"], case["dynamic"], case["skip"])
    assert phases is not None
    sim.add_buffer(PH_BASE, phases.reshape(-1))
    n_slots = case["n_slots"]
    acts = np.full(n_slots * N_TILES64 * 64 * 256 // 2, 0xFFFFFFFF, np.uint32)
    sim.add_buffer(ACT_BASE, acts)
    sim.add_buffer(MASK_BASE, case["masks"].reshape(-1))
    n_points_buf = 128 * 3
    dx = np.full(n_points_buf * case["ld"], 0x7FC00000, np.uint32)
    sim.add_buffer(DX_BASE, dx)
    sim.mem_written = {ACT_BASE: np.zeros(acts.size, bool), DX_BASE: np.zeros(dx.size, bool)}
    lds_h = sim.lds.view(np.float16)
    for r in range(128):
        base = (LDS_X + r * gb.LDH_B) // 2
        lds_h[base:base + 256] = case["T0"][r]
    sim.lds.view(np.float32)[LDS_INV // 4:LDS_INV // 4 + 128] = case["inv"]
    lds_h[LDS_REL // 2:LDS_REL // 2 + 128] = case["rel1"]
    lds_h[LDS_REL // 2 + 128:LDS_REL // 2 + 256] = case["rel2"]
    D = case["D"]
    I_S, I_V = gb.IN_SB, gb.IN_VB
    for w in sim.waves:
        tid = 64 * w.id + np.arange(64)

        def s64(name, val):
            w.s[I_S[name].i], w.s[I_S[name].i + 1] = val & 0xFFFFFFFF, val >> 32
        s64("pk", PK_BASE); s64("phases", PH_BASE)
        s64("act", ACT_BASE + ((D - 1) * N_TILES64 + 2) * 64 * 256 * 2)
        s64("mask", MASK_BASE + (((D - 1) * N_TILES64 + 2) * 256 + 64 * w.id) * 8)
        s64("dxin", DX_BASE + 128 * case["ld"] * 4)
        rows_here = 64 * w.id < case["xin_rows"]
        for name, val in (("lds", LDS_X), ("stash", LDS_STASH), ("wave", w.id), ("invlds", LDS_INV), ("rellds", LDS_REL),
                          ("ld4", case["ld"] * 4), ("nvalid", case["n_valid"] if rows_here else 0)):
            w.s[I_S[name].i] = int(val)
        w.v[I_V["tid"].i] = tid
        # the pre-issue statement's operands
        w.s[gb.PRE_S["off0"].i] = case["steps"][0]["off"] + w.id * 4 * 2048
        w.v[gb.PRE_V["lane16"].i] = (tid & 63) << 4
    t0 = time.time()
    sim.run()
    dt = time.time() - t0
    # the body ends without waiting for the vector-memory queue: only stores may be outstanding
    for wid, pend in sim.pending_at_end.items():
        assert not pend, (kind, "wave", wid, "ends with loads outstanding into", sorted(pend)[:8])
    tiles, dxin, T_last = reference(case)
    # ---- fragment slots: every tile of the chain, both halves; nothing else written
    a16 = acts.view(np.float16).reshape(n_slots, N_TILES64, 4, 8, 64, 8)
    for slot, T in tiles.items():
        for hb in range(2):
            want = fragments(T[64 * hb:64 * hb + 64], case["rel1"][64 * hb:64 * hb + 64], case["rel2"][64 * hb:64 * hb + 64])
            got = a16[slot, 2 + hb]
            assert np.array_equal(got.view(np.uint16), want.view(np.uint16)), (kind, "slot", slot, "half", hb,
                                                                                 int((got.view(np.uint16) != want.view(np.uint16)).sum()))
    keep = np.ones(a16.shape[:2], bool)
    for slot in tiles:
        keep[slot, 2:4] = False
    assert (a16.view(np.uint16)[keep] == 0xFFFF).all(), "a fragment store outside the chain's slots"
    assert sim.mem_written[ACT_BASE].reshape(n_slots, N_TILES64, -1)[~keep].all()
    # ---- d_xin: rows of the 128 points of this tile, columns < xin_rows, valid points only; bit for bit
    dxf = dx.view(np.float32).reshape(n_points_buf, case["ld"])
    wr = sim.mem_written[DX_BASE].reshape(n_points_buf, case["ld"])
    if dxin is not None:
        nv, xr = case["n_valid"], case["xin_rows"]
        assert np.array_equal(dxf[128:128 + nv, :xr].view(np.uint32), dxin[:nv, :xr].astype(np.float32).view(np.uint32)), kind
        assert wr[128:128 + nv, :xr].all() and wr.sum() == nv * xr, (wr.sum(), nv * xr)
    else:
        assert not wr.any()
    # ---- the last tile is what LDS holds
    got_last = np.stack([lds_h[(LDS_X + r * gb.LDH_B) // 2:(LDS_X + r * gb.LDH_B) // 2 + 256] for r in range(128)])
    assert np.array_equal(got_last.view(np.uint16), T_last.view(np.uint16)), kind
    n_mf = sim.waves[0].n_mfma
    want_mf = sum(s["nks"] for s in case["steps"]) * 8
    if verbose:
        print(f"{kind:9s} phases {len(phases) - 3:2d}  MFMAs/wave {n_mf} (expected {want_mf})  instructions/wave {sim.waves[0].n_inst}  "
              f"slots {sorted(tiles)}  d_xin {'yes' if dxin is not None else 'no'}  bit-identical  ({dt:.1f} s)")
    assert n_mf == want_mf
    assert kind != "overflow" or case["n_clamped"] > 1000, case.get("n_clamped")
    return True


if __name__ == "__main__":
    kinds = sys.argv[1:] or ["static", "dynamic", "noskip", "short", "ragged", "overflow"]
    for k in kinds:
        try:
            run_case(k)
        except SimError as e:
            sys.exit(f"{k}: SIMULATION ERROR: {e}")
    print("backward simulation OK")
