#!/usr/bin/env python
"""Generator of the hand-scheduled body of ``nsff_field_bwd_kernel_h3b`` (nsff_pl_amd/csrc/field_bwd_h3b.hip): the data-gradient
chain of the NSFF field network (autograd of reference models/nerf.py:118-213) for one 128-point tile and ONE trunk.

    python tools/h3asm/gen_bwd.py          # writes nsff_pl_amd/csrc/field_bwd_h3b_body.inc, after linting the stream
    python tools/h3asm/check_bwd.py        # runs the generated stream in the functional simulator against numpy

What the body computes is what ``nsff_field_bwd_kernel`` (csrc/field_bwd.hip, compiler-scheduled, 64 points per workgroup, two
workgroups per CU) computes, bit for bit: per step  acc[row][point] (+)= Wt_seg[row][k] . T[point][k]  with ONE f16 MFMA product
per MAC (v_mfma_f32_32x32x16_f16, fp32 accumulate), T the fp16 gradient tile in LDS, then an epilogue: ReLU mask from the
forward's sign words -> clamp -> fp16 (round to nearest) -> the tile of the next step; every tile also goes to HBM on the global
scale in the weight-gradient GEMM's fragment order; the last steps of a dynamic trunk leave d(trunk input) in fp32.

How it is scheduled (the mirror image of the forward body, tools/h3asm/gen.py):
  * one wave per SIMD (four waves, 512 registers each), 128 points per workgroup as two 64-point HALVES A and B (= two of the
    forward's 64-point tiles: its sign words and the fragment slots are per 64 points); wave w owns output rows [64 w, 64 w + 64);
  * the CURRENT step's transposed weights of those rows -- 16 k-steps x (2 row tiles x 4 registers) -- are RESIDENT in a0..a127
    and multiplied with both halves, one after the other:  A(i) | B(i) | A(i+1) | ...;  a phase is 64 MFMAs (2 048 pipe cycles);
  * what is not an MFMA RIDES in the gaps of the other half's MFMAs: in A(i) the epilogue of B(i-1), in B(i) the epilogue of A(i)
    and the refill of every weight slot with step i+1 as soon as B has used it; the HBM copy of a tile rides in the phase that
    multiplies it (the tile is complete behind the previous phase's barrier and untouched until the next phase's ride);
  * one barrier per phase, in front of the last two k-steps: every fragment of the half has been read by then (the last k-step's
    fragments go to a third buffer a k-step early), the ride's LDS stores are done; the 8 MFMAs behind it cover the first
    fragment reads of the next phase and the dispatch.

Register map (asm-owned; the compiler keeps v0..v23, s0..s39 and VCC):
    v24..v35 addresses / scratch   v36..v39 sign words (A: mt 0, 1; B: mt 0, 1)   v40..v63 epilogue temporaries
    v64..v75, v114..v125 copy temporaries (two sets)   v76..v107 fragments XF[4][2]   v108..v112 d_xin addresses, 1/scale
    v128..v191 acc_A               v192..v255 acc_B
    a[8 ks .. 8 ks + 7] weight slot ks = k-step ks of the resident segment: [row tile 0 | row tile 1] x 4 registers
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa import *          # noqa: F401,F403
from gen import Stream, spread, raw, render, in_s, in_v, IN_S, IN_V      # noqa: E402  (the wait-count model and the text helpers)

LDH_B = 528                 # bytes per LDS row (264 halfs)
HALF_B = 64 * LDH_B         # 33792: rows 64..127 = half B
NT_B = 32 * LDH_B           # second 32-point tile of a half
TILE_B = 128 * LDH_B        # 67584: the stash tile lies behind the gradient tile

# ---- VGPRs
V_RD, V_RDS, V_WR, V_WRS, V_CP, V_CPOFF, V_REL, V_LANE16, V_OFF, V_MOFF, V_INVA, V_DXB = (V(i) for i in range(24, 36))
MSK = {"A": (V(36), V(37)), "B": (V(38), V(39))}
T0 = 40                     # v40..v63
CPT = 64                    # v64..v75
XF0 = 76                    # XF[b][nt] = v[76 + 8 b + 4 nt : +4], b = k-step & 3: four rotating buffers, requested two k-steps ahead
V_DX, V_INV, V_L31 = V(108, 2), V(110, 2), V(113)      # V_DX[nt]: d_xin offsets of the current half
ACC = {"A": 128, "B": 192}

# ---- SGPRs (asm-owned: s40..s99)
S_PK = S(40, 2)
S_LDS, S_STASH, S_WAVE, S_INVLDS = S(42), S(43), S(44), S(45)
S_PH = S(46, 2)
S_SAVE = S(48, 2)
S_R1, S_RELLDS, S_T0, S_T1 = S(50), S(51), S(52), S(53)
S_ACT = S(54, 2)
S_CUR = 56                  # s56..s63 current descriptor
S_NXT = 64                  # s64..s71 next descriptor (being fetched)
S_MASK = S(72, 2)
S_ASTRIDE, S_MSTRIDE = S(74), S(75)
S_DXIN = S(76, 2)
S_LD4, S_NVALID = S(78), S(79)
D_BODY, D_FLAGS, D_R1, D_R1W = range(4)

BODY = dict(END=0, AH=1, BH=2, A16F=3, A16=4, B16=5, A16S=6, B16S=7, B16L=8, EPI_B=9, AX=11, BX=12, AXS=13, BXS=14,
            BXD=15, EPI_DXB=16)

IN_SB = dict(pk=S(0, 2), phases=S(2, 2), lds=S(4), stash=S(5), wave=S(6), invlds=S(7), rellds=S(8), act=S(10, 2), mask=S(12, 2),
             dxin=S(16, 2), ld4=S(18), nvalid=S(19))
IN_VB = dict(tid=V(0))


def in_sb(dst, name):
    src = IN_SB[name]
    op = "s_mov_b64" if dst.n == 2 else "s_mov_b32"
    return Inst(op, f"{op} {dst}, %[{name}]", [src], [dst], "salu", dict(d=dst, s=[src]))


def in_vb(dst, name):
    return Inst("v_mov_b32", f"v_mov_b32 {dst}, %[{name}]", [IN_VB[name]], [dst], "valu", dict(d=dst, s=[IN_VB[name]]))


def xf(b, nt):
    return V(XF0 + 8 * b + 4 * nt, 4)


def acc(half, mt, nt):
    return V(ACC[half] + 16 * (2 * mt + nt), 16)


def wslot(ks, mt):
    return A(8 * ks + 4 * mt, 4)


def other(half):
    return "B" if half == "A" else "A"


def half_off(half):
    return 0 if half == "A" else HALF_B


def vadd_s(d, s_, v_):      # d = s + v  (VOP2: the scalar operand first)
    return I_valu("v_add_u32", d, s_, v_, text=f"v_add_u32_e32 {d}, {s_}, {v_}")


def gstore_nt(voff, data, sbase, off=0):
    """a store of data that is written once and read once by another kernel: non-temporal, as field_bwd.hip's fragment_block"""
    i = I_gstore_s(voff, data, sbase, off)
    if "tcp" not in EXP:            # (H3B_EXP=tcp: plain stores, for A/B)
        i.text += " nt"
    return i


# ----------------------------------------------------------------------------------------------------------------------
def epilogue_mask_unit(half, u, tset, stash):
    """(tile u >> 1 = (mt, nt), quad pair p = u & 1) of `half`'s accumulators: ReLU mask from the sign words (bit 16 nt + 4 q + e of
    word mt: the sign-extended bit ANDed onto the value) -> fp16, round to nearest, clamped to the fp16 range -> two 8-byte LDS
    stores (rows 8 p + 4 h .. + 3 and 16 + 8 p + 4 h .. + 3 of the 32-row tile, h = lane >> 5); stash: the same data once more into
    the stash tile (the skip layer's pre-activation gradient, needed again by the trunk-input steps)."""
    t, p = u >> 1, u & 1
    mt, nt = t >> 1, t & 1
    a = acc(half, mt, nt)
    x = [a.sub(4 * p + e) for e in range(4)] + [a.sub(4 * p + 8 + e) for e in range(4)]
    M = [V(T0 + 12 * tset + k) for k in range(8)]
    H = [V(T0 + 12 * tset + 8 + k) for k in range(4)]
    out = []
    for j in range(8):
        bit = 16 * nt + 4 * (p + 2 * (j >> 2)) + (j & 3)
        out.append(I_valu("v_bfe_i32", M[j], MSK[half][mt], bit, 1))
    for j in range(8):
        out.append(I_valu("v_and_b32", x[j], M[j], x[j], text=f"v_and_b32_e32 {x[j]}, {M[j]}, {x[j]}"))
    for k in range(4):          # (MODE.FP16_OVFL is set while the body runs: a conversion that overflows gives +-65504, not infinity --
                                #  the clamp of field_bwd.hip's epilogue, v_med3_f32 per value, without an instruction)
        out.append(I_valu("v_cvt_pk_f16_f32", H[k], x[2 * k], x[2 * k + 1]))
    off = half_off(half) + NT_B * nt + 64 * mt + 16 * p
    out += [I_ds_write_b64(V_WR, V(H[0].i, 2), off), I_ds_write_b64(V_WR, V(H[2].i, 2), off + 32)]
    if stash:
        out += [I_ds_write_b64(V_WRS, V(H[0].i, 2), off), I_ds_write_b64(V_WRS, V(H[2].i, 2), off + 32)]
    return out


def epilogue_mask(half, stash=False):
    """-> ride items; the marker ('NEED_VM', 'msk' + half) in front makes the phase wait for the sign words"""
    out = [("NEED_VM", "msk" + half)]
    for u in range(8):
        out += epilogue_mask_unit(half, u, u & 1, stash)
    return out


def epilogue_dxin(half):
    """d(trunk input) of `half`: acc / s_point (the per-point block scale, 1 / s in LDS) -> fp32 rows of d_xin, 16-byte stores of four
    consecutive rows; lanes whose point lies past the launch's last point (S_NVALID = valid points of this tile; 0 for the waves
    whose rows the input does not have) store nothing.  Items are instructions or LISTS (exec-masked groups, issued contiguously)."""
    hb = 0 if half == "A" else 1
    out = []
    for nt in range(2):
        first = 64 * hb + 32 * nt
        out += [I_salu("s_mul_i32", S_T0, S_LD4, first), vadd_s(V_DX.sub(nt), S_T0, V_DXB),
                (I_ds_read_b32(V_INV.sub(nt), V_INVA, 4 * first), ("inv", half, nt))]
    for nt in range(2):
        first = 64 * hb + 32 * nt
        for mt in range(2):
            a = acc(half, mt, nt)
            if mt == 0:
                out.append(("NEED_LDS", ("inv", half, nt)))
            for r in range(16):
                out.append(I_valu("v_mul_f32", a.sub(r), V_INV.sub(nt), a.sub(r), text=f"v_mul_f32_e32 {a.sub(r)}, {V_INV.sub(nt)}, {a.sub(r)}"))
            # lanes of valid points: (lane & 31) < nvalid - first
            st = [I_salu("s_sub_u32", S_T0, S_NVALID, first, scc=True), I_salu("s_cselect_b32", S_T0, 0, S_T0),
                  I_v_cmp_gt_u32_vcc(S_T0, V_L31), I_s_and_saveexec(S_SAVE)]
            for q in range(4):
                # (non-temporal like the fragment stores: d_xin is read once, by nsff_field_input_backward -- both trunks 488-502 ->
                #  468-472 us, dynamic alone 256-263 -> 256-259, same box interleaved; H3B_EXP=tdx: plain stores, for A/B)
                st.append(((I_gstore_s if "tdx" in EXP else gstore_nt)(V_DX.sub(nt), a.sub(4 * q, 4), S_DXIN, 4 * (32 * mt + 8 * q)), "dxst"))
            st.append(I_s_mov_exec(S_SAVE))
            out.append(st)
    return out


def copy_groups(half, last_of_slot):
    """The HBM copy of the gradient tile of `half` (64 points x 256 rows -> 32 one-KiB blocks in the fragment order of the
    weight-gradient GEMM, eight per wave; layout and arithmetic of field_bwd.hip::fragment_block: two transposing LDS reads, the two
    per-point power-of-two factors as packed fp16 multiplies, ONE contiguous 16-byte store per lane) as a list of GROUPS of ride
    items.  Software-pipelined two blocks deep: block u + 2's LDS reads are requested where block u is multiplied and stored (two
    register sets, the factors of a 16-point pair likewise), so that an LDS round trip -- 100+ cycles under load, more than the ride
    instructions between two groups cover -- is never waited for.  last_of_slot: half B -- the slot pointer moves on."""
    hb = 0 if half == "A" else 1
    T = [[V(CPT + 4 * b + k) for k in range(4)] for b in range(2)]                       # v64..v71
    REL = [[V(CPT + 8, 2), V(CPT + 10, 2), V(114, 2), V(116, 2)], [V(118, 2), V(120, 2), V(122, 2), V(124, 2)]]     # [pair parity][r1a, r1b, r2a, r2b]

    def tr_reads(u):
        imm = half_off(half) + (u >> 1) * 16 * LDH_B + (u & 1) * 256
        t = T[u & 1]
        return [(I_ds_read_tr(V(t[0].i, 2), V_CP, imm), ("cp", u)), (I_ds_read_tr(V(t[2].i, 2), V_CP, imm + 4 * LDH_B), ("cp", u))]

    def rel_reads(pair):    # the factors of 16-point group `pair`: rel1 / rel2 of points pt0 .. pt0 + 3 and pt0 + 4 .. pt0 + 7 of the lane
        ro = 128 * hb + 32 * pair
        r = REL[pair & 1]
        return [(I_ds_read_b64(r[0], V_REL, ro), ("rel", pair)), (I_ds_read_b64(r[1], V_REL, ro + 8), ("rel", pair)),
                (I_ds_read_b64(r[2], V_REL, ro + 256), ("rel", pair)), (I_ds_read_b64(r[3], V_REL, ro + 264), ("rel", pair))]
    groups = [[I_salu("s_lshl_b32", S_T0, S_WAVE, 10, scc=True), vadd_s(V_CPOFF, S_T0, V_LANE16)] +
              ([I_valu("v_add_u32", V_CPOFF, 32768, V_CPOFF, text=f"v_add_u32_e32 {V_CPOFF}, 0x8000, {V_CPOFF}")] if hb else []) +
              rel_reads(0) + tr_reads(0),
              tr_reads(1) + rel_reads(1)]
    for u in range(8):      # block wave + 4 u = (16-point group u >> 1, 32-row block wave + 4 (u & 1))
        t, r = T[u & 1], REL[(u >> 1) & 1]
        groups.append([("NEED_LDS", ("cp", u)), ("NEED_LDS", ("rel", u >> 1)),
                       I_valu("v_pk_mul_f16", t[0], t[0], r[0].sub(0)), I_valu("v_pk_mul_f16", t[1], t[1], r[0].sub(1)),
                       I_valu("v_pk_mul_f16", t[2], t[2], r[1].sub(0)), I_valu("v_pk_mul_f16", t[3], t[3], r[1].sub(1))])
        g2 = [I_valu("v_pk_mul_f16", t[0], t[0], r[2].sub(0)), I_valu("v_pk_mul_f16", t[1], t[1], r[2].sub(1)),
              I_valu("v_pk_mul_f16", t[2], t[2], r[3].sub(0)), I_valu("v_pk_mul_f16", t[3], t[3], r[3].sub(1)),
              (gstore_nt(V_CPOFF, V(t[0].i, 4), S_ACT, 0), "cpst"),
              I_valu("v_add_u32", V_CPOFF, 4096, V_CPOFF, text=f"v_add_u32_e32 {V_CPOFF}, 0x1000, {V_CPOFF}")]
        if u + 2 < 8:
            g2 += tr_reads(u + 2)
            if (u & 1) == 1:                # (both blocks of pair u >> 1 are done with its factors: the registers take pair (u >> 1) + 2's)
                g2 += rel_reads((u >> 1) + 2)
        groups.append(g2)
    if last_of_slot:
        groups[-1] += [I_salu("s_sub_u32", S(S_ACT.i), S(S_ACT.i), S_ASTRIDE, scc=True), I_salu("s_subb_u32", S(S_ACT.i + 1), S(S_ACT.i + 1), 0, scc=True)]
    return groups


def mask_load(half):
    """the sign words of the step whose epilogue of `half` comes next: 8 bytes per lane at S_MASK (this wave's 64 words of half
    A's 64-point tile; half B: + 2 KiB); behind half B's load the pointer moves on to the next (lower) slot"""
    out = [(I_gload_s(V(MSK[half][0].i, 2), V_MOFF, S_MASK, 2048 if half == "B" else 0), "msk" + half)]     # (as `nt`: measured +-0)
    if half == "B":
        out += [I_salu("s_sub_u32", S(S_MASK.i), S(S_MASK.i), S_MSTRIDE, scc=True), I_salu("s_subb_u32", S(S_MASK.i + 1), S(S_MASK.i + 1), 0, scc=True)]
    return out


def merge_ride(main, groups, tail_groups=0):
    """`groups` spread evenly through the list `main` (each group stays contiguous); the last `tail_groups` groups follow the end of
    `main` -- the copy's last multiplies and stores (no LDS operation among them) then sit between the epilogue's last LDS stores and
    the phase barrier, whose lgkmcnt(0) would otherwise wait out a fresh store's round trip"""
    if not groups:
        return list(main)
    out, n, g = [], len(main), len(groups) - tail_groups
    for k in range(g):
        out += groups[k]
        out += main[k * n // g:(k + 1) * n // g]
    for k in range(g, len(groups)):
        out += groups[k]
    return out


def frag_reads(half, ks, dst, src="x"):
    base = V_RD if src == "x" else V_RDS
    return [I_ds_read_b128(dst(nt), base, half_off(half) + NT_B * nt + 32 * ks) for nt in range(2)]


def mfmas(half, ks, init):
    """the four MFMAs of k-step ks; init: the first k-step starts from zero (the inline constant)"""
    out = []
    for mt in range(2):
        for nt in range(2):
            d = acc(half, mt, nt)
            b = xf(ks & 3, nt)
            out.append(I_mfma(d, wslot(ks, mt), b, d if not (init and ks == 0) else 0))
    return out


def refill(ks):
    """slot ks <- k-step ks of the next segment (stream S_R1 of the dispatcher: V_OFF runs through it, one step per two k-steps):
    [load, load (+ step)]"""
    o = 2048 * (ks & 1)
    return [[I_gload_x4_s(A(8 * ks, 4), V_OFF, S_PK, o)],
            [I_gload_x4_s(A(8 * ks + 4, 4), V_OFF, S_PK, o + 1024)] +
            ([I_valu("v_add_u32", V_OFF, 4096, V_OFF, text=f"v_add_u32_e32 {V_OFF}, 0x1000, {V_OFF}")] if ks & 1 else [])]


def emit_ride(s, item):
    """a riding item: an instruction, (instruction, tag), the markers ('NEED_LDS', tag) / ('NEED_VM', tag), or a LIST of items that
    must be issued contiguously (an exec-masked group)"""
    if isinstance(item, list):
        for it in item:
            emit_ride(s, it)
    elif isinstance(item, tuple) and item[0] == "NEED_LDS":
        s.need_lds(item[1])
    elif isinstance(item, tuple) and item[0] == "NEED_VM":
        s.need_vm(item[1])
    elif isinstance(item, tuple):
        s.emit(item[0], item[1])
    else:
        s.emit(item, "ride")


RIDE_CAP = int(os.environ.get("H3B_RIDE_CAP", "8"))
EXP = os.environ.get("H3B_EXP", "")      # timing experiments (results are garbage; only the time is read): nostore, nocopy, noepi, norefill, nomfma, nostream (AH without the 24 riding weight loads)


class LogStream(Stream):
    """Stream that also keeps the tags of every VMEM operation it issued, in order (the seed of the phase behind it)"""

    def __init__(self):
        super().__init__()
        self.vm_log = []

    def emit(self, i, tag=None, group=None):
        if i.kind == "vmem":
            self.vm_log.append(tag)
        return super().emit(i, tag, group)


def n_insts(items):
    return sum(n_insts(it) if isinstance(it, list) else (0 if (isinstance(it, tuple) and it[0] in ("NEED_LDS", "NEED_VM")) else 1) for it in items)


def phase_body(name, half, nks, init=True, src="x", ride=None, copy=False, refills=None, msk=False, vm_seed=None, tail_src="x",
               tail=True, stream=None):
    """One phase: `nks` k-steps of MFMAs on acc_<half> from the tile (src 'x') or the stash tile ('s') of <half>.
    ride: None | 'mask' | 'mask_stash' | 'dxin' -- the epilogue of the OTHER half in the MFMA gaps
    copy: this half's tile goes to HBM while the phase multiplies it
    refills: None | list of slots refilled (behind the k-step that used them last) with the next segment (B phases)
    msk: request the sign words of THIS half's coming epilogue
    vm_seed: None, or the tags of the VMEM operations that may be outstanding at entry, oldest first (the phases in front, as
             generated): the phase then waits for weight slot ks in front of k-step ks
    tail_src: where the first fragments of the NEXT phase (the other half) are read from behind the barrier; tail=False: none
    stream: None | list of weight slots whose loads ride in this phase from its first gap on (AH: slots 4..15 <- k-steps 4..15 of
            the first 16-k-step segment -- nothing else rides there, and a workgroup's start is bound by what crosses the CU's
            vector-memory path: round 6 measured 2.2 k cycles just to ISSUE all 32 loads of a wave in front of the head stage)
    -> (instructions, tags of the VMEM operations issued, in order)"""
    s = LogStream()
    s.emit(I_label(f"L_{name}"))
    oh = other(half)
    last = nks - 1
    bar = (last - 1, 0)
    for k0 in range(2):         # the phase in front requested the fragments of k-steps 0 and 1 behind its barrier
        for nt in range(2):
            s.lds_q.append((("xf", k0), None))
    for tg in (vm_seed or []):
        s.vm_q.append((tg, None))
    ride_ins = {None: [], "mask": epilogue_mask(oh), "mask_stash": epilogue_mask(oh, True), "dxin": epilogue_dxin(oh)}[ride]
    if "noepi" in EXP and ride in ("mask", "mask_stash"):
        ride_ins = [("NEED_VM", "msk" + oh)]
    if copy and "nocopy" not in EXP:
        cg = copy_groups(half, half == "B")
        if "nostore" in EXP:
            cg = [[it for it in g if not (isinstance(it, tuple) and it[1] == "cpst")] for g in cg]
        ride_ins = merge_ride(ride_ins, [x for g in cg for x in [g]], tail_groups=3 if ride_ins else 0)
    if msk:
        ride_ins = mask_load(half) + ride_ins
    if stream:
        pieces = [[I_salu("s_add_u32", S_T0, S_R1, 2048 * stream[0], scc=True), vadd_s(V_OFF, S_T0, V_LANE16)]]
        for slot in stream:
            pieces += [[(x, ("w", slot)) for x in piece] for piece in refill(slot)]
        ride_ins = merge_ride(ride_ins, pieces)
    if "norefill" in EXP:
        refills = None if refills is None else []
    if refills is not None:
        s.emit(vadd_s(V_OFF, S_R1, V_LANE16))
    # gaps: behind every MFMA in front of the barrier; a ride that does not fit them (short phases) runs RIDE_CAP per gap through
    # the whole phase, the rest behind the last MFMA, and the barrier comes last
    early = [(ks, m) for ks in range(nks) for m in range(4) if (ks, m) < bar and (ks, m) != (0, 0)]
    late_barrier = n_insts(ride_ins) > RIDE_CAP * len(early)
    if late_barrier:
        gaps = [(ks, m) for ks in range(nks) for m in range(4) if (ks, m) != (0, 0)]
        per_gap = {g: RIDE_CAP for g in gaps}
    else:
        per_gap = dict(zip(early, spread(len(ride_ins), len(early)))) if ride_ins else {}
    ri = 0

    def tail_reads():
        if tail:                # (buffers 0 and 1: their last users, k-steps last - 3 and last - 2, were issued long ago)
            for k0 in range(2):
                for r in frag_reads(oh, k0, lambda nt: xf(k0, nt), tail_src):
                    s.emit(r, ("xf'", k0))
    for ks in range(nks):
        for m, mf in enumerate(mfmas(half, ks, init)):
            if m == 0:
                if vm_seed is not None:
                    s.need_vm(("w", ks))
                s.need_lds(("xf", ks))
            if (ks, m) == bar and not late_barrier:
                s.wait(lgkm=0)
                s.emit(I_barrier())
            if "nomfma" in EXP:
                s.emit(raw("s_nop 0"))
            else:
                s.emit(mf)
            if m == 0 and ks + 2 <= last:
                # fragments two k-steps ahead (256 matrix-pipe cycles: an LDS round trip under load), into the buffer k-step ks - 2 used
                nxt = ks + 2
                for r in frag_reads(half, nxt, lambda nt: xf(nxt & 3, nt), src):
                    s.emit(r, ("xf", nxt))
            if refills is not None and ks >= 1 and (ks - 1) in refills and m in (1, 2):
                for r in refill(ks - 1)[m - 1]:
                    s.emit(r, ("w", ks - 1))
            n = per_gap.get((ks, m), 0)
            while n > 0 and ri < len(ride_ins):
                n -= max(1, n_insts([ride_ins[ri]])) if late_barrier else 1
                emit_ride(s, ride_ins[ri])
                ri += 1
            if (ks, m) == bar and not late_barrier:
                tail_reads()
    if refills is not None and last in refills:
        for piece in refill(last):
            for r in piece:
                s.emit(r, ("w", last))
    while ri < len(ride_ins):
        emit_ride(s, ride_ins[ri])
        ri += 1
    if late_barrier:
        s.wait(lgkm=0)
        s.emit(I_barrier())
        tail_reads()
    s.emit(I_branch("s_branch", "L_dispatch"))
    return s.ins, list(s.vm_log)


def bare_epilogue(name, half, kind, vm_seed=None):
    """Epilogue of `half` with no MFMAs beside it; ends with the workgroup barrier that publishes the tile"""
    s = LogStream()
    s.emit(I_label(f"L_{name}"))
    for tg in (vm_seed or []):
        s.vm_q.append((tg, None))
    s.wait(lgkm=0)                              # (fragments the phase in front requested for a next phase that does not exist)
    s.emit(I_nop(7)); s.emit(I_nop(7))          # the last MFMAs on these accumulators were issued a few states ago
    for item in (epilogue_mask(half) if kind == "mask" else epilogue_dxin(half)):
        emit_ride(s, item)
    s.wait(vm=0, lgkm=0)
    s.emit(I_barrier())
    s.emit(I_branch("s_branch", "L_dispatch"))
    return s.ins


def static_tail_body(vm_seed=None):
    """End of a static trunk (no trunk-input steps to ride in): the epilogue of half B with the HBM copy of half A's last tile
    between its instructions (that tile is complete behind the last B phase's barrier), the workgroup barrier that publishes half
    B's tile, then its copy."""
    s = LogStream()
    s.emit(I_label("L_EPI_B"))
    for tg in (vm_seed or []):
        s.vm_q.append((tg, None))
    s.wait(lgkm=0)                              # (fragments the phase in front requested for a next phase that does not exist)
    s.emit(I_nop(7)); s.emit(I_nop(7))          # the last MFMAs on these accumulators were issued a few states ago
    for item in merge_ride(epilogue_mask("B"), [g for g in copy_groups("A", False)]):
        emit_ride(s, item)
    s.wait(lgkm=0)
    s.emit(I_barrier())
    for grp in copy_groups("B", True):
        emit_ride(s, grp)
    s.wait(lgkm=0)
    s.emit(I_branch("s_branch", "L_dispatch"))
    return s.ins


PRE_S = dict(pk=S(0, 2), off0=S(20))
PRE_V = dict(lane16=V(1))


def pre_issue():
    """The FIRST asm statement of the kernel, between the record loads of the C++ head stage and its arithmetic: weight slots 0..3 <-
    the head segment (8 loads per wave; slots 4..15 ride in the AH phase).  Operands: %[pk] s64, %[off0] s32 = head segment + wave
    stride, %[lane16] v32 = 16 (tid & 63).  Only the loads in flight survive the statement."""
    def ins(dst, name, table, op):
        src = table[name]
        return Inst(op, f"{op} {dst}, %[{name}]", [src], [dst], "salu" if op.startswith("s_") else "valu", dict(d=dst, s=[src]))
    o = [ins(S_PK, "pk", PRE_S, "s_mov_b64"), ins(S_T0, "off0", PRE_S, "s_mov_b32"), ins(V_LANE16, "lane16", PRE_V, "v_mov_b32")]
    o.append(vadd_s(V_OFF, S_T0, V_LANE16))
    for ks in range(4):
        o += [x for piece in refill(ks) for x in piece]
    return o


def prologue():
    """The asm statement (behind the C++ head stage and a workgroup barrier).  Operands (IN_SB / IN_VB): %[pk] s64 packed transposed
    weights, %[phases] s64, %[lds] / %[stash] / %[invlds] / %[rellds] s32 LDS byte addresses (gradient tile, stash tile, 1 / scale
    floats, rel1 halfs -- rel2 lies 256 bytes behind), %[wave] s32, %[act] s64 first fragment slot of this trunk at half A's
    64-point tile, %[mask] s64 sign words of that slot at this wave (+ 512 wave), %[dxin] s64 d_xin row of the tile's first point,
    %[ld4] s32 bytes per d_xin row, %[nvalid] s32 valid points of the tile for THIS wave's rows (0: rows the input does not have, or
    no d_xin wanted), %[tid] v32.  Descriptor 0 of the phase program is not a phase: it carries, per trunk, [2] / [3] the first
    16-k-step segment (byte offset, wave stride), [4] / [5] the head segment, [6] / [7] the bytes per fragment / sign-word slot."""
    o = []
    e = o.append
    e(in_sb(S_PK, "pk")); e(in_sb(S_PH, "phases")); e(in_sb(S_LDS, "lds")); e(in_sb(S_STASH, "stash")); e(in_sb(S_WAVE, "wave"))
    e(in_sb(S_INVLDS, "invlds")); e(in_sb(S_RELLDS, "rellds")); e(in_sb(S_ACT, "act")); e(in_sb(S_MASK, "mask"))
    e(in_sb(S_DXIN, "dxin")); e(in_sb(S_LD4, "ld4"))
    V_TID = V(T0 + 5)
    e(in_sb(S_NVALID, "nvalid")); e(in_vb(V_TID, "tid"))
    # descriptor 0 -> cur (the trunk's constants), descriptor 1 -> nxt
    e(I_s_load(S(S_CUR, 8), S_PH, 0))
    e(I_s_load(S(S_NXT, 8), S_PH, 32))
    e(I_salu("s_add_u32", S(46), S(46), 64, scc=True)); e(I_salu("s_addc_u32", S(47), S(47), 0, scc=True))
    e(I_wait(lgkm=0))
    e(I_salu("s_mov_b32", S_ASTRIDE, S(S_CUR + 6))); e(I_salu("s_mov_b32", S_MSTRIDE, S(S_CUR + 7)))
    lane, l31, h = V(T0), V_L31, V(T0 + 2)
    e(I_valu("v_and_b32", lane, 63, V_TID)); e(I_valu("v_and_b32", l31, 31, V_TID)); e(I_valu("v_lshrrev_b32", h, 5, lane))
    e(I_valu("v_lshlrev_b32", V_LANE16, 4, lane))
    # (the weight slots were requested by the pre-issue statement in front of the head stage)
    # rd = lds + l31 * 528 + 16 h ; rds = stash + ...
    t3 = V(T0 + 3)
    e(I_valu("v_mul_u32_u24", V_RD, LDH_B, l31)); e(I_valu("v_lshlrev_b32", t3, 4, h)); e(I_valu("v_add_u32", V_RD, V_RD, t3))
    e(vadd_s(V_RDS, S_STASH, V_RD)); e(vadd_s(V_RD, S_LDS, V_RD))
    # wr = lds + l31 * 528 + 128 wave + 8 h ; wrs likewise
    e(I_valu("v_mul_u32_u24", V_WR, LDH_B, l31)); e(I_valu("v_lshlrev_b32", t3, 3, h)); e(I_valu("v_add_u32", V_WR, V_WR, t3))
    e(I_salu("s_lshl_b32", S_T0, S_WAVE, 7, scc=True)); e(vadd_s(V_WR, S_T0, V_WR))
    e(vadd_s(V_WRS, S_STASH, V_WR)); e(vadd_s(V_WR, S_LDS, V_WR))
    # copy: lane i of 16-lane group g reads four rows of point 8 (g >> 1) + (i >> 2): cp = lds + that row * 528 + 32 (g & 1) + 8 (i & 3) + 64 wave
    a_, b_ = V(T0 + 3), V(T0 + 4)
    e(I_valu("v_and_b32", a_, 15, lane)); e(I_valu("v_lshrrev_b32", a_, 2, a_))
    e(I_valu("v_lshrrev_b32", b_, 5, lane)); e(I_valu("v_lshlrev_b32", b_, 3, b_))
    e(I_valu("v_add_u32", a_, a_, b_)); e(I_valu("v_mul_u32_u24", V_CP, LDH_B, a_))
    e(I_valu("v_lshrrev_b32", a_, 4, lane)); e(I_valu("v_and_b32", a_, 1, a_)); e(I_valu("v_lshlrev_b32", a_, 5, a_))
    e(I_valu("v_and_b32", b_, 3, lane)); e(I_valu("v_lshlrev_b32", b_, 3, b_))
    e(I_valu("v_add_u32", a_, a_, b_)); e(I_valu("v_add_u32", V_CP, V_CP, a_))
    e(I_salu("s_lshl_b32", S_T0, S_WAVE, 6, scc=True)); e(I_salu("s_add_u32", S_T0, S_T0, S_LDS, scc=True))
    e(vadd_s(V_CP, S_T0, V_CP))
    # rel: the lane's eight points of a 16-point group start at 8 (g >> 1): rel = rellds + 2 * that
    e(I_valu("v_lshrrev_b32", a_, 5, lane)); e(I_valu("v_lshlrev_b32", a_, 4, a_)); e(vadd_s(V_REL, S_RELLDS, a_))
    # sign words: 8 bytes per lane
    e(I_valu("v_lshlrev_b32", V_MOFF, 3, lane))
    # d_xin: row of point (lane & 31), columns 64 wave + 4 h: V_DXB = ((lane & 31) * ld + 64 wave + 4 h) * 4 ; 1 / scale: V_INVA = invlds + 4 (lane & 31)
    e(I_valu("v_mul_lo_u32", V_DXB, S_LD4, l31, text=f"v_mul_lo_u32 {V_DXB}, {S_LD4}, {l31}"))
    e(I_valu("v_lshlrev_b32", a_, 4, h)); e(I_valu("v_add_u32", V_DXB, V_DXB, a_))
    e(I_salu("s_lshl_b32", S_T0, S_WAVE, 8, scc=True)); e(vadd_s(V_DXB, S_T0, V_DXB))
    e(I_valu("v_lshlrev_b32", a_, 2, l31)); e(vadd_s(V_INVA, S_INVLDS, a_))
    # MODE.FP16_OVFL (bit 23): an fp16 conversion that overflows is clamped to +-65504 (restored at L_end)
    e(Inst("s_setreg", "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1", [], [], "salu", dict(d=None, s=[1], field="fp16_ovfl")))
    e(I_wait(lgkm=0))
    for k0 in range(2):
        for r in frag_reads("A", k0, lambda nt: xf(k0, nt)):
            e(r)
    return o


TIMING = False              # --timing: every dispatcher visit stores an s_memtime stamp (lane 0) at *s[80:81] (+= 4): debug builds only


def timing_store():
    """lane 0 stores the low dword of s_memtime at s[80:81] and advances the pointer.  The store is one more entry of the in-order
    VMEM queue: counted waits for older loads only get stricter."""
    return [raw("s_memtime s[82:83]"), raw("s_waitcnt lgkmcnt(0)"), raw("v_mov_b32 v126, s82"), raw("v_mov_b32 v127, 0"),
            raw("s_mov_b64 s[84:85], exec"), raw("s_mov_b64 exec, 1"), raw("global_store_dword v127, v126, s[80:81]"),
            raw("s_mov_b64 exec, s[84:85]"), raw("s_add_u32 s80, s80, 4"), raw("s_addc_u32 s81, s81, 0")]


DISPATCH_ORDER = ("A16", "B16", "A16S", "B16S", "AH", "BH", "A16F", "B16L", "EPI_B", "AX", "BX", "AXS", "BXS", "BXD", "EPI_DXB")


def dispatcher(bodies):
    o = [I_label("L_dispatch")]
    e = o.append
    if TIMING:
        o.extend(timing_store())
    for k in range(8):
        e(I_salu("s_mov_b32", S(S_CUR + k), S(S_NXT + k)))
    e(I_s_load(S(S_NXT, 8), S_PH, 0))
    e(I_salu("s_add_u32", S(46), S(46), 32, scc=True)); e(I_salu("s_addc_u32", S(47), S(47), 0, scc=True))
    e(I_salu("s_mul_i32", S_T0, S_WAVE, S(S_CUR + D_R1W))); e(I_salu("s_add_u32", S_R1, S(S_CUR + D_R1), S_T0, scc=True))
    for name in DISPATCH_ORDER:
        if name not in bodies:
            continue
        e(I_s_cmp("s_cmp_eq_u32", S(S_CUR + D_BODY), BODY[name]))
        e(I_branch("s_cbranch_scc1", f"L_{name}"))
    e(I_branch("s_branch", "L_end"))
    return o


def build():
    """-> (pre-issue statement, program, bodies).  The VMEM operations a phase may find outstanding at its entry are those of the phase(s) in front of
    it, as generated: the seed of its wait-count model (every counted wait is then exact or stricter, never too weak -- and the
    simulator, which keeps the real queues, runs every phase program the host builder can emit)."""
    bodies = {}

    def gen(name, *a, **k):
        bodies[name], log = phase_body(name, *a, **k)
        return log
    pro_vm = [("w", ks) for ks in range(4) for _ in range(2)]
    ah = gen("AH", "A", 4, msk=True, vm_seed=pro_vm, stream=None if "nostream" in EXP else list(range(4, 16)))
    bh = gen("BH", "B", 4, ride="mask", refills=[0, 1, 2, 3], msk=True, vm_seed=pro_vm + ah)
    a_like = ["mskA"] + ["cpst"] * 8            # what an A phase of a 16-k-step layer issues (every B16* follows one)
    gen("A16F", "A", 16, ride="mask", copy=True, msk=True, vm_seed=pro_vm + ah + bh)
    b16 = gen("B16", "B", 16, ride="mask", copy=True, refills=list(range(16)), msk=True, vm_seed=a_like)
    assert gen("B16S", "B", 16, ride="mask_stash", copy=True, refills=list(range(16)), msk=True, vm_seed=a_like) == b16
    a16 = gen("A16", "A", 16, ride="mask", copy=True, msk=True, vm_seed=b16)
    assert EXP or a16 == a_like, a16
    assert gen("A16S", "A", 16, ride="mask_stash", copy=True, msk=True, vm_seed=b16) == a16
    gen("B16L", "B", 16, ride="mask", copy=True, msk=True, vm_seed=a_like)
    bodies["EPI_B"] = static_tail_body(vm_seed=["mskB"] + ["cpst"] * 8)
    # the trunk-input steps of a dynamic trunk: x0 from the tile (the layer-0 pre-activation gradient), then -- one skip layer --
    # the skip layer's input part from the stash, accumulated on top; d_xin leaves in fp32
    ax = gen("AX", "A", 16, ride="mask", copy=True, vm_seed=b16)
    assert EXP or ax == ["cpst"] * 8
    bx = gen("BX", "B", 16, copy=True, refills=list(range(16)), tail_src="s")
    gen("AXS", "A", 16, init=False, src="s", vm_seed=bx, tail_src="s")
    gen("BXS", "B", 16, init=False, src="s", ride="dxin", tail=False)
    gen("BXD", "B", 16, ride="dxin", copy=True, tail=False)
    bodies["EPI_DXB"] = bare_epilogue("EPI_DXB", "B", "dxin")
    prog = ([raw("s_mov_b64 s[80:81], %[dbg]")] if TIMING else []) + prologue()
    prog.append(I_branch("s_branch", "L_dispatch"))
    for name in bodies:
        prog += bodies[name]
    prog += dispatcher(bodies)
    prog.append(I_label("L_end"))
    # No wait for the vector-memory queue here: what is outstanding when the last phase ends are STORES (fragments, d_xin) --
    # check_bwd.py asserts that no register has a load outstanding into it at this point --, and a persistent workgroup's next
    # item starts its head stage while they drain.  (The kernel's record DMA for that item was issued before this statement and
    # is older than every load the phases waited for: it has landed.)
    prog.append(I_wait(lgkm=0) if not TIMING else I_wait(vm=0, lgkm=0))
    prog.append(Inst("s_setreg", "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 0", [], [], "salu", dict(d=None, s=[0], field="fp16_ovfl")))
    if TIMING:
        prog += timing_store() + [I_wait(vm=0, lgkm=0)]
    return pre_issue(), prog, bodies


def render_b(prog):
    lines = []
    for ins in prog:
        t = ins.text
        if ins.kind == "label":
            t = t.replace("L_", "L_h3b_%=_")
        elif ins.kind == "branch":
            t = t.replace(" L_", " L_h3b_%=_")
        lines.append('    "' + t + '\\n\\t"')
    return "\n".join(lines) + "\n"


def lint(bodies):
    errs = []
    for name, ins in bodies.items():
        errs += lint_straight(ins, name)
        for i in ins:       # gfx90a and later: a VGPR / AGPR tuple starts at an even register
            for x in list(i.args.values()) + [y for v_ in i.args.values() if isinstance(v_, list) for y in v_]:
                if isinstance(x, Reg) and x.f in ("v", "a") and x.n >= 2 and x.i % 2:
                    errs.append(f"{name}: `{i.text}` uses the odd-aligned tuple {x}")
    return errs


def main():
    global TIMING
    TIMING = "--timing" in sys.argv
    pre, prog, bodies = build()
    errs = lint(bodies)
    for e_ in errs[:40]:
        print("LINT:", e_)
    if errs:
        sys.exit(f"{len(errs)} hazard(s)")
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "nsff_pl_amd", "csrc",
                       "field_bwd_h3b_body_timing.inc" if TIMING else (f"field_bwd_h3b_body_{EXP}.inc" if EXP else "field_bwd_h3b_body.inc"))
    clob = ", ".join([f'"v{i}"' for i in range(24, 256)] + [f'"a{i}"' for i in range(128)] + [f'"s{i}"' for i in range(40, 100)] +
                     ['"vcc"', '"scc"', '"memory"'])
    consts = "".join(f"#define H3B_BODY_{k} {v}\n" for k, v in BODY.items())
    macro = lambda name, insts: f"#define {name} \\\n" + render_b(insts).replace("\n", " \\\n").rstrip(" \\\n") + "\n"
    pre_wr = sorted({r for i in pre for r in i.wr if r[0] in ("v", "a") or (r[0] == "s" and r[1] < 100)})
    pre_clob = ", ".join([f'"{f}{i}"' for f, i in pre_wr] + ['"scc"', '"memory"'])
    text = ("// GENERATED by tools/h3asm/gen_bwd.py -- do not edit.  The hand-scheduled body of nsff_field_bwd_kernel_h3b (the data-gradient\n"
            "// chain of one 128-point tile and one trunk; registers v24..v255, a0..a127, s40..s99 are its own while it runs) and H3B_PRE,\n"
            "// the statement in front of the head stage that requests the first sixteen weight slots.\n" + consts +
            "#define H3B_PRE_CLOBBERS " + pre_clob + "\n" + macro("H3B_PRE", pre) +
            "#define H3B_CLOBBERS " + clob + "\n" + macro("H3B_BODY", prog))
    with open(out, "w") as f:
        f.write(text)
    n_m = sum(1 for i in prog if i.kind == "mfma")
    n_all = sum(1 for i in prog if i.kind not in ("label", "other"))
    print(f"wrote {os.path.normpath(out)}: {n_all} instructions, {n_m} MFMAs")
    for name, ins in bodies.items():
        print(f"  {name:9s} {sum(1 for i in ins if i.kind not in ('label', 'other')):5d} instructions, {sum(1 for i in ins if i.kind == 'mfma'):4d} MFMAs")


if __name__ == "__main__":
    main()
