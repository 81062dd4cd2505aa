#!/usr/bin/env python
"""Generator of the hand-scheduled trunk body of ``nsff_field_kernel_h3a`` (nsff_pl_amd/csrc/field_h3a.hip).

    python tools/h3asm/gen.py            # writes nsff_pl_amd/csrc/field_h3a_body.inc, after linting the stream
    python tools/h3asm/check.py          # runs the generated stream in the functional simulator against numpy

What the body is: one wave per SIMD (four waves, 512 registers each), 128 points per workgroup as
two 64-point HALVES A and B.  A wave owns 64 neurons; the CURRENT layer's weights of those neurons (hi + lo halfs, 16 k-steps x
16 registers) are RESIDENT in its 256 accumulation registers and are multiplied with both halves, half a layer apart:

    ... | A(l): MFMAs on acc_A from X_A, riding: epilogue of B(l-1) | B(l): MFMAs on acc_B from X_B, riding: epilogue of A(l),
          refill of every weight slot with layer l+1 as soon as B has used it | A(l+1) ...

so the matrix pipe never waits for an epilogue, a barrier-to-barrier phase is 192 MFMAs (6 144 pipe cycles), every weight byte
crosses the CU's vector-memory path once per 128 points and every activation fragment is read from LDS once per wave.  Short
segments (the 64- / 128-column input layers, the skip layer's input part) run as bare phases; the skip layer's input tile is
restored from a register stash (xyz part) and from the time-code rows (dynamic trunk) while the matrix pipe runs.

The stream is produced as ``Inst`` objects (isa.py): the same list is printed, linted for wait-state hazards and executed by the
simulator.  Register map (asm-owned; the compiler keeps v0..v23, s0..s39 and VCC):
    v24..v39  addresses      v40..v63 epilogue temporaries     v64..v95 input stash     v96..v111 XH[2][2] fragments
    v112..v119 XL[2]         v120..v127 spare                  v128..v191 acc_A         v192..v255 acc_B
    a[16 j .. 16 j + 15] weight slot j = k-step j of the resident segment: [mt0 hi | mt0 lo | mt1 hi | mt1 lo] x 4 registers
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa import *          # noqa: F401,F403

LDH_B = 528                 # bytes per LDS row (264 halfs)
HALF_B = 64 * LDH_B         # 33792: rows 64..127 = half B
NT_B = 32 * LDH_B           # 16896: second 32-point tile of a half
PLANE_B = 128 * LDH_B       # 67584: lo plane behind the hi plane

# ---- VGPRs
V_RD_H, V_RD_L, V_WR_H, V_WR_L, V_BIAS, V_LANE16, V_ST_H, V_ST_L = (V(i) for i in range(24, 32))
V_OFF, V_BADDR = V(32), V(33)
V_TPA, V_TPB = V(34, 2), V(36, 2)
V_N4, V_TMP = V(38), V(39)
T0 = 40                     # temporaries v40..v63
STASH = 64                  # v64..v95: [half][plane][8 dwords]
XH0 = 96                    # XH[b][nt] = v[96 + 8 b + 4 nt : +4]
XL0 = 112                   # XL[nt] = v[112 + 4 nt : +4]
XL2 = 120                   # v120..v127: lo fragments of a phase's LAST k-step (so that the phase barrier can sit a k-step early)
ACC = {"A": 128, "B": 192}

# ---- SGPRs (asm-owned: s40..s99)
S_PK = S(40, 2)
S_LDS, S_BIASLDS, S_WAVE, S_INT = S(42), S(43), S(44), S(45)
S_PH = S(46, 2)             # pointer to the NEXT phase descriptor to fetch
S_SAVE = S(48, 2)
S_R1, S_R2, S_SEL = S(50), S(51), S(52)
S_T0, S_T1 = S(53), S(54)
S_RAWLDS = S(55)            # byte address of the raw-record image in LDS (the heads' pre-activation sums go there)
S_CUR = 56                  # s56..s63 current descriptor
S_NXT = 64                  # s64..s71 next descriptor (being fetched)
D_BODY, D_FLAGS, D_BIAS, D_N1, D_R1, D_R1W, D_R2, D_R2W = range(8)
# flag bits: tail initialises the other accumulator from the bias table; the rebuild has a time-code part; the phase restores
# an input tile at all (A4 / A8 also run the first layer, where the tile is still the encoder's)
F_INIT, F_REBUILD_T, F_REBUILD = 0, 1, 2

BODY = dict(END=0, A16R=1, B16R=2, B16X=3, A4=4, A8=5, B4=6, B8=7, EPI_B=9, A4F=10, A8F=11, B16L=12, HEAD=13, B16RS=14, A16RS=15,
            SAVE_LAST=16, B16LP=17)

# ---- the SAVE build (build(save=True) -> H3A_BODY_SAVE, the body of nsff_field_kernel_h3a_save: the TRAINING forward).  The
# same phases; additionally
#   * every phase over a trunk activation (A16R, B16R, B16X, B16L) copies ITS OWN half's input tile -- the activation the
#     backward pass needs -- to HBM in the fragment order of the weight-gradient GEMM: 8 one-KiB blocks per wave, each = four
#     transposing LDS reads (ds_read_b64_tr_b16, two per plane), four packed adds (hi + lo, rounded to nearest) and one 16-byte
#     store, riding in the phase's MFMA gaps between the epilogue's instructions (the tile is complete -- the barrier in front
#     of the phase -- and stays untouched until the next phase's ride overwrites it); the LAST activation, which no MFMA phase
#     reads, is copied by a phase of its own (SAVE_LAST) in front of the heads;
#   * every epilogue also collects the ReLU sign bits (bit 16 nt + 4 q + e of word mt, the order nsff_field_backward reads) --
#     v_min_u32 1, x and v_lshl_or_b32 per value -- and stores the two words per lane.
# Destinations are running pointers (asm-owned SGPRs): S_ACT / S_MASK point at the current activation slot of the tile of
# half A (half B: + 32 KiB / + 2 KiB) and advance by one slot (S_ASTRIDE / S_MSTRIDE bytes) when half B's copy / sign words
# of a layer are out.  Registers v14..v23 are the body's too in this build.
SAVE = False
V_CP_H, V_CP_L, V_CPOFF, V_MSKW = V(14), V(15), V(16), V(17)
S_ACT, S_MASK, S_ASTRIDE, S_MSTRIDE = S(88, 2), S(90, 2), S(92), S(93)
CPT = 18                    # v18..v23: the copy's temporaries (v40..v63 belong to the epilogue and to the skip layer's rebuild)


# ---- the sigma ride (view-direction static trunk: static_sigma reads the LAST TRUNK activation, nerf.py:169, one layer before the
# trunk's end): the epilogues of that layer also accumulate  sum_n w_sigma[n] relu(acc[n])  over this lane's 32 neurons of each
# point in fp32 (v_fmac on the values the ReLU just produced, weights in the registers of the input stash, which is dead behind
# the last skip layer) and leave the 8 partial sums per point (4 waves x 2 lane halves) at floats 4..11 of the point's raw record
# in LDS; the kernel's records loop adds them up.  B16RS = B16R + the ride on half A's epilogue (it also loads the weights from
# the bias-table row (flags >> 16)), A16RS = A16R + the ride on half B's epilogue.
SIG = {"A": (V(56), V(57)), "B": (V(58), V(59))}     # [nt] partial sums of the half's two 32-point tiles
V_SIGADDR = V(60)


def ws(mt, q, e):            # sigma weight of neuron 64 wave + 32 mt + 8 q + 4 (lane >> 5) + e: v64..v95 (the stash registers)
    return V(STASH + 16 * mt + 4 * q + e)


def I_v_fmac(d, a, b):
    return Inst("v_fmac_f32", f"v_fmac_f32_e32 {d}, {a}, {b}", [a, b, d], [d], "valu", dict(d=d, s=[a, b, d]))


def xh(b, nt):
    return V(XH0 + 8 * b + 4 * nt, 4)


def xl(nt):
    return V(XL0 + 4 * nt, 4)


def acc(half, mt, nt):
    return V(ACC[half] + 16 * (2 * mt + nt), 16)


def wslot(ks, mt, part):     # part 0 = hi, 1 = lo
    return A(16 * ks + 8 * mt + 4 * part, 4)


def other(half):
    return "B" if half == "A" else "A"


def half_off(half):
    return 0 if half == "A" else HALF_B


# ----------------------------------------------------------------------------------------------------------------------
class Stream:
    """Instruction list with the two in-order return queues modelled, so that every s_waitcnt count is COMPUTED from what
    is outstanding (``need_lds(tag)`` / ``need_vm(tag)``: wait until the operation tagged `tag` has returned)."""

    def __init__(self):
        self.ins = []
        self.lds_q = []      # tags of outstanding LDS operations, oldest first
        self.vm_q = []

    def emit(self, i, tag=None, group=None):
        """group: name of the guarded cluster the instruction belongs to (it is issued only when the cluster's flag is set), or
        None.  A counted wait for an operation may only count the later operations that are CERTAIN to have been issued with
        it: the unguarded ones and those of its own cluster -- then the wait is exact or stricter, never too weak."""
        self.ins.append(i)
        if i.kind in ("lds_r", "lds_w"):
            self.lds_q.append((tag, group))
        elif i.kind == "vmem":
            self.vm_q.append((tag, group))
        return i

    @staticmethod
    def _behind(q, tag):
        pos = max(i for i, (t, _) in enumerate(q) if t == tag)
        grp = q[pos][1]
        return sum(1 for (_, g) in q[pos + 1:] if g is None or g == grp)

    def need_lds(self, tag):
        """emit a wait that covers the LDS op `tag` (no-op if it is not outstanding)"""
        if not any(t == tag for t, _ in self.lds_q):
            return
        self.wait(lgkm=min(self._behind(self.lds_q, tag), 15))

    def need_vm(self, tag):
        if not any(t == tag for t, _ in self.vm_q):
            return
        self.wait(vm=min(self._behind(self.vm_q, tag), 63))

    def wait(self, vm=None, lgkm=None):
        # (two waits in a row for different counters are one instruction)
        last = self.ins[-1] if self.ins else None
        if last is not None and last.kind == "wait" and ((vm is None) or last.args["vm"] is None) and ((lgkm is None) or last.args["lgkm"] is None):
            self.ins[-1] = I_wait(vm if vm is not None else last.args["vm"], lgkm if lgkm is not None else last.args["lgkm"])
        else:
            self.emit(I_wait(vm, lgkm))
        # afterwards at most `n` operations are outstanding: the newest n of the model (guarded ones included: if they were not
        # issued, older ones may still be in flight -- keeping the newest n entries would forget those, so unguarded entries
        # are only dropped while n unguarded newer ones remain)
        if lgkm is not None:
            self.lds_q = self._keep(self.lds_q, lgkm)
        if vm is not None:
            self.vm_q = self._keep(self.vm_q, vm)

    @staticmethod
    def _keep(q, n):
        if n == 0:
            return []
        kept, sure = [], 0
        for ent in reversed(q):
            if sure >= n:
                break
            kept.append(ent)
            if ent[1] is None:
                sure += 1
        return list(reversed(kept))


# ----------------------------------------------------------------------------------------------------------------------
NOSWAP_STORES = os.environ.get("H3A_NOSWAP_STORES", "1") == "1"    # 0: the v_permlane32_swap + 16-byte store form (A/B builds)
EXP = os.environ.get("H3A_EXP", "")          # timing experiments (results are garbage): nomix, noswap, nowrite, noride, andsub, nolo, nolh


def epilogue_unit(half, u, tset, sig=False):
    """ReLU -> hi / lo split -> four 8-byte LDS stores (two per plane) of (tile u>>1, quad pair p = u&1) of `half`'s accumulators.
    In place on the accumulator registers; 8 temporaries from set `tset`.  (Until late in round 4: lanes i, i+32 traded halves with
    v_permlane32_swap for two 16-byte stores -- two instructions more per unit, +0.6 % on the C2 step.)"""
    t, p = u >> 1, u & 1
    mt, nt = t >> 1, t & 1
    a = acc(half, mt, nt)
    x = [a.sub(4 * p + e) for e in range(4)] + [a.sub(4 * p + 8 + e) for e in range(4)]
    H = [V(T0 + 8 * tset + k) for k in range(4)]
    L = [V(T0 + 8 * tset + 4 + k) for k in range(4)]
    out = [I_v_max0(r, r) for r in x]
    if SAVE:                # sign bits of the eight values: bit 16 nt + 4 q + e of word mt (one word at a time: V_MSKW)
        if u in (0, 4):
            out.append(I_valu("v_mov_b32", V_MSKW, 0))
        for j in range(8):
            out += [I_valu("v_min_u32", V_TMP, 1, x[j]), I_v_lshl_or(V_MSKW, V_TMP, 16 * nt + 4 * (p + 2 * (j >> 2)) + (j & 3), V_MSKW)]
    if sig:                 # x[j] = neuron 8 q + 4 h + e of the tile, q = p + 2 (j >> 2), e = j & 3
        out += [I_v_fmac(SIG[half][nt], ws(mt, p + 2 * (j >> 2), j & 3), x[j]) for j in range(8)]
    out += [I_v_cvt_pkrtz(H[k], x[2 * k], x[2 * k + 1]) for k in range(4)]
    for k in range(4):
        out += [I_v_sub_lo_half(x[2 * k], H[k], x[2 * k]), I_v_sub_hi_half(x[2 * k + 1], H[k], x[2 * k + 1])]
    out += [I_v_cvt_pkrtz(L[k], x[2 * k], x[2 * k + 1]) for k in range(4)]
    off = half_off(half) + NT_B * nt + 64 * mt + 16 * p
    if NOSWAP_STORES:
        # a lane holds neurons 8 p + 4 h .. + 3 and 16 + 8 p + 4 h .. + 3 of the 32-neuron tile (h = lane >> 5): four 8-byte stores
        # (write base + 8 h) instead of four v_permlane32_swap + two 16-byte stores (write base + 32 h)
        out += [I_ds_write_b64(V_WR_H, V(H[0].i, 2), off), I_ds_write_b64(V_WR_H, V(H[2].i, 2), off + 32),
                I_ds_write_b64(V_WR_L, V(L[0].i, 2), off), I_ds_write_b64(V_WR_L, V(L[2].i, 2), off + 32)]
    else:
        out += [I_v_permlane32_swap(H[0], H[2]), I_v_permlane32_swap(H[1], H[3]),
                I_v_permlane32_swap(L[0], L[2]), I_v_permlane32_swap(L[1], L[3])]
        out += [I_ds_write_b128(V_WR_H, V(H[0].i, 4), off), I_ds_write_b128(V_WR_L, V(L[0].i, 4), off)]
    if "nomix" in EXP:
        out = [I_valu("v_mov_b32", i.args["d"], i.args["s"][1]) if i.op.startswith("v_fma_mix") else i for i in out]
    if "noswap" in EXP:
        out = [i for i in out if i.op != "v_permlane32_swap"]
    if "nowrite" in EXP:
        out = [i for i in out if i.kind != "lds_w"]
    if "nocvt" in EXP:
        out = [I_valu("v_mov_b32", i.args["d"], i.args["s"][0]) if i.op == "v_cvt_pkrtz_f16_f32" else i for i in out]
    if "nomax" in EXP:
        out = [i for i in out if i.op != "v_max_f32"]
    if "noride" in EXP:
        out = []
    if SAVE and u in (3, 7):   # word mt complete: lane's 4 bytes at S_MASK + 8 (lane) + 4 mt (+ 2 KiB: the tile of half B)
        out += [I_valu("v_lshrrev_b32", V_TMP, 1, V_LANE16),
                _nt(I_gstore_s(V_TMP, V_MSKW, S_MASK, 4 * mt + (2048 if half == "B" else 0)))]
    return out


def _nt(store):
    """a SAVE-build store (activation fragments, sign words) as non-temporal: written once, read once by another kernel -- as the
    compiler-scheduled kernels' fragment_block and the data-gradient body do.  Same box, interleaved (round 6): training forward
    of 196 608 points x both trunks 996-1007 -> 989-994 us, C2 training step 5.81 -> 5.76 ms.  (H3A_EXP=tsave: plain stores, for A/B)"""
    if "tsave" not in EXP:
        store.text += " nt"
    return store


def copy_groups(half):
    """The HBM copy of X_<half> (see SAVE) as a list of GROUPS of ride items -- the caller spreads them between other riding
    instructions so that an LDS round trip lies between a group's reads and the next group's ('NEED_LDS', tag) marker."""
    hb = 0 if half == "A" else 1
    t = [V(CPT + k) for k in range(6)]
    groups = [[I_salu("s_lshl_b32", S_T0, S_WAVE, 10, scc=True),
               I_valu("v_add_u32", V_CPOFF, S_T0, V_LANE16, text=f"v_add_u32_e32 {V_CPOFF}, {S_T0}, {V_LANE16}")] +
              ([I_valu("v_add_u32", V_CPOFF, 32768, V_CPOFF, text=f"v_add_u32_e32 {V_CPOFF}, 0x8000, {V_CPOFF}")] if hb else [])]
    for b in range(8):      # block wave + 4 b = (16-point group b >> 1, 32-neuron block wave + 4 (b & 1))
        imm = half_off(half) + (b >> 1) * 16 * LDH_B + (b & 1) * 256
        groups[-1] += [(I_ds_read_tr(V(t[0].i, 2), V_CP_H, imm), ("cp", b, 0)), (I_ds_read_tr(V(t[2].i, 2), V_CP_L, imm), ("cp", b, 0))]
        groups.append([("NEED_LDS", ("cp", b, 0)), I_v_pk_add_f16(t[0], t[0], t[2]), I_v_pk_add_f16(t[1], t[1], t[3]),
                       (I_ds_read_tr(V(t[2].i, 2), V_CP_H, imm + 4 * LDH_B), ("cp", b, 1)),
                       (I_ds_read_tr(V(t[4].i, 2), V_CP_L, imm + 4 * LDH_B), ("cp", b, 1))])
        groups.append([("NEED_LDS", ("cp", b, 1)), I_v_pk_add_f16(t[2], t[2], t[4]), I_v_pk_add_f16(t[3], t[3], t[5]),
                       _nt(I_gstore_s(V_CPOFF, V(t[0].i, 4), S_ACT, 0)),
                       I_valu("v_add_u32", V_CPOFF, 4096, V_CPOFF, text=f"v_add_u32_e32 {V_CPOFF}, 0x1000, {V_CPOFF}")])
    if hb:                  # both halves of this slot are out: the pointer moves on to the next slot
        groups[-1] += [I_salu("s_add_u32", S(S_ACT.i), S(S_ACT.i), S_ASTRIDE, scc=True), I_salu("s_addc_u32", S(S_ACT.i + 1), S(S_ACT.i + 1), 0, scc=True)]
    return groups


def merge_ride(main, groups):
    """`groups` spread evenly through the list `main` (each group stays contiguous)"""
    if not groups:
        return list(main)
    out, n, g = [], len(main), len(groups)
    for k in range(g):
        out += groups[k]
        out += main[k * n // g:(k + 1) * n // g]
    return out


def epilogue_stream(half, sig=False, wait_ws=False):
    """sig: with the sigma ride (see SIG); wait_ws: the phase requested the sigma weights at its start -- the marker
    ('NEED_LDS', 'ws') in front of the first unit makes the phase wait for them."""
    out = []
    if sig:
        out += [I_valu("v_mov_b32", SIG[half][0], 0), I_valu("v_mov_b32", SIG[half][1], 0)]
        if wait_ws:
            out.append(("NEED_LDS", "ws"))
    for u in range(8):
        out += epilogue_unit(half, u, u & 1, sig)
    if sig:
        # record of point 64 hb + 32 nt + (lane & 31): rawlds + 64 point + 16 + 8 wave + 4 (lane >> 5)
        t0 = V_SIGADDR
        out += [I_valu("v_and_b32", t0, 0x1f0, V_LANE16), I_valu("v_lshlrev_b32", t0, 2, t0),
                I_valu("v_lshrrev_b32", V_TMP, 7, V_LANE16), I_valu("v_and_b32", V_TMP, 4, V_TMP),
                I_valu("v_add_u32", t0, t0, V_TMP),
                I_salu("s_lshl_b32", S_T0, S_WAVE, 3, scc=True), I_salu("s_add_u32", S_T0, S_T0, S_RAWLDS, scc=True),
                I_valu("v_add_u32", t0, S_T0, t0, text=f"v_add_u32_e32 {t0}, {S_T0}, {t0}")]
        hb = 0 if half == "A" else 1
        out += [I_ds_write_b32(t0, SIG[half][nt], 16 + 4096 * hb + 2048 * nt) for nt in range(2)]
    if SAVE and half == "B":  # the sign words of both halves of this layer are out: next slot
        out += [I_salu("s_add_u32", S(S_MASK.i), S(S_MASK.i), S_MSTRIDE, scc=True), I_salu("s_addc_u32", S(S_MASK.i + 1), S(S_MASK.i + 1), 0, scc=True)]
    return out


def sigma_weight_reads():
    """WS := row (flags >> 16) of the bias table, this lane's 32 neurons in accumulator order (the mapping of init_reads)"""
    out = [I_salu("s_lshr_b32", S_T0, S(S_CUR + D_FLAGS), 16, scc=True),
           I_valu("v_add_u32", V_TMP, S_T0, V_BIAS, text=f"v_add_u32_e32 {V_TMP}, {S_T0}, {V_BIAS}")]
    for mt in range(2):
        for q in range(4):
            out.append(I_ds_read_b128(V(STASH + 16 * mt + 4 * q, 4), V_TMP, 128 * mt + 32 * q))
    return out


def init_reads(half):
    """acc_<half> := bias of the segment whose table offset is in V_BADDR (16 x ds_read_b128 straight into the accumulators)"""
    out = []
    for mt in range(2):
        for nt in range(2):
            for q in range(4):
                out.append(I_ds_read_b128(acc(half, mt, nt).sub(4 * q, 4), V_BADDR, 128 * mt + 32 * q))
    return out


def frag_reads_h(half, ks, b):
    return [I_ds_read_b128(xh(b, nt), V_RD_H, half_off(half) + NT_B * nt + 32 * ks) for nt in range(2)]


def xl2(nt):
    return V(XL2 + 4 * nt, 4)


def frag_reads_l(half, ks, second=False):
    return [I_ds_read_b128(xl2(nt) if second else xl(nt), V_RD_L, half_off(half) + NT_B * nt + 32 * ks) for nt in range(2)]


def mfmas(half, ks, second_xl=False):
    b = ks & 1
    out = []
    for part, xsel in ((1, "h"), (0, "h"), (0, "l")):      # Wl.xh, Wh.xh, Wh.xl: the lo fragments are needed last
        for mt in range(2):
            for nt in range(2):
                d = acc(half, mt, nt)
                bop = xh(b, nt) if xsel == "h" else (xl2(nt) if second_xl else xl(nt))
                # (timing experiments, results garbage: "nolo" drops the Wh.xl products -- what a two-slot scheme would issue --,
                #  "nolh" the Wl.xh products as well: one MFMA per product)
                if ("nolo" in EXP or "nolh" in EXP) and xsel == "l" or ("nolh" in EXP and part == 1):
                    out.append(raw("s_nop 0"))
                else:
                    out.append(I_mfma(d, wslot(ks, mt, part), bop, d))
    return out


def refill(ks, one=False):
    """slot ks <- slice ks of the next segment that uses it: r1 for ks < n1 (the next segment), r2 beyond (the one after).
    Returns the pieces [address arithmetic, load, load, load, load] (one piece per MFMA gap).
    one: the phase requests from ONE stream (a 16-k-step segment follows: every B phase but the one in front of a skip layer's
    input part) -- V_OFF runs through it (set by refill_start), 5 instead of 9 instructions per slot."""
    loads = [[I_gload_x4_s(A(16 * ks + 4 * c, 4), V_OFF, S_PK, 1024 * c)] for c in range(4)]
    if one:
        loads[3].append(I_valu("v_add_u32", V_OFF, 4096, V_OFF, text=f"v_add_u32_e32 {V_OFF}, 0x1000, {V_OFF}"))
        return [[]] + loads
    head = [I_s_cmp("s_cmp_gt_u32", S(S_CUR + D_N1), ks),
            I_salu("s_cselect_b32", S_SEL, S_R1, S_R2),
            I_valu("v_add_u32", V_OFF, S_SEL, V_LANE16, text=f"v_add_u32_e32 {V_OFF}, {S_SEL}, {V_LANE16}"),
            I_salu("s_add_u32", S_R1, S_R1, 4096, scc=True), I_salu("s_add_u32", S_R2, S_R2, 4096, scc=True)]
    return [head] + loads


def refill_start(stream):
    """V_OFF := start of the phase's single refill stream (S_R1 or S_R2 of the dispatcher) + this lane's 16 bytes"""
    return I_valu("v_add_u32", V_OFF, stream, V_LANE16, text=f"v_add_u32_e32 {V_OFF}, {stream}, {V_LANE16}")


def refill_flat(ks, one=False):
    return [i for piece in refill(ks, one) for i in piece]


def rebuild_parts(half, name):
    """Restore the input tile of `half` (flag F_REBUILD): xyz part from the register stash (4 stores); time-code part (dynamic
    trunk, flag F_REBUILD_T) re-read from its rows -- part 1 requests the rows, part 2 (a few k-steps later) splits them into
    hi / lo halfs and stores them.  Each part is a guarded cluster (forward branch when its flag is clear).
    Items are instructions or the marker ('NEED_VM', tag)."""
    hb = 0 if half == "A" else 1
    ho = half_off(half)
    st = STASH + 16 * hb
    ptr = V_TPA if half == "A" else V_TPB
    F = [V(T0 + 4 * j, 4) for j in range(4)]                # four float4 of time code: v40..v55
    p1 = [I_s_cmp("s_bitcmp1_b32", S(S_CUR + D_FLAGS), F_REBUILD), I_branch("s_cbranch_scc0", f"L_{name}_nr1"),
          I_ds_write_b128(V_ST_H, V(st, 4), ho), I_ds_write_b128(V_ST_H, V(st + 4, 4), ho + 16),
          I_ds_write_b128(V_ST_L, V(st + 8, 4), ho), I_ds_write_b128(V_ST_L, V(st + 12, 4), ho + 16),
          I_s_cmp("s_bitcmp1_b32", S(S_CUR + D_FLAGS), F_REBUILD_T), I_branch("s_cbranch_scc0", f"L_{name}_nr1")]
    for j in range(4):
        p1 += [I_valu("v_mov_b32", F[j].sub(e), 0) for e in range(4)]
    for j in range(4):
        p1 += [I_v_cmp_lt_u32_vcc(j, V_N4), I_s_and_saveexec(S_SAVE), ("TLOAD", I_gload_x4_v(F[j], ptr, 16 * j)), I_s_mov_exec(S_SAVE)]
    p1.append(I_label(f"L_{name}_nr1"))
    H = [V(T0 + 16 + k) for k in range(8)]                  # v56..v63: packed hi halfs of the 16 columns
    p2 = [I_s_cmp("s_bitcmp1_b32", S(S_CUR + D_FLAGS), F_REBUILD_T), I_branch("s_cbranch_scc0", f"L_{name}_nr2"),
          ("NEED_VM", "tload")]
    # lo halfs are packed back into the float4 registers they came from: columns 0..7 -> F[0], columns 8..15 -> F[2]
    for j in range(4):
        dst = F[j & 2]
        for pr in range(2):
            a_, b_ = F[j].sub(2 * pr), F[j].sub(2 * pr + 1)
            h = H[2 * j + pr]
            p2 += [I_v_cvt_pkrtz(h, a_, b_), I_v_sub_lo_half(a_, h, a_), I_v_sub_hi_half(b_, h, b_)]
        # (both pairs converted before either lo pack: the packs of float4 j = 1, 3 land in the upper half of F[j - 1])
        p2 += [I_v_cvt_pkrtz(dst.sub(2 * (j & 1)), F[j].sub(0), F[j].sub(1)),
               I_v_cvt_pkrtz(dst.sub(2 * (j & 1) + 1), F[j].sub(2), F[j].sub(3))]
    p2 += [I_ds_write_b128(V_ST_H, V(H[0].i, 4), ho + 128), I_ds_write_b128(V_ST_H, V(H[4].i, 4), ho + 144),
           I_ds_write_b128(V_ST_L, F[0], ho + 128), I_ds_write_b128(V_ST_L, F[2], ho + 144)]
    p2.append(I_label(f"L_{name}_nr2"))
    return p1, p2


def spread(n_items, n_gaps):
    """items per gap, as even as possible"""
    return [(g + 1) * n_items // n_gaps - g * n_items // n_gaps for g in range(n_gaps)]


TIMING = False              # --timing: every dispatcher visit stores stamps of the phase before it (debug builds only)
RIDE_CAP = int(os.environ.get("H3A_RIDE_CAP", "7"))    # riding instructions per MFMA gap where the ride does not fit the gaps (short B phases)


def guarded_init(s, name, half):
    """acc_<half> := bias table row of the descriptor (flag F_INIT)"""
    skip = f"L_{name}_noinit"
    s.emit(I_s_cmp("s_bitcmp1_b32", S(S_CUR + D_FLAGS), F_INIT))
    s.emit(I_branch("s_cbranch_scc0", skip))
    s.emit(I_valu("v_add_u32", V_BADDR, S(S_CUR + D_BIAS), V_BIAS, text=f"v_add_u32_e32 {V_BADDR}, {S(S_CUR + D_BIAS)}, {V_BIAS}"))
    for r in init_reads(half):
        s.emit(r, "init", group="init")
    s.emit(I_label(skip))


def emit_rebuild(s, part, first):
    for r in part:
        if isinstance(r, tuple) and r[0] == "NEED_VM":
            s.need_vm(r[1])
        elif isinstance(r, tuple):
            s.emit(r[1], "tload", group="rebuild_t")
        else:
            # the stash stores are issued under F_REBUILD, everything of the time-code part under F_REBUILD_T
            s.emit(r, "rebuild", group="rebuild_x" if (first and r.kind == "lds_w") else "rebuild_t")


def emit_ride(s, item):
    """a riding item: an instruction, (instruction, tag) or the marker ('NEED_LDS', tag)"""
    if isinstance(item, tuple) and item[0] == "NEED_LDS":
        s.need_lds(item[1])
    elif isinstance(item, tuple):
        s.emit(item[0], item[1])
    else:
        s.emit(item, "ride")


def phase_body(name, half, nks, ride=None, refills=False, tail_init=False, rebuild=None, vm_mode=None, stream_slots=None,
               one_stream=False, copy=False, refill_slots=None):
    """One phase: `nks` k-steps of MFMAs on acc_<half> from X_<half>.  The phase's barrier sits in front of MFMA 8 of the
    LAST BUT ONE k-step: by then every fragment of this half has been read (the last k-step's lo fragments go to the second
    XL buffer) and the ride's stores are done, and 16 MFMAs remain to cover what follows the barrier -- the bias-table
    reads into the other half's accumulators (flag F_INIT) and the first fragments of the next phase.
    ride: None | 'epi' (epilogue of the other half in the gaps of k-steps 1 .. barrier) | 'epi_sig' (with the sigma ride) |
          'epi_sig_ws' (... whose weights this phase requests at its start)
    refills: weight slot refills behind every k-step (B phases)
    rebuild: None | half whose input tile is restored by two guarded clusters
    vm_mode: None | 'formula' (A phase behind a B phase that issued its 16 refills in slot order: vmcnt(4 (15 - ks)) in front
             of k-step ks) | 'model' (first segment: slots 0..7 were requested before the encoder ran, slots 8..15 ride here)
    stream_slots: weight slots whose loads ride in this phase, one piece per gap from the first gap on
    refill_slots: the slots a refilling phase requests (default: all sixteen).  B16LP -- the last segment's B phase of a PERSISTENT
             workgroup -- requests slots 0..7 of the trunk's FIRST segments (its descriptor carries descriptor 0's stream fields):
             the next tile of the workgroup finds them resident, as a first tile finds what the pre-issue statement requested"""
    s = Stream()
    s.emit(I_label(f"L_{name}"))
    oh = other(half)
    last = nks - 1
    bar = (last - 1, 8)
    for nt in range(2):
        s.lds_q.append((("xh", 0), None))
    for nt in range(2):
        s.lds_q.append((("xl", 0), None))
    if vm_mode == "model":
        for ks in range(8):
            for c in range(4):
                s.vm_q.append((("w", ks), None))

    def stamp(k):           # timing build: s[78 + 2 k : 79 + 2 k] = s_memtime (k = 0 entry, 1 first MFMA, 2 / 3 around the barrier, 4 end)
        if TIMING:
            s.emit(raw(f"s_memtime s[{78 + 2 * k}:{79 + 2 * k}]"))
    stamp(0)
    ride_ins = {None: [], "epi": epilogue_stream(oh), "epi_sig": epilogue_stream(oh, sig=True),
                "epi_sig_ws": epilogue_stream(oh, sig=True, wait_ws=True)}[ride]
    if ride == "epi_sig_ws":
        for r in sigma_weight_reads():
            s.emit(r, "ws")
    if copy:                # (SAVE) this half's input tile goes to HBM while the phase multiplies it
        ride_ins = merge_ride(ride_ins, copy_groups(half))
    rb_parts = rebuild_parts(rebuild, name) if rebuild is not None else None
    ride_gaps = [(ks, m) for ks in range(1, nks) for m in range(12) if (ks, m) < (bar[0], bar[1] - 1)]
    per_gap = dict(zip(ride_gaps, spread(len(ride_ins), len(ride_gaps)))) if ride_ins else {}
    pieces = []
    if stream_slots is not None:
        # (the first A phase: 128 KiB ride here, the CU's vector-memory path is the limit -- the general form of the refill, whose
        # scalar arithmetic spaces the loads, measured 0.3 k cycles faster than the one-stream form in this phase)
        s.emit(I_salu("s_add_u32", S_R1, S_R1, 4096 * stream_slots[0], scc=True))
        s.emit(I_salu("s_add_u32", S_R2, S_R2, 4096 * stream_slots[0], scc=True))
        for slot in stream_slots:
            pieces += [(slot, pc) for pc in refill(slot)]
        every = max(1, (12 * nks - 2) // len(pieces))
    if refills and one_stream:
        s.emit(refill_start(S_R1))
    ri = pi = gi = 0
    for ks in range(nks):
        for m, mf in enumerate(mfmas(half, ks, second_xl=(ks == last))):
            # ---- in front of the MFMA
            if m == 0:
                if vm_mode == "formula":
                    # (+ the stores this phase has issued so far: they are younger entries of the same in-order queue)
                    s.wait(vm=min(4 * (15 - ks) + sum(1 for i_ in s.ins if i_.op == "gstore_s"), 63))
                elif vm_mode == "model":
                    s.need_vm(("w", ks))
                s.need_lds(("xh", ks))
            if m == 8:
                s.need_lds(("xl", ks))
            if (ks, m) == bar:
                stamp(2)
                s.wait(lgkm=0)
                s.emit(I_barrier())
                stamp(3)
            if ks == 0 and m == 0:
                stamp(1)
            s.emit(mf)
            # ---- behind it
            if m == 0 and ks < last:
                for r in frag_reads_h(half, ks + 1, (ks + 1) & 1):
                    s.emit(r, ("xh", ks + 1))
                if ks + 1 == last:                      # the last k-step's lo fragments: second buffer, requested a k-step early
                    for r in frag_reads_l(half, last, second=True):
                        s.emit(r, ("xl", last))
            if m == 11 and ks + 1 < last:
                for r in frag_reads_l(half, ks + 1):
                    s.emit(r, ("xl", ks + 1))
            if rebuild is not None and m == 2 and ks in (0, max(1, (nks - 1) // 2)):
                emit_rebuild(s, rb_parts[0 if ks == 0 else 1], ks == 0)
            if refills and ks >= 1 and 3 <= m <= 7 and (refill_slots is None or ks - 1 in refill_slots):
                for r in refill(ks - 1, one_stream)[m - 3]:
                    s.emit(r, ("w", ks - 1))
            if pieces and gi % every == 0 and pi < len(pieces):
                for r in pieces[pi][1]:
                    s.emit(r, ("w", pieces[pi][0]))
                pi += 1
            gi += 1
            for _ in range(per_gap.get((ks, m), 0)):
                emit_ride(s, ride_ins[ri])
                ri += 1
            if (ks, m) == bar:
                if tail_init:
                    guarded_init(s, name, oh)
                for r in frag_reads_h(oh, 0, 0):
                    s.emit(r, ("xh'", 0))
            if ks == last and m == 0:
                for r in frag_reads_l(oh, 0):
                    s.emit(r, ("xl'", 0))
            if ks == last and m == 11 and refills and (refill_slots is None or last in refill_slots):
                for r in refill_flat(last, one_stream):
                    s.emit(r, ("w", last))
    assert ri == len(ride_ins) and pi == len(pieces)
    stamp(4)
    s.emit(I_branch("s_branch", "L_dispatch"))
    return s.ins


def short_b_body(name, nks):
    """B phase of a short segment (the first layer, a skip layer's input part): MFMAs on acc_B, the weight-slot refills, and the
    epilogue of half A -- RIDE_CAP instructions per gap from k-step 1 on, the rest behind the last MFMA.  The phase barrier
    comes last (every read of X_B and every store into X_A of this wave is done), then the bias-table reads into acc_A (flag)
    and the first fragments of the next A phase."""
    s = Stream()
    s.emit(I_label(f"L_{name}"))
    last = nks - 1
    for nt in range(2):
        s.lds_q.append((("xh", 0), None))
    for nt in range(2):
        s.lds_q.append((("xl", 0), None))

    def stamp(k):
        if TIMING:
            s.emit(raw(f"s_memtime s[{78 + 2 * k}:{79 + 2 * k}]"))
    stamp(0)
    s.emit(refill_start(S_R1))          # (a 16-k-step segment follows a short one: one refill stream)
    ride_ins = epilogue_stream("A")
    ri = 0
    for ks in range(nks):
        for m, mf in enumerate(mfmas("B", ks)):
            if m == 0:
                s.need_lds(("xh", ks))
            if m == 8:
                s.need_lds(("xl", ks))
            if ks == 0 and m == 0:
                stamp(1)
            s.emit(mf)
            if m == 0 and ks < last:
                for r in frag_reads_h("B", ks + 1, (ks + 1) & 1):
                    s.emit(r, ("xh", ks + 1))
            if m == 11 and ks < last:
                for r in frag_reads_l("B", ks + 1):
                    s.emit(r, ("xl", ks + 1))
            if ks >= 1 and 3 <= m <= 7:
                for r in refill(ks - 1, True)[m - 3]:
                    s.emit(r, ("w", ks - 1))
            if ks >= 1:
                for _ in range(RIDE_CAP if not (3 <= m <= 7) else RIDE_CAP - 1):
                    if ri < len(ride_ins):
                        emit_ride(s, ride_ins[ri])
                        ri += 1
    for r in refill_flat(last, True):
        s.emit(r, ("w", last))
    stamp(2)
    while ri < len(ride_ins):
        emit_ride(s, ride_ins[ri])
        ri += 1
    s.wait(vm=0, lgkm=0)
    s.emit(I_barrier())
    stamp(3)
    guarded_init(s, name, "A")
    for r in frag_reads_h("A", 0, 0) + frag_reads_l("A", 0):
        s.emit(r)
    stamp(4)
    s.emit(I_branch("s_branch", "L_dispatch"))
    return s.ins


HEAD_ACC = [V(128 + 16 * c, 16) for c in range(3)]     # three accumulator chains of the heads (acc_A's registers)


def head_loads():
    """The head tile [k-step][hi, lo][lane = tile row + 32 k-half][8 halfs] (32 KiB, stream r1 of the descriptor) -> a128..a255
    (weight slots 8..15: slots 0..7 may already hold the NEXT tile's first segments, B16LP), requested in front of the last
    epilogue: k-step ks' hi fragment in a[128 + 8 ks : 128 + 8 ks + 3], lo in a[128 + 8 ks + 4 : 128 + 8 ks + 7].  Only
    the lanes whose row (lane & 31) is one of the n1 head rows load (the others keep stale, finite weights: their output rows
    are never read); the last MFMAs of the trunk were issued long before the first load can return.
    Returns [setup (VCC = the loading lanes; nothing of the epilogue touches VCC), group 0, ..., group 7]: a group is four loads
    under the lane mask -- spread over the epilogue, so that the wave is not held at the issue stage for all 32 at once."""
    row16 = V(T0 + 16)
    setup = [I_valu("v_and_b32", row16, 0x1f0, V_LANE16), I_salu("s_lshl_b32", S_T0, S(S_CUR + D_N1), 4, scc=True),
             I_v_cmp_gt_u32_vcc(S_T0, row16)]
    groups = []
    for g in range(8):
        o = [I_s_and_saveexec(S_SAVE), I_valu("v_add_u32", V_OFF, S_R1, V_LANE16, text=f"v_add_u32_e32 {V_OFF}, {S_R1}, {V_LANE16}")]
        o += [I_gload_x4_s(A(128 + 16 * g + 4 * c, 4), V_OFF, S_PK, 1024 * c) for c in range(4)]
        o += [I_salu("s_add_u32", S_R1, S_R1, 4096, scc=True), I_s_mov_exec(S_SAVE)]
        groups.append(o)
    return [setup] + groups


def head_body():
    """HEAD: the narrow output layers on the trunk's last activation (both planes complete: EPI_B ended with a barrier).  Wave w
    multiplies the 32-row head tile (a128..a255, see head_loads) with the fragments of points 32 w .. 32 w + 31 -- three
    accumulator chains Wl.xh, Wh.xl, Wh.xh of sixteen MFMAs -- and leaves  (chain0 + chain1) + chain2  of head row
    r < n1 at float D_BIAS / 4 + r of the point's raw record in LDS; bias and activation are applied where the records are stored."""
    s = Stream()
    e = s.emit
    e(I_label("L_HEAD"))
    vbh, vbl, vw, v4h, t0 = V(T0), V(T0 + 1), V(T0 + 2), V(T0 + 3), V(T0 + 4)
    e(I_salu("s_mul_i32", S_T0, S_WAVE, NT_B))
    e(I_valu("v_add_u32", vbh, S_T0, V_RD_H, text=f"v_add_u32_e32 {vbh}, {S_T0}, {V_RD_H}"))
    e(I_valu("v_add_u32", vbl, S_T0, V_RD_L, text=f"v_add_u32_e32 {vbl}, {S_T0}, {V_RD_L}"))

    # fragments of four k-steps in registers (three in flight ahead of the MFMAs: 96 matrix-pipe cycles per k-step do not cover
    # an LDS round trip): hi in XH[b][nt], lo in XL / XL2
    fh = lambda ks: xh((ks >> 1) & 1, ks & 1)
    fl = lambda ks: (xl, xl2)[(ks >> 1) & 1](ks & 1)

    def reads(ks):
        return [I_ds_read_b128(fh(ks), vbh, 32 * ks), I_ds_read_b128(fl(ks), vbl, 32 * ks)]
    for k0 in range(3):
        for r in reads(k0):
            e(r, ("x", k0))
    s.wait(vm=0)
    a0, a1, a2 = HEAD_ACC
    for ks in range(16):
        s.need_lds(("x", ks))
        e(I_mfma(a0, A(128 + 8 * ks + 4, 4), fh(ks), a0 if ks else 0))
        e(I_mfma(a1, A(128 + 8 * ks, 4), fl(ks), a1 if ks else 0))
        e(I_mfma(a2, A(128 + 8 * ks, 4), fh(ks), a2 if ks else 0))
        if ks + 3 < 16:                 # (into the registers of k-step ks - 1: its MFMAs were issued 96 cycles ago)
            for r in reads(ks + 3):
                e(r, ("x", ks + 3))
    # record address of this lane's point: rawlds + (32 wave + (lane & 31)) * 64 + 16 (lane >> 5) + 4 slot0
    e(I_valu("v_and_b32", t0, 0x1f0, V_LANE16)); e(I_valu("v_lshlrev_b32", t0, 2, t0))
    e(I_valu("v_lshrrev_b32", v4h, 5, V_LANE16)); e(I_valu("v_and_b32", v4h, 0x10, v4h))
    e(I_valu("v_add_u32", t0, t0, v4h))
    e(I_salu("s_lshl_b32", S_T0, S_WAVE, 11, scc=True)); e(I_salu("s_add_u32", S_T0, S_T0, S_RAWLDS, scc=True))
    e(I_salu("s_add_u32", S_T0, S_T0, S(S_CUR + D_BIAS), scc=True))
    e(I_valu("v_add_u32", vw, S_T0, t0, text=f"v_add_u32_e32 {vw}, {S_T0}, {t0}"))
    e(I_valu("v_lshrrev_b32", v4h, 2, v4h))                                     # 4 (lane >> 5)
    e(I_nop(7)); e(I_nop(3))                                                    # (the last MFMAs' results)
    for r in range(8):
        e(I_valu("v_add_f32", a0.sub(r), a0.sub(r), a1.sub(r)))
    for r in range(8):
        e(I_valu("v_add_f32", a0.sub(r), a0.sub(r), a2.sub(r)))
    for r in range(8):              # accumulator register r of a lane = head row (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
        c_r = (r & 3) + 8 * (r >> 2)
        e(I_salu("s_sub_u32", S_T1, S(S_CUR + D_N1), c_r, scc=True))
        e(I_salu("s_cselect_b32", S_T1, 0, S_T1))                               # (n1 <= c_r: no lane)
        e(I_v_cmp_gt_u32_vcc(S_T1, v4h))
        e(I_s_and_saveexec(S_SAVE))
        e(I_ds_write_b32(vw, a0.sub(r), 4 * c_r), "hw")
        e(I_s_mov_exec(S_SAVE))
    s.wait(lgkm=0)
    e(I_branch("s_branch", "L_dispatch"))
    return s.ins


def bare_epilogue(name, half):
    """Epilogue of `half` with no MFMAs beside it (end of the trunk), behind the requests of the head tile; ends with the
    workgroup barrier that publishes the last activation (everything this wave started has landed)."""
    s = Stream()
    s.emit(I_label(f"L_{name}"))
    hl = head_loads()
    for i in hl[0] + hl[1]:
        s.emit(i, "head")
    s.emit(I_nop(7)); s.emit(I_nop(7))          # the last MFMAs on these accumulators were issued a few states ago
    # two units at a time, instruction by instruction: independent chains for the in-order VALU; a group of head-tile loads
    # in front of and in the middle of every pair
    units = [epilogue_unit(half, u, u & 1) for u in range(8)]
    g = 2
    for u in range(0, 8, 2):
        n = len(units[u])
        # (the SAVE build's units share the sign-word registers: one after the other there)
        pairs = list(zip(units[u], units[u + 1])) if not SAVE else [(x, None) for x in units[u]] + [(y, None) for y in units[u + 1]]
        for k, (x, y) in enumerate(pairs):
            if k == n // 2 or (k == 0 and u > 0):
                if g < len(hl):
                    for i in hl[g]:
                        s.emit(i, "head")
                    g += 1
            s.emit(x, "ride")
            if y is not None:
                s.emit(y, "ride")
    if SAVE:                # (the sign words of half B of the last layer were the slot's last: epilogue_stream's pointer step is not needed)
        pass
    while g < len(hl):
        for i in hl[g]:
            s.emit(i, "head")
        g += 1
    s.wait(vm=0, lgkm=0)
    s.emit(I_barrier())
    s.emit(I_branch("s_branch", "L_dispatch"))
    return s.ins


def save_last_body():
    """SAVE build: the trunk's last activation (both halves, complete behind EPI_B's barrier, read next by the HEAD phase) goes to
    its slot -- a phase of its own, nothing to ride on."""
    s = Stream()
    s.emit(I_label("L_SAVE_LAST"))
    for half in ("A", "B"):
        for grp in copy_groups(half):
            for item in grp:
                emit_ride(s, item)
    s.wait(lgkm=0)
    s.emit(I_branch("s_branch", "L_dispatch"))
    return s.ins


def raw(text, wr=(), rd=()):
    return Inst("raw", text, rd, wr, "other")


# Inline-asm operands and the registers the simulator's harness presets in their place
IN_S = dict(pk=S(0, 2), phases=S(2, 2), lds=S(4), biaslds=S(5), wave=S(6), in_t=S(7), r1=S(8), r1w=S(9), r2=S(10), r2w=S(11), n1=S(12),
            rawlds=S(13), act=S(14, 2), mask=S(16, 2), astride=S(18), mstride=S(19))
IN_V = dict(tid=V(0), tpa0=V(1), tpa1=V(2), tpb0=V(3), tpb1=V(4))


def in_s(dst, name):
    src = IN_S[name]
    op = "s_mov_b64" if dst.n == 2 else "s_mov_b32"
    return Inst(op, f"{op} {dst}, %[{name}]", [src], [dst], "salu", dict(d=dst, s=[src]))


def in_v(dst, name):
    return Inst("v_mov_b32", f"v_mov_b32 {dst}, %[{name}]", [IN_V[name]], [dst], "valu", dict(d=dst, s=[IN_V[name]]))


def pre_issue():
    """The FIRST asm statement of the kernel, in front of the input encoder: request weight slots 0..7 (the first segment, and
    the start of the second where the first is four k-steps long) so that the 128 KiB cross the CU's vector-memory path while
    the encoder computes.  Nothing but the loads in flight survives this statement; the compiler-generated code between the
    two statements does not touch accumulation registers (tools/h3asm/audit.py checks that).
    Operands: %[pk] s64, %[wave] s32, %[tid] v32, %[r1] / %[r1w] / %[r2] / %[r2w] / %[n1] s32 (fields of phase descriptor 0)."""
    o = [in_s(S_PK, "pk"), in_s(S_WAVE, "wave"), in_v(V_TMP, "tid"), in_s(S(S_CUR + D_N1), "n1"),
         in_s(S_R1, "r1"), in_s(S_R2, "r2"), in_s(S_T0, "r1w"), in_s(S_T1, "r2w")]
    o += [I_salu("s_mul_i32", S_T0, S_WAVE, S_T0), I_salu("s_add_u32", S_R1, S_R1, S_T0, scc=True),
          I_salu("s_mul_i32", S_T1, S_WAVE, S_T1), I_salu("s_add_u32", S_R2, S_R2, S_T1, scc=True),
          I_valu("v_and_b32", V_LANE16, 63, V_TMP), I_valu("v_lshlrev_b32", V_LANE16, 4, V_LANE16)]
    for ks in range(8):
        o += refill_flat(ks)
    return o


def prologue():
    """The second asm statement (behind the encoder and a workgroup barrier).  Operands: %[pk] s64, %[phases] s64 (phase
    descriptors), %[lds] s32 (byte address of the activation tile), %[biaslds] s32, %[wave] s32, %[in_t] s32, %[tid] v32,
    %[tpa0/1] %[tpb0/1] v32 (time-code row pointers)."""
    o = []
    e = o.append
    if "align64" in EXP:
        e(raw(".p2align 6"))
    if "shift4" in EXP:
        e(raw(".p2align 6")); e(raw("s_nop 0"))
    if TIMING:
        e(raw("s_mov_b64 s[76:77], %[dbg]")); e(raw("s_memtime s[74:75]"))
    e(in_s(S_PK, "pk")); e(in_s(S_PH, "phases")); e(in_s(S_LDS, "lds")); e(in_s(S_BIASLDS, "biaslds"))
    e(in_s(S_WAVE, "wave")); e(in_s(S_INT, "in_t")); e(in_s(S_RAWLDS, "rawlds")); e(in_v(V_TMP, "tid"))
    e(in_v(V(34), "tpa0")); e(in_v(V(35), "tpa1")); e(in_v(V(36), "tpb0")); e(in_v(V(37), "tpb1"))
    if not SAVE:
        # the NEXT tile's point of this thread (a persistent workgroup's tile loop, field_h3.hip): %[nxa] v64 its address, %[nx0..2]
        # three of the compiler's registers (outputs of the statement, valid behind L_end's vmcnt(0)).  Not in the wait model:
        # operations the model does not know make a counted wait stricter, never weaker (the queue returns in order).
        for j in range(3):
            e(raw(f"global_load_dword %[nx{j}], %[nxa], off" + (f" offset:{4 * j}" if j else "")))
    # descriptor 0 -> cur (bias row of segment 0), descriptor 1 -> nxt (fetched now, valid after the lgkmcnt(0) below)
    e(I_s_load(S(S_CUR, 8), S_PH, 0))
    e(I_s_load(S(S_NXT, 8), S_PH, 32))
    e(I_salu("s_add_u32", S(46), S(46), 64, scc=True)); e(I_salu("s_addc_u32", S(47), S(47), 0, scc=True))
    # lane = tid & 63; l31 = lane & 31; h = lane >> 5
    lane, l31, h = V(T0), V(T0 + 1), V(T0 + 2)
    e(I_valu("v_and_b32", lane, 63, V_TMP)); e(I_valu("v_and_b32", l31, 31, V_TMP)); e(I_valu("v_lshrrev_b32", h, 5, lane))
    e(I_valu("v_lshlrev_b32", V_LANE16, 4, lane))
    # rd_h = lds + l31 * 528 + 16 h ; rd_l = rd_h + PLANE
    e(I_valu("v_mul_u32_u24", V_RD_H, LDH_B, l31))
    e(I_valu("v_lshlrev_b32", V(T0 + 3), 4, h)); e(I_valu("v_add_u32", V_RD_H, V_RD_H, V(T0 + 3)))
    e(I_valu("v_add_u32", V_RD_H, S_LDS, V_RD_H, text=f"v_add_u32_e32 {V_RD_H}, {S_LDS}, {V_RD_H}"))
    e(I_valu("v_add_u32", V_RD_L, PLANE_B, V_RD_H, text=f"v_add_u32_e32 {V_RD_L}, {PLANE_B}, {V_RD_H}"))
    # wr_h = lds + l31 * 528 + 128 wave + 32 h ; wr_l = wr_h + PLANE
    e(I_valu("v_mul_u32_u24", V_WR_H, LDH_B, l31))
    e(I_valu("v_lshlrev_b32", V(T0 + 3), 3 if NOSWAP_STORES else 5, h)); e(I_valu("v_add_u32", V_WR_H, V_WR_H, V(T0 + 3)))
    e(I_salu("s_lshl_b32", S_T0, S_WAVE, 7, scc=True))
    e(I_salu("s_add_u32", S_T0, S_T0, S_LDS, scc=True))
    e(I_valu("v_add_u32", V_WR_H, S_T0, V_WR_H, text=f"v_add_u32_e32 {V_WR_H}, {S_T0}, {V_WR_H}"))
    e(I_valu("v_add_u32", V_WR_L, PLANE_B, V_WR_H, text=f"v_add_u32_e32 {V_WR_L}, {PLANE_B}, {V_WR_H}"))
    # bias = biaslds + 256 wave + 16 h
    e(I_salu("s_lshl_b32", S_T0, S_WAVE, 8, scc=True)); e(I_salu("s_add_u32", S_T0, S_T0, S_BIASLDS, scc=True))
    e(I_valu("v_lshlrev_b32", V_BIAS, 4, h))
    e(I_valu("v_add_u32", V_BIAS, S_T0, V_BIAS, text=f"v_add_u32_e32 {V_BIAS}, {S_T0}, {V_BIAS}"))
    # stash address: row = tid >> 2 (0..63), quarter = tid & 3:  st_h = lds + row * 528 + 32 q ; st_l = st_h + PLANE
    e(I_valu("v_lshrrev_b32", V(T0 + 3), 2, V_TMP)); e(I_valu("v_and_b32", V(T0 + 4), 3, V_TMP))
    e(I_valu("v_mul_u32_u24", V_ST_H, LDH_B, V(T0 + 3)))
    e(I_valu("v_lshlrev_b32", V(T0 + 5), 5, V(T0 + 4))); e(I_valu("v_add_u32", V_ST_H, V_ST_H, V(T0 + 5)))
    e(I_valu("v_add_u32", V_ST_H, S_LDS, V_ST_H, text=f"v_add_u32_e32 {V_ST_H}, {S_LDS}, {V_ST_H}"))
    e(I_valu("v_add_u32", V_ST_L, PLANE_B, V_ST_H, text=f"v_add_u32_e32 {V_ST_L}, {PLANE_B}, {V_ST_H}"))
    # n4 = float4s of time code this thread restores: clamp((in_t - 16 q) / 4, 0, 4)
    e(I_valu("v_lshlrev_b32", V(T0 + 5), 4, V(T0 + 4)))                       # 16 q
    e(I_valu("v_mov_b32", V_N4, S_INT, text=f"v_mov_b32_e32 {V_N4}, {S_INT}"))
    e(I_valu("v_min_u32", V(T0 + 5), V(T0 + 5), V_N4))                        # min(16 q, in_t)
    e(I_valu("v_sub_u32", V_N4, V_N4, V(T0 + 5)))                             # in_t - min(16 q, in_t) >= 0
    e(I_valu("v_lshrrev_b32", V_N4, 2, V_N4)); e(I_valu("v_min_u32", V_N4, 4, V_N4))
    if SAVE:
        # %[act] / %[mask] s64: this tile's (half A's) first activation slot / sign words of THIS WAVE (+ 512 wave);
        # %[astride] / %[mstride] s32: bytes per slot.  Copy addresses: lane i of a 16-lane group g reads 4 neurons of point
        # 8 (g >> 1) + (i >> 2): cp_h = lds + that row * 528 + 32 (g & 1) + 8 (i & 3) + 64 wave
        e(in_s(S_ACT, "act")); e(in_s(S_MASK, "mask")); e(in_s(S_ASTRIDE, "astride")); e(in_s(S_MSTRIDE, "mstride"))
        a_, b_ = V(T0 + 3), V(T0 + 4)
        e(I_valu("v_and_b32", a_, 15, lane)); e(I_valu("v_lshrrev_b32", a_, 2, a_))             # i >> 2
        e(I_valu("v_lshrrev_b32", b_, 5, lane)); e(I_valu("v_lshlrev_b32", b_, 3, b_))          # 8 (g >> 1)
        e(I_valu("v_add_u32", a_, a_, b_)); e(I_valu("v_mul_u32_u24", V_CP_H, LDH_B, a_))
        e(I_valu("v_lshrrev_b32", a_, 4, lane)); e(I_valu("v_and_b32", a_, 1, a_)); e(I_valu("v_lshlrev_b32", a_, 5, a_))   # 32 (g & 1)
        e(I_valu("v_and_b32", b_, 3, lane)); e(I_valu("v_lshlrev_b32", b_, 3, b_))              # 8 (i & 3)
        e(I_valu("v_add_u32", a_, a_, b_)); e(I_valu("v_add_u32", V_CP_H, V_CP_H, a_))
        e(I_salu("s_lshl_b32", S_T0, S_WAVE, 6, scc=True)); e(I_salu("s_add_u32", S_T0, S_T0, S_LDS, scc=True))
        e(I_valu("v_add_u32", V_CP_H, S_T0, V_CP_H, text=f"v_add_u32_e32 {V_CP_H}, {S_T0}, {V_CP_H}"))
        e(I_valu("v_add_u32", V_CP_L, PLANE_B, V_CP_H, text=f"v_add_u32_e32 {V_CP_L}, {PLANE_B}, {V_CP_H}"))
    # stash of the input tile (the C++ encoder built it and synchronised the workgroup before this statement)
    for hb in range(2):
        st, ho = STASH + 16 * hb, hb * HALF_B
        e(I_ds_read_b128(V(st, 4), V_ST_H, ho)); e(I_ds_read_b128(V(st + 4, 4), V_ST_H, ho + 16))
        e(I_ds_read_b128(V(st + 8, 4), V_ST_L, ho)); e(I_ds_read_b128(V(st + 12, 4), V_ST_L, ho + 16))
    e(I_wait(lgkm=0))                                                         # descriptors + stash
    # acc_A, acc_B := bias of segment 0 (flag), first fragments of half A
    e(I_s_cmp("s_bitcmp1_b32", S(S_CUR + D_FLAGS), F_INIT)); e(I_branch("s_cbranch_scc0", "L_pro_noinit"))
    e(I_valu("v_add_u32", V_BADDR, S(S_CUR + D_BIAS), V_BIAS, text=f"v_add_u32_e32 {V_BADDR}, {S(S_CUR + D_BIAS)}, {V_BIAS}"))
    o.extend(init_reads("A"))
    # (half B's row of segment 0 sits (flags >> 16) bytes behind half A's: the same row, or the next ray's where the time code is
    # folded into per-ray rows)
    e(I_salu("s_lshr_b32", S_T0, S(S_CUR + D_FLAGS), 16, scc=True))
    e(I_valu("v_add_u32", V_BADDR, S_T0, V_BADDR, text=f"v_add_u32_e32 {V_BADDR}, {S_T0}, {V_BADDR}"))
    o.extend(init_reads("B"))
    e(I_label("L_pro_noinit"))
    o.extend(frag_reads_h("A", 0, 0)); o.extend(frag_reads_l("A", 0))
    return o


def timing_store():
    """lane 0 stores six dwords at *s[76:77] (+= 24): the stamp of the previous dispatcher visit (s74, issued one phase ago) and
    the five stamps the phase in between left in s78..s87; then a new visit stamp is requested.  The stores are entries of the
    in-order VMEM queue: counted waits for older loads only get stricter."""
    o = [raw("v_mov_b32 v32, 0"), raw("s_mov_b64 exec, 1")]
    for k, sr in enumerate((74, 78, 80, 82, 84, 86)):
        o += [raw(f"v_mov_b32 v39, s{sr}"), raw(f"global_store_dword v32, v39, s[76:77] offset:{4 * k}")]
    o += [raw("s_mov_b64 exec, -1"), raw("s_add_u32 s76, s76, 24"), raw("s_addc_u32 s77, s77, 0"), raw("s_memtime s[74:75]")]
    return o


DISPATCH_ORDER = ("A16R", "B16R", "B16X", "A4", "B4", "A8", "B8", "A4F", "A8F", "B16L", "B16LP", "EPI_B", "HEAD", "B16RS", "A16RS", "SAVE_LAST")


def dispatcher():
    o = [I_label("L_dispatch")]
    e = o.append
    if TIMING:
        o.extend(timing_store())
    # cur <- nxt (valid: every body ends its LDS / SMEM traffic with lgkmcnt(0) at its barrier)
    for k in range(8):
        e(I_salu("s_mov_b32", S(S_CUR + k), S(S_NXT + k)))
    e(I_s_load(S(S_NXT, 8), S_PH, 0))
    e(I_salu("s_add_u32", S(46), S(46), 32, scc=True)); e(I_salu("s_addc_u32", S(47), S(47), 0, scc=True))
    e(I_salu("s_mul_i32", S_T0, S_WAVE, S(S_CUR + D_R1W))); e(I_salu("s_add_u32", S_R1, S(S_CUR + D_R1), S_T0, scc=True))
    e(I_salu("s_mul_i32", S_T0, S_WAVE, S(S_CUR + D_R2W))); e(I_salu("s_add_u32", S_R2, S(S_CUR + D_R2), S_T0, scc=True))
    for name in DISPATCH_ORDER:
        if name not in dispatcher.bodies:
            continue
        e(I_s_cmp("s_cmp_eq_u32", S(S_CUR + D_BODY), BODY[name]))
        e(I_branch("s_cbranch_scc1", f"L_{name}"))
    e(I_branch("s_branch", "L_end"))
    return o


def build(save=False):
    """-> (pre-issue statement, main statement, bodies); save: the training-forward body (see SAVE)"""
    global SAVE
    SAVE = bool(save)
    try:
        return _build()
    finally:
        SAVE = False


def _build():
    cp = SAVE
    bodies = {
        "A16R": phase_body("A16R", "A", 16, ride="epi", tail_init=True, vm_mode="formula", copy=cp),
        "B16R": phase_body("B16R", "B", 16, ride="epi", refills=True, tail_init=True, one_stream=True, copy=cp),
        "B16L": phase_body("B16L", "B", 16, ride="epi", copy=cp),        # the trunk's last segment: nothing left to request
        "B16X": phase_body("B16X", "B", 16, refills=True, rebuild="A", copy=cp),
        "A4": phase_body("A4", "A", 4, rebuild="B", vm_mode="formula"),
        "A8": phase_body("A8", "A", 8, rebuild="B", vm_mode="formula"),
        "A4F": phase_body("A4F", "A", 4, vm_mode="model", stream_slots=list(range(8, 16))),
        "A8F": phase_body("A8F", "A", 8, vm_mode="model", stream_slots=list(range(8, 16))),
        "B4": short_b_body("B4", 4),
        "B8": short_b_body("B8", 8),
        "EPI_B": bare_epilogue("EPI_B", "B"),
        "HEAD": head_body(),
    }
    # the sigma ride of a view-direction static trunk: the last trunk layer's B phase and the A phase behind it (round 6: in the SAVE
    # build too -- the reference's documented training configuration, README.md:226-233, is a view-direction model)
    bodies["B16RS"] = phase_body("B16RS", "B", 16, ride="epi_sig_ws", refills=True, tail_init=True, one_stream=True, copy=cp)
    bodies["A16RS"] = phase_body("A16RS", "A", 16, ride="epi_sig", tail_init=True, vm_mode="formula", copy=cp)
    if SAVE:
        bodies["SAVE_LAST"] = save_last_body()
    else:
        # the last segment's B phase of a persistent workgroup: the next tile's weight slots 0..7 behind its k-steps 1..8
        bodies["B16LP"] = phase_body("B16LP", "B", 16, ride="epi", refills=True, refill_slots=range(8))
    dispatcher.bodies = set(bodies)
    prog = prologue()
    prog.append(I_branch("s_branch", "L_dispatch"))
    for name in bodies:
        prog += bodies[name]
    prog += dispatcher()
    prog.append(I_label("L_end"))
    prog.append(I_wait(vm=0, lgkm=0))
    if TIMING:
        prog += timing_store() + [I_wait(vm=0, lgkm=0)]
    return pre_issue(), prog, bodies


def render(prog):
    lines = []
    for ins in prog:
        t = ins.text
        if ins.kind == "label":
            t = t.replace("L_", "L_h3a_%=_")
        elif ins.kind == "branch":
            t = t.replace(" L_", " L_h3a_%=_")
        lines.append('    "' + t + '\\n\\t"')
    return "\n".join(lines) + "\n"


def lint(bodies, prog):
    errs = []
    for name, ins in bodies.items():
        errs += lint_straight(ins, name)
        # ... and with every guarded cluster skipped
        kept, skip_to = [], None
        for i in ins:
            if skip_to is not None:
                if i.kind == "label" and i.args["name"] == skip_to:
                    skip_to = None
                continue
            if i.kind == "branch" and i.op == "s_cbranch_scc0":
                skip_to = i.args["target"]
                continue
            kept.append(i)
        errs += lint_straight(kept, name + "(guards skipped)")
    return errs


def main():
    global TIMING
    TIMING = "--timing" in sys.argv
    pre, prog, bodies = build()
    timing, TIMING = TIMING, False               # (the stamps are the inference body's: `make timing` measures that one)
    _, prog_save, bodies_save = build(save=True)
    TIMING = timing
    errs = lint(bodies, prog) + lint(bodies_save, prog_save)
    for e_ in errs[:40]:
        print("LINT:", e_)
    if errs:
        sys.exit(f"{len(errs)} hazard(s)")
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "nsff_pl_amd", "csrc",
                       "field_h3a_body_timing.inc" if TIMING else (f"field_h3a_body_{EXP}.inc" if EXP else
                       (f"field_h3a_body_cap{RIDE_CAP}.inc" if "H3A_RIDE_CAP" in os.environ else
                        ("field_h3a_body_swap.inc" if not NOSWAP_STORES else "field_h3a_body.inc"))))
    clob = ", ".join([f'"v{i}"' for i in range(24, 256)] + [f'"a{i}"' for i in range(256)] + [f'"s{i}"' for i in range(40, 100)] +
                     ['"vcc"', '"scc"', '"memory"'])
    clob_save = ", ".join([f'"v{i}"' for i in range(14, 256)] + [f'"a{i}"' for i in range(256)] + [f'"s{i}"' for i in range(40, 100)] +
                          ['"vcc"', '"scc"', '"memory"'])
    pre_wr = sorted({r for i in pre for r in i.wr if r[0] in ("v", "a") or (r[0] == "s" and r[1] < 100)})
    pre_clob = ", ".join([f'"{f}{i}"' for f, i in pre_wr] + ['"scc"', '"memory"'])
    consts = "".join(f"#define H3A_BODY_{k} {v}\n" for k, v in BODY.items())
    consts += f"#define H3A_F_INIT {1 << F_INIT}\n#define H3A_F_REBUILD_T {1 << F_REBUILD_T}\n#define H3A_F_REBUILD {1 << F_REBUILD}\n"
    macro = lambda name, insts: f"#define {name} \\\n" + render(insts).replace("\n", " \\\n").rstrip(" \\\n") + "\n"
    # the same 32 loads as eight statements, one weight slot each, for the kernel to spread over its encoder: statement k takes
    # %[pk] s64, %[off] s32 = (k < n1 ? r1 + wave r1w : r2 + wave r2w) + 4096 k, %[lane16] v32 = 16 (tid & 63)
    slots = ""
    for k in range(8):
        body = f'    "v_add_u32 v32, %[off], %[lane16]\\n\\t" \\\n'
        body += " \\\n".join(f'    "global_load_dwordx4 a[{16 * k + 4 * c}:{16 * k + 4 * c + 3}], v32, %[pk]' + (f" offset:{1024 * c}" if c else "") + '\\n\\t"'
                               for c in range(4))
        slots += f"#define H3A_PRE_SLOT{k} \\\n{body}\n"
        slots += f"#define H3A_PRE_SLOT{k}_CLOBBERS " + ", ".join(['"v32"'] + [f'"a{16 * k + j}"' for j in range(16)] + ['"memory"']) + "\n"
    text = ("// GENERATED by tools/h3asm/gen.py -- do not edit.  The hand-scheduled trunk body of nsff_field_kernel_h3a:\n"
            "// H3A_PRE (weight slots 0..7 requested in front of the encoder; H3A_PRE_SLOT0..7: the same loads one slot per statement)\n"
            "// and H3A_BODY (the trunk; registers v24..v255, a0..a255, s40..s99 are its own while it runs); H3A_BODY_SAVE: the training\n"
            "// forward's body (activation copies + ReLU sign words riding in the phases; v14..v23 are its own too).\n" + consts +
            "#define H3A_PRE_CLOBBERS " + pre_clob + "\n" + macro("H3A_PRE", pre) + slots +
            "#define H3A_CLOBBERS " + clob + "\n" + macro("H3A_BODY", prog) +
            "#define H3A_SAVE_CLOBBERS " + clob_save + "\n" + macro("H3A_BODY_SAVE", prog_save))
    with open(out, "w") as f:
        f.write(text)
    n_m = sum(1 for i in prog if i.kind == "mfma")
    n_all = sum(1 for i in prog if i.kind not in ("label", "other"))
    print(f"wrote {os.path.normpath(out)}: {n_all} instructions, {n_m} MFMAs")
    for name, ins in bodies.items():
        sv = bodies_save.get(name)
        print(f"  {name:9s} {sum(1 for i in ins if i.kind not in ('label', 'other')):5d} instructions, "
              f"{sum(1 for i in ins if i.kind == 'mfma'):4d} MFMAs" +
              (f"   (save build: {sum(1 for i in sv if i.kind not in ('label', 'other'))})" if sv is not None else ""))


if __name__ == "__main__":
    main()
