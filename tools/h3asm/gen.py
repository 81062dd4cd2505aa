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
SPARE = 120
ACC = {"A": 128, "B": 192}

# ---- SGPRs (asm-owned: s40..s99)
S_PK = S(40, 2)
S_LDS, S_BIASLDS, S_WAVE, S_INT = S(42), S(43), S(44), S(45)
S_PH = S(46, 2)             # pointer to the NEXT phase descriptor to fetch
S_SAVE = S(48, 2)
S_R1, S_R2, S_SEL = S(50), S(51), S(52)
S_T0, S_T1 = S(53), S(54)
S_CUR = 56                  # s56..s63 current descriptor
S_NXT = 64                  # s64..s71 next descriptor (being fetched)
D_BODY, D_FLAGS, D_BIAS, D_N1, D_R1, D_R1W, D_R2, D_R2W = range(8)
# flag bits: tail initialises the other accumulator from the bias table; the rebuild has a time-code part; the phase restores
# an input tile at all (A4 / A8 also run the first layer, where the tile is still the encoder's)
F_INIT, F_REBUILD_T, F_REBUILD = 0, 1, 2

BODY = dict(END=0, A16R=1, B16R=2, B16X=3, A4=4, A8=5, B4=6, B8=7, EPI_A=8, EPI_B=9)


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
def epilogue_unit(half, u, tset):
    """ReLU -> hi / lo split -> lanes i, i+32 trade halves -> two 16-byte LDS stores of (tile u>>1, quad pair p = u&1) of
    `half`'s accumulators.  In place on the accumulator registers; 8 temporaries from set `tset`."""
    t, p = u >> 1, u & 1
    mt, nt = t >> 1, t & 1
    a = acc(half, mt, nt)
    x = [a.sub(4 * p + e) for e in range(4)] + [a.sub(4 * p + 8 + e) for e in range(4)]
    H = [V(T0 + 8 * tset + k) for k in range(4)]
    L = [V(T0 + 8 * tset + 4 + k) for k in range(4)]
    out = [I_v_max0(r, r) for r in x]
    out += [I_v_cvt_pkrtz(H[k], x[2 * k], x[2 * k + 1]) for k in range(4)]
    for k in range(4):
        out += [I_v_sub_lo_half(x[2 * k], H[k], x[2 * k]), I_v_sub_hi_half(x[2 * k + 1], H[k], x[2 * k + 1])]
    out += [I_v_cvt_pkrtz(L[k], x[2 * k], x[2 * k + 1]) for k in range(4)]
    out += [I_v_permlane32_swap(H[0], H[2]), I_v_permlane32_swap(H[1], H[3]),
            I_v_permlane32_swap(L[0], L[2]), I_v_permlane32_swap(L[1], L[3])]
    off = half_off(half) + NT_B * nt + 64 * mt + 16 * p
    out += [I_ds_write_b128(V_WR_H, V(H[0].i, 4), off), I_ds_write_b128(V_WR_L, V(L[0].i, 4), off)]
    return out


def epilogue_stream(half):
    out = []
    for u in range(8):
        out += epilogue_unit(half, u, u & 1)
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


def frag_reads_l(half, ks):
    return [I_ds_read_b128(xl(nt), V_RD_L, half_off(half) + NT_B * nt + 32 * ks) for nt in range(2)]


def mfmas(half, ks):
    b = ks & 1
    out = []
    for part, xsel in ((1, "h"), (0, "h"), (0, "l")):      # Wl.xh, Wh.xh, Wh.xl: the lo fragments are needed last
        for mt in range(2):
            for nt in range(2):
                d = acc(half, mt, nt)
                out.append(I_mfma(d, wslot(ks, mt, part), xh(b, nt) if xsel == "h" else xl(nt), d))
    return out


def refill(ks):
    """slot ks <- slice ks of the next segment that uses it: r1 for ks < n1 (the next segment), r2 beyond (the one after).
    Returns the pieces [address arithmetic, load, load, load, load] (one piece per MFMA gap)."""
    head = [I_s_cmp("s_cmp_gt_u32", S(S_CUR + D_N1), ks),
            I_salu("s_cselect_b32", S_SEL, S_R1, S_R2),
            I_valu("v_add_u32", V_OFF, S_SEL, V_LANE16, text=f"v_add_u32_e32 {V_OFF}, {S_SEL}, {V_LANE16}"),
            I_salu("s_add_u32", S_R1, S_R1, 4096, scc=True), I_salu("s_add_u32", S_R2, S_R2, 4096, scc=True)]
    return [head] + [[I_gload_x4_s(A(16 * ks + 4 * c, 4), V_OFF, S_PK, 1024 * c)] for c in range(4)]


def refill_flat(ks):
    return [i for piece in refill(ks) for i in piece]


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
    Lr = [V(SPARE + k) for k in range(8)]                   # v120..v127: packed lo halfs
    p2 = [I_s_cmp("s_bitcmp1_b32", S(S_CUR + D_FLAGS), F_REBUILD_T), I_branch("s_cbranch_scc0", f"L_{name}_nr2"),
          ("NEED_VM", "tload")]
    for j in range(4):
        for pr in range(2):
            a_, b_ = F[j].sub(2 * pr), F[j].sub(2 * pr + 1)
            h = H[2 * j + pr]
            p2 += [I_v_cvt_pkrtz(h, a_, b_), I_v_sub_lo_half(a_, h, a_), I_v_sub_hi_half(b_, h, b_),
                   I_v_cvt_pkrtz(Lr[2 * j + pr], a_, b_)]
    p2 += [I_ds_write_b128(V_ST_H, V(H[0].i, 4), ho + 128), I_ds_write_b128(V_ST_H, V(H[4].i, 4), ho + 144),
           I_ds_write_b128(V_ST_L, V(Lr[0].i, 4), ho + 128), I_ds_write_b128(V_ST_L, V(Lr[4].i, 4), ho + 144)]
    p2.append(I_label(f"L_{name}_nr2"))
    return p1, p2


def spread(n_items, n_gaps):
    """items per gap, as even as possible"""
    return [(g + 1) * n_items // n_gaps - g * n_items // n_gaps for g in range(n_gaps)]


BARRIER_AT = 8              # the phase barrier sits in front of MFMA `BARRIER_AT` of the last k-step


def phase_body(name, half, nks, ride=None, refills=False, tail_init=False, prefetch=True, rebuild=None, vm_waits=False):
    """One phase: `nks` k-steps of MFMAs on acc_<half> from X_<half>.
    ride: None | 'epi' (epilogue of the other half rides in k-steps 1 .. nks-1)
    refills: weight slot refills behind every k-step (B phases)
    rebuild: None | half whose input tile is restored by a guarded cluster in k-step 1
    tail_init: bias-table reads into the other half's accumulators behind the barrier (guarded by F_INIT)
    prefetch: read the first fragments of the other half behind the barrier
    vm_waits: A phases -- wait for slot ks's weights in front of k-step ks (counted against the 16-slot refill order)"""
    s = Stream()
    s.emit(I_label(f"L_{name}"))
    oh = other(half)
    # fragments of k-step 0 were requested by the previous phase: model them as outstanding
    for nt in range(2):
        s.lds_q.append((("xh", 0), None))
    for nt in range(2):
        s.lds_q.append((("xl", 0), None))
    if ride == "epi":
        # (the previous tail's bias reads into the ACCUMULATORS OF THIS PHASE precede the fragment prefetch in the queue)
        pass
    ride_ins = epilogue_stream(oh) if ride == "epi" else []
    rb_parts = rebuild_parts(rebuild, name) if rebuild is not None else None
    # the ride occupies the gaps of k-steps 1 .. last (up to the barrier)
    last = nks - 1
    ride_gaps = [(ks, m) for ks in range(1, nks) for m in range(12) if not (ks == last and m >= BARRIER_AT - 1)]
    per_gap = dict(zip(ride_gaps, spread(len(ride_ins), len(ride_gaps)))) if ride_ins else {}
    ri = 0
    for ks in range(nks):
        ms = mfmas(half, ks)
        for m, mf in enumerate(ms):
            # ---- in front of the MFMA
            if m == 0:
                if vm_waits:
                    s.wait(vm=min(4 * (15 - ks), 63))
                s.need_lds(("xh", ks))
            if m == 8:
                s.need_lds(("xl", ks))
            if ks == last and m == BARRIER_AT:
                s.wait(lgkm=0)
                s.emit(I_barrier())
            s.emit(mf)
            # ---- behind it
            if m == 0 and ks < last:
                for r in frag_reads_h(half, ks + 1, (ks + 1) & 1):
                    s.emit(r, ("xh", ks + 1))
            if m == 11 and ks < last:
                for r in frag_reads_l(half, ks + 1):
                    s.emit(r, ("xl", ks + 1))
            if rebuild is not None and m == 2 and ks in (0, nks // 2):
                for r in rb_parts[0 if ks == 0 else 1]:
                    if isinstance(r, tuple) and r[0] == "NEED_VM":
                        s.need_vm(r[1])
                    elif isinstance(r, tuple):
                        s.emit(r[1], "tload", group="rebuild_t")
                    else:
                        # the stash stores are issued under F_REBUILD, everything of the time-code part under F_REBUILD_T
                        s.emit(r, "rebuild", group="rebuild_x" if (ks == 0 and r.kind == "lds_w") else "rebuild_t")
            if refills and ks >= 1 and 3 <= m <= 7:
                for r in refill(ks - 1)[m - 3]:
                    s.emit(r, ("w", ks - 1))
            n = per_gap.get((ks, m), 0)
            for _ in range(n):
                s.emit(ride_ins[ri], "ride")
                ri += 1
            if ks == last and m == BARRIER_AT:
                if tail_init:
                    skip = f"L_{name}_noinit"
                    s.emit(I_s_cmp("s_bitcmp1_b32", S(S_CUR + D_FLAGS), F_INIT))
                    s.emit(I_branch("s_cbranch_scc0", skip))
                    s.emit(I_valu("v_add_u32", V_BADDR, S(S_CUR + D_BIAS), V_BIAS,
                                  text=f"v_add_u32_e32 {V_BADDR}, {S(S_CUR + D_BIAS)}, {V_BIAS}"))
                    for r in init_reads(oh):
                        s.emit(r, "init", group="init")
                    s.emit(I_label(skip))
                if prefetch:
                    for r in frag_reads_h(oh, 0, 0):
                        s.emit(r, ("xh'", 0))
            if ks == last and m == 11:
                if refills:
                    for r in refill_flat(last):
                        s.emit(r, ("w", last))
                if prefetch:
                    for r in frag_reads_l(oh, 0):
                        s.emit(r, ("xl'", 0))
    assert ri == len(ride_ins)
    if not prefetch:
        s.wait(lgkm=0)
    s.emit(I_branch("s_branch", "L_dispatch"))
    return s.ins


def bare_epilogue(name, half, end):
    """Epilogue of `half` with no MFMAs beside it.  EPI_A: then barrier, bias init of acc_A (flag) and the first fragments of A.
    EPI_B (end of the trunk): everything this wave started has landed when the body is left."""
    s = Stream()
    s.emit(I_label(f"L_{name}"))
    s.emit(I_nop(7)); s.emit(I_nop(7))          # the last MFMAs on these accumulators were issued a few states ago
    for r in epilogue_stream(half):
        s.emit(r, "ride")
    s.wait(vm=0, lgkm=0)
    if not end:
        s.emit(I_barrier())
        skip = f"L_{name}_noinit"
        s.emit(I_s_cmp("s_bitcmp1_b32", S(S_CUR + D_FLAGS), F_INIT))
        s.emit(I_branch("s_cbranch_scc0", skip))
        s.emit(I_valu("v_add_u32", V_BADDR, S(S_CUR + D_BIAS), V_BIAS, text=f"v_add_u32_e32 {V_BADDR}, {S(S_CUR + D_BIAS)}, {V_BIAS}"))
        for r in init_reads(half):
            s.emit(r, "init", group="init")
        s.emit(I_label(skip))
        for r in frag_reads_h(half, 0, 0) + frag_reads_l(half, 0):
            s.emit(r)
    s.emit(I_branch("s_branch", "L_dispatch"))
    return s.ins


def raw(text, wr=(), rd=()):
    return Inst("raw", text, rd, wr, "other")


def prologue():
    """Inputs (inline-asm operands): %[pk] s64, %[phases] s64 (phase descriptors), %[lds] s32 (byte address of the activation
    tile), %[biaslds] s32, %[wave] s32, %[in_t] s32, %[tid] v32, %[tpa] v64, %[tpb] v64, %[first] (descriptor 0 is the prologue's:
    weight streams of the first two segments, bias offset of segment 0)."""
    o = []
    e = o.append
    e(raw("s_mov_b64 s[40:41], %[pk]", [S_PK]))
    e(raw("s_mov_b64 s[46:47], %[phases]", [S_PH]))
    e(raw("s_mov_b32 s42, %[lds]", [S_LDS]))
    e(raw("s_mov_b32 s43, %[biaslds]", [S_BIASLDS]))
    e(raw("s_mov_b32 s44, %[wave]", [S_WAVE]))
    e(raw("s_mov_b32 s45, %[in_t]", [S_INT]))
    e(raw("v_mov_b32 v39, %[tid]", [V_TMP]))
    e(raw("v_mov_b32 v34, %[tpa0]", [V(34)])); e(raw("v_mov_b32 v35, %[tpa1]", [V(35)]))
    e(raw("v_mov_b32 v36, %[tpb0]", [V(36)])); e(raw("v_mov_b32 v37, %[tpb1]", [V(37)]))
    # descriptor 0 -> cur, descriptor 1 -> nxt (fetched now, valid after the first lgkmcnt(0) below)
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
    e(I_valu("v_lshlrev_b32", V(T0 + 3), 5, h)); e(I_valu("v_add_u32", V_WR_H, V_WR_H, V(T0 + 3)))
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
    # n4 = float4s of time code this thread restores: clamp((in_t - 16 q) / 4, 0, 4)   (v_n4 = min(4, max(0, ...)) via u32 tricks)
    e(I_valu("v_lshlrev_b32", V(T0 + 5), 4, V(T0 + 4)))                       # 16 q
    e(I_valu("v_mov_b32", V_N4, S_INT, text=f"v_mov_b32_e32 {V_N4}, {S_INT}"))
    e(I_valu("v_min_u32", V(T0 + 5), V(T0 + 5), V_N4))                        # min(16 q, in_t)
    e(I_valu("v_sub_u32", V_N4, V_N4, V(T0 + 5)))                             # in_t - min(16 q, in_t) >= 0
    e(I_valu("v_lshrrev_b32", V_N4, 2, V_N4)); e(I_valu("v_min_u32", V_N4, 4, V_N4))
    # stash of the input tile (the C++ encoder built it and synchronised the workgroup before the asm)
    for hb in range(2):
        st, ho = STASH + 16 * hb, hb * HALF_B
        e(I_ds_read_b128(V(st, 4), V_ST_H, ho)); e(I_ds_read_b128(V(st + 4, 4), V_ST_H, ho + 16))
        e(I_ds_read_b128(V(st + 8, 4), V_ST_L, ho)); e(I_ds_read_b128(V(st + 12, 4), V_ST_L, ho + 16))
    e(I_wait(lgkm=0))                                                         # descriptors + stash
    # weights: descriptor 0 names the streams -- r1 = segment 0 (slots < n1), r2 = segment 1 (slots >= n1)
    e(I_salu("s_mul_i32", S_T0, S_WAVE, S(S_CUR + D_R1W))); e(I_salu("s_add_u32", S_R1, S(S_CUR + D_R1), S_T0, scc=True))
    e(I_salu("s_mul_i32", S_T0, S_WAVE, S(S_CUR + D_R2W))); e(I_salu("s_add_u32", S_R2, S(S_CUR + D_R2), S_T0, scc=True))
    for ks in range(16):
        o.extend(refill_flat(ks))
    # acc_A, acc_B := bias of segment 0 (flag), first fragments of half A
    e(I_s_cmp("s_bitcmp1_b32", S(S_CUR + D_FLAGS), F_INIT)); e(I_branch("s_cbranch_scc0", "L_pro_noinit"))
    e(I_valu("v_add_u32", V_BADDR, S(S_CUR + D_BIAS), V_BIAS, text=f"v_add_u32_e32 {V_BADDR}, {S(S_CUR + D_BIAS)}, {V_BIAS}"))
    o.extend(init_reads("A")); o.extend(init_reads("B"))
    e(I_label("L_pro_noinit"))
    o.extend(frag_reads_h("A", 0, 0)); o.extend(frag_reads_l("A", 0))
    return o


def dispatcher():
    o = [I_label("L_dispatch")]
    e = o.append
    # cur <- nxt (valid: every body ends its LDS / SMEM traffic with lgkmcnt(0) before it comes here or before its barrier)
    for k in range(8):
        e(I_salu("s_mov_b32", S(S_CUR + k), S(S_NXT + k)))
    e(I_s_load(S(S_NXT, 8), S_PH, 0))
    e(I_salu("s_add_u32", S(46), S(46), 32, scc=True)); e(I_salu("s_addc_u32", S(47), S(47), 0, scc=True))
    e(I_salu("s_mul_i32", S_T0, S_WAVE, S(S_CUR + D_R1W))); e(I_salu("s_add_u32", S_R1, S(S_CUR + D_R1), S_T0, scc=True))
    e(I_salu("s_mul_i32", S_T0, S_WAVE, S(S_CUR + D_R2W))); e(I_salu("s_add_u32", S_R2, S(S_CUR + D_R2), S_T0, scc=True))
    for name in ("A16R", "B16R", "EPI_A", "A4", "B4", "A8", "B8", "B16X", "EPI_B"):
        e(I_s_cmp("s_cmp_eq_u32", S(S_CUR + D_BODY), BODY[name]))
        e(I_branch("s_cbranch_scc1", f"L_{name}"))
    e(I_branch("s_branch", "L_end"))
    return o


def build():
    prog = []
    prog += prologue()
    prog.append(I_branch("s_branch", "L_dispatch"))
    bodies = {
        "A16R": phase_body("A16R", "A", 16, ride="epi", tail_init=True, vm_waits=True),
        "B16R": phase_body("B16R", "B", 16, ride="epi", refills=True, tail_init=True),
        "B16X": phase_body("B16X", "B", 16, refills=True, rebuild="A"),
        "A4": phase_body("A4", "A", 4, rebuild="B", vm_waits=True),
        "A8": phase_body("A8", "A", 8, rebuild="B", vm_waits=True),
        "B4": phase_body("B4", "B", 4, refills=True, prefetch=False),
        "B8": phase_body("B8", "B", 8, refills=True, prefetch=False),
        "EPI_A": bare_epilogue("EPI_A", "A", end=False),
        "EPI_B": bare_epilogue("EPI_B", "B", end=True),
    }
    for name in ("A16R", "B16R", "B16X", "A4", "A8", "B4", "B8", "EPI_A", "EPI_B"):
        prog += bodies[name]
    prog += dispatcher()
    prog.append(I_label("L_end"))
    prog.append(I_wait(vm=0, lgkm=0))
    return prog, bodies


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
    prog, bodies = build()
    errs = lint(bodies, prog)
    for e_ in errs[:40]:
        print("LINT:", e_)
    if errs:
        sys.exit(f"{len(errs)} hazard(s)")
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "nsff_pl_amd", "csrc", "field_h3a_body.inc")
    text = ("// GENERATED by tools/h3asm/gen.py -- do not edit.  The hand-scheduled trunk body of nsff_field_kernel_h3a\n"
            "// (one asm statement; registers v24..v255, a0..a255, s40..s99 are its own).\n" + render(prog))
    with open(out, "w") as f:
        f.write(text)
    n_m = sum(1 for i in prog if i.kind == "mfma")
    n_all = sum(1 for i in prog if i.kind not in ("label", "other"))
    print(f"wrote {os.path.normpath(out)}: {n_all} instructions, {n_m} MFMAs")
    for name, ins in bodies.items():
        print(f"  {name:6s} {sum(1 for i in ins if i.kind not in ('label', 'other')):5d} instructions, "
              f"{sum(1 for i in ins if i.kind == 'mfma'):4d} MFMAs")


if __name__ == "__main__":
    main()
