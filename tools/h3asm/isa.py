"""A small model of the gfx950 instructions the hand-scheduled field-kernel body uses (tools/h3asm/gen.py).

Three things live here:
  * ``Inst`` + the ``I_*`` constructors: one object per emitted instruction, carrying the registers it reads / writes,
    so that the generator can (a) print the assembly text, (b) LINT the final stream for the software-visible hazards
    of the part (MFMA result -> VALU read wait states, VALU -> v_permlane32_swap, waitcnt discipline) and
  * ``Sim``: a functional simulator of ONE workgroup (four waves, 64 lanes) executing such a stream: registers, LDS,
    a flat global memory, in-order LDS / VMEM return queues (a register written by an outstanding load may not be
    touched before an ``s_waitcnt`` has covered it), ``s_barrier`` rendezvous and a cross-wave LDS race detector
    (two waves touching the same dword between the same pair of barriers, at least one of them writing).

It is test infrastructure for the generator: the product is the generated text, assembled by hipcc.
Register notation: ('v', i), ('a', i), ('s', i) with a width in dwords where an operand is a tuple of registers.
"""
import numpy as np

NL = 64  # lanes


class Reg:
    __slots__ = ("f", "i", "n")

    def __init__(self, f, i, n=1):
        self.f, self.i, self.n = f, int(i), int(n)

    def __repr__(self):
        if self.n == 1:
            return f"{self.f}{self.i}"
        return f"{self.f}[{self.i}:{self.i + self.n - 1}]"

    def regs(self):
        return [(self.f, self.i + k) for k in range(self.n)]

    def sub(self, k, n=1):
        assert 0 <= k and k + n <= self.n
        return Reg(self.f, self.i + k, n)


def V(i, n=1):
    return Reg("v", i, n)


def A(i, n=1):
    return Reg("a", i, n)


def S(i, n=1):
    return Reg("s", i, n)


VCC = Reg("s", 106, 2)      # vcc_lo / vcc_hi as s106 / s107 in this model
EXEC = Reg("s", 126, 2)


class Inst:
    """op: mnemonic key; text: assembly line; rd / wr: lists of (file, index); kind: 'mfma' | 'valu' | 'salu' | 'lds_r' |
    'lds_w' | 'vmem' | 'smem' | 'wait' | 'barrier' | 'branch' | 'label' | 'nop' | 'other'"""
    __slots__ = ("op", "text", "rd", "wr", "kind", "args", "tag")

    def __init__(self, op, text, rd=(), wr=(), kind="other", args=None, tag=None):
        self.op, self.text, self.kind, self.args, self.tag = op, text, kind, args or {}, tag
        self.rd = [r for x in rd for r in (x.regs() if isinstance(x, Reg) else [x])]
        self.wr = [r for x in wr for r in (x.regs() if isinstance(x, Reg) else [x])]

    def __repr__(self):
        return self.text


def _src(x):
    if isinstance(x, Reg):
        return repr(x)
    if isinstance(x, float):
        return repr(x)
    return str(x)


def _is_reg(x):
    return isinstance(x, Reg)


# ---------------------------------------------------------------- constructors ------------------------------------
def I_mfma(d, a, b, c):
    """v_mfma_f32_32x32x16_f16 D(16), A(4), B(4), C(16 or the inline constant 0)"""
    rd = [a, b] + ([c] if _is_reg(c) else [])
    return Inst("mfma", f"v_mfma_f32_32x32x16_f16 {d}, {a}, {b}, {_src(c)}", rd, [d], "mfma", dict(d=d, a=a, b=b, c=c))


def I_ds_read_b128(d, addr, off=0):
    assert 0 <= off < 65536 and d.n == 4
    return Inst("ds_read_b128", f"ds_read_b128 {d}, {addr} offset:{off}", [addr], [d], "lds_r", dict(d=d, addr=addr, off=off))


def I_ds_write_b128(addr, data, off=0):
    assert 0 <= off < 65536 and data.n == 4
    return Inst("ds_write_b128", f"ds_write_b128 {addr}, {data} offset:{off}", [addr, data], [], "lds_w",
                dict(addr=addr, data=data, off=off))


def I_ds_write_b64(addr, data, off=0):
    assert 0 <= off < 65536 and data.n == 2
    return Inst("ds_write_b64", f"ds_write_b64 {addr}, {data} offset:{off}", [addr, data], [], "lds_w",
                dict(addr=addr, data=data, off=off))


def I_ds_write_b32(addr, data, off=0):
    assert 0 <= off < 65536 and data.n == 1
    return Inst("ds_write_b32", f"ds_write_b32 {addr}, {data} offset:{off}", [addr, data], [], "lds_w",
                dict(addr=addr, data=data, off=off))


def I_ds_read_tr(d, addr, off=0):
    """ds_read_b64_tr_b16 d[2], addr offset: within each group of 16 lanes, lane i supplies the address of 4 consecutive halfs
    (one row of a 16 x 4 block); lane l receives element j = the half (l & 3) of the row supplied by lane 4 j + ((l & 15) >> 2) --
    a transposing read (tools/debug/probes/tr_read_probe.hip prints the mapping on the part)."""
    assert 0 <= off < 65536 and d.n == 2
    return Inst("ds_read_b64_tr_b16", f"ds_read_b64_tr_b16 {d}, {addr} offset:{off}", [addr], [d], "lds_r", dict(d=d, addr=addr, off=off))


def I_ds_read_b64(d, addr, off=0):
    assert 0 <= off < 65536 and d.n == 2
    return Inst("ds_read_b64", f"ds_read_b64 {d}, {addr} offset:{off}", [addr], [d], "lds_r", dict(d=d, addr=addr, off=off))


def I_ds_read_b32(d, addr, off=0):
    assert 0 <= off < 65536 and d.n == 1
    return Inst("ds_read_b32", f"ds_read_b32 {d}, {addr} offset:{off}", [addr], [d], "lds_r", dict(d=d, addr=addr, off=off))


def I_gload_s(d, voff, sbase, off=0):
    """global_load_dword[x2|x4] d, voff, s[base:base+1] offset  (address = sbase + zext(voff) + off)"""
    assert -4096 <= off <= 4095 and d.n in (1, 2, 4) and sbase.n == 2
    suffix = {1: "dword", 2: "dwordx2", 4: "dwordx4"}[d.n]
    return Inst("gload_s", f"global_load_{suffix} {d}, {voff}, {sbase} offset:{off}", [voff, sbase], [d], "vmem",
                dict(d=d, voff=voff, sbase=sbase, off=off))


def I_gstore_s(voff, data, sbase, off=0):
    """global_store_dword[x2|x4] voff, data, s[base:base+1] offset  (address = sbase + zext(voff) + off)"""
    assert 0 <= off <= 4095 and data.n in (1, 2, 4) and sbase.n == 2
    suffix = {1: "dword", 2: "dwordx2", 4: "dwordx4"}[data.n]
    return Inst("gstore_s", f"global_store_{suffix} {voff}, {data}, {sbase} offset:{off}", [voff, data, sbase], [], "vmem",
                dict(data=data, voff=voff, sbase=sbase, off=off))


def I_v_pk_add_f16(d, a, b):
    return Inst("v_pk_add_f16", f"v_pk_add_f16 {d}, {a}, {b}", [a, b], [d], "valu", dict(d=d, s=[a, b]))


def I_v_lshl_or(d, a, sh, c):            # d = (a << sh) | c
    return Inst("v_lshl_or_b32", f"v_lshl_or_b32 {d}, {a}, {sh}, {c}", [x for x in (a, c) if _is_reg(x)], [d], "valu", dict(d=d, s=[a, sh, c]))


def I_gload_x4_s(d, voff, sbase, off=0):
    """global_load_dwordx4 d, voff, s[base:base+1] offset  (address = sbase + zext(voff) + off)"""
    assert -4096 <= off <= 4095 and d.n == 4 and sbase.n == 2
    return Inst("gload_s", f"global_load_dwordx4 {d}, {voff}, {sbase} offset:{off}", [voff, sbase], [d], "vmem",
                dict(d=d, voff=voff, sbase=sbase, off=off))


def I_gload_x4_v(d, vaddr, off=0):
    """global_load_dwordx4 d, v[addr:addr+1], off offset"""
    assert -4096 <= off <= 4095 and d.n == 4 and vaddr.n == 2
    return Inst("gload_v", f"global_load_dwordx4 {d}, {vaddr}, off offset:{off}", [vaddr], [d], "vmem",
                dict(d=d, vaddr=vaddr, off=off))


def I_valu(op, d, *srcs, text=None, extra_wr=()):
    t = text or f"{op} {d}, " + ", ".join(_src(s) for s in srcs)
    return Inst(op, t, [s for s in srcs if _is_reg(s)], [d] + list(extra_wr), "valu", dict(d=d, s=list(srcs)))


def I_v_max0(d, s):                      # ReLU in place or not
    return I_valu("v_max_f32", d, 0, s)


def I_v_cvt_pkrtz(d, a, b):
    return I_valu("v_cvt_pkrtz_f16_f32", d, a, b)


def I_v_sub_lo_half(d, h, v):            # d = v - float(h.lo16)
    return I_valu("v_fma_mix_lo", d, h, v, text=f"v_fma_mix_f32 {d}, {h}, -1.0, {v} op_sel_hi:[1,0,0]")


def I_v_sub_hi_half(d, h, v):            # d = v - float(h.hi16)
    return I_valu("v_fma_mix_hi", d, h, v, text=f"v_fma_mix_f32 {d}, {h}, -1.0, {v} op_sel:[1,0,0] op_sel_hi:[1,0,0]")


def I_v_permlane32_swap(a, b):           # a.hi <-> b.lo
    return Inst("v_permlane32_swap", f"v_permlane32_swap_b32 {a}, {b}", [a, b], [a, b], "valu", dict(a=a, b=b))


def I_v_cmp_lt_u32_vcc(a, b):            # vcc = a < b   (a: constant or reg, b: VGPR)
    return Inst("v_cmp_lt_u32", f"v_cmp_lt_u32_e32 vcc, {_src(a)}, {b}", [x for x in (a, b) if _is_reg(x)], [VCC], "valu",
                dict(a=a, b=b))


def I_v_cmp_gt_u32_vcc(a, b):            # vcc = a > b   (a: constant or SGPR or reg, b: VGPR)
    return Inst("v_cmp_gt_u32", f"v_cmp_gt_u32_e32 vcc, {_src(a)}, {b}", [x for x in (a, b) if _is_reg(x)], [VCC], "valu",
                dict(a=a, b=b))


def I_salu(op, d, *srcs, text=None, scc=False):
    t = text or f"{op} {d}, " + ", ".join(_src(s) for s in srcs)
    return Inst(op, t, [s for s in srcs if _is_reg(s)], [d] if d is not None else [], "salu", dict(d=d, s=list(srcs), scc=scc))


def I_s_cmp(op, a, b):                   # sets SCC
    return Inst(op, f"{op} {_src(a)}, {_src(b)}", [x for x in (a, b) if _is_reg(x)], [], "salu", dict(a=a, b=b))


def I_s_and_saveexec(d):                 # d = exec; exec &= vcc
    return Inst("s_and_saveexec_b64", f"s_and_saveexec_b64 {d}, vcc", [VCC, EXEC], [d, EXEC], "salu", dict(d=d))


def I_s_mov_exec(src):                   # exec = src (s pair) or -1
    return Inst("s_mov_exec", f"s_mov_b64 exec, {_src(src)}", [src] if _is_reg(src) else [], [EXEC], "salu", dict(s=src))


def I_s_load(d, base, off):
    assert d.n in (1, 2, 4, 8, 16)
    suffix = {1: "dword", 2: "dwordx2", 4: "dwordx4", 8: "dwordx8", 16: "dwordx16"}[d.n]
    return Inst("s_load", f"s_load_{suffix} {d}, {base}, {hex(off)}", [base], [d], "smem", dict(d=d, base=base, off=off))


def I_wait(vm=None, lgkm=None):
    parts = []
    if vm is not None:
        assert 0 <= vm <= 63
        parts.append(f"vmcnt({vm})")
    if lgkm is not None:
        assert 0 <= lgkm <= 15
        parts.append(f"lgkmcnt({lgkm})")
    return Inst("s_waitcnt", "s_waitcnt " + " ".join(parts), [], [], "wait", dict(vm=vm, lgkm=lgkm))


def I_barrier():
    return Inst("s_barrier", "s_barrier", [], [], "barrier")


def I_nop(n):
    assert 0 <= n <= 7
    return Inst("s_nop", f"s_nop {n}", [], [], "nop", dict(n=n))


def I_label(name):
    return Inst("label", f"{name}:", [], [], "label", dict(name=name))


def I_branch(op, target):                # s_branch / s_cbranch_scc0 / s_cbranch_scc1
    return Inst(op, f"{op} {target}", [], [], "branch", dict(target=target))


def I_comment(text):
    return Inst("comment", f"; {text}", [], [], "other")


def I_memtime(d):
    return Inst("s_memtime", f"s_memtime {d}", [], [d], "smem", dict(d=d))


# ---------------------------------------------------------------- lint ---------------------------------------------
MFMA_TO_VALU_STATES = 11      # 8-pass XDL write -> VALU / LDS / VMEM access of the result: passes + 2 (+1 on gfx950)
VALU_TO_PERMLANE_STATES = 2


def states(inst):
    if inst.kind in ("label", "other"):
        return 0
    if inst.op == "s_nop":
        return inst.args["n"] + 1
    return 1


def lint_straight(insts, what=""):
    """Wait-state hazards inside one straight-line stream (every body of the generator is straight-line apart from forward
    skips over guarded clusters, which only remove instructions -- so a skipped cluster can shorten a distance: the lint is
    run on the stream WITH and WITHOUT every guarded cluster by the caller)."""
    last_mfma_wr = {}       # reg -> state index of the MFMA that wrote it
    last_valu_wr = {}
    t = 0
    errs = []
    for k, ins in enumerate(insts):
        if ins.kind == "mfma":
            a = ins.args
            # A / B / C operands must not be fresh MFMA results of another tile (C == D accumulate chain is forwarded)
            for r in ins.rd:
                if r in last_mfma_wr and not (_is_reg(a["c"]) and r in a["c"].regs() and a["c"].i == a["d"].i):
                    if t - last_mfma_wr[r] < MFMA_TO_VALU_STATES:
                        errs.append(f"{what}[{k}] {ins.text}: reads {r} {t - last_mfma_wr[r]} states after the MFMA that wrote it")
        elif ins.kind in ("valu", "lds_r", "lds_w", "vmem"):
            for r in ins.rd + ins.wr:
                if r in last_mfma_wr and t - last_mfma_wr[r] < MFMA_TO_VALU_STATES:
                    errs.append(f"{what}[{k}] {ins.text}: touches {r} {t - last_mfma_wr[r]} states after the MFMA that wrote it")
        if ins.op == "v_permlane32_swap":
            for r in ins.rd:
                if r in last_valu_wr and t - last_valu_wr[r] <= VALU_TO_PERMLANE_STATES:
                    errs.append(f"{what}[{k}] {ins.text}: reads {r} {t - last_valu_wr[r] - 1} states after a VALU write")
        t += states(ins)
        if ins.kind == "mfma":
            for r in ins.wr:
                last_mfma_wr[r] = t
        elif ins.kind == "valu":
            for r in ins.wr:
                last_valu_wr[r] = t
                last_mfma_wr.pop(r, None)
        else:
            for r in ins.wr:
                last_mfma_wr.pop(r, None)
    return errs


# ---------------------------------------------------------------- simulator ----------------------------------------
class SimError(Exception):
    pass


def f16_rtz_pack(a, b):
    """v_cvt_pkrtz_f16_f32: two fp32 arrays -> packed u32 (round toward zero)."""
    def rtz(x):
        x = x.astype(np.float32)
        h = x.astype(np.float16)                         # round to nearest even
        back = h.astype(np.float32)
        over = np.abs(back) > np.abs(x)                  # rounded away from zero: step one ulp toward zero
        hb = h.view(np.uint16).copy()
        hb[over] -= 1
        h2 = hb.view(np.float16)
        inf = np.isinf(h2) & np.isfinite(x)
        hb2 = h2.view(np.uint16).copy()
        hb2[inf] = (hb2[inf] & 0x8000) | 0x7BFF
        return hb2
    lo, hi = rtz(a), rtz(b)
    return lo.astype(np.uint32) | (hi.astype(np.uint32) << 16)


def halfs_of(u):
    """packed u32 array -> (lo half as f32, hi half as f32)"""
    lo = (u & 0xFFFF).astype(np.uint16).view(np.float16).astype(np.float32)
    hi = (u >> 16).astype(np.uint16).view(np.float16).astype(np.float32)
    return lo, hi


class Wave:
    def __init__(self, wid):
        self.id = wid
        self.v = np.zeros((256, NL), np.uint32)
        self.a = np.zeros((256, NL), np.uint32)
        self.s = np.zeros(128, np.uint32)
        self.exec = np.ones(NL, bool)
        self.scc = 0
        self.pc = 0
        self.epoch = 0
        self.lds_q = []          # outstanding LDS ops in order: list of sets of dest regs (empty set for writes)
        self.vm_q = []
        self.sm_q = []           # outstanding SMEM (may return out of order: only lgkmcnt(0) resolves them)
        self.pending = {}        # (file, idx) -> queue name
        self.n_inst = 0
        self.n_mfma = 0
        self.done = False


class Sim:
    def __init__(self, prog, lds_bytes=160 * 1024):
        self.prog = prog
        self.labels = {ins.args["name"]: k for k, ins in enumerate(prog) if ins.kind == "label"}
        self.lds = np.zeros(lds_bytes // 4, np.uint32)
        self.lds_w_epoch = np.full(lds_bytes // 4, -1, np.int64)
        self.lds_w_wave = np.full(lds_bytes // 4, -1, np.int64)
        self.lds_r_epoch = np.full((4, lds_bytes // 4), -1, np.int64)
        self.mem = {}            # base address -> np.uint32 array
        self.mem_written = None  # {base: bool array}: set by the harness to catch a global dword stored twice
        self.waves = [Wave(w) for w in range(4)]
        self.trace = None

    # ---- memory ----
    def add_buffer(self, base, arr_u32):
        self.mem[base] = np.ascontiguousarray(arr_u32).view(np.uint32).reshape(-1)

    def _gread(self, addr, n_dw=4):
        """addr: (NL,) uint64 byte addresses, 4 n_dw-byte loads -> (n_dw, NL) u32"""
        out = np.zeros((n_dw, NL), np.uint32)
        nb = np.uint64(4 * n_dw)
        for base, arr in self.mem.items():
            m = (addr >= base) & (addr + nb <= base + arr.size * 4)
            if m.any():
                idx = ((addr[m] - base) // 4).astype(np.int64)
                if ((addr[m] - base) % 4).any():
                    raise SimError("unaligned global load")
                for k in range(n_dw):
                    out[k, m] = arr[idx + k]
        ok = np.zeros(NL, bool)
        for base, arr in self.mem.items():
            ok |= (addr >= base) & (addr + nb <= base + arr.size * 4)
        return out, ok

    # ---- register access with pending checks ----
    def _chk(self, w, regs, ins, write=False):
        for r in regs:
            if r in w.pending:
                raise SimError(f"wave {w.id} pc {w.pc}: `{ins.text}` {'writes' if write else 'reads'} {r[0]}{r[1]} while a "
                               f"{w.pending[r]} load into it is outstanding")

    def _rv(self, w, reg, k=0):
        f = w.v if reg.f == "v" else w.a
        return f[reg.i + k]

    def _val(self, w, x, as_float=False):
        """scalar-or-vector source as a (NL,) uint32 array"""
        if isinstance(x, Reg):
            if x.f == "s":
                return np.full(NL, w.s[x.i], np.uint32)
            return self._rv(w, x).copy()
        if isinstance(x, float):
            return np.full(NL, np.float32(x).view(np.uint32), np.uint32)
        return np.full(NL, np.uint32(x & 0xFFFFFFFF), np.uint32)

    def _sval(self, w, x):
        if isinstance(x, Reg):
            assert x.f == "s"
            return int(w.s[x.i])
        return int(x) & 0xFFFFFFFF

    def _wv(self, w, reg, val, k=0, masked=True):
        f = w.v if reg.f == "v" else w.a
        if masked:
            f[reg.i + k][w.exec] = val[w.exec]
        else:
            f[reg.i + k] = val

    # ---- LDS with race detection ----
    def _lds_access(self, w, byte_addr, n_dw, write, data=None, lanes=None):
        lanes = w.exec if lanes is None else lanes
        if (byte_addr[lanes] % (4 * n_dw)).any():
            raise SimError(f"wave {w.id}: unaligned {4 * n_dw}-byte LDS access")
        idx0 = (byte_addr // 4).astype(np.int64)
        out = np.zeros((n_dw, NL), np.uint32)
        for k in range(n_dw):
            idx = idx0[lanes] + k
            if (idx < 0).any() or (idx >= self.lds.size).any():
                raise SimError(f"wave {w.id}: LDS access out of range")
            if write:
                # race: another wave read or wrote this dword in the same epoch
                other_r = np.zeros(idx.shape, bool)
                for ow in range(4):
                    if ow != w.id:
                        other_r |= self.lds_r_epoch[ow, idx] == w.epoch
                other_w = (self.lds_w_epoch[idx] == w.epoch) & (self.lds_w_wave[idx] != w.id)
                if other_r.any() or other_w.any():
                    bad = idx[(other_r | other_w)][0]
                    raise SimError(f"LDS RACE: wave {w.id} writes dword {bad} (byte {bad * 4}) in epoch {w.epoch} that another "
                                   f"wave {'read' if other_r.any() else 'wrote'} in the same epoch (pc {w.pc})")
                self.lds[idx] = data[k][lanes]
                self.lds_w_epoch[idx] = w.epoch
                self.lds_w_wave[idx] = w.id
            else:
                other_w = (self.lds_w_epoch[idx] == w.epoch) & (self.lds_w_wave[idx] != w.id)
                if other_w.any():
                    bad = idx[other_w][0]
                    raise SimError(f"LDS RACE: wave {w.id} reads dword {bad} (byte {bad * 4}) in epoch {w.epoch} that wave "
                                   f"{self.lds_w_wave[bad]} wrote in the same epoch (pc {w.pc})")
                self.lds_r_epoch[w.id, idx] = w.epoch
                out[k, lanes] = self.lds[idx]
        return out

    # ---- execution ----
    def step(self, w):
        ins = self.prog[w.pc]
        w.pc += 1
        k = ins.kind
        if k in ("label", "other", "nop"):
            return None
        w.n_inst += 1
        a = ins.args
        if k == "wait":
            if a["lgkm"] is not None:
                n = a["lgkm"]
                if n == 0:
                    for regs in w.sm_q:
                        for r in regs:
                            w.pending.pop(r, None)
                    w.sm_q = []
                # LDS ops return in order; outstanding SMEM ops count too but can only make the wait stricter
                allow = max(0, n - len(w.sm_q))
                while len(w.lds_q) > allow:
                    for r in w.lds_q.pop(0):
                        w.pending.pop(r, None)
            if a["vm"] is not None:
                while len(w.vm_q) > a["vm"]:
                    for r in w.vm_q.pop(0):
                        w.pending.pop(r, None)
            return None
        if k == "barrier":
            if w.lds_q or w.sm_q:
                # legal on hardware, but every barrier of this kernel is meant to publish finished LDS traffic
                raise SimError(f"wave {w.id} pc {w.pc}: s_barrier with {len(w.lds_q)} LDS operations outstanding")
            return "barrier"
        if k == "branch":
            op = ins.op
            take = op == "s_branch" or (op == "s_cbranch_scc1" and w.scc) or (op == "s_cbranch_scc0" and not w.scc)
            if take:
                w.pc = self.labels[a["target"]]
            return None
        self._chk(w, ins.rd, ins)
        self._chk(w, ins.wr, ins, write=True)
        if k == "mfma":
            w.n_mfma += 1
            d, A_, B_, C_ = a["d"], a["a"], a["b"], a["c"]

            def frag(reg):       # (4 regs, NL) u32 -> [32 rows][16 k] f32:  lane l: row l%32, k = 8*(l//32) + t
                m = np.zeros((32, 16), np.float32)
                for r4 in range(4):
                    lo, hi = halfs_of(self._rv(w, reg, r4))
                    for hb in range(2):
                        m[:, 8 * hb + 2 * r4] = lo[32 * hb:32 * hb + 32]
                        m[:, 8 * hb + 2 * r4 + 1] = hi[32 * hb:32 * hb + 32]
                return m
            Am, Bm = frag(A_), frag(B_)                   # A[i][k], B[j][k]
            prod = (Am.astype(np.float64) @ Bm.astype(np.float64).T)      # [i][j]
            for r in range(16):
                acc = np.zeros(NL, np.float32) if not isinstance(C_, Reg) else self._rv(w, C_, r).view(np.float32).copy()
                for hb in range(2):
                    i = 8 * (r // 4) + 4 * hb + (r % 4)
                    acc[32 * hb:32 * hb + 32] = (acc[32 * hb:32 * hb + 32].astype(np.float64) + prod[i, :]).astype(np.float32)
                self._wv(w, d, acc.view(np.uint32), r, masked=False)       # (MFMA ignores EXEC)
            return None
        if k == "lds_r" and ins.op == "ds_read_b64_tr_b16":
            addr = self._rv(w, a["addr"]).astype(np.int64) + a["off"]
            raw = self._lds_access(w, addr, 2, False)                  # (2, NL) dwords = 4 halfs per lane
            halfs = np.zeros((NL, 4), np.uint16)
            halfs[:, 0], halfs[:, 1] = raw[0] & 0xFFFF, raw[0] >> 16
            halfs[:, 2], halfs[:, 3] = raw[1] & 0xFFFF, raw[1] >> 16
            got = np.zeros((NL, 4), np.uint16)
            for l in range(NL):
                g0 = l & ~15
                for j in range(4):
                    got[l, j] = halfs[g0 + 4 * j + ((l & 15) >> 2), l & 3]
            self._wv(w, a["d"], got[:, 0].astype(np.uint32) | (got[:, 1].astype(np.uint32) << 16), 0)
            self._wv(w, a["d"], got[:, 2].astype(np.uint32) | (got[:, 3].astype(np.uint32) << 16), 1)
            regs = set(a["d"].regs())
            w.lds_q.append(regs)
            for r in regs:
                w.pending[r] = "LDS"
            return None
        if k == "lds_r":
            addr = self._rv(w, a["addr"]).astype(np.int64) + a["off"]
            n_dw = a["d"].n
            out = self._lds_access(w, addr, n_dw, False)
            for r4 in range(n_dw):
                self._wv(w, a["d"], out[r4], r4)
            regs = set(a["d"].regs())
            w.lds_q.append(regs)
            for r in regs:
                w.pending[r] = "LDS"
            return None
        if k == "lds_w":
            addr = self._rv(w, a["addr"]).astype(np.int64) + a["off"]
            n_dw = a["data"].n
            data = np.stack([self._rv(w, a["data"], r4) for r4 in range(n_dw)])
            self._lds_access(w, addr, n_dw, True, data)
            w.lds_q.append(set())
            return None
        if k == "vmem" and ins.op == "gstore_s":
            base = int(w.s[a["sbase"].i]) | (int(w.s[a["sbase"].i + 1]) << 32)
            addr = base + self._rv(w, a["voff"]).astype(np.uint64) + np.uint64(a["off"])
            n_dw = a["data"].n
            for l in range(NL):
                if not w.exec[l]:
                    continue
                hit = False
                for b0, arr in self.mem.items():
                    if b0 <= int(addr[l]) and int(addr[l]) + 4 * n_dw <= b0 + arr.size * 4:
                        if (int(addr[l]) - b0) % (4 * n_dw):
                            raise SimError(f"unaligned global store ({ins.text})")
                        o = (int(addr[l]) - b0) // 4
                        if self.mem_written is not None:
                            if self.mem_written[b0][o:o + n_dw].any():
                                raise SimError(f"wave {w.id}: `{ins.text}` writes global dword {o} of buffer {b0:#x} a second time")
                            self.mem_written[b0][o:o + n_dw] = True
                        for kk in range(n_dw):
                            arr[o + kk] = self._rv(w, a["data"], kk)[l]
                        hit = True
                if not hit:
                    raise SimError(f"wave {w.id} pc {w.pc}: `{ins.text}` writes unmapped global memory (lane {l}, address {int(addr[l]):#x})")
            w.vm_q.append(set())
            return None
        if k == "vmem":
            if ins.op == "gload_s":
                base = int(w.s[a["sbase"].i]) | (int(w.s[a["sbase"].i + 1]) << 32)
                addr = base + self._rv(w, a["voff"]).astype(np.uint64) + np.uint64(a["off"] & 0xFFFFFFFFFFFFFFFF if a["off"] >= 0 else 0)
                if a["off"] < 0:
                    addr = addr - np.uint64(-a["off"])
            else:
                lo = self._rv(w, a["vaddr"], 0).astype(np.uint64)
                hi = self._rv(w, a["vaddr"], 1).astype(np.uint64)
                addr = (lo | (hi << np.uint64(32))) + np.uint64(a["off"])
            out, ok = self._gread(addr, a["d"].n)
            if not ok[w.exec].all():
                raise SimError(f"wave {w.id} pc {w.pc}: `{ins.text}` reads unmapped global memory (lane {int(np.argmin(ok | ~w.exec))}, "
                               f"address {int(addr[np.argmin(ok | ~w.exec)]):#x})")
            for r4 in range(a["d"].n):
                self._wv(w, a["d"], out[r4], r4)
            regs = set(a["d"].regs())
            w.vm_q.append(regs)
            for r in regs:
                w.pending[r] = "VMEM"
            return None
        if k == "smem":
            if ins.op == "s_memtime":
                w.s[a["d"].i] = w.n_inst & 0xFFFFFFFF
                w.s[a["d"].i + 1] = 0
                return None
            base = int(w.s[a["base"].i]) | (int(w.s[a["base"].i + 1]) << 32)
            addr = base + a["off"]
            n = a["d"].n
            got = None
            for b0, arr in self.mem.items():
                if b0 <= addr and addr + 4 * n <= b0 + arr.size * 4:
                    got = arr[(addr - b0) // 4:(addr - b0) // 4 + n]
            if got is None:
                raise SimError(f"s_load from unmapped memory {addr:#x}")
            w.s[a["d"].i:a["d"].i + n] = got
            regs = set(a["d"].regs())
            w.sm_q.append(regs)
            for r in regs:
                w.pending[r] = "SMEM"
            return None
        if k == "valu":
            self._valu(w, ins)
            return None
        if k == "salu":
            self._salu(w, ins)
            return None
        raise SimError(f"unhandled instruction {ins.text}")

    def _valu(self, w, ins):
        op, a = ins.op, ins.args
        f32 = lambda u: u.view(np.float32)
        if op == "v_permlane32_swap":
            x, y = self._rv(w, a["a"]).copy(), self._rv(w, a["b"]).copy()
            nx, ny = x.copy(), y.copy()
            nx[32:] = y[:32]
            ny[:32] = x[32:]
            self._wv(w, a["a"], nx, masked=False)
            self._wv(w, a["b"], ny, masked=False)
            return
        if op in ("v_cmp_lt_u32", "v_cmp_gt_u32"):
            r = self._val(w, a["a"]) < self._val(w, a["b"]) if op == "v_cmp_lt_u32" else self._val(w, a["a"]) > self._val(w, a["b"])
            bits = 0
            for l in range(NL):
                if r[l] and w.exec[l]:
                    bits |= 1 << l
            w.s[VCC.i], w.s[VCC.i + 1] = bits & 0xFFFFFFFF, bits >> 32
            return
        d, s = a["d"], [self._val(w, x) for x in a["s"]]
        if op == "v_max_f32":
            out = np.maximum(f32(s[0]), f32(s[1])).view(np.uint32)
        elif op == "v_cvt_pkrtz_f16_f32":
            out = f16_rtz_pack(f32(s[0]), f32(s[1]))
        elif op == "v_fma_mix_lo":
            out = (f32(s[1]) - halfs_of(s[0])[0]).astype(np.float32).view(np.uint32)
        elif op == "v_fma_mix_hi":
            out = (f32(s[1]) - halfs_of(s[0])[1]).astype(np.float32).view(np.uint32)
        elif op == "v_mov_b32":
            out = s[0]
        elif op == "v_add_f32":
            out = (f32(s[0]) + f32(s[1])).astype(np.float32).view(np.uint32)
        elif op == "v_pk_add_f16":
            lo = (halfs_of(s[0])[0] + halfs_of(s[1])[0]).astype(np.float16)       # (the exact fp32 sum of two halfs, rounded to nearest even)
            hi = (halfs_of(s[0])[1] + halfs_of(s[1])[1]).astype(np.float16)
            out = lo.view(np.uint16).astype(np.uint32) | (hi.view(np.uint16).astype(np.uint32) << 16)
        elif op == "v_lshl_or_b32":
            out = ((s[0] << (s[1] & 31)) | s[2]).astype(np.uint32)
        elif op == "v_fmac_f32":
            out = (f32(s[0]).astype(np.float64) * f32(s[1]).astype(np.float64) + f32(s[2]).astype(np.float64)).astype(np.float32).view(np.uint32)
        elif op == "v_add_u32":
            out = s[0] + s[1]
        elif op == "v_sub_u32":
            out = s[0] - s[1]
        elif op == "v_and_b32":
            out = s[0] & s[1]
        elif op == "v_or_b32":
            out = s[0] | s[1]
        elif op == "v_lshrrev_b32":
            out = s[1] >> (s[0] & 31)
        elif op == "v_lshlrev_b32":
            out = s[1] << (s[0] & 31)
        elif op == "v_mul_u32_u24":
            out = ((s[0] & 0xFFFFFF).astype(np.uint64) * (s[1] & 0xFFFFFF).astype(np.uint64)).astype(np.uint32)
        elif op == "v_mul_lo_u32":
            out = (s[0].astype(np.uint64) * s[1].astype(np.uint64)).astype(np.uint32)
        elif op == "v_mad_u32_u24":
            out = (((s[0] & 0xFFFFFF).astype(np.uint64) * (s[1] & 0xFFFFFF).astype(np.uint64)) + s[2]).astype(np.uint32)
        elif op == "v_min_u32":
            out = np.minimum(s[0], s[1])
        elif op == "v_bfe_i32":              # sign-extended bit field: src, offset, width
            off_, wid = int(s[1][0]) & 31, int(s[2][0]) & 31
            fld = (s[0] >> np.uint32(off_)) & np.uint32((1 << wid) - 1)
            out = np.where((fld >> np.uint32(wid - 1)) & 1, fld | np.uint32((0xFFFFFFFF << wid) & 0xFFFFFFFF), fld).astype(np.uint32)
        elif op == "v_med3_f32":
            out = np.median(np.stack([f32(s[0]), f32(s[1]), f32(s[2])]), axis=0).astype(np.float32).view(np.uint32)
        elif op == "v_mul_f32":
            out = (f32(s[0]) * f32(s[1])).astype(np.float32).view(np.uint32)
        elif op == "v_cvt_pk_f16_f32":       # round to nearest even (the mode register's default); overflow -> inf, or +-65504 under FP16_OVFL
            with np.errstate(over="ignore"):
                lo = f32(s[0]).astype(np.float16)
                hi = f32(s[1]).astype(np.float16)
            if getattr(w, "fp16_ovfl", False):
                for h_, src_ in ((lo, f32(s[0])), (hi, f32(s[1]))):
                    m_ = np.isinf(h_) & np.isfinite(src_)
                    h_[m_] = np.where(src_[m_] > 0, np.float16(65504.0), np.float16(-65504.0))
            out = lo.view(np.uint16).astype(np.uint32) | (hi.view(np.uint16).astype(np.uint32) << 16)
        elif op == "v_pk_mul_f16":
            with np.errstate(over="ignore", under="ignore"):
                lo = (halfs_of(s[0])[0] * halfs_of(s[1])[0]).astype(np.float16)       # (the exact fp32 product of two halfs, rounded once)
                hi = (halfs_of(s[0])[1] * halfs_of(s[1])[1]).astype(np.float16)
            out = lo.view(np.uint16).astype(np.uint32) | (hi.view(np.uint16).astype(np.uint32) << 16)
        elif op == "v_mbcnt_lo_u32_b32":
            mask = int(s[0][0])
            out = np.array([bin(mask & ((1 << min(l, 32)) - 1)).count("1") for l in range(NL)], np.uint32) + s[1]
        elif op == "v_mbcnt_hi_u32_b32":
            mask = int(s[0][0])
            out = np.array([bin(mask & ((1 << max(l - 32, 0)) - 1)).count("1") for l in range(NL)], np.uint32) + s[1]
        else:
            raise SimError(f"VALU op {op} not modelled")
        self._wv(w, d, out.astype(np.uint32))

    def _salu(self, w, ins):
        op, a = ins.op, ins.args
        if op == "s_setreg":             # (only MODE.FP16_OVFL is modelled: overflowing fp16 conversions clamp instead of giving infinity)
            w.fp16_ovfl = bool(a["s"][0])
            return
        if op == "s_and_saveexec_b64":
            bits = 0
            for l in range(NL):
                if w.exec[l]:
                    bits |= 1 << l
            w.s[a["d"].i], w.s[a["d"].i + 1] = bits & 0xFFFFFFFF, bits >> 32
            vcc = int(w.s[VCC.i]) | (int(w.s[VCC.i + 1]) << 32)
            w.exec = np.array([(bits >> l) & (vcc >> l) & 1 for l in range(NL)], bool)
            w.scc = int(w.exec.any())
            return
        if op == "s_mov_exec":
            if isinstance(a["s"], Reg):
                bits = int(w.s[a["s"].i]) | (int(w.s[a["s"].i + 1]) << 32)
            else:
                bits = (1 << 64) - 1 if a["s"] == -1 else int(a["s"])
            w.exec = np.array([(bits >> l) & 1 for l in range(NL)], bool)
            return
        if op in ("s_cmp_lt_u32", "s_cmp_gt_u32", "s_cmp_eq_u32", "s_cmp_lg_u32", "s_cmp_ge_u32", "s_cmp_le_u32"):
            x, y = self._sval(w, a["a"]), self._sval(w, a["b"])
            w.scc = int({"s_cmp_lt_u32": x < y, "s_cmp_gt_u32": x > y, "s_cmp_eq_u32": x == y, "s_cmp_lg_u32": x != y,
                         "s_cmp_ge_u32": x >= y, "s_cmp_le_u32": x <= y}[op])
            return
        if op == "s_bitcmp1_b32":
            w.scc = (self._sval(w, a["a"]) >> (self._sval(w, a["b"]) & 31)) & 1
            return
        d, s = a["d"], [self._sval(w, x) for x in a["s"]]
        if op == "s_mov_b32":
            r = s[0]
        elif op == "s_mov_b64":
            src = a["s"][0]
            if isinstance(src, Reg):
                w.s[d.i], w.s[d.i + 1] = w.s[src.i], w.s[src.i + 1]
            else:
                w.s[d.i], w.s[d.i + 1] = src & 0xFFFFFFFF, (src >> 32) & 0xFFFFFFFF
            return
        elif op == "s_add_u32":
            t = s[0] + s[1]
            w.scc = int(t >> 32)
            r = t & 0xFFFFFFFF
        elif op == "s_addc_u32":
            t = s[0] + s[1] + w.scc
            w.scc = int(t >> 32)
            r = t & 0xFFFFFFFF
        elif op == "s_sub_u32":
            r = (s[0] - s[1]) & 0xFFFFFFFF
            w.scc = int(s[1] > s[0])
        elif op == "s_mul_i32":
            r = (s[0] * s[1]) & 0xFFFFFFFF
        elif op == "s_mul_hi_u32":
            r = ((s[0] * s[1]) >> 32) & 0xFFFFFFFF
        elif op == "s_subb_u32":
            t = s[0] - s[1] - w.scc
            w.scc = int(t < 0)
            r = t & 0xFFFFFFFF
        elif op == "s_lshl_b32":
            r = (s[0] << (s[1] & 31)) & 0xFFFFFFFF
            w.scc = int(r != 0)
        elif op == "s_lshr_b32":
            r = s[0] >> (s[1] & 31)
            w.scc = int(r != 0)
        elif op == "s_and_b32":
            r = s[0] & s[1]
            w.scc = int(r != 0)
        elif op == "s_or_b32":
            r = s[0] | s[1]
            w.scc = int(r != 0)
        elif op == "s_cselect_b32":
            r = s[0] if w.scc else s[1]
        elif op == "s_min_u32":
            r = min(s[0], s[1])
            w.scc = int(s[0] <= s[1])
        else:
            raise SimError(f"SALU op {op} not modelled")
        w.s[d.i] = r

    def run(self, max_inst=50_000_000):
        """Run the four waves to completion (program end = falling off the list)."""
        n = len(self.prog)
        at_barrier = [False] * 4
        while True:
            progress = False
            for w in self.waves:
                if w.done or at_barrier[w.id]:
                    continue
                while True:
                    if w.pc >= n:
                        w.done = True
                        # registers a load is still outstanding into when the program ends (the code behind an asm statement
                        # may use them): {(file, index): queue}
                        self.pending_at_end = getattr(self, "pending_at_end", {})
                        self.pending_at_end[w.id] = dict(w.pending)
                        break
                    r = self.step(w)
                    progress = True
                    if r == "barrier":
                        at_barrier[w.id] = True
                        break
                    if w.n_inst > max_inst:
                        raise SimError("instruction budget exceeded (endless loop?)")
            if all(w.done for w in self.waves):
                return
            live = [w for w in self.waves if not w.done]
            if all(at_barrier[w.id] for w in live):
                if len(live) != 4:
                    raise SimError("some waves finished while others wait at a barrier")
                for w in self.waves:
                    at_barrier[w.id] = False
                    w.epoch += 1
                progress = True
            if not progress:
                raise SimError("deadlock")
