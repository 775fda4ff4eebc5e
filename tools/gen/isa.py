"""A small gfx950 assembly DSL with a numpy emulator, for generating the steady-state loop of olasave_kernel
(tools/gen/gen_olasave.py) with hand-allocated registers.  Every instruction is built by a function that records both its
assembly text and its operands; `Emu` executes a program for one 256-thread workgroup (4 waves) instruction by instruction
in lockstep -- a legal interleaving of a race-free program -- and checks the s_waitcnt discipline: a register that is the
destination of a load / ds_read still in flight must not be read or written, and a store's data register must not be
overwritten before the store has left the queue position the ISA guarantees (we are conservative: until it is waited for).

Subset: v_pk_{add,mul,fma}_f32 (op_sel / op_sel_hi / neg_lo / neg_hi), v_mov_b32 (+ DPP wave_shr:1), v_readlane_b32,
integer VALU for address set-up, ds_{read,write}_b64, global_{load,store}_dword[x2] (saddr + voffset + imm), SALU moves /
adds / compares / branches, s_waitcnt, s_barrier.
"""
import numpy as np


class V:  # one VGPR
    def __init__(self, n): self.n = n
    def __repr__(self): return "v%d" % self.n


class VP:  # aligned VGPR pair
    def __init__(self, n):
        assert n % 2 == 0, "64-bit VGPR operands are even-aligned on gfx950"
        self.n = n
    def __repr__(self): return "v[%d:%d]" % (self.n, self.n + 1)
    @property
    def lo(self): return V(self.n)
    @property
    def hi(self): return V(self.n + 1)


class S:
    def __init__(self, n): self.n = n
    def __repr__(self): return "s%d" % self.n


class SP:
    def __init__(self, n):
        assert n % 2 == 0
        self.n = n
    def __repr__(self): return "s[%d:%d]" % (self.n, self.n + 1)
    @property
    def lo(self): return S(self.n)
    @property
    def hi(self): return S(self.n + 1)


class _VccHalf:
    def __init__(self, name, n): self.name, self.n = name, n
    def __repr__(self): return self.name


class VccPair:  # VCC as a lane-mask register pair (s106:107 on gfx9)
    n = 106
    def __repr__(self): return "vcc"
    lo = _VccHalf("vcc_lo", 106)
    hi = _VccHalf("vcc_hi", 107)


VCC = VccPair()


class Lit:  # 32-bit literal / inline constant (float or int)
    def __init__(self, v): self.v = v
    def __repr__(self):
        if isinstance(self.v, float):
            if self.v in (0.5, -0.5, 1.0, -1.0, 2.0, -2.0, 4.0, -4.0):   # inline constants (no literal slot taken)
                return repr(self.v)
            return "0x%08x" % np.float32(self.v).view(np.uint32)
        return str(self.v)
    def bits(self):
        if isinstance(self.v, float):
            return np.uint32(np.float32(self.v).view(np.uint32))
        return np.uint32(self.v & 0xffffffff)


class Ins:
    def __init__(self, op, text, **kw):
        self.op, self.text = op, text
        self.__dict__.update(kw)


def _mods(op_sel=None, op_sel_hi=None, neg_lo=None, neg_hi=None, n=2):
    s = ""
    if op_sel is not None and any(op_sel): s += " op_sel:[%s]" % ",".join(str(x) for x in op_sel)
    if op_sel_hi is not None and not all(op_sel_hi): s += " op_sel_hi:[%s]" % ",".join(str(x) for x in op_sel_hi)
    if neg_lo is not None and any(neg_lo): s += " neg_lo:[%s]" % ",".join(str(x) for x in neg_lo)
    if neg_hi is not None and any(neg_hi): s += " neg_hi:[%s]" % ",".join(str(x) for x in neg_hi)
    return s


class Prog:
    """Instruction list with AUTOMATIC s_waitcnt insertion: vector-memory and LDS operations return in order within their
    counter, so before an instruction touches a register that an operation still in flight will write (or, for loads'
    destinations, that it reads), the minimal `s_waitcnt vmcnt(n)` / `lgkmcnt(n)` is emitted: n = operations issued after
    the youngest one needed.  At a loop label the caller states what may be outstanding (reset_queues)."""
    def __init__(self):
        self.ins = []
        self.labels = {}
        self.vmq, self.lgq = [], []   # in flight, oldest first: (set of VGPR numbers the op will WRITE)
        self.auto = True

    def reset_queues(self, vm_stores=0, lg=0):
        self.vmq = [set() for _ in range(vm_stores)]
        self.lgq = [set() for _ in range(lg)]

    @staticmethod
    def _regs(x):
        if isinstance(x, V): return {x.n}
        if isinstance(x, VP): return {x.n, x.n + 1}
        return set()

    def _need(self, regs):
        """emit waits so that no op in flight writes any of `regs`"""
        for q, name in ((self.vmq, "vmcnt"), (self.lgq, "lgkmcnt")):
            last = -1
            for idx, w in enumerate(q):
                if w & regs:
                    last = idx
            if last >= 0:
                left = len(q) - 1 - last
                lim = 63 if name == "vmcnt" else 15
                left = min(left, lim)
                self.ins.append(Ins("s_waitcnt", "s_waitcnt %s(%d)" % (name, left), vmcnt=left if name == "vmcnt" else None,
                                    lgkmcnt=left if name == "lgkmcnt" else None))
                del q[:len(q) - left]

    def add(self, i):
        if self.auto and i.op not in ("comment", "label", "s_waitcnt", "probe"):
            regs = set()
            for k in ("d", "a", "b", "sh", "addr", "data", "voff"):
                if hasattr(i, k):
                    regs |= self._regs(getattr(i, k))
            for x in getattr(i, "srcs", []):
                regs |= self._regs(x)
            self._need(regs)
            if i.op in ("gload2", "gload1"):
                self.vmq.append(self._regs(i.d))
            elif i.op == "gstore2":
                self.vmq.append(set())
            elif i.op in ("ds_read_b64", "ds_read_b32"):
                if len(self.lgq) >= 15:   # the counter has 4 bits: the 16th operation in flight would stall the wave anyway
                    pass
                self.lgq.append(self._regs(i.d))
            elif i.op in ("ds_write_b64", "ds_write_b32"):
                self.lgq.append(set())
            # (s_atomic_add is NOT counted: scalar memory returns out of order with LDS, so a count that leaned on its position
            # could be satisfied with the LDS operation in front of it still in flight; a wait computed from the LDS operations
            # alone is at worst stricter than needed while the atomic is out.  Its result is waited for with an explicit lgkmcnt(0).)
        elif i.op == "s_waitcnt":
            if i.vmcnt is not None: del self.vmq[:max(0, len(self.vmq) - i.vmcnt)]
            if i.lgkmcnt is not None: del self.lgq[:max(0, len(self.lgq) - i.lgkmcnt)]
        self.ins.append(i)
        return i

    # ---- packed fp32 ----
    def pk(self, op, d, srcs, op_sel=None, op_sel_hi=None, neg_lo=None, neg_hi=None):
        n = len(srcs)
        op_sel = list(op_sel) if op_sel is not None else [0] * n
        op_sel_hi = list(op_sel_hi) if op_sel_hi is not None else [1] * n
        neg_lo = list(neg_lo) if neg_lo is not None else [0] * n
        neg_hi = list(neg_hi) if neg_hi is not None else [0] * n
        assert sum(isinstance(x, SP) for x in srcs) <= 1 or len(set(x.n for x in srcs if isinstance(x, SP))) == 1, "one SGPR source per VOP3P"
        t = "v_pk_%s_f32 %r, %s%s" % (op, d, ", ".join(repr(x) for x in srcs), _mods(op_sel, op_sel_hi, neg_lo, neg_hi))
        return self.add(Ins("pk_" + op, t, d=d, srcs=srcs, op_sel=op_sel, op_sel_hi=op_sel_hi, neg_lo=neg_lo, neg_hi=neg_hi))

    # ---- 32-bit VALU ----
    def v_mov(self, d, a): return self.add(Ins("v_mov", "v_mov_b32 %r, %r" % (d, a), d=d, a=a))
    def v_mov_dpp_shr1(self, d, a):
        return self.add(Ins("v_mov_dpp_shr1", "v_mov_b32_dpp %r, %r wave_shr:1 row_mask:0xf bank_mask:0xf" % (d, a), d=d, a=a))
    def v_mov_dpp_rowshl(self, d, a, k):
        """lanes 0..3 of every wave: d[lane] = a[lane + k] (row 0, bank 0 only; the other lanes keep d)"""
        assert 1 <= k <= 15
        return self.add(Ins("v_mov_dpp_rowshl", "v_mov_b32_dpp %r, %r row_shl:%d row_mask:0x1 bank_mask:0x1" % (d, a, k), d=d, a=a, k=k))
    def v_mul_f32(self, d, a, b): return self.add(Ins("v_mul_f32", "v_mul_f32 %r, %r, %r" % (d, a, b), d=d, a=a, b=b))
    def v_readfirstlane(self, sd, v): return self.add(Ins("v_readfirstlane", "v_readfirstlane_b32 %r, %r" % (sd, v), d=sd, a=v))
    def v_readlane(self, sd, v, lane): return self.add(Ins("v_readlane", "v_readlane_b32 %r, %r, %d" % (sd, v, lane), d=sd, a=v, lane=lane))
    def v_and(self, d, a, b): return self.add(Ins("v_and", "v_and_b32 %r, %r, %r" % (d, a, b), d=d, a=a, b=b))
    def v_lshl(self, d, sh, a): return self.add(Ins("v_lshl", "v_lshlrev_b32 %r, %r, %r" % (d, sh, a), d=d, a=a, sh=sh))
    def v_lshr(self, d, sh, a): return self.add(Ins("v_lshr", "v_lshrrev_b32 %r, %r, %r" % (d, sh, a), d=d, a=a, sh=sh))
    def v_add_u32(self, d, a, b): return self.add(Ins("v_add_u32", "v_add_u32 %r, %r, %r" % (d, a, b), d=d, a=a, b=b))
    def v_mul_u24(self, d, a, b): return self.add(Ins("v_mul_u24", "v_mul_u32_u24 %r, %r, %r" % (d, a, b), d=d, a=a, b=b))
    def v_cmp_eq_u32(self, a, b): return self.add(Ins("v_cmp_eq", "v_cmp_eq_u32 vcc, %r, %r" % (a, b), a=a, b=b))      # (a: constant or SGPR, b: VGPR)
    def v_cmp_lt_u32(self, a, b): return self.add(Ins("v_cmp_lt", "v_cmp_lt_u32 vcc, %r, %r" % (a, b), a=a, b=b))      # vcc = a < b
    def v_cmp_lt_f32(self, m, a, b):   # lane mask in the SGPR pair m = a < b (floats; VOP3 form: any SGPR pair, not just VCC)
        return self.add(Ins("v_cmp_lt_f32", "v_cmp_lt_f32_e64 %r, %r, %r" % (m, a, b), m=m, a=a, b=b))
    def v_mbcnt(self, d, m):
        """d = number of set bits of the lane mask m (an SGPR pair) below this lane"""
        self.add(Ins("v_mbcnt_lo", "v_mbcnt_lo_u32_b32 %r, %r, 0" % (d, m.lo), d=d, m=m))
        return self.add(Ins("v_mbcnt_hi", "v_mbcnt_hi_u32_b32 %r, %r, %r" % (d, m.hi, d), d=d, a=d, m=m))
    def v_max3_abs(self, d, a, b):   # d = max(|a|, |b|, d)
        return self.add(Ins("v_max3_abs", "v_max3_f32 %r, |%r|, |%r|, %r" % (d, a, b, d), d=d, a=a, b=b))
    def s_bcnt1(self, d, m): return self.add(Ins("s_bcnt1", "s_bcnt1_i32_b64 %r, %r" % (d, m), d=d, m=m))
    def s_lshl_b32(self, d, a, sh): return self.add(Ins("s_lshl", "s_lshl_b32 %r, %r, %r" % (d, a, sh), d=d, a=a, sh=sh))
    def exec_from(self, m): return self.add(Ins("exec_from", "s_mov_b64 exec, %r" % m, m=m))
    def exec_all(self): return self.add(Ins("exec_all", "s_mov_b64 exec, -1"))
    def exec_lane0(self): return self.add(Ins("exec_lane0", "s_mov_b64 exec, 1"))
    def v_cndmask(self, d, a, b): return self.add(Ins("v_cndmask", "v_cndmask_b32 %r, %r, %r, vcc" % (d, a, b), d=d, a=a, b=b))  # d = vcc ? b : a

    # ---- LDS ----
    def ds_write_b64(self, addr, data, off=0):
        assert 0 <= off < 65536
        return self.add(Ins("ds_write_b64", "ds_write_b64 %r, %r%s" % (addr, data, " offset:%d" % off if off else ""), addr=addr, data=data, off=off))
    def ds_read_b32(self, d, addr, off=0):
        assert 0 <= off < 65536
        return self.add(Ins("ds_read_b32", "ds_read_b32 %r, %r%s" % (d, addr, " offset:%d" % off if off else ""), d=d, addr=addr, off=off))
    def ds_read_b64(self, d, addr, off=0):
        assert 0 <= off < 65536
        return self.add(Ins("ds_read_b64", "ds_read_b64 %r, %r%s" % (d, addr, " offset:%d" % off if off else ""), d=d, addr=addr, off=off))

    # ---- global ----
    def _goff(self, off):
        assert -4096 <= off <= 4095, off
        return " offset:%d" % off if off else ""
    def global_load_dwordx2(self, d, voff, sbase, off=0, nt=False):
        return self.add(Ins("gload2", "global_load_dwordx2 %r, %r, %r%s%s" % (d, voff, sbase, self._goff(off), " nt" if nt else ""), d=d, voff=voff, sbase=sbase, off=off))
    def global_load_dword(self, d, voff, sbase, off=0, nt=False):
        return self.add(Ins("gload1", "global_load_dword %r, %r, %r%s%s" % (d, voff, sbase, self._goff(off), " nt" if nt else ""), d=d, voff=voff, sbase=sbase, off=off))
    def global_store_dwordx2(self, voff, data, sbase, off=0, nt=False):
        return self.add(Ins("gstore2", "global_store_dwordx2 %r, %r, %r%s%s" % (voff, data, sbase, self._goff(off), " nt" if nt else ""), voff=voff, data=data, sbase=sbase, off=off))

    # ---- SALU / control ----
    def s_mov(self, d, a): return self.add(Ins("s_mov", "s_mov_b32 %r, %r" % (d, a), d=d, a=a))
    def s_mov64(self, d, a): return self.add(Ins("s_mov64", "s_mov_b64 %r, %r" % (d, a), d=d, a=a))
    def s_add_u32(self, d, a, b): return self.add(Ins("s_add_u32", "s_add_u32 %r, %r, %r" % (d, a, b), d=d, a=a, b=b))
    def s_addc_u32(self, d, a, b): return self.add(Ins("s_addc_u32", "s_addc_u32 %r, %r, %r" % (d, a, b), d=d, a=a, b=b))
    def s_sub_u32(self, d, a, b): return self.add(Ins("s_sub_u32", "s_sub_u32 %r, %r, %r" % (d, a, b), d=d, a=a, b=b))
    def s_subb_u32(self, d, a, b): return self.add(Ins("s_subb_u32", "s_subb_u32 %r, %r, %r" % (d, a, b), d=d, a=a, b=b))
    def s_cselect(self, d, a, b): return self.add(Ins("s_cselect", "s_cselect_b32 %r, %r, %r" % (d, a, b), d=d, a=a, b=b))
    def s_cmp_lg_u32(self, a, b): return self.add(Ins("s_cmp_lg", "s_cmp_lg_u32 %r, %r" % (a, b), a=a, b=b))
    def s_cmp_eq_u32(self, a, b): return self.add(Ins("s_cmp_eq", "s_cmp_eq_u32 %r, %r" % (a, b), a=a, b=b))
    def s_cmp_lt_u32(self, a, b): return self.add(Ins("s_cmp_lt", "s_cmp_lt_u32 %r, %r" % (a, b), a=a, b=b))
    def s_cmp_ge_u32(self, a, b): return self.add(Ins("s_cmp_ge", "s_cmp_ge_u32 %r, %r" % (a, b), a=a, b=b))
    def s_mul_i32(self, d, a, b): return self.add(Ins("s_mul_i32", "s_mul_i32 %r, %r, %r" % (d, a, b), d=d, a=a, b=b))
    def s_mul_hi_u32(self, d, a, b): return self.add(Ins("s_mul_hi_u32", "s_mul_hi_u32 %r, %r, %r" % (d, a, b), d=d, a=a, b=b))
    def s_mul_hi_i32(self, d, a, b): return self.add(Ins("s_mul_hi_i32", "s_mul_hi_i32 %r, %r, %r" % (d, a, b), d=d, a=a, b=b))
    def s_and_b32(self, d, a, b): return self.add(Ins("s_and_b32", "s_and_b32 %r, %r, %r" % (d, a, b), d=d, a=a, b=b))
    def s_min_u32(self, d, a, b): return self.add(Ins("s_min_u32", "s_min_u32 %r, %r, %r" % (d, a, b), d=d, a=a, b=b))
    def s_atomic_add(self, d, sbase):
        """d (holding the addend) <- the 32-bit word at sbase before the add (glc: a wave-uniform fetch-and-add through the scalar
        cache; returns through lgkmcnt, OUT of order with LDS operations: only lgkmcnt(0) says it has arrived)"""
        return self.add(Ins("s_atomic_add", "s_atomic_add %r, %r, 0x0 glc" % (d, sbase), d=d, sbase=sbase))
    def ds_write_b32(self, addr, data, off=0):
        assert 0 <= off < 65536
        return self.add(Ins("ds_write_b32", "ds_write_b32 %r, %r%s" % (addr, data, " offset:%d" % off if off else ""), addr=addr, data=data, off=off))
    def s_cbranch_scc1(self, label): return self.add(Ins("s_cbranch_scc1", "s_cbranch_scc1 %s" % label, label=label))
    def s_cbranch_scc0(self, label): return self.add(Ins("s_cbranch_scc0", "s_cbranch_scc0 %s" % label, label=label))
    def s_branch(self, label): return self.add(Ins("s_branch", "s_branch %s" % label, label=label))
    def label(self, name):
        self.labels[name] = len(self.ins)
        return self.add(Ins("label", "%s:" % name, name=name))
    def s_waitcnt(self, vmcnt=None, lgkmcnt=None):
        parts = []
        if vmcnt is not None:
            assert 0 <= vmcnt <= 63
            parts.append("vmcnt(%d)" % vmcnt)
        if lgkmcnt is not None:
            assert 0 <= lgkmcnt <= 15
            parts.append("lgkmcnt(%d)" % lgkmcnt)
        return self.add(Ins("s_waitcnt", "s_waitcnt " + " ".join(parts), vmcnt=vmcnt, lgkmcnt=lgkmcnt))
    def s_wait_lgkm0_here(self):
        """an explicit `s_waitcnt lgkmcnt(0)` inside a region only some waves execute (a scalar atomic's result): seen by the emulator,
        not by the generator's own wait bookkeeping (which describes every wave) nor by assert_no_waits_since"""
        return self.add(Ins("s_waitcnt_x", "s_waitcnt lgkmcnt(0)", vmcnt=None, lgkmcnt=0))
    def s_barrier(self): return self.add(Ins("s_barrier", "s_barrier"))
    def s_nop(self, n=0): return self.add(Ins("s_nop", "s_nop %d" % n))
    def s_setprio(self, n): return self.add(Ins("s_nop", "s_setprio %d" % n))
    def comment(self, txt): return self.add(Ins("comment", "; " + txt))
    def probe(self, name, regs):   # emulator only: snapshot of registers (debugging a generator)
        return self.add(Ins("probe", "; probe " + name, name=name, regs=list(regs)))
    def raw(self, txt):  # text the emulator ignores (input copies in the C++ wrapper's dialect)
        return self.add(Ins("comment", txt))

    # ---- wave-uniform branches: the wait bookkeeping of two alternative instruction sequences
    def settle(self, regs):
        """nothing in flight may still write these registers from here on (call in front of a region some waves skip: a wait
        inserted INSIDE it would be skipped with it)"""
        r = set()
        for x in regs:
            r |= self._regs(x)
        self._need(r)

    def fork(self):
        return ([set(x) for x in self.vmq], [set(x) for x in self.lgq], sum(1 for i in self.ins if i.op == "s_waitcnt"))

    def restore(self, st):
        self.vmq, self.lgq = [set(x) for x in st[0]], [set(x) for x in st[1]]

    def join(self, other):
        """merge the bookkeeping of the alternative that was emitted last with `other` (the state at the end of the one before):
        both must have put the same number of operations in flight; a register either of them loads counts as loading"""
        assert len(self.vmq) == len(other[0]) and len(self.lgq) == len(other[1]), "alternatives differ in what they leave in flight"
        self.vmq = [a | b for a, b in zip(self.vmq, other[0])]
        self.lgq = [a | b for a, b in zip(self.lgq, other[1])]

    def assert_no_waits_since(self, st):
        assert sum(1 for i in self.ins if i.op == "s_waitcnt") == st[2], "a wait was inserted inside a region that some waves skip"

    def text(self, label_suffix=""):
        out = []
        for i in self.ins:
            t = i.text
            if i.op == "probe":
                continue
            if i.op == "label":
                t = "%s%s:" % (i.name, label_suffix)
            elif hasattr(i, "label"):
                t = t.replace(i.label, i.label + label_suffix)
            out.append(t)
        return out

    def stats(self):
        c = {}
        for i in self.ins:
            c[i.op] = c.get(i.op, 0) + 1
        return c


# ------------------------------------------------------------------------------------------------ emulator
class _Wave:
    def __init__(self):
        self.v = np.zeros((256, 64), np.uint32)
        self.s = np.zeros(128, np.uint32)
        self.scc = 0
        self.vcc = np.zeros(64, bool)
        self.exec = np.ones(64, bool)
        self.masks = {}                     # lane masks kept in SGPR pairs (v_cmp ... s[n:n+1]): first SGPR number -> bool[64]
        self.mask_wr = {}                   # ... and how many instructions ago a VALU wrote them
        self.pc = 0
        self.vm_q, self.lgkm_q = [], []     # in-flight ops: (kind, regs)
        self.recent = []                    # VGPRs written by the last two VALU instructions (DPP hazard: 2 wait states)
        self.sgpr_wr = 99                   # instructions since a VALU wrote an SGPR (v_readlane / v_readfirstlane)
        self.done = False
        self.smem_regs = set()              # SGPRs a scalar atomic in flight will write


class Emu:
    """One workgroup of NT threads (NT / 64 waves).  Every wave has its own program counter, registers, SCC and wait
    queues; the waves run ONE AFTER THE OTHER from barrier to barrier (a legal interleaving of a race-free program), so
    wave-uniform branches may diverge between waves.  That the program IS race-free in LDS is checked: a slot written by
    one wave and touched by another between the same two barriers is an error (the exchanges that skip the workgroup
    barrier must stay inside one wave).  The old interface (emu.v[reg][thread], emu.s[wave, reg]) is kept as views."""
    def __init__(self, lds_bytes=65536, check_waits=True, n_threads=256):
        assert n_threads % 64 == 0
        self.NT = n_threads
        self.NW = n_threads // 64
        self.waves = [_Wave() for _ in range(self.NW)]
        self.lds = np.zeros(lds_bytes // 4, np.uint32)
        self.mem = {}                                   # base address -> (np.uint32 array)
        self.check = check_waits
        self.epoch = 0
        self.probes = {}
        self.lds_w_epoch = np.full(lds_bytes // 4, -1, np.int64)   # last write: barrier epoch, wave
        self.lds_w_wave = np.zeros(lds_bytes // 4, np.int64)
        self.lds_r_epoch = np.full(lds_bytes // 4, -1, np.int64)   # reads in the current epoch: epoch, bit mask of waves
        self.lds_r_mask = np.zeros(lds_bytes // 4, np.int64)
        self.v = _VView(self)
        self.s = _SView(self)

    # memory regions
    def map(self, base, arr_u32):
        self.mem[base] = arr_u32

    def _region(self, addr):
        for b, a in self.mem.items():
            if b <= addr < b + a.size * 4:
                return b, a
        raise RuntimeError("unmapped address 0x%x" % addr)

    # LDS race discipline
    def _lds_write(self, w, words, what):
        E = self.epoch
        other_w = (self.lds_w_epoch[words] == E) & (self.lds_w_wave[words] != w)
        other_r = (self.lds_r_epoch[words] == E) & ((self.lds_r_mask[words] & ~(1 << w)) != 0)
        if other_w.any() or other_r.any():
            raise RuntimeError("LDS race: wave %d %s a slot another wave %s since the last barrier" % (w, what, "wrote" if other_w.any() else "read"))
        self.lds_w_epoch[words] = E
        self.lds_w_wave[words] = w

    def _lds_read(self, w, words, what):
        E = self.epoch
        other_w = (self.lds_w_epoch[words] == E) & (self.lds_w_wave[words] != w)
        if other_w.any():
            raise RuntimeError("LDS race: wave %d %s a slot another wave wrote since the last barrier" % (w, what))
        stale = self.lds_r_epoch[words] != E
        self.lds_r_mask[words] = np.where(stale, 0, self.lds_r_mask[words]) | (1 << w)
        self.lds_r_epoch[words] = E

    @staticmethod
    def _regs(x):
        if isinstance(x, V): return [x.n]
        if isinstance(x, VP): return [x.n, x.n + 1]
        return []

    def run(self, prog, max_steps=20_000_000):
        steps = 0
        for wv in self.waves:
            wv.pc, wv.done = 0, False
        while not all(wv.done for wv in self.waves):
            at_barrier = 0
            for w, wv in enumerate(self.waves):
                if wv.done:
                    continue
                n, hit = self._run_wave(prog, w, wv, max_steps - steps)
                steps += n
                at_barrier += hit
            live = sum(not wv.done for wv in self.waves)
            if at_barrier not in (0, self.NW) and live != at_barrier:
                raise RuntimeError("some waves ended while others wait at a barrier")
            if at_barrier and at_barrier != self.NW:
                raise RuntimeError("barrier reached by %d of %d waves" % (at_barrier, self.NW))
            self.epoch += 1
        return steps

    def _run_wave(self, prog, w, W_, budget):
        """runs wave w until its next s_barrier (returns (steps, 1)) or the end of the program (steps, 0)"""
        ins = prog.ins
        f32 = lambda u: u.view(np.float32)
        v, s = W_.v, W_.s
        steps = 0

        def busy_regs():
            busy = {}
            for kind, regs in W_.vm_q + W_.lgkm_q:
                if kind in ("store", "dswrite"):   # data is read when the instruction issues
                    continue
                for r in regs:
                    busy[r] = kind
            return busy

        def touch(reads, writes, what):
            if not self.check:
                return
            busy = busy_regs()
            for r in reads:
                if r in busy and busy[r] in ("load", "dsread"):
                    raise RuntimeError("%s reads v%d while a %s into it is in flight" % (what, r, busy[r]))
            for r in writes:
                if r in busy:
                    raise RuntimeError("%s writes v%d while a %s on it is in flight" % (what, r, busy[r]))

        def rd32(x):
            if isinstance(x, V): return v[x.n].copy()
            if isinstance(x, S): return np.full(64, s[x.n], np.uint32)
            if isinstance(x, Lit): return np.full(64, x.bits(), np.uint32)
            if isinstance(x, (int, np.integer)): return np.full(64, np.uint32(x & 0xffffffff), np.uint32)
            raise TypeError(x)

        def rd64(x):
            if isinstance(x, VP): return v[x.n].copy(), v[x.n + 1].copy()
            if isinstance(x, SP): return np.full(64, s[x.n], np.uint32), np.full(64, s[x.n + 1], np.uint32)
            raise TypeError(x)

        def srd(x):
            if isinstance(x, S) and x.n in W_.smem_regs:
                raise RuntimeError("s%d is read while the scalar atomic that writes it is in flight (no lgkmcnt(0) since)" % x.n)
            if isinstance(x, S): return int(s[x.n])
            if isinstance(x, Lit): return int(x.bits())
            if isinstance(x, (int, np.integer)): return int(x) & 0xffffffff
            raise TypeError(x)

        regs_of = self._regs
        while W_.pc < len(ins):
            i = ins[W_.pc]
            W_.pc += 1
            steps += 1
            if steps > budget:
                raise RuntimeError("emulation did not terminate")
            op = i.op
            if op in ("comment", "label"):
                continue
            if op == "probe":
                snap = self.probes.setdefault(i.name, {})
                if w not in snap:   # (first visit)
                    snap[w] = {r.n: (v[r.n].copy(), v[r.n + 1].copy()) if isinstance(r, VP) else v[r.n].copy() for r in i.regs}
                continue
            if op == "s_barrier":
                return steps, 1
            if op == "s_nop":
                W_.recent = (W_.recent + [set()])[-2:]
                W_.sgpr_wr += 5
                continue
            # measured on gfx950: a VALU instruction reading an SGPR directly behind a VALU write of an SGPR reads garbage
            if op.startswith("pk_") or (op.startswith("v_") and op not in ("v_readlane", "v_readfirstlane")):
                srcs = list(getattr(i, "srcs", [])) + [getattr(i, k) for k in ("a", "b", "sh") if hasattr(i, k)]
                if W_.sgpr_wr < 3 and any(isinstance(x, (S, SP)) for x in srcs):
                    raise RuntimeError("SGPR hazard: %s reads an SGPR %d instruction(s) after a VALU wrote one" % (i.text, W_.sgpr_wr + 1))
            W_.sgpr_wr = 0 if op in ("v_readlane", "v_readfirstlane") else W_.sgpr_wr + 1
            for k_ in list(W_.mask_wr):
                W_.mask_wr[k_] += 1
            if op in ("v_mbcnt_lo", "v_mbcnt_hi", "s_bcnt1", "exec_from") and W_.mask_wr.get(i.m.n, 99) < 5 + (op == "v_mbcnt_hi"):
                # (round 4, found on the hardware as one frame of a survivor list lost now and then: the same family as the
                # v_readfirstlane hazard below -- an SGPR written by a VALU instruction read back too early is stale)
                raise RuntimeError("lane-mask hazard: %s reads a mask %d instruction(s) after the compare that wrote it" % (i.text, W_.mask_wr[i.m.n]))
            if op in ("v_readlane", "v_readfirstlane"):
                # measured on gfx950 (round 4: waves 4-6 of a workgroup read a stale register, wave 7 the new value): a lane read
                # directly behind the VALU write of its source sees the old contents -- same two wait states as DPP
                for wr in W_.recent:
                    if wr & set(regs_of(i.a)):
                        raise RuntimeError("lane-read hazard: %s within 2 wait states of a VALU write of its source" % i.text)
            if op in ("v_mov_dpp_shr1", "v_mov_dpp_rowshl"):
                used = set(regs_of(i.a) + regs_of(i.d))
                for wr in W_.recent:
                    if wr & used:
                        raise RuntimeError("DPP hazard: %s within 2 wait states of a VALU write of its register" % i.text)
            if op.startswith("pk_") or op.startswith("v_"):
                W_.recent = (W_.recent + [set(regs_of(i.d)) if op not in ("v_readlane", "v_readfirstlane", "v_cmp_eq", "v_cmp_lt", "v_cmp_lt_f32") else set()])[-2:]
            else:
                W_.recent = (W_.recent + [set()])[-2:]
            if op.startswith("pk_"):
                reads = sum((regs_of(x) for x in i.srcs), [])
                touch(reads, regs_of(i.d), i.text)
                vals = [rd64(x) for x in i.srcs]
                def pick(k, sel, neg):
                    lo, hi = vals[k]
                    x = f32(hi if sel else lo).astype(np.float64)
                    return -x if neg else x
                res = []
                for half, (sels, negs) in enumerate(((i.op_sel, i.neg_lo), (i.op_sel_hi, i.neg_hi))):
                    a = pick(0, sels[0], negs[0])
                    b = pick(1, sels[1], negs[1])
                    if op == "pk_add": r = (a + b)
                    elif op == "pk_mul": r = (a * b)
                    elif op == "pk_fma": r = (a * b + pick(2, sels[2], negs[2]))
                    else: raise RuntimeError(op)
                    res.append(r.astype(np.float32).view(np.uint32))
                v[i.d.n], v[i.d.n + 1] = res[0], res[1]
            elif op == "v_mov":
                touch(regs_of(i.a), regs_of(i.d), i.text)
                v[i.d.n] = rd32(i.a)
            elif op == "v_mov_dpp_shr1":
                touch(regs_of(i.a) + regs_of(i.d), regs_of(i.d), i.text)
                src = v[i.a.n].copy()
                v[i.d.n][1:] = src[:-1]
            elif op == "v_mov_dpp_rowshl":
                touch(regs_of(i.a) + regs_of(i.d), regs_of(i.d), i.text)
                src = v[i.a.n].copy()
                for lane in range(4):
                    if lane + i.k <= 15:
                        v[i.d.n][lane] = src[lane + i.k]
            elif op == "v_mul_f32":
                touch(regs_of(i.a) + regs_of(i.b), regs_of(i.d), i.text)
                a, b = rd32(i.a).view(np.float32), rd32(i.b).view(np.float32)
                v[i.d.n] = (a * b).astype(np.float32).view(np.uint32)
            elif op == "v_readfirstlane":
                touch(regs_of(i.a), [], i.text)
                s[i.d.n] = v[i.a.n][0]
            elif op == "v_readlane":
                touch(regs_of(i.a), [], i.text)
                s[i.d.n] = v[i.a.n][i.lane]
            elif op in ("v_and", "v_add_u32", "v_mul_u24"):
                touch(regs_of(i.a) + regs_of(i.b), regs_of(i.d), i.text)
                a, b = rd32(i.a).astype(np.uint64), rd32(i.b).astype(np.uint64)
                r = (a & b) if op == "v_and" else (a + b) if op == "v_add_u32" else ((a & 0xffffff) * (b & 0xffffff))
                v[i.d.n] = (r & 0xffffffff).astype(np.uint32)
            elif op in ("v_cmp_eq", "v_cmp_lt"):
                touch(regs_of(i.a) + regs_of(i.b), [], i.text)
                a, b = rd32(i.a), rd32(i.b)
                W_.vcc = (a == b) if op == "v_cmp_eq" else (a < b)
            elif op == "v_cmp_lt_f32":
                touch(regs_of(i.a) + regs_of(i.b), [], i.text)
                W_.masks[i.m.n] = rd32(i.a).view(np.float32) < rd32(i.b).view(np.float32)
                W_.mask_wr[i.m.n] = 0
            elif op == "v_mbcnt_lo":
                touch([], regs_of(i.d), i.text)
                mk = W_.masks[i.m.n]
                c = np.concatenate([[0], np.cumsum(mk[:32])[:-1]]).astype(np.uint32)
                v[i.d.n] = np.concatenate([c, np.full(32, int(mk[:32].sum()), np.uint32)])
            elif op == "v_mbcnt_hi":
                touch(regs_of(i.a), regs_of(i.d), i.text)
                mk = W_.masks[i.m.n]
                c = np.concatenate([[0], np.cumsum(mk[32:])[:-1]]).astype(np.uint32)
                v[i.d.n] = rd32(i.a) + np.concatenate([np.zeros(32, np.uint32), c])
            elif op == "v_max3_abs":
                touch(regs_of(i.a) + regs_of(i.b) + regs_of(i.d), regs_of(i.d), i.text)
                m = np.maximum(np.maximum(np.abs(rd32(i.a).view(np.float32)), np.abs(rd32(i.b).view(np.float32))), rd32(i.d).view(np.float32))
                v[i.d.n] = m.astype(np.float32).view(np.uint32)
            elif op == "s_bcnt1":
                s[i.d.n] = int(W_.masks[i.m.n].sum())
                W_.scc = int(W_.masks[i.m.n].any())
            elif op == "s_lshl":
                r = (srd(i.a) << (srd(i.sh) & 31)) & 0xffffffff
                s[i.d.n] = r
                W_.scc = int(r != 0)
            elif op == "exec_from":
                W_.exec = W_.masks[i.m.n].copy()
            elif op == "exec_all":
                W_.exec = np.ones(64, bool)
            elif op == "exec_lane0":
                W_.exec = np.zeros(64, bool)
                W_.exec[0] = True
            elif op == "v_cndmask":
                touch(regs_of(i.a) + regs_of(i.b), regs_of(i.d), i.text)
                v[i.d.n] = np.where(W_.vcc, rd32(i.b), rd32(i.a))
            elif op in ("v_lshl", "v_lshr"):
                touch(regs_of(i.a), regs_of(i.d), i.text)
                a, sh = rd32(i.a).astype(np.uint64), rd32(i.sh).astype(np.uint64) & 31
                r = (a << sh) if op == "v_lshl" else (a >> sh)
                v[i.d.n] = (r & 0xffffffff).astype(np.uint32)
            elif op == "ds_write_b64":
                touch(regs_of(i.addr) + regs_of(i.data), [], i.text)
                a = (rd32(i.addr).astype(np.int64) + i.off)
                assert np.all(a % 8 == 0) and a.max() + 8 <= self.lds.size * 4, "LDS address"
                wd = a // 4
                assert len(set(wd.tolist())) == 64, "two lanes store to one LDS slot"
                self._lds_write(w, np.concatenate([wd, wd + 1]), "writes")
                lo, hi = rd64(i.data)
                self.lds[wd], self.lds[wd + 1] = lo, hi
                W_.lgkm_q.append(("dswrite", regs_of(i.data)))
            elif op == "ds_write_b32":
                touch(regs_of(i.addr) + regs_of(i.data), [], i.text)
                a = (rd32(i.addr).astype(np.int64) + i.off)[W_.exec]
                assert np.all(a % 4 == 0) and a.max() + 4 <= self.lds.size * 4, "LDS address"
                wd = a // 4
                assert len(set(wd.tolist())) == len(wd), "two lanes store to one LDS word"
                self._lds_write(w, wd, "writes")
                self.lds[wd] = rd32(i.data)[W_.exec]
                W_.lgkm_q.append(("dswrite", regs_of(i.data)))
            elif op == "ds_read_b64":
                touch(regs_of(i.addr), regs_of(i.d), i.text)
                a = (rd32(i.addr).astype(np.int64) + i.off)
                assert np.all(a % 8 == 0) and a.max() + 8 <= self.lds.size * 4, "LDS address"
                wd = a // 4
                self._lds_read(w, np.concatenate([wd, wd + 1]), "reads")
                v[i.d.n], v[i.d.n + 1] = self.lds[wd], self.lds[wd + 1]
                W_.lgkm_q.append(("dsread", regs_of(i.d)))
            elif op == "ds_read_b32":
                touch(regs_of(i.addr), regs_of(i.d), i.text)
                a = (rd32(i.addr).astype(np.int64) + i.off)
                assert np.all(a % 4 == 0) and a.max() + 4 <= self.lds.size * 4, "LDS address"
                self._lds_read(w, a // 4, "reads")
                v[i.d.n] = self.lds[a // 4]
                W_.lgkm_q.append(("dsread", regs_of(i.d)))
            elif op in ("gload2", "gload1", "gstore2"):
                base = srd(i.sbase.lo) | (srd(i.sbase.hi) << 32)
                addr = base + rd32(i.voff).astype(np.int64) + i.off
                act = addr[W_.exec] if op == "gstore2" else addr
                if act.size == 0:
                    W_.vm_q.append(("store", regs_of(i.data)))
                    continue
                b0, arr = self._region(int(act.min()))
                assert int(act.max()) + (4 if op == "gload1" else 8) <= b0 + arr.size * 4, "global access out of its buffer"
                addr = np.where(W_.exec, addr, b0) if op == "gstore2" else addr
                assert np.all(addr % 4 == 0)
                wd = ((addr - b0) // 4).astype(np.int64)
                if op == "gload2":
                    touch(regs_of(i.voff), regs_of(i.d), i.text)
                    v[i.d.n], v[i.d.n + 1] = arr[wd], arr[wd + 1]
                    W_.vm_q.append(("load", regs_of(i.d)))
                elif op == "gload1":
                    touch(regs_of(i.voff), regs_of(i.d), i.text)
                    v[i.d.n] = arr[wd]
                    W_.vm_q.append(("load", regs_of(i.d)))
                else:
                    touch(regs_of(i.voff) + regs_of(i.data), [], i.text)
                    lo, hi = rd64(i.data)
                    m = W_.exec
                    arr[wd[m]], arr[wd[m] + 1] = lo[m], hi[m]
                    W_.vm_q.append(("store", regs_of(i.data)))
            elif op in ("s_waitcnt", "s_waitcnt_x"):
                if i.vmcnt is not None:
                    while len(W_.vm_q) > i.vmcnt: W_.vm_q.pop(0)
                if i.lgkmcnt is not None:
                    # (scalar memory returns out of order with LDS: with one in flight only lgkmcnt(0) guarantees anything about IT;
                    # the LDS operations in front of the allowed count are complete either way -- an smem entry is kept at the
                    # queue's end until a wait for zero)
                    if i.lgkmcnt == 0:
                        W_.lgkm_q.clear()
                        W_.smem_regs.clear()
                    else:
                        # the case that matters for safety: the atomic has already returned, so only LDS operations hold the count
                        lds_q = [x for x in W_.lgkm_q if x[0] != "smem"]
                        n_sm = len(W_.lgkm_q) - len(lds_q)
                        while len(lds_q) > i.lgkmcnt: lds_q.pop(0)
                        W_.lgkm_q = lds_q + [("smem", [])] * n_sm
            elif op == "s_mov":
                s[i.d.n] = srd(i.a)
            elif op == "s_mov64":
                s[i.d.n], s[i.d.n + 1] = srd(i.a.lo), srd(i.a.hi)
            elif op in ("s_add_u32", "s_addc_u32", "s_sub_u32", "s_subb_u32"):
                a, b = srd(i.a), srd(i.b)
                if op == "s_add_u32": r = a + b
                elif op == "s_addc_u32": r = a + b + W_.scc
                elif op == "s_sub_u32": r = a + ((1 << 32) - b)
                else: r = a + ((1 << 32) - b - W_.scc) if (b + W_.scc) else a + (1 << 32)
                if op == "s_sub_u32": W_.scc = int(b > a)
                elif op == "s_subb_u32": W_.scc = int(b + W_.scc > a)
                else: W_.scc = (r >> 32) & 1
                s[i.d.n] = r & 0xffffffff
            elif op == "s_cselect":
                s[i.d.n] = srd(i.a) if W_.scc else srd(i.b)
            elif op in ("s_cmp_lg", "s_cmp_eq", "s_cmp_lt", "s_cmp_ge"):
                a, b = srd(i.a), srd(i.b)
                W_.scc = int((a != b) if op == "s_cmp_lg" else (a < b) if op == "s_cmp_lt" else (a >= b) if op == "s_cmp_ge" else (a == b))
            elif op == "s_mul_i32":
                s[i.d.n] = (srd(i.a) * srd(i.b)) & 0xffffffff
            elif op == "s_mul_hi_u32":
                s[i.d.n] = ((srd(i.a) * srd(i.b)) >> 32) & 0xffffffff
            elif op == "s_mul_hi_i32":
                sg = lambda u: u - (1 << 32) if u & 0x80000000 else u
                s[i.d.n] = ((sg(srd(i.a)) * sg(srd(i.b))) >> 32) & 0xffffffff
            elif op == "s_and_b32":
                r = srd(i.a) & srd(i.b)
                s[i.d.n] = r
                W_.scc = int(r != 0)
            elif op == "s_min_u32":
                a, b = srd(i.a), srd(i.b)
                s[i.d.n] = min(a, b)
                W_.scc = int(a <= b)
            elif op == "s_atomic_add":
                # (the waves of the emulated workgroup run one after the other: the read-modify-write is atomic by construction;
                # `atomic_stride` > 1 stands for the other workgroups of a launch that draw from the same counter in between)
                base = srd(i.sbase.lo) | (srd(i.sbase.hi) << 32)
                b0, arr = self._region(base)
                wd = (base - b0) // 4
                old = int(arr[wd])
                arr[wd] = (old + srd(i.d) * getattr(self, "atomic_stride", 1)) & 0xffffffff
                s[i.d.n] = old
                W_.smem_regs.add(i.d.n)
                W_.lgkm_q.append(("smem", []))
            elif op in ("s_cbranch_scc1", "s_cbranch_scc0", "s_branch"):
                take = True if op == "s_branch" else (W_.scc == 1) if op == "s_cbranch_scc1" else (W_.scc == 0)
                if take:
                    W_.pc = prog.labels[i.label]
            else:
                raise RuntimeError("emulator: unknown op " + op)
        W_.done = True
        return steps, 0


class _VView:   # emu.v[reg] -> all threads' values of one VGPR, as the pre-round-4 emulator exposed them (read / assign)
    def __init__(self, emu): self.e = emu
    def __getitem__(self, n): return np.concatenate([w.v[n] for w in self.e.waves])
    def __setitem__(self, n, val):
        val = np.broadcast_to(np.asarray(val, np.uint32), (self.e.NT,))
        for k, w in enumerate(self.e.waves):
            w.v[n] = val[64 * k:64 * k + 64]


class _SView:   # emu.s[:, n] / emu.s[wave, n]
    def __init__(self, emu): self.e = emu
    def __setitem__(self, key, val):
        wsel, n = key
        ws = range(self.e.NW) if isinstance(wsel, slice) else [wsel]
        for w in ws:
            self.e.waves[w].s[n] = np.uint32(int(val) & 0xffffffff)
    def __getitem__(self, key):
        wsel, n = key
        if isinstance(wsel, slice):
            return np.array([w.s[n] for w in self.e.waves], np.uint32)
        return self.e.waves[wsel].s[n]
