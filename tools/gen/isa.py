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


class Lit:  # 32-bit literal / inline constant (float or int)
    def __init__(self, v): self.v = v
    def __repr__(self):
        if isinstance(self.v, float):
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
        if self.auto and i.op not in ("comment", "label", "s_waitcnt"):
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
            elif i.op == "ds_write_b64":
                self.lgq.append(set())
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
    def s_cselect(self, d, a, b): return self.add(Ins("s_cselect", "s_cselect_b32 %r, %r, %r" % (d, a, b), d=d, a=a, b=b))
    def s_cmp_lg_u32(self, a, b): return self.add(Ins("s_cmp_lg", "s_cmp_lg_u32 %r, %r" % (a, b), a=a, b=b))
    def s_cmp_eq_u32(self, a, b): return self.add(Ins("s_cmp_eq", "s_cmp_eq_u32 %r, %r" % (a, b), a=a, b=b))
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
    def s_barrier(self): return self.add(Ins("s_barrier", "s_barrier"))
    def s_nop(self, n=0): return self.add(Ins("s_nop", "s_nop %d" % n))
    def s_setprio(self, n): return self.add(Ins("s_nop", "s_setprio %d" % n))
    def comment(self, txt): return self.add(Ins("comment", "; " + txt))
    def raw(self, txt):  # text the emulator ignores (input copies in the C++ wrapper's dialect)
        return self.add(Ins("comment", txt))

    def text(self, label_suffix=""):
        out = []
        for i in self.ins:
            t = i.text
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
class Emu:
    NT = 256

    def __init__(self, lds_bytes=65536, check_waits=True):
        self.v = np.zeros((256, self.NT), np.uint32)
        self.s = np.zeros((4, 128), np.uint32)          # per wave
        self.scc = 0
        self.lds = np.zeros(lds_bytes // 4, np.uint32)
        self.mem = {}                                   # base address -> (np.uint32 array)
        self.check = check_waits
        self.vm_q, self.lgkm_q = [], []                 # in-flight ops: (kind, regs)
        self.v[255, :] = 0

    # memory regions
    def map(self, base, arr_u32):
        self.mem[base] = arr_u32

    def _region(self, addr):
        for b, a in self.mem.items():
            if b <= addr < b + a.size * 4:
                return b, a
        raise RuntimeError("unmapped address 0x%x" % addr)

    # waits discipline
    def _busy_regs(self):
        busy = {}
        for kind, regs in self.vm_q + self.lgkm_q:
            if kind in ("store", "dswrite"):   # data is read when the instruction issues
                continue
            for r in regs:
                busy[r] = kind
        return busy

    def _touch(self, reads, writes, what):
        if not self.check:
            return
        busy = self._busy_regs()
        for r in reads:
            if r in busy and busy[r] in ("load", "dsread"):
                raise RuntimeError("%s reads v%d while a %s into it is in flight" % (what, r, busy[r]))
        for r in writes:
            if r in busy:
                raise RuntimeError("%s writes v%d while a %s on it is in flight" % (what, r, busy[r]))

    @staticmethod
    def _regs(x):
        if isinstance(x, V): return [x.n]
        if isinstance(x, VP): return [x.n, x.n + 1]
        return []

    # operand access
    def _rd32(self, x):
        if isinstance(x, V): return self.v[x.n].copy()
        if isinstance(x, S): return np.repeat(self.s[:, x.n], 64)
        if isinstance(x, Lit): return np.full(self.NT, x.bits(), np.uint32)
        if isinstance(x, (int, np.integer)): return np.full(self.NT, np.uint32(x & 0xffffffff), np.uint32)
        raise TypeError(x)

    def _rd64(self, x):
        if isinstance(x, VP): return self.v[x.n].copy(), self.v[x.n + 1].copy()
        if isinstance(x, SP): return np.repeat(self.s[:, x.n], 64), np.repeat(self.s[:, x.n + 1], 64)
        raise TypeError(x)

    def _srd(self, x):  # scalar read, per wave [4]
        if isinstance(x, S): return self.s[:, x.n].copy()
        if isinstance(x, Lit): return np.full(4, x.bits(), np.uint32)
        if isinstance(x, (int, np.integer)): return np.full(4, np.uint32(x & 0xffffffff), np.uint32)
        raise TypeError(x)

    def run(self, prog, max_steps=10_000_000):
        pc, steps = 0, 0
        ins = prog.ins
        recent = []   # VGPRs written by the last two VALU instructions (DPP hazard: 2 wait states)
        sgpr_wr = 99  # instructions since a VALU wrote an SGPR (v_readlane / v_readfirstlane)
        f32 = lambda u: u.view(np.float32)
        while pc < len(ins):
            i = ins[pc]
            pc += 1
            steps += 1
            if steps > max_steps:
                raise RuntimeError("emulation did not terminate")
            op = i.op
            if op in ("comment", "label", "s_barrier"):
                continue
            if op == "s_nop":
                recent = (recent + [set()])[-2:]
                sgpr_wr += 5
                continue
            # measured on gfx950: a VALU instruction reading an SGPR directly behind a VALU write of an SGPR reads garbage
            if op.startswith("pk_") or (op.startswith("v_") and op not in ("v_readlane", "v_readfirstlane")):
                srcs = list(getattr(i, "srcs", [])) + [getattr(i, k) for k in ("a", "b", "sh") if hasattr(i, k)]
                if sgpr_wr < 3 and any(isinstance(x, (S, SP)) for x in srcs):
                    raise RuntimeError("SGPR hazard: %s reads an SGPR %d instruction(s) after a VALU wrote one" % (i.text, sgpr_wr + 1))
            sgpr_wr = 0 if op in ("v_readlane", "v_readfirstlane") else sgpr_wr + (5 if op == "s_nop" else 1)
            if op in ("v_mov_dpp_shr1", "v_mov_dpp_rowshl"):
                used = set(self._regs(i.a) + self._regs(i.d))
                for w in recent:
                    if w & used:
                        raise RuntimeError("DPP hazard: %s within 2 wait states of a VALU write of its register" % i.text)
            if op.startswith("pk_") or op.startswith("v_"):
                recent = (recent + [set(self._regs(i.d)) if op not in ("v_readlane", "v_readfirstlane") else set()])[-2:]
            else:
                recent = (recent + [set()])[-2:]
            if op.startswith("pk_"):
                reads = sum((self._regs(x) for x in i.srcs), [])
                self._touch(reads, self._regs(i.d), i.text)
                vals = [self._rd64(x) for x in i.srcs]
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
                self.v[i.d.n], self.v[i.d.n + 1] = res[0], res[1]
            elif op == "v_mov":
                self._touch(self._regs(i.a), self._regs(i.d), i.text)
                self.v[i.d.n] = self._rd32(i.a)
            elif op == "v_mov_dpp_shr1":
                self._touch(self._regs(i.a) + self._regs(i.d), self._regs(i.d), i.text)
                src = self.v[i.a.n].reshape(4, 64)
                dst = self.v[i.d.n].reshape(4, 64).copy()
                dst[:, 1:] = src[:, :-1]
                self.v[i.d.n] = dst.reshape(-1)
            elif op == "v_mov_dpp_rowshl":
                self._touch(self._regs(i.a) + self._regs(i.d), self._regs(i.d), i.text)
                src = self.v[i.a.n].reshape(4, 64)
                dst = self.v[i.d.n].reshape(4, 64).copy()
                for lane in range(4):
                    if lane + i.k <= 15:
                        dst[:, lane] = src[:, lane + i.k]
                self.v[i.d.n] = dst.reshape(-1)
            elif op == "v_mul_f32":
                self._touch(self._regs(i.a) + self._regs(i.b), self._regs(i.d), i.text)
                a, b = self._rd32(i.a).view(np.float32), self._rd32(i.b).view(np.float32)
                self.v[i.d.n] = (a * b).astype(np.float32).view(np.uint32)
            elif op == "v_readfirstlane":
                self._touch(self._regs(i.a), [], i.text)
                self.s[:, i.d.n] = self.v[i.a.n].reshape(4, 64)[:, 0]
            elif op == "v_readlane":
                self._touch(self._regs(i.a), [], i.text)
                self.s[:, i.d.n] = self.v[i.a.n].reshape(4, 64)[:, i.lane]
            elif op in ("v_and", "v_add_u32", "v_mul_u24"):
                self._touch(self._regs(i.a) + self._regs(i.b), self._regs(i.d), i.text)
                a, b = self._rd32(i.a).astype(np.uint64), self._rd32(i.b).astype(np.uint64)
                r = (a & b) if op == "v_and" else (a + b) if op == "v_add_u32" else ((a & 0xffffff) * (b & 0xffffff))
                self.v[i.d.n] = (r & 0xffffffff).astype(np.uint32)
            elif op in ("v_lshl", "v_lshr"):
                self._touch(self._regs(i.a), self._regs(i.d), i.text)
                a, sh = self._rd32(i.a).astype(np.uint64), self._rd32(i.sh).astype(np.uint64) & 31
                r = (a << sh) if op == "v_lshl" else (a >> sh)
                self.v[i.d.n] = (r & 0xffffffff).astype(np.uint32)
            elif op == "ds_write_b64":
                self._touch(self._regs(i.addr) + self._regs(i.data), [], i.text)
                a = (self._rd32(i.addr).astype(np.int64) + i.off)
                assert np.all(a % 8 == 0) and a.max() + 8 <= self.lds.size * 4, "LDS address"
                w = a // 4
                assert len(set(w.tolist())) == self.NT, "two lanes store to one LDS slot"
                lo, hi = self._rd64(i.data)
                self.lds[w], self.lds[w + 1] = lo, hi
                self.lgkm_q.append(("dswrite", self._regs(i.data)))
            elif op == "ds_read_b64":
                self._touch(self._regs(i.addr), self._regs(i.d), i.text)
                a = (self._rd32(i.addr).astype(np.int64) + i.off)
                assert np.all(a % 8 == 0) and a.max() + 8 <= self.lds.size * 4, "LDS address"
                w = a // 4
                self.v[i.d.n], self.v[i.d.n + 1] = self.lds[w], self.lds[w + 1]
                self.lgkm_q.append(("dsread", self._regs(i.d)))
            elif op == "ds_read_b32":
                self._touch(self._regs(i.addr), self._regs(i.d), i.text)
                a = (self._rd32(i.addr).astype(np.int64) + i.off)
                assert np.all(a % 4 == 0) and a.max() + 4 <= self.lds.size * 4, "LDS address"
                self.v[i.d.n] = self.lds[a // 4]
                self.lgkm_q.append(("dsread", self._regs(i.d)))
            elif op in ("gload2", "gload1", "gstore2"):
                base = (self._srd(i.sbase.lo).astype(np.uint64) | (self._srd(i.sbase.hi).astype(np.uint64) << 32))
                addr = np.repeat(base, 64).astype(np.int64) + self._rd32(i.voff).astype(np.int64) + i.off
                b0, arr = self._region(int(addr.min()))
                assert int(addr.max()) + (4 if op == "gload1" else 8) <= b0 + arr.size * 4, "global access out of its buffer"
                assert np.all(addr % 4 == 0)
                w = ((addr - b0) // 4).astype(np.int64)
                if op == "gload2":
                    self._touch(self._regs(i.voff), self._regs(i.d), i.text)
                    self.v[i.d.n], self.v[i.d.n + 1] = arr[w], arr[w + 1]
                    self.vm_q.append(("load", self._regs(i.d)))
                elif op == "gload1":
                    self._touch(self._regs(i.voff), self._regs(i.d), i.text)
                    self.v[i.d.n] = arr[w]
                    self.vm_q.append(("load", self._regs(i.d)))
                else:
                    self._touch(self._regs(i.voff) + self._regs(i.data), [], i.text)
                    lo, hi = self._rd64(i.data)
                    arr[w], arr[w + 1] = lo, hi
                    self.vm_q.append(("store", self._regs(i.data)))
            elif op == "s_waitcnt":
                if i.vmcnt is not None:
                    while len(self.vm_q) > i.vmcnt: self.vm_q.pop(0)
                if i.lgkmcnt is not None:
                    while len(self.lgkm_q) > i.lgkmcnt: self.lgkm_q.pop(0)
            elif op == "s_mov":
                self.s[:, i.d.n] = self._srd(i.a)
            elif op == "s_mov64":
                self.s[:, i.d.n], self.s[:, i.d.n + 1] = self._srd(i.a.lo), self._srd(i.a.hi)
            elif op in ("s_add_u32", "s_addc_u32", "s_sub_u32"):
                a, b = self._srd(i.a).astype(np.uint64), self._srd(i.b).astype(np.uint64)
                if op == "s_add_u32": r = a + b
                elif op == "s_addc_u32": r = a + b + np.uint64(self.scc)
                else: r = (a + (np.uint64(1 << 32) - b))
                if op == "s_sub_u32":
                    self.scc = int(b[0] > a[0])
                else:
                    self.scc = int(r[0] >> 32) & 1
                self.s[:, i.d.n] = (r & 0xffffffff).astype(np.uint32)
            elif op == "s_cselect":
                self.s[:, i.d.n] = self._srd(i.a) if self.scc else self._srd(i.b)
            elif op in ("s_cmp_lg", "s_cmp_eq"):
                a, b = self._srd(i.a), self._srd(i.b)
                self.scc = int((a[0] != b[0]) if op == "s_cmp_lg" else (a[0] == b[0]))
            elif op in ("s_cbranch_scc1", "s_cbranch_scc0", "s_branch"):
                take = True if op == "s_branch" else (self.scc == 1) if op == "s_cbranch_scc1" else (self.scc == 0)
                if take:
                    pc = prog.labels[i.label]
            else:
                raise RuntimeError("emulator: unknown op " + op)
        return steps
