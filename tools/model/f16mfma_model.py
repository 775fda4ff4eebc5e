#!/usr/bin/env python3
"""Lane-level numpy model of prot_f16mfma.hip (round 6): the 4096-point transform pair of the polyphase overlap-add block with its
six radix-16 passes written as 32 x 32 real matrix products in the operand layout of v_mfma_f32_32x32x16_f16 -- which lane holds
which element of which column in every pass, every LDS address, every twiddle index -- checked against numpy's FFT.  The HIP
kernel is a transcription of this file; python3 tools/model/f16mfma_model.py runs the self-test (float64, so only the index
logic is on trial)."""
import numpy as np

NW, NL = 4, 64
L1, G = 272, 17


def onat(h, u):          # output slot u of lane half h -> output index (identity row order of the matrix)
    return 4 * (u >> 1) + 2 * h + (u & 1)


def enat(h, s):          # input slot s of lane half h -> element index (identity column order)
    return 8 * (s >> 2) + 4 * h + (s & 3)


def perm_from_sets(e0, e1):
    """Q with Q[8 (s >> 2) + 4 h + (s & 3)] = e_h[s]"""
    q = [0] * 16
    for h, e in ((0, e0), (1, e1)):
        for s in range(8):
            q[8 * (s >> 2) + 4 * h + (s & 3)] = e[s]
    assert sorted(q) == list(range(16))
    return q


def outperm_from_sets(o0, o1):
    """P with P[4 (u >> 1) + 2 h + (u & 1)] = o_h[u]"""
    p = [0] * 16
    for h, o in ((0, o0), (1, o1)):
        for u in range(8):
            p[4 * (u >> 1) + 2 * h + (u & 1)] = o[u]
    assert sorted(p) == list(range(16))
    return p


DH_OF = {8: 4, 10: 3, 12: 2, 14: 1}     # (dry_delay - conv_delay) / 512 of the engines that use HS rows per block


def rows_for(HS):
    """(input rows E_0, E_1 of the first forward pass; output rows O_0, O_1 of the last inverse pass) for HS new rows per block.
    A lane half owns 8 of the 16 rows of its columns on either side: the non-zero input rows split [0, HS/2) | [HS/2, HS); its
    output rows are its input rows DH later (mod HS) -- the dry frame of output row k is the input pair of row k - DH, of this block
    or the one before: the lane already holds it -- and a row k < 16 - HS sits with its overlap-add partner k + HS.  Slot order of
    the outputs: the rows with a tail first (ascending), then the others, then the tails."""
    NT, DH, HH = 16 - HS, DH_OF[HS], HS // 2
    e = [list(range(0, HH)), list(range(HH, HS))]
    o = []
    for h in range(2):
        rows = [(k + DH) % HS for k in e[h]]
        withtail = sorted(k for k in rows if k < NT)
        others = sorted(k for k in rows if k >= NT)
        assert len(withtail) == NT // 2, (HS, rows)
        o.append(withtail + others + [k + HS for k in withtail])
    z = list(range(HS, 16))
    e[0] += z[:8 - HH]
    e[1] += z[len(z) - (8 - HH):]
    return e[0], e[1], o[0], o[1]


def col_a(w, g, n):
    """pass A / inverse A: the lane's two columns are NEIGHBOURS (one 8-byte access fetches, exchanges and stores both)"""
    return 64 * w + 2 * n + g


def matrix(direction, Q, P):
    """the real 32 x 32 matrix of a radix-16 pass: row M' = 2 * (row slot) + part, column K' = 2 * (column slot) + part"""
    M = np.zeros((32, 32))
    for mp in range(32):
        o, po = P[mp >> 1], mp & 1
        for kp in range(32):
            e, pe = Q[kp >> 1], kp & 1
            th = 2 * np.pi * o * e / 16.0
            c, s = np.cos(th), np.sin(th) * (1 if direction > 0 else -1)      # forward: W = c - i s
            M[mp, kp] = (c if po == pe else (s if (po, pe) == (0, 1) else -s))
    return M


def mfma_pass(M, vin):
    """vin[w][l][g][s] complex -> vout[w][l][g][u]: what two pairs of v_mfma_f32_32x32x16_f16 (K halves) compute per wave"""
    vout = np.zeros_like(vin)
    for w in range(NW):
        for g in range(2):
            X = np.zeros((32, 32))               # [K'][n]
            for l in range(NL):
                h, n = l >> 5, l & 31
                for s in range(8):
                    kh, i2 = s >> 2, s & 3      # MFMA kh, lane supplies K = 8 h + 2 i2 (+1) of that K half
                    K = 16 * kh + 8 * h + 2 * i2
                    X[K, n], X[K + 1, n] = vin[w, l, g, s].real, vin[w, l, g, s].imag
            D = M @ X
            for l in range(NL):
                h, n = l >> 5, l & 31
                for r in range(16):
                    mp = 8 * (r >> 2) + 4 * h + (r & 3)
                    if r & 1:
                        vout[w, l, g, r >> 1] += 1j * D[mp, n]
                    else:
                        vout[w, l, g, r >> 1] += D[mp, n]
    return vout


def block_model(z, Hs, HS=10):
    """z: 256 * HS complex input samples of one block (rows of 256); Hs[0..2048]: the spectrum / 4096 of g.  Returns the 4096
    complex values of the circular convolution, [row][col] order as out[256 * k + t] -- through the lane-level pipeline."""
    E0, E1, O0, O1 = rows_for(HS)
    QA, PA = perm_from_sets(E0, E1), outperm_from_sets(O0, O1)
    Qn = list(range(16))
    Pn = list(range(16))
    Qp = perm_from_sets([onat(0, s) for s in range(8)], [onat(1, s) for s in range(8)])   # input slots hold what output slots held
    M1, M2 = matrix(1, QA, Pn), matrix(1, Qn, Pn)
    M3, M4, M5 = matrix(-1, Qp, Pn), matrix(-1, Qn, Pn), matrix(-1, Qp, PA)
    x = np.zeros(4096, complex)
    x[:256 * HS] = z
    lds = np.zeros(16 * L1, complex)
    W = lambda N, k: np.exp(-2j * np.pi * k / N)
    idx = lambda: ((w, l, g) for w in range(NW) for l in range(NL) for g in range(2))
    v = np.zeros((NW, NL, 2, 8), complex)
    # ---- forward pass A: column t = 64 w + 32 g + n, input rows E_h
    for w, l, g in idx():
        h, n = l >> 5, l & 31
        t = col_a(w, g, n)
        for s in range(8):
            k = (E0, E1)[h][s]
            v[w, l, g, s] = x[t + 256 * k]
    v = mfma_pass(M1, v)
    for w, l, g in idx():                       # twiddle W4096^{t a}, exchange 1: slot a * L1 + t
        h, n = l >> 5, l & 31
        t = col_a(w, g, n)
        for u in range(8):
            a = onat(h, u)
            lds[a * L1 + t] = v[w, l, g, u] * W(4096, t * a)
    # barrier
    for w, l, g in idx():                       # pass B: column (j0, a) = (n & 15, 4 w + 2 g + (n >> 4)), element j1
        h, n = l >> 5, l & 31
        j0, a = n & 15, 4 * w + 2 * g + (n >> 4)
        for s in range(8):
            v[w, l, g, s] = lds[a * L1 + j0 + 16 * enat(h, s)]
    v = mfma_pass(M2, v)
    for w, l, g in idx():                       # twiddle W256^{j0 b}; wave-local exchange 2: slot a * L1 + b * G + j0
        h, n = l >> 5, l & 31
        j0, a = n & 15, 4 * w + 2 * g + (n >> 4)
        for u in range(8):
            b = onat(h, u)
            lds[a * L1 + b * G + j0] = v[w, l, g, u] * W(256, j0 * b)
    for w, l, g in idx():                       # pass C: column (a, b) = (4 w + 2 g + (n >> 4), n & 15), element j0
        h, n = l >> 5, l & 31
        b, a = n & 15, 4 * w + 2 * g + (n >> 4)
        for s in range(8):
            v[w, l, g, s] = lds[a * L1 + b * G + enat(h, s)]
    v = mfma_pass(M2, v)
    # ---- spectrum: the lane holds bins kappa = a + 16 b + 256 c, c = onat(h, u)
    spec = np.zeros(4096, complex)
    for w, l, g in idx():
        h, n = l >> 5, l & 31
        b, a = n & 15, 4 * w + 2 * g + (n >> 4)
        for u in range(8):
            c = onat(h, u)
            kap = a + 16 * b + 256 * c
            spec[kap] = v[w, l, g, u]
            Hk = Hs[kap] if kap <= 2048 else np.conj(Hs[4096 - kap])
            v[w, l, g, u] *= Hk
    # ---- inverse pass C (input slots = the output slots just held), twiddle conj W256^{j0 b}, wave-local exchange
    v = mfma_pass(M3, v)
    for w, l, g in idx():
        h, n = l >> 5, l & 31
        b, a = n & 15, 4 * w + 2 * g + (n >> 4)
        for u in range(8):
            j0 = onat(h, u)
            lds[a * L1 + b * G + j0] = v[w, l, g, u] * np.conj(W(256, j0 * b))
    for w, l, g in idx():                       # inverse pass B: column (j0, a), element b
        h, n = l >> 5, l & 31
        j0, a = n & 15, 4 * w + 2 * g + (n >> 4)
        for s in range(8):
            v[w, l, g, s] = lds[a * L1 + enat(h, s) * G + j0]
    v = mfma_pass(M4, v)
    for w, l, g in idx():                       # exchange 1': slot a * L1 + j0 + 16 j1 (no twiddle yet: it belongs to (t, a), applied by the reader)
        h, n = l >> 5, l & 31
        j0, a = n & 15, 4 * w + 2 * g + (n >> 4)
        for u in range(8):
            lds[a * L1 + j0 + 16 * onat(h, u)] = v[w, l, g, u]
    # barrier
    for w, l, g in idx():                       # inverse pass A: column t, elements a = onat(h, s) (the slots this lane wrote in exchange 1), conj W4096^{t a}
        h, n = l >> 5, l & 31
        t = col_a(w, g, n)
        for s in range(8):
            a = onat(h, s)
            v[w, l, g, s] = lds[a * L1 + t] * np.conj(W(4096, t * a))
    v = mfma_pass(M5, v)
    out = np.zeros(4096, complex)
    for w, l, g in idx():
        h, n = l >> 5, l & 31
        t = col_a(w, g, n)
        for u in range(8):
            out[t + 256 * (O0, O1)[h][u]] = v[w, l, g, u]
    return out, spec


def self_test():
    rng = np.random.default_rng(1)
    for HS in (8, 10, 12, 14):
        E0, E1, O0, O1 = rows_for(HS)
        NT = 16 - HS
        assert sorted(E0 + E1) == list(range(16)) and sorted(O0 + O1) == list(range(16))
        for O in (O0, O1):
            for k in O:
                if k < NT:
                    assert k + HS in O, (HS, O)
        assert len([k for k in E0 if k < HS]) == HS // 2 == len([k for k in E1 if k < HS])
        z = rng.standard_normal(256 * HS) + 1j * rng.standard_normal(256 * HS)
        g = rng.standard_normal(256 * NT + 1)
        Hfull = np.fft.fft(np.concatenate([g, np.zeros(4096 - len(g))])) / 4096.0
        # (the kernel's table holds bins 0..2048 and mirrors the rest as conj: that is a REAL g's spectrum)
        out, spec = block_model(z, Hfull[:2049], HS)
        x = np.concatenate([z, np.zeros(4096 - len(z))])
        assert np.abs(spec - np.fft.fft(x)).max() < 1e-9, HS
        ref = np.fft.ifft(np.fft.fft(x) * np.fft.fft(np.concatenate([g, np.zeros(4096 - len(g))])))
        err = np.abs(out - ref).max()
        print("HS %2d: rows in %s | %s, rows out %s | %s: max |model - fft| = %.2e" % (HS, E0, E1, O0, O1, err))
        assert err < 1e-9


if __name__ == "__main__":
    self_test()


def cxx_tables():
    """the constexpr tables of prot_f16mfma.hip (MfRows<HS>): output rows by lane half and slot, and where the dry frame of output
    slot u comes from -- slot >= 0: the current block's input pair of that slot; < 0: the carried pair -1 - slot of the block before
    (the carried pairs are the last DH input rows of a block: slots HS/2 - DH .. HS/2 - 1 of lane half 1)"""
    out = []
    for HS in (8, 10, 12, 14):
        E0, E1, O0, O1 = rows_for(HS)
        DH, HH = DH_OF[HS], HS // 2
        src = []
        for h, O in ((0, O0), (1, O1)):
            row = []
            for u in range(HH):
                r = O[u] - DH
                if r >= 0:
                    assert HH * h <= r < HH * (h + 1)
                    row.append(r - HH * h)
                else:
                    r += HS
                    assert h == 1 and r >= HS - DH
                    row.append(-1 - (r - (HS - DH)))
            src.append(row)
        f = lambda a: "{ " + ", ".join("%d" % x for x in a) + " }"
        out.append("template <> struct MfRows<%d> {\n\tstatic constexpr int DH = %d;\n\tstatic constexpr int O[2][8] = { %s, %s };\n"
                   "\tstatic constexpr int DSRC[2][%d] = { %s, %s };\n};" % (HS, DH, f(O0), f(O1), HH, f(src[0]), f(src[1])))
    return "\n".join(out)


if __name__ == "__main__" and "--tables" in __import__("sys").argv:
    print(cxx_tables())
