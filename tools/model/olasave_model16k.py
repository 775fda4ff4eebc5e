"""numpy model of the 16384-point overlap-save block (round 5; olasave16k_kernel): 1024 threads, 16 points per thread, three
radix-16 passes over the digits (a, b, c) / (k1, k2, k3) plus ONE radix-4 layer over the top two bits e of the thread index
(T = 256 e + 16 b + c; k = k1 + 16 ke + 64 k2 + 1024 k3), the forward transform again evaluated at the half-shifted index
p - 1/2 so that the odd-tap polyphase filter g needs no spectrum table (olasave_model.py: the 4096-point form, olasave_model8k.py:
the 8192-point one, whose radix-2 layer this generalises).

A block holds 16 rows of 1024 complex samples; NT = len(g) / 1024 rows are context.  The CLI's default engine at 88.2 / 96 kHz
(blksiz 16384, cli/phase-rotate.cc:749-755) has 8192 odd taps: NT = 8, HS = 8 rows (16384 frames) new per block -- ONE table-free
partition in ONE launch where olapoly_kernel takes two accumulating launches of two 2048-tap partitions each."""
import numpy as np

N = 16384


def W(num, den):
    return np.exp(-2j * np.pi * np.asarray(num, np.float64) / den)


def hilbert_fir(n_taps):
    n = np.arange(n_taps); m = n - n_taps // 2
    fir = np.zeros(n_taps)
    odd = (m % 2 != 0)
    fir[odd] = -(2 / n_taps) / np.tan(np.pi * m[odd] / n_taps) * 0.5 * (1 - np.cos(2 * np.pi * n[odd] / n_taps))
    return fir.astype(np.float32).astype(np.float64)


def tables(g):
    Lg = len(g)
    assert Lg % 1024 == 0 and np.all(g == -g[::-1])
    H = np.fft.fft(np.concatenate([g, np.zeros(N - Lg)]))
    k = np.arange(N)
    B = (H / (W((Lg // 2) * k, N) * W(-k, 2 * N)) / (-1j))
    assert np.abs(B.imag).max() < 1e-10
    return B.real


F16 = W(np.outer(np.arange(16), np.arange(16)), 16)   # DFT-16 kernel [out][in]
F4 = W(np.outer(np.arange(4), np.arange(4)), 4)


def fwd_shifted(w):
    """w[1024 a + T] -> Z'[k] = sum_p w[p] W^{(p - 1/2) k}, k = k1 + 16 ke + 64 k2 + 1024 k3; returns [k1][ke][k2][k3]"""
    x = w.reshape(16, 1024)                                  # [a][T]
    u = np.einsum('ka,at->kt', F16, x)                       # pass 1: [k1][T]
    T = np.arange(1024)
    u = u * W((T[None, :] - 0.5) * np.arange(16)[:, None], N)            # twiddle 1: W_N^{(T - 1/2) k1}
    u = u.reshape(16, 4, 16, 16)                             # [k1][e][b][c]
    v = np.einsum('qe,kebc->kqbc', F4, u)                    # radix 4 over e -> ke: [k1][ke][b][c]
    v = v * W(np.arange(16)[None, None, :, None] * np.arange(4)[None, :, None, None], 64)   # W64^{b ke}
    y2 = np.einsum('jb,kebc->kejc', F16, v)                  # pass 2: [k1][ke][k2][c]
    cc = np.arange(16)[None, None, None, :] - 0.5
    y2 = y2 * W(cc * np.arange(4)[None, :, None, None], 1024) * W(cc * np.arange(16)[None, None, :, None], 256)
    y3 = np.einsum('lc,kejc->kejl', F16, y2)                 # pass 3: [k1][ke][k2][k3]
    return y3 * W(-np.arange(16), 32)[None, None, None, :]   # W32^{-k3}


def inv_plain(Y):
    """Y[k1][ke][k2][k3] -> y[1024 a + T] = sum_k Y[k] W^{-pk} (unnormalised)"""
    Fi, F4i = np.conj(F16), np.conj(F4)
    r = np.einsum('cl,kejl->kejc', Fi, Y)                    # over k3 -> c
    cc = np.arange(16)[None, None, None, :]
    r = r * np.conj(W(cc * np.arange(4)[None, :, None, None], 1024) * W(cc * np.arange(16)[None, None, :, None], 256))
    s = np.einsum('bj,kejc->kebc', Fi, r)                    # over k2 -> b
    s = s * np.conj(W(np.arange(16)[None, None, :, None] * np.arange(4)[None, :, None, None], 64))
    t = np.einsum('eq,kqbc->kebc', F4i, s)                   # over ke -> e: [k1][e][b][c]
    t = t.reshape(16, 1024)
    T = np.arange(1024)
    t = t * np.conj(W(T[None, :] * np.arange(16)[:, None], N))
    return np.einsum('ak,kt->at', Fi, t).reshape(N)          # over k1 -> a


def kindex():
    return (np.arange(16)[:, None, None, None] + 16 * np.arange(4)[None, :, None, None] +
            64 * np.arange(16)[None, None, :, None] + 1024 * np.arange(16)[None, None, None, :])


def block(w, B, exact_B=False):
    """one block; valid outputs are the whole rows R0 .. R0 + HS - 1"""
    Zp = fwd_shifted(w)
    k = kindex()
    Bk = B[k]
    if not exact_B:                                          # |B| = 1 outside the rows next to DC: k3 = 0 and k3 = 15
        keep = (k < 1024) | (k >= N - 1024)
        Bk = np.where(keep, Bk, np.sign(Bk))
    return inv_plain(Zp * (-1j) * Bk) / N


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    w = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    Z = np.fft.fft(w) * W(-np.arange(N), 2 * N)
    assert np.abs(fwd_shifted(w) - Z[kindex()]).max() < 1e-8
    assert np.abs(inv_plain(Z[kindex()] / W(-np.arange(N), 2 * N)[kindex()]) / N - w).max() < 1e-9
    for n_taps, hs in [(16384, 8), (8192, 12)]:
        fir = hilbert_fir(n_taps)
        g = fir[1::2]
        Lg = len(g); NT = 16 - hs
        assert Lg == 1024 * NT
        R0 = NT // 2
        B = tables(g)
        dev = np.abs(np.abs(B) - 1)
        rows = [float(dev[1024 * r:1024 * (r + 1)].max()) for r in range(3)]
        z = 0.3 * (rng.standard_normal(N) + 1j * rng.standard_normal(N))
        ref = np.convolve(z, g)
        want = ref[Lg // 2:Lg // 2 + N]
        lo, hi = 1024 * R0, 1024 * (R0 + hs)
        e_exact = np.abs(block(z, B, True) - want)[lo:hi].max()
        e1 = np.abs(block(z, B, False) - want)[lo:hi].max()
        print("taps %5d HS %2d NT %2d: |B|-1 by 1024-bin row %s  err exact %.2e  table-free %.2e  sign(B) %s" % (
            n_taps, hs, NT, ["%.1e" % r for r in rows], e_exact, e1, set(np.sign(B[1024:N - 1024]).tolist())))
