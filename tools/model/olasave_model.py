"""numpy model of the overlap-save polyphase block of olasave_kernel (round 3): a 4096-point complex transform pair as three
radix-16 passes over the digits (a, b, c) / (k1, k2, k3), the forward one evaluated at the half-shifted time index n - 1/2
so that the Hilbert FIR's odd-tap polyphase filter g (antisymmetric about its centre, i.e. a half-sample-shift all-pass
away from DC) needs no spectrum table: H[k] = -i B[k] W^{(Lg/2) k} W^{-k/2}, B real, |B| = 1 to 7e-8 for |k| >= 256.
Checked here against the direct convolution in float64; the HIP kernel follows this file pass by pass."""
import numpy as np

N = 4096


def hilbert_fir(n_taps):
    n = np.arange(n_taps); m = n - n_taps // 2
    fir = np.zeros(n_taps)
    odd = (m % 2 != 0)
    fir[odd] = -(2 / n_taps) / np.tan(np.pi * m[odd] / n_taps) * 0.5 * (1 - np.cos(2 * np.pi * n[odd] / n_taps))
    return fir.astype(np.float32).astype(np.float64)


def W(num, den):
    return np.exp(-2j * np.pi * np.asarray(num, np.float64) / den)


def tables(g):
    """per-engine constants of the table-free form; g = odd taps (float64 copies of the float taps)"""
    Lg = len(g)
    assert Lg % 512 == 0 and np.all(g == -g[::-1])
    H = np.fft.fft(np.concatenate([g, np.zeros(N - Lg)]))            # sum g[s] W^{sk}
    k = np.arange(N)
    # H = (-i B) W^{(Lg/2) k} W^{-k/2}
    B = (H / (W((Lg // 2) * k, N) * W(-k, 2 * N)) / (-1j))
    assert np.abs(B.imag).max() < 1e-12
    return B.real


def fwd_shifted(z):
    """z[256 a + t] -> Z'[k1 + 16 k2 + 256 k3] = sum_n z[n] W^{(n - 1/2) k}; returns array [k1][k2][k3]"""
    x = z.reshape(16, 16, 16)                    # [a][b][c], t = 16 b + c
    a = np.arange(16)
    F = W(np.outer(a, a), 16)                    # DFT-16 kernel [out][in]
    y1 = np.einsum('ka,abc->kbc', F, x)          # [k1][b][c]
    t = (16 * np.arange(16)[:, None] + np.arange(16)[None, :])  # [b][c]
    y1 = y1 * W((t[None, :, :] - 0.5) * a[:, None, None], N)   # W^{(t - 1/2) k1}
    y2 = np.einsum('jb,kbc->kjc', F, y1)         # [k1][k2][c]
    y2 = y2 * W((np.arange(16)[None, None, :] - 0.5) * a[None, :, None], 256)  # W256^{(c - 1/2) k2}
    y3 = np.einsum('lc,kjc->kjl', F, y2)         # [k1][k2][k3]
    y3 = y3 * W(-a, 32)[None, None, :]           # W32^{-k3}
    return y3


def inv_plain(Y):
    """Y[k1][k2][k3] -> y[256 a + t] = sum_k Y[k] W^{-nk} (unnormalised)"""
    a = np.arange(16)
    Fi = np.conj(W(np.outer(a, a), 16))
    u = np.einsum('cl,kjl->kjc', Fi, Y)                           # over k3 -> c
    u = u * np.conj(W(np.arange(16)[None, None, :] * a[None, :, None], 256))  # conj W256^{c k2}
    v = np.einsum('bj,kjc->kbc', Fi, u)                           # over k2 -> b
    t = (16 * np.arange(16)[:, None] + np.arange(16)[None, :])
    v = v * np.conj(W(t[None, :, :] * a[:, None, None], N))       # conj W^{t k1}
    y = np.einsum('ak,kbc->abc', Fi, v)                           # over k1 -> a
    return y.reshape(N)


def block(z, B, Lg, exact_B=False, na_rows=1):
    """one overlap-save block: valid outputs are local samples [Lg/2, N - Lg/2) = rows R0 .. 16 - R0"""
    Zp = fwd_shifted(z)                          # Z[k] W^{-k/2}
    k = (np.arange(16)[:, None, None] + 16 * np.arange(16)[None, :, None] + 256 * np.arange(16)[None, None, :])
    Bk = B[k]
    if not exact_B:                              # |B| = 1 outside the rows next to DC
        sgn = np.sign(Bk)
        keep = (k < 256 * na_rows) | (k >= N - 256 * na_rows)
        Bk = np.where(keep, Bk, sgn)
    Y = Zp * (-1j) * Bk
    return inv_plain(Y) / N


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    for n_taps, hs in [(3072, 10), (4096, 8), (2048, 12), (1024, 14)]:
        fir = hilbert_fir(n_taps)
        g = fir[1::2]
        Lg = len(g); R0 = (16 - hs) // 2
        assert Lg == 256 * (16 - hs)
        B = tables(g)
        dev = np.abs(np.abs(B) - 1)
        rows = [float(dev[256 * r:256 * (r + 1)].max()) for r in range(4)]
        z = rng.standard_normal(N) + 1j * rng.standard_normal(N)
        z *= 0.3
        ref = np.convolve(z, g)                  # ref[n] = sum g[s] z[n - s]
        # block output local n corresponds to ref[n + Lg/2]
        want = ref[Lg // 2:Lg // 2 + N]
        lo, hi = 256 * R0, N - 256 * R0
        e_exact = np.abs(block(z, B, Lg, True) - want)[lo:hi].max()
        e1 = np.abs(block(z, B, Lg, False, 1) - want)[lo:hi].max()
        e2 = np.abs(block(z, B, Lg, False, 2) - want)[lo:hi].max()
        print("taps %5d HS %2d: |B|-1 by row %s  err exact %.2e  na=1 %.2e  na=2 %.2e  sign(B) %s" % (
            n_taps, hs, ["%.1e" % r for r in rows], e_exact, e1, e2, set(np.sign(B[256:3840]).tolist())))
