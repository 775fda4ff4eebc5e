"""numpy model of the SHORT-CHAIN overlap-save block (round 6: csrc/prot_olasave512.hip): the 4096-point transform pair of
olasave_model.py as FOUR radix-8 passes on 512 threads (8 points per thread) -- a lone workgroup's dependent chain per thread is
half as long as with 256 threads x 16 points, and two waves share every SIMD.  Forward at the half-shifted index n - 1/2 over the
digits n = 512 a + 64 b + 8 c + d -> k = k1 + 8 k2 + 64 k3 + 512 k4; the inverse is the same decimation run from the k side
(k4 first), so every twiddle is w^j of a thread-fixed w and the register index j.  The model follows the kernel thread by thread
(who holds what between the passes, which LDS slot every exchange uses) and is checked against the direct convolution."""
import numpy as np

N, NT8 = 4096, 512


def W(num, den):
    return np.exp(-2j * np.pi * np.asarray(num, np.float64) / den)


F8 = W(np.outer(np.arange(8), np.arange(8)), 8)      # [out][in]


def tables(B):
    """the per-engine constants as the host lays them out (prot_engine.hip save_tables512): six twiddle tables [j - 1][index] for
    j = 1..7, and per thread (k1, k2, k3) the two |B| factors of its k4 = 0 and k4 = 7 bins (1 outside the rows next to DC)"""
    j = np.arange(1, 8)[:, None]
    t = {}
    t["fA"] = W((np.arange(512)[None, :] - 0.5) * j, 4096)        # forward pass 1: index T
    t["fB"] = W((np.arange(64)[None, :] - 0.5) * j, 512)          # forward pass 2: index t' = T & 63
    t["fC"] = W((np.arange(8)[None, :] - 0.5) * j, 64)            # forward pass 3: index d = T & 7
    t["fD"] = W(-np.arange(8), 16)                                 # forward pass 4: W16^{-k4} (constants)
    t["iA"] = np.conj(W(np.arange(512)[None, :] * j, 4096))       # inverse pass 1: index t'' = 64 k3 + 8 k2 + k1, register d = j
    t["iB"] = np.conj(W(np.arange(64)[None, :] * j, 512))         # inverse pass 2: index 8 k2 + k1, register c = j
    t["iC"] = np.conj(W(np.arange(8)[None, :] * j, 64))           # inverse pass 3: index k1 (wave-uniform), register b = j
    k = np.arange(512)                                             # thread (k1, k2, k3): k1 = wave, lane = 8 k2 + k3
    k1, k2, k3 = k >> 6, (k >> 3) & 7, k & 7
    kk = k1 + 8 * k2 + 64 * k3
    assert np.all(B[256:3840] < 0) and np.abs(np.abs(B[256:3840]) - 1).max() < 2e-7     # B = -1 away from DC (both halves: the half-sample shift carries the sign)
    rho = -B
    t["rho"] = np.stack([np.where(k3 < 4, rho[kk], 1.0), np.where(k3 >= 4, rho[kk + 3584], 1.0)], 1)
    return t


def block(z, tb, check_slots=True):
    """one block: z[4096] -> y[4096] = IFFT((-i) B FFT_{n - 1/2}(z)) / N, thread by thread"""
    T = np.arange(512)
    # thread T holds z[512 a + T]
    v = z.reshape(8, 512)                                          # [a][T]
    v = F8 @ v                                                     # [k1][T]
    v[1:] *= tb["fA"]                                              # W4096^{(T - 1/2) k1}
    # exchange 1 (cross-wave): writer (b = T >> 6, t' = T & 63) register k1 -> slot (8 k1 + b) 64 + t'; reader U = 64 k1 + t' register b
    lds = np.zeros(4096, complex)
    for k1 in range(8):
        lds[(8 * k1 + (T >> 6)) * 64 + (T & 63)] = v[k1]
    U = np.arange(512)
    v = np.stack([lds[(8 * (U >> 6) + b) * 64 + (U & 63)] for b in range(8)])    # [b][U], U = (k1, t')
    v = F8 @ v                                                     # [k2][U]
    v[1:] *= tb["fB"][:, U & 63]
    # exchange 2 (inside wave k1; chunk of 576 slots): writer lane t' = (c, d) register k2 -> slot (9 c + k2) 8 + d; reader lane (k2, d) register c
    lane, wave = U & 63, U >> 6
    ch = np.zeros((8, 576), complex)
    for k2 in range(8):
        ch[wave, (9 * (lane >> 3) + k2) * 8 + (lane & 7)] = v[k2]
    v = np.stack([ch[wave, (9 * c + (lane >> 3)) * 8 + (lane & 7)] for c in range(8)])   # [c][U], lane = (k2, d)
    v = F8 @ v                                                     # [k3][U]
    v[1:] *= tb["fC"][:, lane & 7]
    # exchange 3: writer lane (k2, d) register k3 -> slot (8 k2 + d) 9 + k3; reader lane (k2, k3) register d
    ch = np.zeros((8, 576), complex)
    for k3 in range(8):
        ch[wave, lane * 9 + k3] = v[k3]
    v = np.stack([ch[wave, (8 * (lane >> 3) + d) * 9 + (lane & 7)] for d in range(8)])   # [d][U], lane = (k2, k3)
    v = F8 @ v                                                     # [k4][U]
    v *= tb["fD"][:, None]
    # spectrum: thread (k1 = wave, k2, k3), register k4: (-i) B[k] = i rho[k], rho = 1 outside the rows next to DC (k4 = 0, k3 < 4 and k4 = 7, k3 >= 4)
    Y = v * 1j
    Y[0] *= tb["rho"][:, 0]
    Y[7] *= tb["rho"][:, 1]
    # inverse pass 1 over k4 -> d, twiddle conj W4096^{t'' d}, t'' = 64 k3 + 8 k2 + k1
    v = np.conj(F8) @ Y                                            # [d][U]
    tpp = 64 * (lane & 7) + 8 * (lane >> 3) + wave
    v[1:] *= tb["iA"][:, tpp]
    # exchange 3': writer lane (k2, k3) register d -> slot (8 k2 + d) 9 + k3; reader lane (k2, d) register k3
    ch = np.zeros((8, 576), complex)
    for d in range(8):
        ch[wave, (8 * (lane >> 3) + d) * 9 + (lane & 7)] = v[d]
    v = np.stack([ch[wave, lane * 9 + k3] for k3 in range(8)])     # [k3][U], lane = (k2, d)
    v = np.conj(F8) @ v                                            # [c][U]
    # twiddle conj W512^{(8 k2 + k1) c}
    v[1:] *= tb["iB"][:, 8 * (lane >> 3) + wave]
    # exchange 2': writer lane (k2, d) register c -> slot (9 c + k2) 8 + d; reader lane t' = (c, d) register k2
    ch = np.zeros((8, 576), complex)
    for c in range(8):
        ch[wave, (9 * c + (lane >> 3)) * 8 + (lane & 7)] = v[c]
    v = np.stack([ch[wave, (9 * (lane >> 3) + k2) * 8 + (lane & 7)] for k2 in range(8)])   # [k2][U], lane = (c, d)
    v = np.conj(F8) @ v                                            # [b][U]
    v[1:] *= tb["iC"][:, wave]                                     # conj W64^{k1 b}
    # exchange 1' (cross-wave): writer (k1 = wave, t') register b -> slot (8 k1 + b) 64 + t'; reader T = 64 b + t' register k1
    lds = np.zeros(4096, complex)
    for b in range(8):
        lds[(8 * wave + b) * 64 + lane] = v[b]
    v = np.stack([lds[(8 * k1 + (T >> 6)) * 64 + (T & 63)] for k1 in range(8)])   # [k1][T]
    v = np.conj(F8) @ v                                            # [a][T]
    return v.reshape(4096) / N


if __name__ == "__main__":
    import olasave_model as M
    rng = np.random.default_rng(2)
    for n_taps, hs in [(3072, 10), (4096, 8), (2048, 12)]:
        fir = M.hilbert_fir(n_taps)
        g = fir[1::2]
        Lg, R0 = len(g), (16 - hs) // 2
        B = M.tables(g)
        tb = tables(B)
        z = 0.3 * (rng.standard_normal(N) + 1j * rng.standard_normal(N))
        want = np.convolve(z, g)[Lg // 2:Lg // 2 + N]
        lo, hi = 256 * R0, N - 256 * R0
        e = np.abs(block(z, tb) - want)[lo:hi].max()
        e0 = np.abs(M.block(z, B, Lg, False, 1) - want)[lo:hi].max()
        print("taps %5d HS %2d: radix-8 block err %.2e (radix-16 model %.2e)" % (n_taps, hs, e, e0))
        assert e < 2e-6
    print("ok")
