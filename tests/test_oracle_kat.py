"""CPU: pin the oracle (oracle/prot_oracle.c) against every known answer we have
for this path -- tests/golden/known_answers.json, float64 models and the
reference's own dsp_peak_calc.h built as oracle/_ref/libref_peak.so."""
import json
import os

import numpy as np
import pytest

import orc
from conftest import harmonic_stack, xorshift_uniform

KA = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "known_answers.json")))


def db(x):
    return 20.0 * np.log10(np.maximum(np.asarray(x, np.float64), 1e-30))


# ---------------------------------------------------------------- FFT -----
@pytest.mark.parametrize("n", [8, 64, 512, 3072, 4096, 16384])
def test_fft_matches_numpy(n):
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n).astype(np.float32)
    X = orc.r2c(x)
    Xn = np.fft.rfft(x.astype(np.float64))
    assert np.abs(X - Xn).max() / np.abs(Xn).max() < 5e-7
    S = rng.standard_normal(n // 2 + 1) + 1j * rng.standard_normal(n // 2 + 1)
    y = orc.c2r(S.astype(np.complex64), n)
    S2 = S.copy()
    S2[0] = S2[0].real   # c2r ignores Im(DC) and Im(Nyquist) like FFTW
    S2[-1] = S2[-1].real
    yn = np.fft.irfft(S2, n) * n  # unnormalised
    assert np.abs(y - yn).max() / np.abs(yn).max() < 5e-7


# ---------------------------------------------------------------- FIR -----
@pytest.mark.parametrize("n", [2048, 3072, 4096, 8192, 16384])
def test_fir_known_answers(n):
    f = orc.fir_design(n)  # the reference's construction: spectrum -> c2r -> Hann
    cf = orc.fir_closed_form(n)
    assert np.abs(f - cf).max() < 2e-7
    assert np.abs(f).sum() == pytest.approx(KA["fir_abs_sum"][str(n)], abs=6e-5)
    tol = KA["fir_centre_tap"]["tolerance"]
    assert f[n // 2 - 1] == pytest.approx(KA["fir_centre_tap"]["value"], abs=tol)
    assert f[n // 2 + 1] == pytest.approx(-KA["fir_centre_tap"]["value"], abs=tol)
    even = (np.arange(n) - n // 2) % 2 == 0
    assert np.abs(f[even]).max() < 2e-7  # exactly 0 in the closed form


def test_lv2_filter_response():
    ka = KA["lv2_48k_filter_response"]
    f = orc.fir_design(3072).astype(np.float64)
    for hz in ("20", "50", "100", "1000", "23000"):
        w = 2 * np.pi * float(hz) / 48000
        h = np.sum(f * np.exp(-1j * w * (np.arange(3072) - 1536)))
        assert abs(h) == pytest.approx(ka[hz], abs=ka["tolerance"])
        assert np.degrees(np.angle(h)) == pytest.approx(90.0, abs=0.01)


# --------------------------------------------------------------- peak -----
def test_peak_matches_reference_header_bit_exact():
    ref = orc.load_ref_peak()
    if ref is None:
        pytest.skip("oracle/_ref/libref_peak.so not built (needs /root/reference at build time)")
    rng = np.random.default_rng(4)
    for n in (0, 1, 7, 8, 15, 16, 17, 255, 2048, 4099):
        for off in (0, 1, 3):  # unaligned heads exercise the scalar prologue of the AVX variant
            buf = (rng.standard_normal(n + off).astype(np.float32) * 0.7)[off:]
            buf = np.ascontiguousarray(buf)
            for cur in (0.0, 0.3, 5.0):
                a = orc.lib.orc_compute_peak(buf, n, cur)
                b = ref.ref_dsp_compute_peak(buf, n, cur)
                assert a == b, (n, off, cur)


def test_sincos_lut_sign_and_range():
    # Q7: negative radians, half-degree steps
    s, c = orc.sincos_lut(0)
    assert (abs(s), c) == (0.0, 1.0)
    s, c = orc.sincos_lut(180)  # 90 degrees
    assert s == pytest.approx(-1.0, abs=1e-7) and abs(c) < 1e-7
    s, c = orc.sincos_lut(60)
    assert s == pytest.approx(-0.5, abs=1e-7) and c == pytest.approx(np.sqrt(3) / 2, abs=1e-7)


# ---------------------------------------------------------------- LV2 -----
@pytest.mark.parametrize("rate", [44100, 48000, 88200, 96000, 176400, 192000])
def test_lv2_latency_and_geometry(rate):
    e = orc.Lv2Engine(rate, 1)
    assert e.latency == KA["lv2_latency"][str(rate)]
    key = "48000" if rate < 64000 else "96000" if rate < 128000 else "192000"
    assert e.info() == KA["lv2_geometry"][key]


@pytest.mark.parametrize("rate,bufsize,angle", [(48000, 1024, 90.0), (48000, 64, -120.0),
                                                (48000, 4096, 180.0), (96000, 1024, 45.0),
                                                (192000, 333, -77.0)])
def test_lv2_steady_state_matches_float64_lti(rate, bufsize, angle):
    """y[t] = cos th x[t-L] + sin th (fir*x)[t-P], th = -2 pi angle/360."""
    e = orc.Lv2Engine(rate, 1)
    info = e.info()
    L, N, P = e.latency, info["firlen"], info["parsiz"]
    n = 20000
    x = xorshift_uniform(n, seed=int(rate) + bufsize)
    y = e.run_stream(x[None, :], angle, bufsize)[0]
    u = np.convolve(x.astype(np.float64), orc.fir_closed_form(N))[:n]
    xd = np.concatenate([np.zeros(L), x[:n - L]])
    ud = np.concatenate([np.zeros(P), u[:n - P]])
    th = -2 * np.pi * angle / 360
    ym = np.cos(th) * xd + np.sin(th) * ud
    s = 6000  # past the initial ramp from 0 degrees (Q10)
    assert np.abs(y[s:] - ym[s:]).max() < 2 * KA["steady_state_tolerance"]["lv2"]
    assert np.all(y[:P] == 0)


def test_lv2_sinusoid_phase_lag():
    rate, f0, ang = 48000, 1000.0, 60.0
    n = 16384
    t = np.arange(n) / rate
    x = np.cos(2 * np.pi * f0 * t).astype(np.float32)
    e = orc.Lv2Engine(rate, 1)
    y = e.run_stream(x[None, :], ang, 1024)[0]
    L = e.latency
    seg = slice(8192, 16384)
    ref = np.exp(-1j * 2 * np.pi * f0 * (t[seg] - L / rate))
    ph = np.degrees(np.angle(np.sum(y[seg] * ref)))  # phase of y relative to the delayed input
    assert ph == pytest.approx(-ang, abs=KA["lv2_sinusoid"]["tolerance_deg"])


def test_lv2_initial_ramp_q10_and_clamp():
    """First blocks ramp from 0 deg at <= parsiz*1e-6 turns/sample; a 180 deg target
    takes 0.5 / (256e-6) = 1953 frames."""
    e = orc.Lv2Engine(48000, 1)
    n = 8192
    x = np.ones(n, np.float32) * 0.25
    y = e.run_stream(x[None, :], 180.0, 256)[0]
    # after the ramp and the latency: cos(pi) * 0.25 (the Hilbert branch of DC is 0)
    assert y[-1] == pytest.approx(-0.25, abs=1e-5)
    assert np.all(y[:256] == 0)


def test_lv2_meters_follow_output():
    e = orc.Lv2Engine(48000, 1)
    x = xorshift_uniform(8192, seed=12)[None, :]
    y = e.run_stream(x, 30.0, 1024)
    m, p, d = e.meters(0)
    assert p[1] == pytest.approx(np.abs(y).max(), rel=1e-6)
    assert p[0] <= np.abs(x).max() + 1e-7


# ---------------------------------------------------------------- CLI -----
@pytest.mark.parametrize("blksiz,a", [(2048, 90), (2048, 299), (4096, 180), (8192, 340)])
def test_cli_apply_matches_float64_model(blksiz, a):
    n = blksiz * 6
    x = xorshift_uniform(n, seed=blksiz + a)
    y = orc.cli_apply_file_mono(blksiz, x, a)
    u = np.convolve(x.astype(np.float64), orc.fir_closed_form(blksiz))
    th = -a * np.pi / 360
    ym = np.cos(th) * x + np.sin(th) * u[blksiz // 2:blksiz // 2 + n]
    assert np.abs(y - ym).max() < 2 * KA["steady_state_tolerance"]["cli"]


def test_cli_apply_angle_wrap_q9():
    ka = KA["cli_apply_angle_wrap_q9"]
    rate, f0, P = 48000, 1000.0, 2048
    n = P * 8
    t = np.arange(n) / rate
    x = np.cos(2 * np.pi * f0 * t).astype(np.float32)
    for deg in ("30", "-30", "180", "-180"):
        a = int(round(float(deg) * 2))  # cli/phase-rotate.cc:730
        y = orc.cli_apply_file_mono(P, x, a)
        seg = slice(2 * P, 6 * P)
        ph = np.degrees(np.angle(np.sum(y[seg] * np.exp(-1j * 2 * np.pi * f0 * t[seg]))))
        assert (ph - ka[deg] + 180) % 360 - 180 == pytest.approx(0, abs=ka["tolerance_deg"])


def test_cli_first_block_quirk_q1():
    ka = KA["cli_first_block_quirk_q1"]
    P = 2048
    x = np.zeros(P * 3, np.float32)
    x[ka["impulse_index"]] = ka["impulse_value"]
    e = orc.CliEngine(P, 1)
    e.analyze_file(x, 0, 360, 1)
    pk = e.peaks()[0]
    assert db(pk[0]) == pytest.approx(ka["db_at_0deg"], abs=0.01)
    assert db(pk[2]) < ka["db_at_1deg_below"]  # 1 degree = index 2
    assert -45 < db(pk[2]) < -35


def test_cli_angle_loop_bounds_q5():
    import importlib
    from __graft_entry__ import load_package
    load_package()
    sd = importlib.import_module("phaserotate_lv2_amd.sweep_dist")
    assert len(sd.angle_list(0, 360, 24)) == KA["cli_angle_loop_q5"]["coarse_count"]
    assert sd.angle_list(0, 360, 24)[-1] == 336
    fine = sd.angle_list(100 - 12, 100 + 12 + 1, 1)
    assert len(fine) == KA["cli_angle_loop_q5"]["fine_count"] and fine[0] == 88 and fine[-1] == 112
    # the oracle visits exactly those slots
    P = 2048
    x = xorshift_uniform(P * 2, seed=1)
    e = orc.CliEngine(P, 1)
    e.analyze_file(x, 0, 360, 24)
    pk = e.peaks()[0]
    assert set(np.nonzero(pk)[0]) == set(range(0, 360, 24))


def test_cli_sweep_matches_direct_float64():
    """pk[a] = max |cos th_a x[t - P/2] + sin th_a u[t]| with the Q1/Q2/Q4 rules."""
    P = 2048
    n = P * 5 + 300
    x = (0.8 * harmonic_stack(n) + 0.1 * xorshift_uniform(n, seed=2)).astype(np.float32)
    e = orc.CliEngine(P, 1)
    e.analyze_file(x, 0, 360, 1)
    pk = e.peaks()[0]
    nb = (n + P - 1) // P + 1
    xp = np.zeros(nb * P)
    xp[:n] = x
    u = np.convolve(xp, orc.fir_closed_form(P))[:nb * P]
    d = np.concatenate([np.zeros(P // 2), xp[:nb * P - P // 2]])
    d[:P] = 0  # Q1: the first block pairs hil[P/2..P) with silence and skips hil[0..P/2)
    for a in (1, 2, 77, 180, 233, 359):
        th = -a * np.pi / 360
        v = np.abs(np.cos(th) * d + np.sin(th) * u)[P // 2:]
        assert db(pk[a]) == pytest.approx(db(v.max()), abs=1e-3)
    assert pk[0] == np.abs(x).max()  # Q4


def test_cli_find_min_stride1_equals_argmin_and_coarse_fine_agree():
    P = 2048
    n = 48000
    x = (0.8 * harmonic_stack(n) + 0.02 * xorshift_uniform(n, seed=5)).astype(np.float32)
    ma1, fo1, rz1, rm1, tbl1 = orc.cli_find_min(P, 1, x, stride=1)
    e = orc.CliEngine(P, 1)
    e.analyze_file(x, 0, 360, 1)
    pk = e.peaks()[0]
    # stride 1: the LAST global minimum wins (ascending std::map, :859-865)
    last_min = int(np.nonzero(pk == pk.min())[0][-1])
    assert (ma1[0] + 360) % 360 == last_min and fo1[0] == 1
    assert rm1[0] == pk.min() and rz1[0] == pk[0]
    ma24, fo24, _, rm24, _ = orc.cli_find_min(P, 1, x, stride=24)
    assert fo24[0] == 1
    assert rm24[0] <= pk[::24].min()
    # the fine pass lands on a local minimum of the full table
    a = (ma24[0] + 360) % 360
    assert pk[a] <= pk[(a - 1) % 360] and pk[a] <= pk[(a + 1) % 360]


def test_oracle_fft_agrees_with_an_fftw_interface_library_if_the_host_has_one():
    """A further pin of the restated FFTW semantics (unnormalised r2c / c2r, Im(DC) and Im(Nyquist) ignored, bin order):
    when the host has libfftw3f.so.3 or MKL's FFTW3 interface (libmkl_rt.so: present in this image), the two engines
    give the same samples on it as on the oracle's own FFT.  (bench.py's cpu_baseline times the oracle on that library.)"""
    import ctypes as C
    import os
    os.environ.setdefault("MKL_NUM_THREADS", "1")
    orc.lib.orc_fft_use_external.restype = C.c_int
    orc.lib.orc_fft_backend_name.restype = C.c_char_p
    x = xorshift_uniform(40000, seed=5).reshape(1, -1)
    y_own = orc.Lv2Engine(48000, 1).run_stream(x, -60.0, 1024)
    blk = xorshift_uniform(2048 * 6, seed=6)
    o = orc.CliEngine(2048, 1)
    c_own = np.concatenate([o.apply(blk[b * 2048:(b + 1) * 2048], [77]) for b in range(6)])
    kind = orc.lib.orc_fft_use_external(1)
    try:
        if kind == 0:
            pytest.skip("no libfftw3f.so.3 / libmkl_rt.so on this host")
        fir_ext = orc.fir_design(3072)
        y_ext = orc.Lv2Engine(48000, 1).run_stream(x, -60.0, 1024)
        o = orc.CliEngine(2048, 1)
        c_ext = np.concatenate([o.apply(blk[b * 2048:(b + 1) * 2048], [77]) for b in range(6)])
    finally:
        orc.lib.orc_fft_use_external(0)
    assert np.abs(fir_ext - orc.fir_design(3072)).max() < 5e-7   # the 3072-point design transform (2^10 x 3); taps up to 0.64
    assert np.abs(y_ext - y_own).max() < 1e-6
    assert np.abs(c_ext - c_own).max() < 1e-6
