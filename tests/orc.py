"""ctypes binding of the CPU oracle (oracle/liborc.so) -- TEST INFRASTRUCTURE.

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() import
this.  The product path (phaserotate.lv2_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
MAXSAMPLE = 360

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build():
    """(Re)build oracle/liborc.so and, if /root/reference exists, oracle/_ref."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True,
                   stdout=subprocess.DEVNULL)


def _load():
    so = os.path.join(ORACLE_DIR, "liborc.so")
    if not os.path.exists(so):
        build()
    lib = C.CDLL(so)
    lib.orc_fft_r2c.argtypes = [C.c_int, _f32p, _f32p]
    lib.orc_fft_c2r.argtypes = [C.c_int, _f32p, _f32p]
    lib.orc_fft_r2c_f64.argtypes = [C.c_int, _f64p, _f64p]
    lib.orc_fft_c2r_f64.argtypes = [C.c_int, _f64p, _f64p]
    lib.orc_compute_peak.argtypes = [_f32p, C.c_uint32, C.c_float]
    lib.orc_compute_peak.restype = C.c_float
    lib.orc_sincos_lut.argtypes = [C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.orc_fir_design.argtypes = [C.c_int, _f32p]
    lib.orc_fir_closed_form.argtypes = [C.c_int, _f64p]
    lib.orc_cli_proc_new.restype = C.c_void_p
    lib.orc_cli_proc_new.argtypes = [C.c_uint32]
    lib.orc_cli_proc_free.argtypes = [C.c_void_p]
    lib.orc_cli_proc_hilbert.argtypes = [C.c_void_p, _f32p, _f32p, _f32p]
    lib.orc_cli_proc_process.argtypes = [C.c_void_p, _f32p, _f32p, _f32p, C.c_int]
    lib.orc_cli_proc_ffir.restype = C.POINTER(C.c_float)
    lib.orc_cli_proc_ffir.argtypes = [C.c_void_p]
    lib.orc_cli_new.restype = C.c_void_p
    lib.orc_cli_new.argtypes = [C.c_uint32, C.c_uint32]
    lib.orc_cli_free.argtypes = [C.c_void_p]
    lib.orc_cli_reset.argtypes = [C.c_void_p]
    lib.orc_cli_analyze.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.orc_cli_apply.argtypes = [C.c_void_p, _f32p, _i32p]
    lib.orc_cli_peak.restype = C.c_float
    lib.orc_cli_peak.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.orc_cli_peak_all.restype = C.c_float
    lib.orc_cli_peak_all.argtypes = [C.c_void_p, C.c_int]
    lib.orc_cli_analyze_file.argtypes = [C.c_void_p, _f32p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.orc_cli_find_min.argtypes = [C.c_uint32, C.c_uint32, _f32p, C.c_uint64, C.c_int, C.c_int,
                                     _i32p, _i32p, _f32p, _f32p, C.c_void_p]
    lib.orc_cli_apply_file_mono.argtypes = [C.c_uint32, _f32p, C.c_uint64, C.c_int, _f32p]
    lib.orc_lv2_new.restype = C.c_void_p
    lib.orc_lv2_new.argtypes = [C.c_double, C.c_uint32]
    lib.orc_lv2_free.argtypes = [C.c_void_p]
    lib.orc_lv2_activate.argtypes = [C.c_void_p]
    lib.orc_lv2_latency.restype = C.c_uint32
    lib.orc_lv2_latency.argtypes = [C.c_void_p]
    lib.orc_lv2_info.argtypes = [C.c_void_p] + [C.POINTER(C.c_uint32)] * 4
    lib.orc_lv2_run.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                C.c_uint32, _f32p, C.c_void_p]
    lib.orc_lv2_meters.argtypes = [C.c_void_p, C.c_uint32, _f32p, _f32p, _f32p]
    return lib


lib = _load()


def load_ref_peak():
    """The reference's own dsp_compute_peak (oracle/_ref/libref_peak.so) or None."""
    so = os.path.join(ORACLE_DIR, "_ref", "libref_peak.so")
    if not os.path.exists(so):
        return None
    r = C.CDLL(so)
    r.ref_dsp_compute_peak.argtypes = [_f32p, C.c_uint32, C.c_float]
    r.ref_dsp_compute_peak.restype = C.c_float
    return r


# ---------------------------------------------------------------- helpers --
def r2c(x):
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty(x.size + 2, np.float32)
    lib.orc_fft_r2c(x.size, x, out)
    return out[0::2] + 1j * out[1::2]


def c2r(spec, n):
    s = np.empty(n + 2, np.float32)
    s[0::2] = spec.real
    s[1::2] = spec.imag
    out = np.empty(n, np.float32)
    lib.orc_fft_c2r(n, s, out)
    return out


def fir_design(n):
    f = np.empty(n, np.float32)
    lib.orc_fir_design(n, f)
    return f


def fir_closed_form(n):
    f = np.empty(n, np.float64)
    lib.orc_fir_closed_form(n, f)
    return f


def sincos_lut(a):
    s, c = C.c_float(), C.c_float()
    lib.orc_sincos_lut(a, C.byref(s), C.byref(c))
    return s.value, c.value


class CliEngine:
    """PhaseRotate (cli/phase-rotate.cc:251-485) restated on the CPU."""

    def __init__(self, blksiz, n_chn):
        self.blksiz, self.n_chn = blksiz, n_chn
        self.h = lib.orc_cli_new(blksiz, n_chn)

    def __del__(self):
        if getattr(self, "h", None):
            lib.orc_cli_free(self.h)
            self.h = None

    def reset(self):
        lib.orc_cli_reset(self.h)

    def analyze(self, buf, ang_start=0, ang_end=180, ang_stride=1, chn=-1, start=False):
        buf = np.ascontiguousarray(buf, np.float32).reshape(-1)
        assert buf.size == self.blksiz * self.n_chn
        lib.orc_cli_analyze(self.h, buf, ang_start, ang_end, ang_stride, chn, int(start))

    def apply(self, buf, angles):
        buf = np.ascontiguousarray(buf, np.float32).reshape(-1).copy()
        assert buf.size == self.blksiz * self.n_chn
        lib.orc_cli_apply(self.h, buf, np.asarray(angles, np.int32))
        return buf

    def analyze_file(self, data, ang_start, ang_end, ang_stride, chn=-1):
        data = np.ascontiguousarray(data, np.float32).reshape(-1)
        lib.orc_cli_analyze_file(self.h, data, data.size // self.n_chn,
                                 ang_start, ang_end, ang_stride, chn)

    def peaks(self):
        return np.array([[lib.orc_cli_peak(self.h, c, a) for a in range(MAXSAMPLE)]
                         for c in range(self.n_chn)], np.float32)


def cli_find_min(blksiz, n_chn, data, stride=24, link=False):
    data = np.ascontiguousarray(data, np.float32).reshape(-1)
    ma = np.zeros(n_chn, np.int32)
    fo = np.zeros(n_chn, np.int32)
    rz = np.zeros(n_chn, np.float32)
    rm = np.zeros(n_chn, np.float32)
    tbl = np.zeros((n_chn, MAXSAMPLE), np.float32)
    lib.orc_cli_find_min(blksiz, n_chn, data, data.size // n_chn, stride, int(link),
                         ma, fo, rz, rm, tbl.ctypes.data_as(C.c_void_p))
    return ma, fo, rz, rm, tbl


def cli_apply_file_mono(blksiz, data, angle):
    data = np.ascontiguousarray(data, np.float32)
    out = np.zeros_like(data)
    lib.orc_cli_apply_file_mono(blksiz, data, data.size, int(angle), out)
    return out


class Lv2Engine:
    """LV2 plugin DSP (src/phaserotate.c) restated on the CPU."""

    def __init__(self, rate, n_chn=1):
        self.n_chn = n_chn
        self.h = lib.orc_lv2_new(float(rate), n_chn)
        assert self.h
        lib.orc_lv2_activate(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            lib.orc_lv2_free(self.h)
            self.h = None

    @property
    def latency(self):
        return lib.orc_lv2_latency(self.h)

    def info(self):
        v = [C.c_uint32() for _ in range(4)]
        lib.orc_lv2_info(self.h, *[C.byref(x) for x in v])
        return dict(zip(("fftlen", "firlen", "parsiz", "n_segm"), [x.value for x in v]))

    def activate(self):
        lib.orc_lv2_activate(self.h)

    def run(self, x, angle_deg):
        """x: [n_chn, n] float32 -> y [n_chn, n]; one run() call."""
        x = np.ascontiguousarray(np.atleast_2d(x), np.float32)
        y = x.copy()
        n = x.shape[1]
        ptrs = (C.c_void_p * self.n_chn)(*[y[c].ctypes.data for c in range(self.n_chn)])
        ang = np.ascontiguousarray(np.broadcast_to(np.asarray(angle_deg, np.float32), (self.n_chn,)))
        lvl = np.zeros((self.n_chn, 2), np.float32)
        lib.orc_lv2_run(self.h, ptrs, ptrs, n, ang, lvl.ctypes.data_as(C.c_void_p))
        self.last_levels = lvl
        return y

    def run_stream(self, x, angle_deg, bufsize):
        x = np.ascontiguousarray(np.atleast_2d(x), np.float32)
        out = np.empty_like(x)
        for s in range(0, x.shape[1], bufsize):
            out[:, s:s + bufsize] = self.run(x[:, s:s + bufsize], angle_deg)
        return out

    def meters(self, chn=0):
        m, p, d = (np.zeros(2, np.float32) for _ in range(3))
        lib.orc_lv2_meters(self.h, chn, m, p, d)
        return m, p, d
