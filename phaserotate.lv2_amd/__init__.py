"""phaserotate.lv2_amd -- host-side mirror of the reference's interfaces over the
C ABI of libprot_hip.so (include/prot_hip.h).

This package is plumbing for tests and bench.py: ctypes bindings that look like
the reference's own classes (PhaseRotate of cli/phase-rotate.cc:251-485, the
LV2 instance of src/phaserotate.c).  All arithmetic happens in the gfx950
kernels behind the C ABI.  There is NO fallback: if the shared library is
missing, or no HIP device is usable, importing / constructing fails loudly.

The directory name contains a dot, so import it through
``__graft_entry__.load_package()`` (module name ``phaserotate_lv2_amd``).
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
# PROT_LIB_PATH: tuning only (tools/ab_bench.py times alternative builds of the same sources)
# the SHIPPED library (what the LV2 plugin and the CLI link: include/prot_hip.h's symbols and nothing else) and the test-hook build of
# the same sources (-DPROT_TESTHOOKS: + csrc/prot_internal.h), which this mirror -- tests/ and bench.py plumbing -- loads when it is there
SHIPPED_LIB_PATH = os.path.join(_PKG, "lib", "libprot_hip.so")
_HOOKS = os.path.join(_PKG, "lib", "libprot_hip_hooks.so")
LIB_PATH = os.environ.get("PROT_LIB_PATH") or (_HOOKS if os.path.exists(_HOOKS) else SHIPPED_LIB_PATH)

MAXSAMPLE = 360          # cli/phase-rotate.cc:39
SUBSAMPLE = 2            # cli/phase-rotate.cc:38
KIND_CLI, KIND_LV2 = 0, 1
F_STATELESS, F_HILBERT, F_STREAM_START, F_STREAM_END = 1, 2, 4, 8
CFG_SINGLE_FFT_PER_HOP = 1
CFG_DIRECT_FORM = 2
CFG_POLYPHASE = 4
CFG_OVERLAP_ADD = 8
CFG_PIPELINE = 16
CFG_BLOCK4096 = 32
CFG_STATIC_RUNS = 64
FORM_DIRECT, FORM_POLY_OLA, FORM_POLY_SAVE, FORM_POLY_SAVE8K, FORM_POLY_SAVE16K = 0, 1, 2, 3, 4

# every symbol include/prot_hip.h declares (tests check the .so exports them all)
ABI_SYMBOLS = (
    "prot_abi_version", "prot_strerror", "prot_last_error", "prot_sincos_lut",
    "prot_engine_create", "prot_engine_destroy", "prot_engine_reset", "prot_engine_info",
    "prot_engine_fir", "prot_engine_sync",
    "prot_apply_dev", "prot_apply_dev_ramp", "prot_apply_dev_f16", "prot_apply_dev_f16_ramp", "prot_sweep_dev",
    "prot_lv2_run", "prot_lv2_set_max_block", "prot_lv2_set_deferred", "prot_lv2_deferred_stats", "prot_lv2_get_angle", "prot_cli_apply", "prot_cli_analyze", "prot_cli_analyze_file",
    "prot_cli_apply_file", "prot_cli_peak", "prot_cli_peak_all", "prot_cli_peaks",
    "prot_cli_peaks_dev", "prot_allreduce_peaks", "prot_cli_analyze_file_multi", "prot_sweep_stats",
    "prot_sweep_stats_ex", "prot_sweep_stats_fused", "prot_cli_file_resident",
    "prot_sweep_seed_dev", "prot_engine_prime_dev", "prot_sweep_shard_seed_dev",
)


class ProtError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("kind", C.c_int32), ("param", C.c_double), ("n_chn", C.c_uint32),
                ("device", C.c_int32), ("fftlen", C.c_uint32), ("flags", C.c_uint32)]


class Info(C.Structure):
    _fields_ = [("kind", C.c_int32), ("n_chn", C.c_uint32), ("fftlen", C.c_uint32),
                ("hop", C.c_uint32), ("firlen", C.c_uint32), ("latency", C.c_uint32),
                ("dry_delay", C.c_uint32), ("conv_delay", C.c_uint32), ("history", C.c_uint32),
                ("blksiz", C.c_uint32), ("partitions", C.c_uint32), ("run_align", C.c_uint32),
                ("form", C.c_uint32), ("wg_per_cu", C.c_uint32)]


_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_lib = None


def _share_hip_runtime_with_torch():
    """PyTorch wheels bundle their own libamdhip64/libhsa-runtime64.  Two HIP
    runtimes in one process cannot both own the GPU, and a torch stream or
    tensor is only meaningful to the runtime that made it.  So when torch is
    importable, load ITS libamdhip64 with RTLD_GLOBAL before libprot_hip.so:
    the library's HIP calls then bind to that one runtime (global scope is
    searched before a library's own DT_NEEDED), and torch allocations /
    streams / RCCL and the engine share it.  Pure C hosts (the CLI, the LV2
    plugin) never load torch and use /opt/rocm's runtime."""
    try:
        import torch  # noqa: F401
    except ImportError:
        return
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load_library():
    """dlopen libprot_hip.so; raises if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ProtError(
            "HIP extension missing: %s not built (run __graft_entry__.build()); "
            "there is no CPU fallback" % LIB_PATH)
    _share_hip_runtime_with_torch()
    lib = C.CDLL(LIB_PATH)
    lib.prot_abi_version.restype = C.c_int
    lib.prot_strerror.restype = C.c_char_p
    lib.prot_strerror.argtypes = [C.c_int]
    lib.prot_last_error.restype = C.c_char_p
    lib.prot_sincos_lut.argtypes = [C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.prot_engine_create.restype = C.c_void_p
    lib.prot_engine_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_int)]
    lib.prot_engine_destroy.argtypes = [C.c_void_p]
    lib.prot_engine_reset.argtypes = [C.c_void_p]
    lib.prot_engine_info.argtypes = [C.c_void_p, C.POINTER(Info)]
    lib.prot_engine_fir.argtypes = [C.c_void_p, _f32p]
    lib.prot_engine_sync.argtypes = [C.c_void_p]
    lib.prot_apply_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_uint64, C.c_void_p,
                                   C.c_int64, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p]
    lib.prot_apply_dev_ramp.argtypes = lib.prot_apply_dev.argtypes
    lib.prot_apply_dev_f16.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_uint64, C.c_void_p,
                                       C.c_int64, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p]
    lib.prot_apply_dev_f16_ramp.argtypes = lib.prot_apply_dev_f16.argtypes
    lib.prot_lv2_set_max_block.argtypes = [C.c_void_p, C.c_uint32]
    lib.prot_lv2_set_deferred.argtypes = [C.c_void_p, C.c_uint32]
    lib.prot_lv2_deferred_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.prot_sweep_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_uint64, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p]
    lib.prot_sweep_seed_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_uint64, C.c_int, C.c_int,
                                        C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
    lib.prot_sweep_seed_dev.restype = C.c_int
    lib.prot_engine_prime_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_uint64, C.c_void_p]
    lib.prot_engine_prime_dev.restype = C.c_int
    lib.prot_sweep_shard_seed_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_uint64, C.c_void_p, C.c_int64, C.c_uint64, C.c_int, C.c_int,
                                              C.c_int, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
    lib.prot_sweep_shard_seed_dev.restype = C.c_int
    lib.prot_lv2_run.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                 C.c_uint32, _f32p]
    lib.prot_lv2_get_angle.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_float)]
    lib.prot_sweep_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.prot_sweep_stats_ex.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    lib.prot_sweep_stats_fused.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    lib.prot_cli_file_resident.argtypes = [C.c_void_p, C.c_int]
    lib.prot_cli_apply.argtypes = [C.c_void_p, _f32p, _i32p]
    lib.prot_cli_analyze.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.prot_cli_analyze_file.argtypes = [C.c_void_p, _f32p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.prot_cli_apply_file.argtypes = [C.c_void_p, _f32p, C.c_uint64, _i32p, _f32p]
    lib.prot_cli_peak.restype = C.c_float
    lib.prot_cli_peak.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.prot_cli_peak_all.restype = C.c_float
    lib.prot_cli_peak_all.argtypes = [C.c_void_p, C.c_int]
    lib.prot_cli_peaks.argtypes = [C.c_void_p, _f32p]
    lib.prot_cli_peaks_dev.restype = C.c_void_p
    lib.prot_cli_peaks_dev.argtypes = [C.c_void_p]
    lib.prot_allreduce_peaks.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    lib.prot_cli_analyze_file_multi.argtypes = [C.POINTER(C.c_void_p), C.c_int, _f32p, C.c_uint64,
                                                C.c_int, C.c_int, C.c_int, C.c_int]
    _lib = lib
    return lib


# csrc/prot_internal.h: process-wide switches for tests and bench.py (NOT part of the drop-in boundary)
TEST_SWEEP_UNPRUNED, TEST_EDGE_PATH_ONLY, TEST_LV2_STAGED, TEST_ALLREDUCE, TEST_SWEEP_UNFUSED, TEST_LV2_NO_SERVER, TEST_LV2_NO_BAR, TEST_SEED_MFMA, TEST_FUSED_SPLIT_LOG2, TEST_F16_NO_MFMA = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9
TEST_NO_SHORT_CHAIN = 10


def set_test_option(option, value):
    """prot_internal_set_test_option: the shipped library reads no environment variables; what tests and the bench
    switch (unpruned sweep, every block through the staged path, ...) goes through this hook."""
    lib = load_library()
    lib.prot_internal_set_test_option.argtypes = [C.c_int, C.c_int]
    _check(lib.prot_internal_set_test_option(int(option), int(value)), "prot_internal_set_test_option")


def _check(rc, what):
    if rc != 0:
        lib = load_library()
        raise ProtError("%s failed: %s (%s)" % (
            what, lib.prot_strerror(rc).decode(), lib.prot_last_error().decode()))


def sincos_lut(a):
    """SinCosLut::sincos (cli/phase-rotate.cc:57-67) as the device uses it."""
    lib = load_library()
    s, c = C.c_float(), C.c_float()
    lib.prot_sincos_lut(a, C.byref(s), C.byref(c))
    return s.value, c.value


class Engine:
    """Opaque prot_engine handle + the device-resident streaming API."""

    def __init__(self, kind, param, n_chn=1, device=0, fftlen=0, flags=0):
        self._lib = load_library()
        cfg = Config(kind, float(param), n_chn, device, fftlen, flags)
        st = C.c_int(0)
        self._h = self._lib.prot_engine_create(C.byref(cfg), C.byref(st))
        if not self._h:
            raise ProtError("prot_engine_create failed: %s (%s)" % (
                self._lib.prot_strerror(st.value).decode(), self._lib.prot_last_error().decode()))
        inf = Info()
        _check(self._lib.prot_engine_info(self._h, C.byref(inf)), "prot_engine_info")
        self.info = inf
        self.n_chn = n_chn

    def close(self):
        if getattr(self, "_h", None):
            self._lib.prot_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        _check(self._lib.prot_engine_reset(self._h), "prot_engine_reset")

    def sync(self):
        _check(self._lib.prot_engine_sync(self._h), "prot_engine_sync")

    def fir(self):
        f = np.empty(self.info.firlen, np.float32)
        _check(self._lib.prot_engine_fir(self._h, f), "prot_engine_fir")
        return f

    # -- device-resident path: arguments are raw device addresses (ints) ------
    def apply_dev(self, d_in, in_stride, n_in, d_out, out_stride, n_out, rot, flags=0, stream=0):
        rot = np.ascontiguousarray(rot, np.float32).reshape(-1)
        assert rot.size == 2 * self.n_chn
        _check(self._lib.prot_apply_dev(self._h, d_in, in_stride, n_in, d_out, out_stride, n_out,
                                        rot.ctypes.data, flags, stream), "prot_apply_dev")

    def apply_dev_f16(self, d_in, in_stride, n_in, d_out, out_stride, n_out, rot, stream=0, flags=F_STATELESS):
        """fp16 sample path: device arrays of torch.float16 / IEEE half (flags=0: the engine carries the stream's history)."""
        rot = np.ascontiguousarray(rot, np.float32).reshape(-1)
        assert rot.size == 2 * self.n_chn
        _check(self._lib.prot_apply_dev_f16(self._h, d_in, in_stride, n_in, d_out, out_stride, n_out,
                                            rot.ctypes.data, flags, stream), "prot_apply_dev_f16")

    def apply_dev_f16_ramp(self, d_in, in_stride, n_in, d_out, out_stride, n_out, d_rot, flags=0, stream=0):
        """fp16 samples, per-frame rotation: d_rot = device float32 [n_chn][n_out][2] {ca, sa}."""
        _check(self._lib.prot_apply_dev_f16_ramp(self._h, d_in, in_stride, n_in, d_out, out_stride, n_out,
                                                 d_rot, flags, stream), "prot_apply_dev_f16_ramp")

    def lv2_set_max_block(self, frames):
        _check(self._lib.prot_lv2_set_max_block(self._h, frames), "prot_lv2_set_max_block")

    def lv2_set_deferred(self, frames):
        """deferred run(): a call posts its block and returns the output `frames` frames older without waiting (0: off)"""
        _check(self._lib.prot_lv2_set_deferred(self._h, frames), "prot_lv2_set_deferred")

    def lv2_deferred_stats(self):
        late, dropped = C.c_uint64(0), C.c_uint64(0)
        _check(self._lib.prot_lv2_deferred_stats(self._h, C.byref(late), C.byref(dropped)), "prot_lv2_deferred_stats")
        return {"late": late.value, "dropped": dropped.value}

    def hilbert_dev(self, d_in, in_stride, n_in, d_out, out_stride, n_out, flags=0, stream=0):
        _check(self._lib.prot_apply_dev(self._h, d_in, in_stride, n_in, d_out, out_stride, n_out,
                                        None, flags | F_HILBERT, stream), "prot_apply_dev")

    def apply_dev_ramp(self, d_in, in_stride, n_in, d_out, out_stride, n_out, d_rot, flags=0, stream=0):
        _check(self._lib.prot_apply_dev_ramp(self._h, d_in, in_stride, n_in, d_out, out_stride,
                                             n_out, d_rot, flags, stream), "prot_apply_dev_ramp")

    def shard_align_blocks(self):
        """Blocks of `blksiz` frames per alignment unit: shards / chunks that start on a multiple of it are
        bit-identical to a single pass (lcm(blksiz, info.run_align) / blksiz)."""
        import math
        P, A = int(self.info.blksiz), max(1, int(self.info.run_align))
        return P // math.gcd(P, A) * A // P

    def sweep_stats(self):
        """(evaluated, seen) 32-frame tiles of the sweep since the last reset (prot_sweep_stats)."""
        a, b = C.c_uint64(0), C.c_uint64(0)
        _check(self._lib.prot_sweep_stats(self._h, C.byref(a), C.byref(b)), "prot_sweep_stats")
        return a.value, b.value

    def sweep_stats_ex(self):
        """{evaluated, seen, untested (tiles), survivors (frames of the tested tiles)} (prot_sweep_stats_ex)."""
        v = (C.c_uint64 * 4)()
        _check(self._lib.prot_sweep_stats_ex(self._h, v), "prot_sweep_stats_ex")
        return dict(evaluated=v[0], seen=v[1], untested=v[2], survivors=v[3])

    def sweep_stats_fused(self):
        """{tested, listed}: channel-frames the fused Hilbert + test pass has seen / kept since the last reset (prot_sweep_stats_fused)."""
        v = (C.c_uint64 * 2)()
        _check(self._lib.prot_sweep_stats_fused(self._h, v), "prot_sweep_stats_fused")
        return dict(tested=v[0], listed=v[1])

    def time_lv2_run(self, frames, calls):
        """mean microseconds per prot_lv2_run call, looped in C (test hook prot_internal_time_lv2_run)"""
        us = (C.c_double * 3)()
        self._lib.prot_internal_time_lv2_run.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.POINTER(C.c_double)]
        self._lib.prot_internal_time_lv2_run.restype = C.c_int
        _check(self._lib.prot_internal_time_lv2_run(self._h, frames, calls, us), "prot_internal_time_lv2_run")
        self.last_worst_call = (us[1], int(us[2]))
        return us[0]

    def time_lv2_run_paced(self, frames, calls, gap_us):
        """(mean, slowest) microseconds of a prot_lv2_run call BY ITSELF, `gap_us` of busy-waiting between calls, looped in C"""
        us = (C.c_double * 3)()
        self._lib.prot_internal_time_lv2_run_paced.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_uint, C.POINTER(C.c_double)]
        self._lib.prot_internal_time_lv2_run_paced.restype = C.c_int
        _check(self._lib.prot_internal_time_lv2_run_paced(self._h, frames, calls, gap_us, us), "prot_internal_time_lv2_run_paced")
        return us[0], us[1]

    def server_stats(self):
        """{calls, starts}: run() calls the resident server kernel has taken / kernels started for it (test hook)."""
        v = (C.c_ulonglong * 2)()
        self._lib.prot_internal_server_stats.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
        self._lib.prot_internal_server_stats.restype = C.c_int
        _check(self._lib.prot_internal_server_stats(self._h, v), "prot_internal_server_stats")
        return dict(calls=v[0], starts=v[1])

    def sweep_dev(self, d_in, in_stride, n_frames, ang_start, ang_end, ang_stride, chn=-1,
                  d_peaks=0, flags=0, stream=0):
        _check(self._lib.prot_sweep_dev(self._h, d_in, in_stride, n_frames, ang_start, ang_end,
                                        ang_stride, chn, d_peaks or None, flags, stream),
               "prot_sweep_dev")

    def sweep_seed_dev(self, d_in, in_stride, n_frames, ang_start, ang_end, ang_stride, chn=-1,
                       d_peaks=0, flags=0, n_blocks=8, stream=0):
        """phase one of a sharded sweep: only the seeding pass of the fused route, over about n_blocks device blocks of the call"""
        _check(self._lib.prot_sweep_seed_dev(self._h, d_in, in_stride, n_frames, ang_start, ang_end,
                                             ang_stride, chn, d_peaks or None, flags, n_blocks, stream),
               "prot_sweep_seed_dev")

    def sweep_shard_seed_dev(self, d_halo, halo_stride, n_halo, d_in, in_stride, n_frames, ang_start, ang_end, ang_stride,
                             d_peaks=0, flags=0, n_blocks=8, stream=0):
        """phase one of a shard in one call: reset, the table zeroed, the history primed with the halo, the seeding pass"""
        _check(self._lib.prot_sweep_shard_seed_dev(self._h, d_halo or None, halo_stride, n_halo, d_in, in_stride, n_frames, ang_start,
                                                   ang_end, ang_stride, d_peaks or None, flags, n_blocks, stream),
               "prot_sweep_shard_seed_dev")

    def prime_dev(self, d_in, in_stride, n_frames, stream=0):
        """n_frames of input pass into the history untransformed (the frames in front of a shard)"""
        _check(self._lib.prot_engine_prime_dev(self._h, d_in, in_stride, n_frames, stream), "prot_engine_prime_dev")

    def peaks(self):
        out = np.empty((self.n_chn, MAXSAMPLE), np.float32)
        _check(self._lib.prot_cli_peaks(self._h, out.reshape(-1)), "prot_cli_peaks")
        return out

    def peaks_dev(self):
        return self._lib.prot_cli_peaks_dev(self._h)


class PhaseRotate(Engine):
    """Mirror of class PhaseRotate (cli/phase-rotate.cc:251-485): same method
    names, argument meaning and defaults; one PhaseRotateProc per channel is
    implied by the engine (cli/phase-rotate.cc:768-777)."""

    def __init__(self, blksiz, n_chn, device=0):
        super().__init__(KIND_CLI, blksiz, n_chn, device)
        self._blksiz = blksiz

    def n_channels(self):
        return self.n_chn

    def blksiz(self):
        return self._blksiz

    def analyze(self, buf, ang_start=0, ang_end=180, ang_stride=1, chn=-1, start=False):
        buf = np.ascontiguousarray(buf, np.float32).reshape(-1)
        if buf.size != self._blksiz * self.n_chn:
            raise ValueError("analyze wants one interleaved block of blksiz frames")
        _check(self._lib.prot_cli_analyze(self._h, buf, ang_start, ang_end, ang_stride, chn, int(start)),
               "prot_cli_analyze")

    def apply(self, buf, angles):
        buf = np.ascontiguousarray(buf, np.float32).reshape(-1).copy()
        if buf.size != self._blksiz * self.n_chn:
            raise ValueError("apply wants one interleaved block of blksiz frames")
        ang = np.ascontiguousarray(angles, np.int32)
        if ang.size != self.n_chn:
            raise ValueError("one angle per channel")
        _check(self._lib.prot_cli_apply(self._h, buf, ang), "prot_cli_apply")
        return buf

    def peak(self, c, a):
        return self._lib.prot_cli_peak(self._h, c, a)

    def peak_all(self, a):
        return self._lib.prot_cli_peak_all(self._h, a)

    # analyze_file / the apply loop of main(), on in-memory interleaved files
    def analyze_file(self, data, ang_start, ang_end, ang_stride, chn=-1):
        data = np.ascontiguousarray(data, np.float32).reshape(-1)
        _check(self._lib.prot_cli_analyze_file(self._h, data, data.size // self.n_chn, ang_start,
                                               ang_end, ang_stride, chn), "prot_cli_analyze_file")

    def file_resident(self, on=True):
        """prot_cli_file_resident: keep the de-interleaved file of analyze_file on the device between passes."""
        _check(self._lib.prot_cli_file_resident(self._h, 1 if on else 0), "prot_cli_file_resident")

    def apply_file(self, data, angles, out=None):
        data = np.ascontiguousarray(data, np.float32).reshape(-1)
        if out is None:
            out = np.zeros_like(data)
        assert out.dtype == np.float32 and out.size == data.size and out.flags["C_CONTIGUOUS"]
        _check(self._lib.prot_cli_apply_file(self._h, data, data.size // self.n_chn,
                                             np.ascontiguousarray(angles, np.int32), out),
               "prot_cli_apply_file")
        return out


def analyze_file_multi(engines, data, ang_start, ang_end, ang_stride, chn=-1):
    """prot_cli_analyze_file_multi: one in-memory file sharded over several PhaseRotate engines."""
    lib = load_library()
    data = np.ascontiguousarray(data, np.float32).reshape(-1)
    hs = (C.c_void_p * len(engines))(*[e._h for e in engines])
    _check(lib.prot_cli_analyze_file_multi(hs, len(engines), data, data.size // engines[0].n_chn,
                                           ang_start, ang_end, ang_stride, chn), "prot_cli_analyze_file_multi")


def time_lv2_run_multi(engines, frames, calls):
    """(mean microseconds per prot_lv2_run call, slowest single call) with the engines called in turn, looped in C (test hook
    prot_internal_time_lv2_run_multi: an LV2 host running several plugin instances per cycle)"""
    lib = load_library()
    us = (C.c_double * 3)()
    hs = (C.c_void_p * len(engines))(*[e._h for e in engines])
    lib.prot_internal_time_lv2_run_multi.argtypes = [C.POINTER(C.c_void_p), C.c_uint, C.c_uint, C.c_uint, C.POINTER(C.c_double)]
    lib.prot_internal_time_lv2_run_multi.restype = C.c_int
    _check(lib.prot_internal_time_lv2_run_multi(hs, len(engines), frames, calls, us), "prot_internal_time_lv2_run_multi")
    return us[0], us[1]


class Lv2Instance(Engine):
    """Mirror of the LV2 instance (src/phaserotate.c): instantiate(rate) ->
    activate() -> run(n_samples) with planar port buffers and the angle
    control ports in degrees."""

    def __init__(self, rate, n_chn=1, device=0):
        super().__init__(KIND_LV2, rate, n_chn, device)

    @property
    def latency(self):
        return self.info.latency

    def activate(self):
        self.reset()

    def run(self, x, angle_deg):
        x = np.ascontiguousarray(np.atleast_2d(x), np.float32)
        assert x.shape[0] == self.n_chn
        y = np.empty_like(x)
        n = x.shape[1]
        pin = (C.c_void_p * self.n_chn)(*[x[c].ctypes.data for c in range(self.n_chn)])
        pout = (C.c_void_p * self.n_chn)(*[y[c].ctypes.data for c in range(self.n_chn)])
        ang = np.ascontiguousarray(np.broadcast_to(np.asarray(angle_deg, np.float32), (self.n_chn,)))
        _check(self._lib.prot_lv2_run(self._h, pin, pout, n, ang), "prot_lv2_run")
        return y

    def run_stream(self, x, angle_deg, bufsize):
        x = np.ascontiguousarray(np.atleast_2d(x), np.float32)
        out = np.empty_like(x)
        for s in range(0, x.shape[1], bufsize):
            out[:, s:s + bufsize] = self.run(x[:, s:s + bufsize], angle_deg)
        return out
