"""The host C/C++ front ends over the C ABI: the LV2 plugin (phaserotate.so,
lv2_descriptor/run) and the phase-rotate CLI.  CPU tests cover what does not
need a device (exports, option validation, error text, failure without a GPU);
GPU tests compare run() and the CLI results with the oracle."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

import orc
from conftest import harmonic_stack, xorshift_uniform

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "phaserotate.lv2_amd", "lib")
PLUGIN = os.path.join(LIBDIR, "phaserotate.so")
CLI = os.path.join(LIBDIR, "phase-rotate")
URI = b"http://gareus.org/oss/lv2/phaserotate"


# ------------------------------------------------------------ LV2 host ----
class Feature(C.Structure):
    _fields_ = [("URI", C.c_char_p), ("data", C.c_void_p)]


MAPFN = C.CFUNCTYPE(C.c_uint32, C.c_void_p, C.c_char_p)


class UridMap(C.Structure):
    _fields_ = [("handle", C.c_void_p), ("map", MAPFN)]


class Descriptor(C.Structure):
    pass


Descriptor._fields_ = [
    ("URI", C.c_char_p),
    ("instantiate", C.CFUNCTYPE(C.c_void_p, C.POINTER(Descriptor), C.c_double, C.c_char_p, C.POINTER(C.POINTER(Feature)))),
    ("connect_port", C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_void_p)),
    ("activate", C.CFUNCTYPE(None, C.c_void_p)),
    ("run", C.CFUNCTYPE(None, C.c_void_p, C.c_uint32)),
    ("deactivate", C.c_void_p),
    ("cleanup", C.CFUNCTYPE(None, C.c_void_p)),
    ("extension_data", C.CFUNCTYPE(C.c_void_p, C.c_char_p)),
]


class Lv2Host:
    """A tiny LV2 host: urid:map, the port buffers of src/phaserotate.h:99-111, run()."""

    def __init__(self, index=0, rate=48000.0, with_map=True):
        self.lib = C.CDLL(PLUGIN)
        self.lib.lv2_descriptor.restype = C.POINTER(Descriptor)
        self.lib.lv2_descriptor.argtypes = [C.c_uint32]
        self.desc = self.lib.lv2_descriptor(index)
        self.n_chn = index + 1
        self.uris = {}

        def do_map(handle, uri):
            return self.uris.setdefault(uri, len(self.uris) + 1)

        self._mapfn = MAPFN(do_map)
        self._map = UridMap(None, self._mapfn)
        feats = []
        if with_map:
            feats.append(Feature(b"http://lv2plug.in/ns/ext/urid#map", C.cast(C.pointer(self._map), C.c_void_p)))
        self._feats = feats
        arr = (C.POINTER(Feature) * (len(feats) + 1))(*[C.pointer(f) for f in feats], None)
        self.handle = self.desc.contents.instantiate(self.desc, rate, b"/tmp", arr)

    def urid(self, uri):
        return self.uris[uri]

    def setup(self, nmax, atom_ports=True):
        d = self.desc.contents
        self.latency = C.c_float(-1)
        self.angle = [C.c_float(0) for _ in range(self.n_chn)]
        self.inbuf = np.zeros((self.n_chn, nmax), np.float32)
        self.outbuf = np.zeros((self.n_chn, nmax), np.float32)
        self.control = np.zeros(64, np.uint32)     # LV2_Atom_Sequence, 256 bytes
        self.notify = np.zeros(2048, np.uint32)    # 8 KiB
        d.connect_port(self.handle, 2, C.addressof(self.latency))
        if atom_ports:
            d.connect_port(self.handle, 0, self.control.ctypes.data)
            d.connect_port(self.handle, 1, self.notify.ctypes.data)
        for c in range(self.n_chn):
            d.connect_port(self.handle, 3 + 3 * c, C.addressof(self.angle[c]))
            d.connect_port(self.handle, 4 + 3 * c, self.inbuf[c].ctypes.data)
            d.connect_port(self.handle, 5 + 3 * c, self.outbuf[c].ctypes.data)
        d.activate(self.handle)

    def set_control(self, otype=None):
        """Empty sequence, or one event carrying an empty object of type `otype`."""
        seq_t = self.uris.get(b"http://lv2plug.in/ns/ext/atom#Sequence", 0)
        if otype is None:
            self.control[:4] = [8, seq_t, 0, 0]
        else:
            obj_t = self.urid(b"http://lv2plug.in/ns/ext/atom#Object")
            # sequence body (8) + event: time (8) + atom hdr (8) + object body (8)
            self.control[:10] = [8 + 24, seq_t, 0, 0, 0, 0, 8, obj_t, 1, otype]

    def run(self, x, angles):
        n = x.shape[1]
        self.inbuf[:, :n] = x
        for c in range(self.n_chn):
            self.angle[c].value = angles[c]
        self.notify[0] = self.notify.nbytes  # capacity, src/phaserotate.c:796
        self.notify[1] = 0
        self.desc.contents.run(self.handle, n)
        return self.outbuf[:, :n].copy()

    def notify_objects(self):
        """Parse the notify sequence into [(otype, {key: value})]."""
        raw = self.notify.tobytes()
        size, _ = struct.unpack_from("<II", raw, 0)
        end = 8 + size
        off = 16
        out = []
        inv = {v: k for k, v in self.uris.items()}
        while off + 16 <= end:
            _t, asz, atype = struct.unpack_from("<qII", raw, off)
            body = off + 16
            _id, otype = struct.unpack_from("<II", raw, body)
            props = {}
            p = body + 8
            while p + 16 <= body + asz:
                key, _ctx, vsz, vtype = struct.unpack_from("<IIII", raw, p)
                vt = inv.get(vtype, b"")
                val = struct.unpack_from("<f" if vt.endswith(b"#Float") else "<i", raw, p + 16)[0]
                props[inv.get(key, key)] = val
                p += (16 + vsz + 7) & ~7
            out.append((inv.get(otype, otype), props))
            off += (16 + asz + 7) & ~7
        return out

    def close(self):
        if self.handle:
            self.desc.contents.cleanup(self.handle)
            self.handle = None


def test_plugin_exports_only_lv2_descriptor():
    out = subprocess.run(["nm", "-D", "--defined-only", PLUGIN], stdout=subprocess.PIPE, text=True).stdout
    syms = [ln.split()[-1] for ln in out.splitlines() if " T " in ln]
    assert syms == ["lv2_descriptor"]  # -fvisibility=hidden, src/phaserotate.c:872-893


def test_descriptor_uris_and_index_range():
    lib = C.CDLL(PLUGIN)
    lib.lv2_descriptor.restype = C.POINTER(Descriptor)
    lib.lv2_descriptor.argtypes = [C.c_uint32]
    assert lib.lv2_descriptor(0).contents.URI == URI
    assert lib.lv2_descriptor(1).contents.URI == URI + b"#stereo"
    assert not lib.lv2_descriptor(2)
    d = lib.lv2_descriptor(0).contents
    assert not d.deactivate and d.extension_data(b"x") is None


def test_instantiate_needs_urid_map():
    h = Lv2Host(0, with_map=False)
    assert not h.handle  # src/phaserotate.c:252-256


def test_instantiate_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = Lv2Host(0)
    assert not h.handle  # engine creation fails -> NULL like every instantiate() failure


@pytest.mark.gpu
@pytest.mark.parametrize("index,angles", [(0, [90.0]), (1, [30.0, -45.0])])
def test_plugin_run_matches_oracle(index, angles):
    n, bs = 16384, 1024
    h = Lv2Host(index)
    assert h.handle
    h.setup(bs)
    h.set_control(None)
    o = orc.Lv2Engine(48000, index + 1)
    x = xorshift_uniform(n * (index + 1), seed=77).reshape(index + 1, n)
    y = np.empty_like(x)
    for s in range(0, n, bs):
        y[:, s:s + bs] = h.run(x[:, s:s + bs], angles)
    assert h.latency.value == 1792.0
    yo = o.run_stream(x, angles, bs)
    assert np.abs(y[:, 4096:] - yo[:, 4096:]).max() < 2e-6
    assert np.abs(y - yo).max() < 1e-5
    assert h.notify_objects() == []  # UI inactive: empty notify sequence
    h.close()


@pytest.mark.gpu
def test_plugin_deferred_run_reports_its_latency_and_returns_the_period_before(monkeypatch):
    """Round 6: PROT_LV2_DEFERRED=N (or a bundle made with make_bundle.py --deferred N) puts the plugin's run() into the mode that
    never waits for the device (prot_lv2_set_deferred): the latency port says latency + N (src/phaserotate.c:788 reports what the
    host must compensate), and the samples are the oracle's, N frames later."""
    n, bs, D = 16384, 1024, 1024
    monkeypatch.setenv("PROT_LV2_DEFERRED", str(D))
    h = Lv2Host(1)
    assert h.handle
    h.setup(bs)
    h.set_control(None)
    o = orc.Lv2Engine(48000, 2)
    x = xorshift_uniform(n * 2, seed=78).reshape(2, n)
    y = np.empty_like(x)
    import time
    for s in range(0, n, bs):
        y[:, s:s + bs] = h.run(x[:, s:s + bs], [30.0, -45.0])
        time.sleep(0.0003)   # (a host calls once per period -- 21 ms for 1024 frames; back to back, 25 us apart, every block is "late")
    assert h.latency.value == 1792.0 + D
    yo = o.run_stream(x, [30.0, -45.0], bs)
    assert not y[:, :D].any()
    assert np.abs(y[:, 4096 + D:] - yo[:, 4096:n - D]).max() < 2e-6
    assert np.abs(y[:, D:] - yo[:, :n - D]).max() < 1e-5
    h.close()


@pytest.mark.gpu
def test_plugin_without_atom_ports_only_forwards_and_reports_latency():
    h = Lv2Host(0)
    h.setup(512, atom_ports=False)
    x = xorshift_uniform(512, seed=1).reshape(1, -1)
    y = h.run(x, [90.0])
    assert np.array_equal(y, x) and h.latency.value == 1792.0  # src/phaserotate.c:779-793
    h.close()


@pytest.mark.gpu
def test_plugin_meters_and_ui_messages():
    bs = 1024
    h = Lv2Host(0)
    h.setup(bs)
    o = orc.Lv2Engine(48000, 1)
    x = xorshift_uniform(bs * 12, seed=5).reshape(1, -1)
    # period 0 carries ui_on: from then on every run() emits one `levels` object (+ one `state`)
    h.set_control(h.urid(URI + b"#ui_on"))
    objs = None
    for s in range(0, x.shape[1], bs):
        h.run(x[:, s:s + bs], [45.0])
        o.run(x[:, s:s + bs], 45.0)
        objs = h.notify_objects()
        if s == 0:
            assert [t for t, _ in objs] == [URI + b"#levels", URI + b"#state"]
            assert objs[1][1][URI + b"#uiscale"] == 1.0 and objs[1][1][URI + b"#link"] == 0
        else:
            assert [t for t, _ in objs] == [URI + b"#levels"]
        h.set_control(None)
    lv = objs[0][1]
    mom, peak, diff = o.meters(0)
    assert lv[URI + b"#l_channel"] == 0
    assert lv[URI + b"#l_in_cur"] == pytest.approx(o.last_levels[0, 0], rel=1e-6)
    assert lv[URI + b"#l_out_cur"] == pytest.approx(o.last_levels[0, 1], rel=2e-5)
    assert lv[URI + b"#l_in_peak"] == pytest.approx(peak[0], rel=1e-6)
    assert lv[URI + b"#l_out_peak"] == pytest.approx(peak[1], rel=2e-5)
    assert lv[URI + b"#l_in_mom"] == pytest.approx(mom[0], rel=1e-5)
    assert lv[URI + b"#l_diff_min"] == pytest.approx(diff[0], rel=1e-4)
    assert lv[URI + b"#l_diff_max"] == pytest.approx(diff[1], rel=1e-4)
    # ui_off silences the notify port again
    h.set_control(h.urid(URI + b"#ui_off"))
    h.run(x[:, :bs], [45.0])
    assert h.notify_objects() == []
    h.close()


# ------------------------------------------------------------------ CLI ---
def write_wav(path, data, rate, bits=32, is_float=True):
    data = np.asarray(data, np.float32)
    if data.ndim == 1:
        data = data[:, None]
    n, ch = data.shape
    if is_float:
        raw = data.astype("<f4").tobytes()
        tag = 3
    else:
        raw = np.clip(np.rint(data * 32767.0), -32768, 32767).astype("<i2").tobytes()
        tag, bits = 1, 16
    hdr = b"RIFF" + struct.pack("<I", 36 + len(raw)) + b"WAVEfmt " + struct.pack(
        "<IHHIIHH", 16, tag, ch, rate, rate * ch * bits // 8, ch * bits // 8, bits) + b"data" + struct.pack("<I", len(raw))
    open(path, "wb").write(hdr + raw)


def read_wav_f32(path):
    raw = open(path, "rb").read()
    assert raw[:4] == b"RIFF" and raw[8:12] == b"WAVE"
    ch = struct.unpack_from("<H", raw, 22)[0]
    n = struct.unpack_from("<I", raw, 40)[0]
    return np.frombuffer(raw, "<f4", n // 4, 44).reshape(-1, ch)


def cli(*args):
    r = subprocess.run([CLI] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    return r.returncode, r.stdout, r.stderr


def test_cli_option_validation_text_and_exit_codes(tmp_path):
    wav = str(tmp_path / "a.wav")
    write_wav(wav, np.zeros(100, np.float32), 48000)
    assert cli() == (1, "", "Error: Missing parameter. See --help for usage information.\n")
    assert cli("-s", "7", wav) == (1, "", "Error: 180 deg is not evenly dividable by given stride.\n")
    assert cli("-s", "0", wav)[0] == 1 and cli("-s", "120", wav)[0] == 1
    assert cli("-f", "512", wav) == (1, "", "Error: fft-len is out of bounds; valid range 1024..32768\n")
    assert cli("-f", "65536", wav)[0] == 1
    assert cli("-a", "10", wav) == (1, "", "Error: -a, --angle option requires an output file to be given.\n")
    rc, out, err = cli("-a", "200", wav, str(tmp_path / "o.wav"))
    assert rc == 1 and err == "Error: Invalid angle speficied, value needs to be -180 .. +180.\n"
    rc, out, err = cli(str(tmp_path / "missing.wav"))
    assert rc == 1 and err.startswith("Cannot open '%s' for reading: " % (tmp_path / "missing.wav"))
    rc, out, err = cli("-x", wav)
    assert rc == 1 and "Error: unrecognized option. See --help for usage information." in err
    rc, out, _ = cli("-V")
    assert rc == 0 and out.startswith("phase-rotate version ")
    rc, out, _ = cli("-h")
    assert rc == 0 and out.startswith("phase-rotate - Audio File Phase Rotation Util.") and "-l, --link-channels" in out
    stereo = str(tmp_path / "st.wav")
    write_wav(stereo, np.zeros((100, 2), np.float32), 48000)
    rc, out, err = cli("-a", "10,20,30", wav, str(tmp_path / "o.wav"))  # more angles than channels is fine
    assert "file has more channels" not in err
    import torch
    if not torch.cuda.is_available():
        rc, out, err = cli("-f", "2048", wav)
        assert rc == 1 and "cannot create the GPU engine" in err  # no CPU fallback


def _fmt_result(ma, fo, rz, rm):
    def cdb(x):
        return -np.inf if x < 1e-15 else np.float32(20.0) * np.log10(np.float32(x))
    lines = ["# Result -- Minimize digital peak"]
    for c in range(len(ma)):
        if not fo[c]:
            lines.append("Channel: %2d Phase:   0 deg # cannot find min." % (c + 1))
            continue
        s = "Channel: %2d Phase: %5.2f deg" % (c + 1, ma[c] / 2.0)
        if ma[c] != 0:
            s += ", gain: %5.2f dB (att. %4.2f to %4.2f dBFS)" % (cdb(rz[c]) - cdb(rm[c]), cdb(rz[c]), cdb(rm[c]))
        lines.append(s)
    return "\n".join(lines) + "\n"


@pytest.mark.gpu
@pytest.mark.parametrize("n_chn,stride,link", [(1, 24, False), (2, 24, False), (2, 1, False), (2, 24, True)])
def test_cli_analysis_result_matches_oracle(tmp_path, n_chn, stride, link):
    rate, n = 48000, 120000
    chans = [0.8 * harmonic_stack(n, rate, f0=110.0 * (1 + 0.29 * c)) + 0.02 * xorshift_uniform(n, seed=60 + c)
             for c in range(n_chn)]
    data = np.stack(chans, axis=1).astype(np.float32)
    wav = str(tmp_path / "in.wav")
    write_wav(wav, data, rate)
    args = ["-f", "2048", "-s", str(stride)] + (["-l"] if link else []) + [wav]
    rc, out, err = cli(*args)
    assert rc == 0, err
    ma, fo, rz, rm, _ = orc.cli_find_min(2048, n_chn, data, stride=stride, link=link)
    # the angle index must be exact; the dB figures agree to the printed precision unless
    # a value sits on a rounding boundary, so compare numerically
    want = _fmt_result(ma, fo, rz, rm)
    got_lines, want_lines = out.splitlines(), want.splitlines()
    assert len(got_lines) == len(want_lines)
    for g, w in zip(got_lines, want_lines):
        assert g.split(",")[0] == w.split(",")[0]  # "Channel: c Phase: x deg"
        gn = [float(t) for t in g.replace("dB", " ").replace("(att.", " ").replace("to", " ").replace("dBFS)", " ").split() if t.replace(".", "").replace("-", "").isdigit()]
        wn = [float(t) for t in w.replace("dB", " ").replace("(att.", " ").replace("to", " ").replace("dBFS)", " ").split() if t.replace(".", "").replace("-", "").isdigit()]
        assert np.allclose(gn, wn, atol=0.011)


@pytest.mark.gpu
def test_cli_default_block_size_48k(tmp_path):
    """No -f: blksiz = pow2ceil(48000 / 8) = 8192 (cli/phase-rotate.cc:749-755) = two FIR partition pairs here."""
    rate, n = 48000, 100000
    x = (0.8 * harmonic_stack(n, rate, f0=82.4) + 0.02 * xorshift_uniform(n, seed=13)).astype(np.float32)
    wav = str(tmp_path / "in.wav")
    write_wav(wav, x, rate)
    rc, out, err = cli("-vv", wav)
    assert rc == 0 and "Process block-size 8192" in err
    ma, fo, rz, rm, _ = orc.cli_find_min(8192, 1, x, stride=24)
    want = _fmt_result(ma, fo, rz, rm).splitlines()[-1]
    got = err.splitlines()[-1]  # -vv sends the result to stderr (verbose_fd)
    assert got.split(",")[0] == want.split(",")[0]


@pytest.mark.gpu
def test_cli_verbose_table_and_apply(tmp_path):
    rate, P = 48000, 2048
    n = P * 12
    x = (0.8 * harmonic_stack(n, rate) + 0.05 * xorshift_uniform(n, seed=9)).astype(np.float32)
    wav, outw = str(tmp_path / "in.wav"), str(tmp_path / "out.wav")
    write_wav(wav, x, rate)
    rc, out, err = cli("-vv", "-f", "2048", "-s", "24", wav)
    assert rc == 0
    assert "Process block-size 2048" in err and "Analyzing using 1 process threads, stride = 24" in err
    rows = [ln.split() for ln in out.splitlines() if ln and ln[0].isdigit()]
    coarse = rows[:15]
    assert [float(r[0]) for r in coarse] == [a / 2.0 for a in range(0, 360, 24)]  # Q5
    o = orc.CliEngine(P, 1)
    o.analyze_file(x, 0, 360, 24)
    pk = o.peaks()[0]
    for r, a in zip(coarse, range(0, 360, 24)):
        assert float(r[2]) == pytest.approx(20 * np.log10(pk[a]), abs=1.1e-3)
    # -a: apply a given angle and write the file (float in -> float out)
    rc, out, err = cli("-a", "45", "-f", "2048", wav, outw)
    assert rc == 0, err
    y = read_wav_f32(outw)[:, 0]
    yo = orc.cli_apply_file_mono(P, x, 90)
    assert y.shape == x.shape and np.abs(y - yo).max() < 2e-6
    # 16-bit PCM in -> 16-bit PCM out
    wav16, out16 = str(tmp_path / "in16.wav"), str(tmp_path / "out16.wav")
    write_wav(wav16, x, rate, is_float=False)
    rc, out, err = cli("-a", "-30.5", "-f", "2048", wav16, out16)
    assert rc == 0, err
    raw = open(out16, "rb").read()
    assert struct.unpack_from("<HH", raw, 20) == (1, 1) and struct.unpack_from("<H", raw, 34)[0] == 16
    y16 = np.frombuffer(raw, "<i2", n, 44).astype(np.float32)
    x16 = np.clip(np.rint(x * 32767.0), -32768, 32767).astype(np.float32) / 32768.0
    yo = orc.cli_apply_file_mono(P, x16.astype(np.float32), -61)
    assert np.abs(y16 - np.clip(np.rint(yo * 32767.0), -32768, 32767)).max() <= 1.0


@pytest.mark.gpu
def test_multi_engine_sharded_analysis_is_bit_identical(pkg):
    """prot_cli_analyze_file_multi with 3 engines (all on device 0: host-merged tables) ==
    prot_cli_analyze_file on one engine, bit for bit; ragged file, stereo, coarse stride."""
    P, n_chn = 2048, 2
    n = P * 37 + 911
    chans = [0.8 * harmonic_stack(n, f0=120.0 * (1 + 0.23 * c)) + 0.05 * xorshift_uniform(n, seed=c) for c in range(n_chn)]
    data = np.stack(chans, axis=1).astype(np.float32)
    one = pkg.PhaseRotate(P, n_chn)
    one.analyze_file(data, 0, 360, 1)
    want = one.peaks()
    engines = [pkg.PhaseRotate(P, n_chn) for _ in range(3)]
    pkg.analyze_file_multi(engines, data, 0, 360, 1)
    for e in engines:
        assert np.array_equal(e.peaks(), want)
    # a single-channel fine pass with negative angles through the same path
    for e in engines + [one]:
        e.reset()
    one.analyze_file(data, -12, 13, 1, 1)
    pkg.analyze_file_multi(engines, data, -12, 13, 1, 1)
    assert np.array_equal(engines[0].peaks(), one.peaks())
    # more engines than block pairs
    tiny = data[:P + 5]
    for e in engines + [one]:
        e.reset()
    one.analyze_file(tiny, 0, 360, 24)
    pkg.analyze_file_multi(engines, tiny, 0, 360, 24)
    assert np.array_equal(engines[2].peaks(), one.peaks())
    # RCCL path with one communicator (n = 1) must be a no-op success
    lib = pkg.load_library()
    hs = (C.c_void_p * 1)(one._h)
    assert lib.prot_allreduce_peaks(hs, 1) == 0
    for e in engines + [one]:
        e.close()


@pytest.mark.gpu
def test_cli_prot_gpus_env_gives_same_result(tmp_path):
    rate, n = 48000, 90000
    x = np.stack([0.8 * harmonic_stack(n, rate, f0=98.0 * (1 + 0.3 * c)) + 0.02 * xorshift_uniform(n, seed=c)
                  for c in range(2)], axis=1).astype(np.float32)
    wav = str(tmp_path / "in.wav")
    write_wav(wav, x, rate)
    rc1, out1, err1 = cli("-f", "2048", wav)
    env = dict(os.environ, PROT_GPUS="3", PROT_GPUS_SAME_DEVICE="1")
    r = subprocess.run([CLI, "-f", "2048", wav], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    assert rc1 == 0 and r.returncode == 0, r.stderr
    assert r.stdout == out1


@pytest.mark.gpu
def test_two_plugin_instances_run_concurrently():
    """SURVEY 8b threading: several instances may run() at the same time (the reference guards only FFTW's planner,
    src/phaserotate.c:43-44, :198-220).  Two hosts -- a mono and a stereo instance -- instantiate, run 200 buffers and
    clean up from two threads at once (ctypes releases the GIL inside run()); each stream must match its oracle."""
    import threading
    n, bs = 200 * 256, 256
    res, errs = {}, []

    def worker(index, angles, seed):
        try:
            h = Lv2Host(index)
            assert h.handle
            h.setup(bs)
            h.set_control(None)
            x = xorshift_uniform(n * (index + 1), seed=seed).reshape(index + 1, n)
            y = np.empty_like(x)
            for s in range(0, n, bs):
                y[:, s:s + bs] = h.run(x[:, s:s + bs], angles)
            h.close()
            res[index] = (x, y, angles)
        except Exception as ex:  # noqa: BLE001 (reported by the asserting thread)
            errs.append(repr(ex))

    for rep in range(3):  # instantiate / cleanup race as well
        th = [threading.Thread(target=worker, args=(0, [77.0], 5 + rep)), threading.Thread(target=worker, args=(1, [-30.0, 120.0], 9 + rep))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        for index, (x, y, angles) in res.items():
            yo = orc.Lv2Engine(48000, index + 1).run_stream(x, angles, bs)
            assert np.abs(y[:, 4096:] - yo[:, 4096:]).max() < 2e-6
            assert np.abs(y - yo).max() < 1e-5


@pytest.mark.gpu
def test_run_longer_than_the_reserved_block_is_chunked(pkg):
    """prot_lv2_run never allocates: buffers are reserved for 8192 frames (or options:maxBlockLength) and longer calls are
    cut up -- same samples as the oracle's run() with the same call sizes."""
    n = 3 * 20000
    x = xorshift_uniform(n, seed=3).reshape(1, n)
    g = pkg.Lv2Instance(48000, 1)
    g.activate()
    o = orc.Lv2Engine(48000, 1)
    yg, yo = np.empty_like(x), np.empty_like(x)
    for s in range(0, n, 20000):  # 20000 > 8192: two full chunks and a rest per call
        yg[:, s:s + 20000] = g.run(x[:, s:s + 20000], -45.0)
        yo[:, s:s + 20000] = o.run(x[:, s:s + 20000], -45.0)
    assert np.abs(yg - yo).max() < 1e-5 and np.abs(yg[:, 8192:] - yo[:, 8192:]).max() < 2e-6
    g.close()
