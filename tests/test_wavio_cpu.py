"""host/wavio.h without a GPU: a small C++ harness (tests/native/wavio_roundtrip.cc) reads a WAV and writes it back.
Files long enough (> 2^18 samples) for the threaded conversion and the mapped-file reader; every format the front end
accepts; metadata chunks in front of and behind the samples.  Integer PCM is read with 1 / 2^(bits-1) and written with
2^(bits-1) - 1 (libsndfile's defaults, cli/phase-rotate.cc:683-702 via SF_INFO), so a round trip moves large samples by
at most one LSB; float formats come back bit for bit."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("wavio") / "wavio_roundtrip")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "phaserotate.lv2_amd", "host"),
                    os.path.join(ROOT, "tests", "native", "wavio_roundtrip.cc"), "-o", exe, "-lpthread"], check=True)
    return exe


def _chunk(cid, body):
    return cid + struct.pack("<I", len(body)) + body + (b"\0" if len(body) & 1 else b"")


def _wav(tag, chn, rate, bits, raw, front=(), back=()):
    fmt = struct.pack("<HHIIHH", tag, chn, rate, rate * chn * bits // 8, chn * bits // 8, bits)
    body = b"WAVE" + _chunk(b"fmt ", fmt) + b"".join(_chunk(*c) for c in front) + _chunk(b"data", raw) + b"".join(_chunk(*c) for c in back)
    return b"RIFF" + struct.pack("<I", len(body)) + body


def _chunks(blob):
    pos, out = 12, []
    while pos + 8 <= len(blob):
        cid, sz = blob[pos:pos + 4], struct.unpack_from("<I", blob, pos + 4)[0]
        out.append((cid, blob[pos + 8:pos + 8 + sz]))
        pos += 8 + sz + (sz & 1)
    return out


@pytest.mark.parametrize("kind,chn", [("pcm16", 2), ("pcm24", 1), ("pcm24", 3), ("pcm32", 2), ("f32", 2), ("f64", 1)])
def test_wavio_round_trip_threaded(harness, tmp_path, kind, chn):
    n = 400_003  # frames: odd, and n * chn > 2^18 so that several threads convert
    rng = np.random.default_rng(5)
    x = (rng.random((n, chn)) * 1.9 - 0.95)
    x[0, 0], x[1, 0] = 0.999999, -1.0  # the extremes
    if kind == "pcm16":
        q = np.rint(x * 32767).astype("<i2"); raw, tag, bits = q.tobytes(), 1, 16
    elif kind == "pcm24":
        q = np.clip(np.rint(x * 8388607), -8388608, 8388607).astype(np.int32).reshape(-1)
        b = np.empty((q.size, 3), np.uint8); b[:, 0], b[:, 1], b[:, 2] = q & 255, (q >> 8) & 255, (q >> 16) & 255
        raw, tag, bits = b.tobytes(), 1, 24
    elif kind == "pcm32":
        q = np.clip(np.rint(x * 2147483647.0), -2147483648, 2147483647).astype("<i4"); raw, tag, bits = q.tobytes(), 1, 32
    elif kind == "f32":
        q = x.astype("<f4"); raw, tag, bits = q.tobytes(), 3, 32
    else:
        q = x.astype("<f4").astype("<f8"); raw, tag, bits = q.tobytes(), 3, 64  # values a float32 holds: the engine's sample type
    src, dst = str(tmp_path / "in.wav"), str(tmp_path / "out.wav")
    front = [(b"LIST", b"INFOINAM\x05\0\0\0take\0\0")]
    back = [(b"cue ", struct.pack("<I", 0)), (b"bext", b"x" * 603)]
    open(src, "wb").write(_wav(tag, chn, 48000, bits, raw, front, back))
    r = subprocess.run([harness, src, dst], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.split() == ["48000", str(chn), str(bits), "1" if tag == 3 else "0", str(n), "3"]
    ch = _chunks(open(dst, "rb").read())
    got = dict(ch)[b"data"]
    assert [c for c, _ in ch if c in (b"LIST", b"cue ", b"bext")] == [b"LIST", b"cue ", b"bext"]  # carried over, in front of the samples
    assert [c for c, _ in ch][-1] == b"data"
    if tag == 3:
        assert got == raw
    elif kind == "pcm24":
        g = np.frombuffer(got, np.uint8).reshape(-1, 3).astype(np.int32)
        gq = ((g[:, 0] | (g[:, 1] << 8) | (g[:, 2] << 16)) << 8) >> 8
        assert np.abs(gq - q).max() <= 1
    else:
        gq = np.frombuffer(got, "<i2" if bits == 16 else "<i4").astype(np.int64)
        qq = np.asarray(q, np.int64).reshape(-1)
        # 32-bit PCM passes through float32 samples (as sf_readf_float does): 24 significant bits
        lim = 1 if bits == 16 else np.maximum(1, np.abs(qq) >> 22)
        assert np.all(np.abs(gq - qq) <= lim)


def test_wavio_refuses_what_it_does_not_read(harness, tmp_path):
    for name, blob in (("short", b"RIFF\x04\0\0\0WAVE"), ("aiff", b"FORM\0\0\0\x04AIFF"),
                       ("mulaw", _wav(7, 1, 8000, 8, b"\0" * 64)), ("nofmt", b"RIFF" + struct.pack("<I", 20) + b"WAVE" + _chunk(b"data", b"\0" * 8))):
        p = str(tmp_path / (name + ".wav"))
        open(p, "wb").write(blob)
        r = subprocess.run([harness, p, str(tmp_path / "o.wav")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        assert r.returncode == 1 and r.stderr.strip(), name
