"""The steps either side of the hot path: the LV2 bundle metadata that make_bundle.py writes, and the
CLI's WAV I/O (sample formats, metadata chunks, hostile headers).  Bundle checks run on the CPU; the CLI ones need the GPU."""
import os
import re
import struct
import subprocess
import sys

import numpy as np
import pytest

import orc
from conftest import harmonic_stack, xorshift_uniform

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "phaserotate.lv2_amd", "host")
CLI = os.path.join(ROOT, "phaserotate.lv2_amd", "lib", "phase-rotate")
URI = "http://gareus.org/oss/lv2/phaserotate"


# ------------------------------------------------------------------ bundle ---
def _ports(ttl, uri):
    """[(index, symbol, text of the port)] of plugin `uri` in the generated Turtle."""
    m = re.search(r"<%s>\n(.*?)\n\n" % re.escape(uri), ttl + "\n\n", re.S)
    assert m, uri
    body = m.group(1)
    out = []
    for pm in re.finditer(r"\[ a (?:atom:AtomPort|lv2:\w+Port)[^\]]*?lv2:index (\d+) ; lv2:symbol \"(\w+)\"(.*?)\](?= ,| \.)", body, re.S):
        out.append((int(pm.group(1)), pm.group(2), pm.group(0)))
    return body, out


def test_bundle_metadata(tmp_path):
    r = subprocess.run([sys.executable, os.path.join(HOST, "make_bundle.py"), str(tmp_path)], stdout=subprocess.PIPE, text=True)
    assert r.returncode == 0
    b = tmp_path / "phaserotate.lv2"
    man, ttl = (b / "manifest.ttl").read_text(), (b / "phaserotate.ttl").read_text()
    for sfx in ("", "#stereo"):  # src/phaserotate.c:872-893: index 0 mono, 1 stereo
        assert "<%s%s> a lv2:Plugin ; lv2:binary <phaserotate.so> ; rdfs:seeAlso <phaserotate.ttl> ." % (URI, sfx) in man
    want = {"": ["control", "notify", "latency", "angle", "in", "out"],
            "#stereo": ["control", "notify", "latency", "angleL", "inL", "outL", "angleR", "inR", "outR"]}
    for sfx, syms in want.items():  # src/phaserotate.h:99-111, lv2ttl/phaserotate.ports.in:14-41
        body, ports = _ports(ttl, URI + sfx)
        assert [(i, s) for i, s, _ in ports] == list(enumerate(syms))
        assert "lv2:requiredFeature urid:map" in body and "hardRTCapable" not in body
        by = {s: t for _, s, t in ports}
        assert "atom:AtomPort, lv2:InputPort" in by["control"] and "atom:bufferType atom:Sequence" in by["control"]
        assert "lv2:OutputPort" in by["notify"] and "rsz:minimumSize 4096" in by["notify"]
        assert "lv2:minimum 0 ; lv2:maximum 8192" in by["latency"] and "lv2:reportsLatency, lv2:integer" in by["latency"]
        for s in syms:
            if s.startswith("angle"):
                t = by[s]
                assert "lv2:default 0.0 ; lv2:minimum -180.0 ; lv2:maximum 180.0" in t and "pprop:rangeSteps 721" in t
                assert "units:unit units:degree" in t
                assert [float(v) for v in re.findall(r"rdf:value (-?[\d.]+)", t)] == [float(d) for d in range(-180, 181, 45)]
            if s.startswith("in"):
                assert "lv2:AudioPort, lv2:InputPort" in by[s]
            if s.startswith("out"):
                assert "lv2:AudioPort, lv2:OutputPort" in by[s]
    # every prefix that is used is declared
    for pfx in set(re.findall(r"\b([a-z]+):[A-Za-z]", ttl)) - {"http"}:
        assert "@prefix %s:" % pfx in ttl, pfx
    # --deferred N: the bundle asks the plugin for its non-waiting run() and declares lv2:hardRTCapable again (lv2ttl/phaserotate.ports.in:7)
    r = subprocess.run([sys.executable, os.path.join(HOST, "make_bundle.py"), str(tmp_path), "--deferred", "1024"], stdout=subprocess.PIPE, text=True)
    assert r.returncode == 0
    ttl_d = (b / "phaserotate.ttl").read_text()
    for sfx in want:
        body, ports = _ports(ttl_d, URI + sfx)
        assert "lv2:optionalFeature lv2:hardRTCapable" in body
        assert [(i, s_) for i, s_, _ in ports] == list(enumerate(want[sfx]))
    assert (b / "deferred").read_text().strip() == "1024"
    r = subprocess.run([sys.executable, os.path.join(HOST, "make_bundle.py"), str(tmp_path)], stdout=subprocess.PIPE, text=True)
    assert r.returncode == 0 and not (b / "deferred").exists() and "hardRTCapable" not in (b / "phaserotate.ttl").read_text()
    # against the reference's own templates, where the tree is present (this container; never on the GPU box)
    ref = "/root/reference/lv2ttl/"
    if os.path.exists(ref + "phaserotate.ports.in"):
        rt = open(ref + "phaserotate.ports.in").read()
        pat = r"lv2:index (\d+)\s*;\s*lv2:symbol \"(\w+)\""
        for sfx, part in (("", "phaserotate.mono.in"), ("#stereo", "phaserotate.stereo.in")):
            both = rt + open(ref + part).read()
            assert [(int(i), s_) for i, s_ in re.findall(pat, both)] == list(enumerate(want[sfx]))
        assert "rsz:minimumSize 4096" in rt and "lv2:maximum 8192" in rt and "lv2:requiredFeature urid:map" in rt
        assert "pprop:rangeSteps 721" in open(ref + "phaserotate.mono.in").read()


# --------------------------------------------------------------------- WAV ---
def _riff(fmt_body, data, pre=(), post=()):
    def ch(cid, body):
        return cid + struct.pack("<I", len(body)) + body + (b"\0" if len(body) & 1 else b"")
    body = b"WAVE" + ch(b"fmt ", fmt_body) + b"".join(ch(c, b) for c, b in pre) + ch(b"data", data) + b"".join(ch(c, b) for c, b in post)
    return b"RIFF" + struct.pack("<I", len(body)) + body


def _fmt(tag, ch, rate, bits, extensible=False):
    base = struct.pack("<HHIIHH", 0xFFFE if extensible else tag, ch, rate, rate * ch * bits // 8, ch * bits // 8, bits)
    if extensible:
        base += struct.pack("<HHIH", 22, bits, 0, tag) + bytes([0, 0, 0, 0, 0x10, 0, 0x80, 0, 0, 0xAA, 0, 0x38, 0x9B, 0x71])
    return base


def _chunks(raw):
    assert raw[:4] == b"RIFF" and raw[8:12] == b"WAVE"
    off, out = 12, []
    while off + 8 <= len(raw):
        cid, sz = raw[off:off + 4], struct.unpack_from("<I", raw, off + 4)[0]
        out.append((cid, raw[off + 8:off + 8 + sz]))
        off += 8 + sz + (sz & 1)
    return out


def _encode(x, kind):
    if kind == "pcm24":
        q = np.clip(np.rint(x.astype(np.float64) * 8388607.0), -8388608, 8388607).astype(np.int32)
        b = np.empty((q.size, 3), np.uint8)
        b[:, 0], b[:, 1], b[:, 2] = q & 255, (q >> 8) & 255, (q >> 16) & 255
        return b.tobytes(), q.astype(np.float32) / 8388608.0
    if kind == "pcm32":
        q = np.clip(np.rint(x.astype(np.float64) * 2147483647.0), -2147483648, 2147483647).astype(np.int64)
        return q.astype("<i4").tobytes(), (q.astype(np.float64) / 2147483648.0).astype(np.float32)
    if kind == "f64":
        return x.astype("<f8").tobytes(), x.astype(np.float32)
    raise ValueError(kind)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,tag,bits,ext", [("pcm24", 1, 24, True), ("pcm32", 1, 32, True), ("f64", 3, 64, False)])
def test_cli_sample_formats_round_trip(tmp_path, kind, tag, bits, ext):
    """24- and 32-bit PCM and 64-bit float in -> the same format out (the reference copies SF_INFO, cli/phase-rotate.cc:697):
    rotation by 0 degrees returns the samples unchanged, a real rotation matches the oracle on the decoded input."""
    P, n, rate = 2048, 2048 * 10, 48000
    x = (0.7 * harmonic_stack(n, rate) + 0.02 * xorshift_uniform(n, seed=31)).astype(np.float32)
    raw, xdec = _encode(x, kind)
    src, dst = str(tmp_path / "in.wav"), str(tmp_path / "out.wav")
    open(src, "wb").write(_riff(_fmt(tag, 1, rate, bits, extensible=(kind == "pcm24")), raw))
    r = subprocess.run([CLI, "-a", "0", "-f", str(P), src, dst], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    out = open(dst, "rb").read()
    ch = dict(_chunks(out))
    f = ch[b"fmt "]
    got_tag = struct.unpack_from("<H", f, 0)[0]
    assert struct.unpack_from("<HI", f, 2) == (1, rate)
    assert struct.unpack_from("<H", f, 14)[0] == bits
    if ext:  # more than 16 bits of PCM: WAVE_FORMAT_EXTENSIBLE with the PCM sub-format
        assert got_tag == 0xFFFE and len(f) == 40 and struct.unpack_from("<H", f, 24)[0] == tag
    else:
        assert got_tag == tag and len(f) == 16
    # angle 0: cos = 1, sin = 0 exactly -- the samples come back unchanged up to the container's own conversion, which
    # like libsndfile's reads integer PCM with 1 / 2^(bits-1) and writes it with 2^(bits-1) - 1 (one LSB for large samples)
    if kind == "pcm24":
        b = np.frombuffer(ch[b"data"], np.uint8).reshape(-1, 3).astype(np.int32)
        q = ((b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)) << 8) >> 8
        assert np.abs(q - np.rint(xdec.astype(np.float64) * 8388608.0)).max() <= 1
    elif kind == "pcm32":
        q = np.frombuffer(ch[b"data"], "<i4").astype(np.float64)
        assert np.abs(q / 2147483648.0 - xdec).max() < 2.0 ** -22  # (the stream is float32 in between)
    else:
        assert ch[b"data"] == raw
    r = subprocess.run([CLI, "-a", "37.5", "-f", str(P), src, dst], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    body = dict(_chunks(open(dst, "rb").read()))[b"data"]
    yo = orc.cli_apply_file_mono(P, xdec, 75)
    if kind == "pcm24":
        b = np.frombuffer(body, np.uint8).reshape(-1, 3).astype(np.int32)
        y = ((b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)) << 8 >> 8).astype(np.float64)
        assert np.abs(y - np.clip(np.rint(yo.astype(np.float64) * 8388607.0), -8388608, 8388607)).max() <= 4  # 2e-6 of tolerance = 17 LSB; seen <= 2
    elif kind == "pcm32":
        y = np.frombuffer(body, "<i4").astype(np.float64) / 2147483647.0
        assert np.abs(y - yo).max() < 2e-6
    else:
        assert np.abs(np.frombuffer(body, "<f8") - yo).max() < 2e-6


@pytest.mark.gpu
def test_cli_copies_metadata_chunks(tmp_path):
    """copy_metadata (cli/phase-rotate.cc:541-563): INFO strings, cue points and broadcast info reach the output file."""
    P, n, rate = 2048, 2048 * 6, 44100
    x = (0.5 * harmonic_stack(n, rate)).astype(np.float32)
    info = b"INFO" + b"INAM" + struct.pack("<I", 10) + b"Test take\0" + b"IART" + struct.pack("<I", 6) + b"x42 fan"[:5] + b"\0"
    cue = struct.pack("<I", 1) + struct.pack("<II4sIII", 1, 4096, b"data", 0, 0, 4096)
    bext = (b"phase rotate test".ljust(256, b"\0") + b"orig".ljust(32, b"\0") + b"ref".ljust(32, b"\0") + b"2026-10-03" + b"12:00:00"
            + struct.pack("<IIH", 0, 0, 1) + bytes(64) + bytes(190))
    src, dst = str(tmp_path / "in.wav"), str(tmp_path / "out.wav")
    open(src, "wb").write(_riff(_fmt(3, 1, rate, 32), x.astype("<f4").tobytes(), pre=[(b"LIST", info), (b"JUNK", bytes(10))], post=[(b"cue ", cue), (b"bext", bext)]))
    r = subprocess.run([CLI, "-a", "30", "-f", str(P), src, dst], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    got = _chunks(open(dst, "rb").read())
    ids = [c for c, _ in got]
    assert ids[0] == b"fmt " and ids.count(b"data") == 1 and b"JUNK" not in ids
    d = dict(got)
    assert d[b"LIST"] == info and d[b"cue "] == cue and d[b"bext"] == bext
    y = np.frombuffer(d[b"data"], "<f4")
    assert y.shape == x.shape and np.abs(y - orc.cli_apply_file_mono(P, x, 60)).max() < 2e-6


@pytest.mark.gpu
def test_cli_hostile_wav_headers_are_refused(tmp_path):
    """A crafted `fmt ` size must not turn into a multi-GiB allocation, and formats outside PCM 16/24/32 and
    float 32/64 are refused with a message and EXIT_FAILURE, like sf_open failing in the reference (cli/phase-rotate.cc:685-688)."""
    bad = b"RIFF" + struct.pack("<I", 100) + b"WAVEfmt " + struct.pack("<I", 0x7FFFFFF0) + bytes(64)
    adpcm = _riff(struct.pack("<HHIIHH", 2, 1, 48000, 48000, 256, 4), bytes(1024))
    eight = _riff(_fmt(1, 1, 48000, 8), bytes(4096))
    for i, blob in enumerate((bad, adpcm, eight, b"FORM" + bytes(60))):
        p = str(tmp_path / ("bad%d.wav" % i))
        open(p, "wb").write(blob)
        r = subprocess.run([CLI, p], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)
        assert r.returncode == 1 and "Cannot open" in r.stderr or "nsupported" in r.stderr or "unknown format" in r.stderr, (i, r.returncode, r.stderr)
