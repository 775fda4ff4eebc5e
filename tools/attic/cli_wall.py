"""Where does the wall time of the `phase-rotate` CLI go on a long file?  Writes a stereo 96 kHz WAV of `minutes` minutes
(PCM 24 or float 32) under /tmp, runs the CLI with PROT_CLI_TIMING=1 (analysis + apply + write), prints the phases.
usage: python tools/cli_wall.py [minutes=10] [pcm24|f32]"""
import os, struct, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CLI = os.path.join(ROOT, "phaserotate.lv2_amd", "lib", "phase-rotate")
minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
kind = sys.argv[2] if len(sys.argv) > 2 else "pcm24"
rate, chn = 96000, 2
n = int(minutes * 60 * rate)
t = np.arange(n, dtype=np.float64) / rate
rng = np.random.default_rng(1)
x = np.zeros((n, chn), np.float32)
for c in range(chn):
    f0 = 110.0 * (1 + 0.25 * c)
    s = sum(np.sin(2 * np.pi * f0 * k * t + 0.7 * k * k) / k for k in range(1, 12))
    x[:, c] = (0.25 * s * (0.6 + 0.4 * np.sin(2 * np.pi * 0.05 * t)) + 0.01 * rng.standard_normal(n)).astype(np.float32)
path, out = "/tmp/cli_wall_in.wav", "/tmp/cli_wall_out.wav"
if kind == "pcm24":
    q = np.clip(np.rint(x.astype(np.float64) * 8388607.0), -8388608, 8388607).astype(np.int32).reshape(-1)
    b = np.empty((q.size, 3), np.uint8)
    b[:, 0], b[:, 1], b[:, 2] = q & 255, (q >> 8) & 255, (q >> 16) & 255
    raw, tag, bits = b.tobytes(), 1, 24
else:
    raw, tag, bits = x.tobytes(), 3, 32
bps = bits // 8
hdr = b"RIFF" + struct.pack("<I", 36 + len(raw)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, tag, chn, rate, rate * bps * chn, bps * chn, bits) + b"data" + struct.pack("<I", len(raw))
open(path, "wb").write(hdr + raw)
print("file: %.1f min stereo 96 kHz %s, %.0f MB" % (minutes, kind, (len(raw) + 44) / 1e6))
env0 = dict(os.environ, PROT_CLI_TIMING="1")
runs = [([CLI, "-s", "1", path], "analysis only (-s 1)", {}),
        ([CLI, path], "analysis only, default stride (coarse pass + fine passes: the file is re-analysed per candidate)", {}),
        ([CLI, path], "same, file streamed from the host for every pass (PROT_CLI_NO_RESIDENT=1)", {"PROT_CLI_NO_RESIDENT": "1"}),
        ([CLI, "-s", "1", path, out], "analysis + apply + write", {}),
        ([CLI, "-s", "1", path, out], "analysis + apply + write, file streamed from the host (PROT_CLI_NO_RESIDENT=1)", {"PROT_CLI_NO_RESIDENT": "1"}),
        ([CLI, "-s", "1", path, out], "analysis + apply + write (again)", {})]
for args, what, extra in runs:
    env = dict(env0, **extra)
    t0 = time.perf_counter()
    r = subprocess.run(args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    w = time.perf_counter() - t0
    print("== %s: wall %.2f s, rc %d" % (what, w, r.returncode))
    print("\n".join(l for l in r.stderr.splitlines() if "timing" in l))
    print("\n".join(r.stdout.splitlines()[:4]))
os.remove(path)
if os.path.exists(out):
    os.remove(out)
