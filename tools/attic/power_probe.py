"""Diagnostic: board power / shader clock while the headline kernel runs back to back (is the chip power-limited?).
Reads the amdgpu hwmon files every 20 ms from a thread while the main thread keeps the stream full for ~4 s."""
import glob, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()

def rd(p):
    try:
        return open(p).read().strip()
    except Exception:
        return None

hw = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
print("hwmon dirs:", hw)
for h in hw:
    print(" ", h, {f: rd(os.path.join(h, f)) for f in ("name", "power1_cap", "power1_cap_max", "power1_average", "power1_input", "freq1_input", "freq2_input", "temp1_input")})
samples, stop = [], False
def poll():
    while not stop:
        row = [time.time()]
        for h in hw[:1]:
            for f in ("power1_average", "power1_input", "freq1_input"):
                row.append(rd(os.path.join(h, f)))
        samples.append(row)
        time.sleep(0.02)

cases = {"poly": (pkg.KIND_LV2, 48000, 0), "cli": (pkg.KIND_CLI, 2048, 0)}
kind, param, fl = cases[sys.argv[1] if len(sys.argv) > 1 else "poly"]
dev = torch.device("cuda:0")
n = 1 << 27
x = (torch.rand(n, device=dev) - 0.5); y = torch.empty_like(x)
rot = np.array([0.0, -1.0], np.float32)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); st = ts.cuda_stream
e = pkg.Engine(kind, param, 1, flags=fl)
th = threading.Thread(target=poll); th.start()
time.sleep(0.5)
t0 = time.time()
k = 0
while time.time() - t0 < 4.0:
    for _ in range(50):
        e.apply_dev(x.data_ptr(), n, n, y.data_ptr(), n, n, rot, pkg.F_STATELESS, st)
    torch.cuda.synchronize(); k += 50
t1 = time.time()
time.sleep(0.5)
stop = True; th.join()
print("launches %d, %.4f ms each (wall, incl. sync gaps)" % (k, (t1 - t0) / k * 1e3))
def col(i, lo, hi):
    v = [float(r[i]) for r in samples if r[i] is not None and lo <= r[0] <= hi]
    return (min(v), float(np.median(v)), max(v)) if v else None
print("idle   (first 0.4 s): power_avg/power_in/freq1 min/med/max:", [col(i, 0, t0 - 0.1) for i in (1, 2, 3)])
print("loaded (middle)     :", [col(i, t0 + 1.0, t1 - 0.2) for i in (1, 2, 3)])
try:
    import subprocess
    print(subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=30).stdout[-1500:])
except Exception as ex:
    print("rocm-smi:", ex)
