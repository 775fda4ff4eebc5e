"""Diagnostic: where a wave of the headline polyphase kernel spends its cycles.
Needs the stamped variant library: `python tools/ab_build.py stamps:-DPROT_STAMPS=1`, then
`PROT_LIB_PATH=phaserotate.lv2_amd/lib/variants/libprot_hip_stamps.so python tools/phase_stamps.py [case]`.
Prints the per-phase average shader cycles per wave-iteration (summed over all waves / wave-iterations)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
lib = pkg.load_library()
lib.prot_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int, C.c_int]
cases = {"poly": (pkg.KIND_LV2, 48000, 0), "cli": (pkg.KIND_CLI, 2048, 0), "cli1024": (pkg.KIND_CLI, 1024, 0),
         "poly_96k": (pkg.KIND_LV2, 96000, 4)}
name = sys.argv[1] if len(sys.argv) > 1 else "poly"
kind, param, fl = cases[name]
dev = torch.device("cuda:0")
n = 1 << 27
x = (torch.rand(n, device=dev) - 0.5); y = torch.empty_like(x)
rot = np.array([0.0, -1.0], np.float32)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); st = ts.cuda_stream
e = pkg.Engine(kind, param, 1, flags=fl)
for _ in range(30):
    e.apply_dev(x.data_ptr(), n, n, y.data_ptr(), n, n, rot, pkg.F_STATELESS, st)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 32)()
lib.prot_debug_stamps(buf, 32, 1)
reps = 10
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(reps):
    e.apply_dev(x.data_ptr(), n, n, y.data_ptr(), n, n, rot, pkg.F_STATELESS, st)
ev1.record(); torch.cuda.synchronize()
lib.prot_debug_stamps(buf, 32, 0)
v = np.array(list(buf), dtype=np.float64)
wave_it, waves = v[16], v[19]
names = ["0 top: wait loads, DPP shift", "1 fwd pass 1 + twiddle + LDS store", "2 barrier A", "3 reads, pass 2, local exchange",
         "4 H prefetch + pass 3", "5 spectrum multiply", "6 issue next loads + dry", "7 inv pass 1, local xchg, pass 2, store",
         "8 barrier B", "9 reads, twiddle, inv pass 3", "10 OLA + rotate + stores"]
print("%s: %.4f ms per launch (stamped build); %d waves, %.1f iterations per wave; clock %.0f MHz" % (
    name, ev0.elapsed_time(ev1) / reps, waves / reps, wave_it / waves, v[17] / v[18] * 100.0))
tot = v[:11].sum()
for i in range(11):
    print("  %-44s %8.0f cycles  %5.1f %%" % (names[i], v[i] / wave_it, 100.0 * v[i] / tot))
print("  %-44s %8.0f cycles per wave-iteration" % ("sum", tot / wave_it))
