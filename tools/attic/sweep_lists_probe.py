"""The fused sweep pass's lists, inspected: two runs over the same stream must leave the same lists (same order, same bits), and their
union must be exactly the frames with d^2 + h^2 above the bound (here 0: every frame with a non-zero pair)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
lib = pkg.load_library()
lib.prot_internal_fused_lists.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
dev = torch.device("cuda:0")
for val in (1.0e30, float("nan")):
    big = torch.full((int(40e9) // 4,), val, device=dev); torch.cuda.synchronize(); del big; torch.cuda.empty_cache()
P, n_chn = 2048, 2
n = 3001 * P - 777
g = torch.Generator(device=dev).manual_seed(2)
x = (torch.rand((n_chn, n), generator=g, device=dev) - 0.5) * 0.9
def lists():
    e = pkg.Engine(pkg.KIND_CLI, P, n_chn)
    pk = torch.zeros((n_chn, 360), device=dev)
    torch.cuda.synchronize()
    e.sweep_dev(x.data_ptr(), n, n, 0, 360, 1, -1, pk.data_ptr(), pkg.F_STREAM_START | pkg.F_STREAM_END)
    e.sync()
    o = (C.c_uint64 * 4)()
    assert lib.prot_internal_fused_lists(e._h, o) == 0
    surv, cntp, nl, cap = int(o[0]), int(o[1]), int(o[2]), int(o[3])
    cnt = np.empty(n_chn * nl, np.uint32)
    buf = np.empty((n_chn * nl * cap, 2), np.float32)
    import ctypes
    hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    hip.hipMemcpy(cnt.ctypes.data_as(C.c_void_p), C.c_void_p(cntp), C.c_size_t(cnt.nbytes), 2)
    hip.hipMemcpy(buf.ctypes.data_as(C.c_void_p), C.c_void_p(surv), C.c_size_t(buf.nbytes), 2)
    e.close()
    return cnt.reshape(n_chn, nl), buf.reshape(n_chn, nl, cap, 2), pk.cpu().numpy()
ref = lists()
print("lists per channel", ref[0].shape[1], "cap", ref[1].shape[2], "entries per channel", ref[0].sum(axis=1))
for rep in range(10):
    cur = lists()
    same_cnt = np.array_equal(cur[0], ref[0])
    bad = []
    if same_cnt:
        for c in range(n_chn):
            for l in range(cur[0].shape[1]):
                k = ref[0][c, l]
                if not np.array_equal(cur[1][c, l, :k], ref[1][c, l, :k]):
                    bad.append((c, l, int(np.argmax((cur[1][c, l, :k] != ref[1][c, l, :k]).any(axis=1)))))
    print("rep", rep, "counts equal", same_cnt, "lists that differ", bad[:6], "table equal", np.array_equal(cur[2], ref[2]),
          "" if same_cnt else np.argwhere(cur[0] != ref[0])[:5].tolist())
c0, c1 = ref[0][0], ref[0][1]
print("channel 0 list counts: first 12", c0[:12].tolist(), "... around 1020:", c0[1016:1032].tolist(), "last 8", c0[-8:].tolist())
print("channel 1 list counts: first 12", c1[:12].tolist(), "... around 1020:", c1[1016:1032].tolist(), "last 8", c1[-8:].tolist())
import collections
print("channel 0 distinct counts", collections.Counter(c0.tolist()).most_common(6))
print("channel 1 distinct counts", collections.Counter(c1.tolist()).most_common(6))
# which frames are missing from a partial list of channel 1?
c = 1
part = [l for l in range(8, ref[0].shape[1] - 8) if ref[0][c, l] < 3072][:3]
xc = x[c].cpu().numpy()
BLK, Dd = 6144, 1024
for l in part:
    g_, w = l // 4, l % 4
    bx = (g_ & ~7) | ((g_ + (g_ >> 3)) & 7)   # run_of_workgroup's XCD permutation
    r0 = bx * 12288
    want = []
    for b in range(2):
        s = r0 + b * BLK
        for k in range(12):
            for e in range(2):
                for lane in range(64):
                    pos = s + 512 * k + 2 * (64 * w + lane) + e
                    src = pos - Dd
                    want.append((pos, xc[src] if 0 <= src < n else 0.0))
    got = ref[1][c, l, :ref[0][c, l], 0]
    gi, missing = 0, []
    for pos, dval in want:
        if gi < len(got) and got[gi] == np.float32(dval):
            gi += 1
        else:
            missing.append(pos)
    m = np.array(missing)
    print("list", l, "count", ref[0][c, l], "matched", gi, "missing", len(m), "first missing pos (rel. to run)", (m[:10] - r0).tolist(), "last", (m[-5:] - r0).tolist(),
          "missing by row:", np.bincount(((m - r0) % BLK) // 512, minlength=12).tolist())
# which frames of the second block ARE listed in a partial list
l = part[0]; g_, w = l // 4, l % 4
bx = (g_ & ~7) | ((g_ + (g_ >> 3)) & 7); r0 = bx * 12288
got = ref[1][c, l, :ref[0][c, l]]
tail = got[1536:]
pos_of = {}
for k in range(12):
    for e in range(2):
        for lane in range(64):
            pos = r0 + BLK + 512 * k + 2 * (64 * w + lane) + e
            pos_of.setdefault(np.float32(xc[pos - Dd]).tobytes(), []).append((k, e, lane, pos - r0))
print("entries of the second block in list", l, ":")
for d_, h_ in tail[:24]:
    print("   d %.6f h %.6f ->" % (d_, h_), pos_of.get(np.float32(d_).tobytes(), "?"))
