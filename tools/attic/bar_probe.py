"""Can the CPU store into device memory (large BAR), and how fast?  The run() server would rather be handed its input than fetch it over
the host link (4 KiB at ~1 GB/s from a lone workgroup): measured here, the FIRST 4 KiB store into a page of device memory takes
20-30 us, every later one ~0.2 us (write-combined, posted), a CPU LOAD through the BAR 5-7 us -- so the mailbox's host -> device half
lives in device memory and is never read by the host (prot_lv2.hip, lv2_run_chunk).
    python3 tools/bar_probe.py   (on the GPU box; each allocation kind in a child process)"""
import ctypes as C, os, sys, time
hip = C.CDLL("libamdhip64.so")
def chk(r, what):
    if r != 0: print(what, "->", r, flush=True)
    return r
for name, flags in (("hipMalloc", None), ("finegrained", 0x1), ("uncached", 0x3)):
    pid = os.fork()
    if pid == 0:
        p = C.c_void_p()
        if flags is None:
            r = hip.hipMalloc(C.byref(p), 1 << 16)
        else:
            hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
            r = hip.hipExtMallocWithFlags(C.byref(p), 1 << 16, flags)
        if chk(r, name): os._exit(2)
        hip.hipMemset(p, 0, 1 << 16); hip.hipDeviceSynchronize()
        src = (C.c_uint * 1024)(*range(1024))
        t0 = time.perf_counter()
        C.memmove(p.value, src, 4096)      # CPU stores into device memory (through the BAR, if it is mapped)
        dt = time.perf_counter() - t0
        back = (C.c_uint * 1024)()
        hip.hipMemcpy(back, p, 4096, 2)    # D2H
        ok = list(back) == list(range(1024))
        # again, now that the pages are mapped: 200 stores of 4 KiB, and of 64 B
        ts = []
        for k in range(200):
            t0 = time.perf_counter(); C.memmove(p.value + 4096 * (k % 8), src, 4096); ts.append(time.perf_counter() - t0)
        ts.sort()
        t64 = []
        for k in range(200):
            t0 = time.perf_counter(); C.memmove(p.value + 64 * (k % 64), src, 64); t64.append(time.perf_counter() - t0)
        t64.sort()
        print("%s: repeated 4 KiB stores: median %.2f us, p90 %.2f; 64 B stores: median %.2f us (ctypes.memmove overhead included)"
              % (name, ts[100] * 1e6, ts[180] * 1e6, t64[100] * 1e6), flush=True)
        hip.hipMemcpy(back, C.c_void_p(p.value + 4096 * 7), 4096, 2)
        print("%s: last repeated store seen by the device: %s" % (name, list(back) == list(range(1024))), flush=True)
        t0 = time.perf_counter(); v = C.c_uint.from_address(p.value + 40).value; dr = time.perf_counter() - t0
        print("%s: host write of 4 KiB took %.2f us, device sees it: %s; host read of one word %.2f us -> %d" % (name, dt * 1e6, ok, dr * 1e6, v), flush=True)
        os._exit(0)
    _, st = os.waitpid(pid, 0)
    print(name, "child status", st, flush=True)
