#!/usr/bin/env python3
"""Where a run() call through the server goes (prot_olasave512.hip, olasave512_server_kernel): 100 MHz stamps of the resident kernel's stages
-- doorbell seen, parameters + input fetched, block done, done word stored -- and of the block's own phases, from a variant library
built with -DPROT_SERVER_STAMPS.
    python3 tools/server_stamps.py --build                      (here: cross-compiles phaserotate.lv2_amd/lib/variants/libprot_hip_stamps.so)
    PROT_LIB_PATH=.../variants/libprot_hip_stamps.so python3 tools/server_stamps.py     (on the GPU box)"""
import sys
if "--build" in sys.argv:
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("prot_build", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "phaserotate.lv2_amd", "build.py"))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    b.build_lib()
    hipcc=b._hipcc()
    fl=[hipcc,"--offload-arch=gfx950","-O3","-fPIC","-std=c++17","-Wno-unused-function","-DPROT_SERVER_STAMPS","-DPROT_TESTHOOKS","-I" + os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")]
    o1=os.path.join(b.OBJDIR,"prot_olasave512_stamps.o")   # (round 6: the server lives with the 512-thread block)
    b._run(fl+["-ffast-math","-fno-slp-vectorize","-mllvm","-amdgpu-kernarg-preload-count=14","-c",os.path.join(b.CSRC,"prot_olasave512.hip"),"-o",o1])
    host=[]
    for u in ("prot_engine","prot_apply","prot_lv2","prot_sweep_host","prot_multi","prot_hooks"):   # (the host units of the C ABI, with the stamps' hook)
        o=os.path.join(b.OBJDIR,u+"_stamps.o"); b._run(fl+["-c",os.path.join(b.CSRC,u+".hip"),"-o",o]); host.append(o)
    lib=os.path.join(b.LIBDIR,"variants","libprot_hip_stamps.so")
    os.makedirs(os.path.dirname(lib),exist_ok=True)
    b._run([hipcc,"--offload-arch=gfx950","-shared","-fPIC","-o",lib,o1]+host+[os.path.join(b.OBJDIR,x) for x in ("prot_kernels.o","prot_olapoly.o","prot_olasave.o","prot_olasave8k.o","prot_olasave16k.o","prot_sweep.o","prot_kernels_f16.o","prot_f16mfma.o")]+["-lpthread"])
    print("built",lib)
    sys.exit(0)
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from __graft_entry__ import load_package
pkg = load_package()
lib = pkg.load_library()
for nchn, bs in ((1, 1024), (1, 64), (2, 1024)):
    g = pkg.Lv2Instance(48000, nchn); g.activate()
    x = (np.random.default_rng(1).random((nchn, bs), dtype=np.float32) - 0.5)
    for _ in range(300): g.run(x, 90.0)
    acc = []
    for _ in range(200):
        t0 = time.perf_counter(); g.run(x, 90.0); dt = (time.perf_counter() - t0) * 1e6
        v = (C.c_uint * 4)()
        lib.prot_internal_server_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_uint)]
        lib.prot_internal_server_stamps(g._h, v)
        acc.append((dt, (v[1] - v[0]) / 100.0, (v[2] - v[1]) / 100.0, (v[3] - v[2]) / 100.0))
    a = np.median(np.array(acc), 0)
    pkg.set_test_option(pkg.TEST_LV2_STAGED, 0)
    g.sync() if hasattr(g, "sync") else None
    tt = (C.c_uint * 16)()
    lib.prot_engine_sync.argtypes = [C.c_void_p]; lib.prot_engine_sync(g._h)
    lib.prot_debug_srv_t(tt)
    print("   body: entry->stage start %.2f, staging %.2f, core %.2f, output %.2f, history %.2f us" % ((tt[0]-tt[5])/100., (tt[1]-tt[0])/100., (tt[2]-tt[1])/100., (tt[3]-tt[2])/100., (tt[4]-tt[3])/100.))
    print("chn %d buf %d: run() %.1f us; device: doorbell->params %.2f us, params->block done %.2f us, block done->done stored %.2f us" % (nchn, bs, *a), flush=True)
    g.close()
