"""Build tuning variants of libprot_hip.so: `python tools/ab_build.py tag1:-DX=1 tag2:-DX=0,-DY=2`
-> phaserotate.lv2_amd/lib/variants/libprot_hip_<tag>.so (same sources, extra flags for prot_kernels.hip)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util
spec = importlib.util.spec_from_file_location("prot_build", os.path.join(ROOT, "phaserotate.lv2_amd", "build.py"))
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
b.build_lib()
vdir = os.path.join(b.LIBDIR, "variants"); os.makedirs(vdir, exist_ok=True)
hipcc = b._hipcc()
for arg in sys.argv[1:]:
    tag, _, fl = arg.partition(":")
    extra = [f for f in fl.split(",") if f]
    o = os.path.join(b.OBJDIR, "prot_kernels_%s.o" % tag)
    b._run([hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-function",
            "-I" + os.path.join(ROOT, "include"), "-ffast-math", "-fno-slp-vectorize", "-mllvm",
            "-amdgpu-mfma-vgpr-form"] + extra + ["-c", os.path.join(b.CSRC, "prot_kernels.hip"), "-o", o])
    lib = os.path.join(vdir, "libprot_hip_%s.so" % tag)
    b._run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, o,
            os.path.join(b.OBJDIR, "prot_kernels_f16.o"), os.path.join(b.OBJDIR, "prot_olasave.o"),
            os.path.join(b.OBJDIR, "prot_capi.o"), "-lpthread"])
    print("built", lib)
