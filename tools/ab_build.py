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
# variant libraries read the tuning switches of the host units from the environment (-DPROT_TUNING); the shipped library never does
HOST = ("prot_engine", "prot_apply", "prot_lv2", "prot_sweep_host", "prot_multi", "prot_hooks")
HOST_T = [os.path.join(b.OBJDIR, u + "_tuning.o") for u in HOST]
for u, o in zip(HOST, HOST_T):
    if b._newer(o, [os.path.join(b.CSRC, u + ".hip"), os.path.join(b.CSRC, "prot_engine.h"), os.path.join(b.CSRC, "prot_kernels.h")]):
        b._run([hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-function", "-DPROT_TUNING", "-DPROT_TESTHOOKS",
                "-I" + os.path.join(ROOT, "include"), "-c", os.path.join(b.CSRC, u + ".hip"), "-o", o])
UNITS = ("prot_kernels", "prot_olapoly", "prot_olasave", "prot_olasave8k", "prot_olasave16k", "prot_sweep", "prot_kernels_f16", "prot_f16mfma", "prot_olasave512")


def link(tag, replaced):
    """libprot_hip_<tag>.so from the shipped objects, `replaced` = {unit: object} swapped in"""
    lib = os.path.join(vdir, "libprot_hip_%s.so" % tag)
    objs = [replaced.get(u, os.path.join(b.OBJDIR, u + ".o")) for u in UNITS]
    b._run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + HOST_T + ["-lpthread"])
    print("built", lib)


for arg in sys.argv[1:]:
    if arg.startswith("sw:"):
        # "sw:<tag>:flag,flag": a variant of prot_sweep.hip
        _, tag, fl = (arg.split(":") + [""])[:3]
        o = os.path.join(b.OBJDIR, "prot_sweep_%s.o" % tag)
        b._run([hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-function", "-I" + os.path.join(ROOT, "include"), "-ffast-math",
                "-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form"] + [f for f in fl.split(",") if f] + ["-c", os.path.join(b.CSRC, "prot_sweep.hip"), "-o", o])
        link(tag, {"prot_sweep": o})
        continue
    if arg.startswith("sv8:"):
        # "sv8:<tag>[:flag,flag]": a variant of prot_olasave8k.hip (the generated loop csrc/variants/prot_olasave8k_asm_<tag>.inc
        # if it exists: python3 tools/gen/gen_olasave8k.py --variant <tag> --opts ...), linked with the shipped objects
        _, tag, fl = (arg.split(":") + [""])[:3]
        extra = [f for f in fl.split(",") if f]
        inc = os.path.join(b.CSRC, "variants", "prot_olasave8k_asm_%s.inc" % tag)
        if os.path.exists(inc):
            extra.append('-DPROT_SV8_INC="%s"' % inc)
        o = os.path.join(b.OBJDIR, "prot_olasave8k_%s.o" % tag)
        b._run([hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-function",
                "-I" + os.path.join(ROOT, "include"), "-ffast-math", "-fno-slp-vectorize"] + extra +
               ["-c", os.path.join(b.CSRC, "prot_olasave8k.hip"), "-o", o])
        link(tag, {"prot_olasave8k": o})
        continue
    if arg.startswith("sv:"):
        # "sv:<tag>[:flag,flag]": a variant of prot_olasave.hip only (the generated loop csrc/variants/prot_olasave_asm_<tag>.inc
        # if it exists: python3 tools/gen/gen_olasave.py --variant <tag> --opts ...), linked with the shipped objects
        _, tag, fl = (arg.split(":") + [""])[:3]
        extra = [f for f in fl.split(",") if f]
        inc_tag = tag
        for f in list(extra):
            if f.startswith("INC="):   # the generated loop of another variant under this tag's flags
                inc_tag = f[4:]
                extra.remove(f)
        inc = os.path.join(b.CSRC, "variants", "prot_olasave_asm_%s.inc" % inc_tag)
        if os.path.exists(inc):
            extra.append('-DPROT_SV_INC="%s"' % inc)
        o = os.path.join(b.OBJDIR, "prot_olasave_%s.o" % tag)
        b._run([hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-function",
                "-I" + os.path.join(ROOT, "include"), "-ffast-math", "-fno-slp-vectorize"] + extra +
               ["-c", os.path.join(b.CSRC, "prot_olasave.hip"), "-o", o])
        link(tag, {"prot_olasave": o})
        continue
    if arg.startswith("s5:"):
        # "s5:<tag>[:flag,flag]": a variant of prot_olasave512.hip (the 512-thread short-chain block)
        _, tag, fl = (arg.split(":") + [""])[:3]
        o = os.path.join(b.OBJDIR, "prot_olasave512_%s.o" % tag)
        b._run([hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-function", "-I" + os.path.join(ROOT, "include"), "-ffast-math",
                "-fno-slp-vectorize"] + ([] if "NOPRELOAD" in fl else ["-mllvm", "-amdgpu-kernarg-preload-count=14"]) + [f for f in fl.split(",") if f and f != "NOPRELOAD"] +
               ["-c", os.path.join(b.CSRC, "prot_olasave512.hip"), "-o", o])
        link(tag, {"prot_olasave512": o})
        continue
    tag, _, fl = arg.partition(":")
    extra = [f for f in fl.split(",") if f]
    o = os.path.join(b.OBJDIR, "prot_kernels_%s.o" % tag)
    b._run([hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-function",
            "-I" + os.path.join(ROOT, "include"), "-ffast-math", "-fno-slp-vectorize", "-mllvm",
            "-amdgpu-mfma-vgpr-form"] + extra + ["-c", os.path.join(b.CSRC, "prot_kernels.hip"), "-o", o])
    link(tag, {"prot_kernels": o})
