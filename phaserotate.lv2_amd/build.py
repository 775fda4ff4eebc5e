"""Build recipes for the gfx950 shared library and the host-side C front ends.

Everything is built IN-TREE (phaserotate.lv2_amd/lib/) so the artefacts travel
with the repo snapshot to the GPU box.  hipcc cross-compiles for gfx950
without a GPU present.
"""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
HOST = os.path.join(PKG, "host")
LIBDIR = os.path.join(PKG, "lib")
OBJDIR = os.path.join(PKG, "build")
LIB = os.path.join(LIBDIR, "libprot_hip.so")              # the shipped library: exports include/prot_hip.h's symbols, nothing else
LIB_HOOKS = os.path.join(LIBDIR, "libprot_hip_hooks.so")  # the same sources with -DPROT_TESTHOOKS: + csrc/prot_internal.h (tests, bench.py)
ARCH = "gfx950"


def declared_symbols():
    """the functions include/prot_hip.h declares"""
    import re
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "prot_hip.h")).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(prot_[a-z0-9_]+)\s*\(", src)))


def _version_script(path, names):
    with open(path, "w") as f:
        f.write("{\n global:\n" + "".join("  %s;\n" % n for n in names) + " local:\n  *;\n};\n")
    return path


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the gfx950 library cannot be built")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("build failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def build_lib(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> phaserotate.lv2_amd/lib/libprot_hip.so"""
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in ("prot_fft.h", "prot_fft4096.h", "prot_kernels.h", "prot_dev.h", "prot_ablate.h", "prot_olasave_asm.inc", "prot_olasave8k_asm.inc", "prot_olasave16k_asm.inc")] + \
           [os.path.join(ROOT, "include", "prot_hip.h")]
    common = [hipcc, "--offload-arch=" + ARCH, "-O3", "-fPIC", "-std=c++17", "-Wall",
              "-Wno-unused-function", "-I" + os.path.join(ROOT, "include")]
    units = [
        # device code: fast-math lets the compiler prune the zero-padded radix-16 inputs;
        # no SLP: hipcc otherwise packs the butterflies into v_pk_*_f32 with ~500 v_mov per
        # iteration to marshal register pairs (measured: 1500 -> 1240 VALU, 255 -> 165 VGPRs);
        # MFMA results in VGPRs (unified file on gfx950) so the sweep's v_max3 reads them directly
        ("prot_kernels.hip", ["-ffast-math", "-fno-slp-vectorize"]),
        ("prot_olapoly.hip", ["-ffast-math", "-fno-slp-vectorize"]),
        ("prot_sweep.hip", ["-ffast-math", "-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form"]),
        ("prot_kernels_f16.hip", ["-ffast-math"]),
        ("prot_f16mfma.hip", ["-ffast-math", "-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form"]),
        ("prot_olasave.hip", ["-ffast-math", "-fno-slp-vectorize"]),
        ("prot_olasave8k.hip", ["-ffast-math", "-fno-slp-vectorize"]),
        ("prot_olasave16k.hip", ["-ffast-math", "-fno-slp-vectorize"]),
        # (the launched kernel's first fourteen argument dwords preloaded into SGPRs at wave launch: prot_olasave512.hip)
        ("prot_olasave512.hip", ["-ffast-math", "-fno-slp-vectorize", "-mllvm", "-amdgpu-kernarg-preload-count=14"]),
    ]
    # the host side of the C ABI (round 6: prot_capi.hip split into its parts; prot_engine.h is what they share).  Host logic mirrors the
    # reference's float arithmetic: no fast-math.  Compiled twice: as shipped, and with -DPROT_TESTHOOKS (prot_internal_* entry points,
    # the test options) for libprot_hip_hooks.so -- the device units are the same object code in both libraries
    HOST_UNITS = ("prot_engine.hip", "prot_apply.hip", "prot_lv2.hip", "prot_sweep_host.hip", "prot_multi.hip")
    HOOK_UNITS = HOST_UNITS + ("prot_hooks.hip",)
    objs = []
    for src, extra in units:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        if force or _newer(o, [s] + hdrs):
            out = _run(common + extra + ["-c", s, "-o", o])
            if verbose and out:
                print(out)
        objs.append(o)
    host_hdrs = hdrs + [os.path.join(CSRC, "prot_engine.h"), os.path.join(CSRC, "prot_internal.h")]
    objs_h = list(objs)
    for src in HOST_UNITS:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        if force or _newer(o, [s] + host_hdrs):
            _run(common + ["-c", s, "-o", o])
        objs.append(o)
    for src in HOOK_UNITS:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJDIR, src.replace(".hip", "_hooks.o"))
        if force or _newer(o, [s] + host_hdrs):
            _run(common + ["-DPROT_TESTHOOKS", "-c", s, "-o", o])
        objs_h.append(o)
    names = declared_symbols()
    internal = ["prot_internal_set_rccl_library", "prot_internal_set_test_option", "prot_internal_fused_lists", "prot_internal_time_lv2_run", "prot_internal_time_lv2_run_paced", "prot_internal_time_lv2_run_multi",
                "prot_internal_server_stats"]
    if force or _newer(LIB, objs + [os.path.join(ROOT, "include", "prot_hip.h")]):
        vs = _version_script(os.path.join(OBJDIR, "exports.map"), names)
        _run([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-Wl,--version-script=" + vs, "-o", LIB] + objs + ["-lpthread"])
    if force or _newer(LIB_HOOKS, objs_h + [os.path.join(ROOT, "include", "prot_hip.h")]):
        vs = _version_script(os.path.join(OBJDIR, "exports_hooks.map"), names + internal)
        _run([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-Wl,--version-script=" + vs, "-o", LIB_HOOKS] + objs_h + ["-lpthread"])
    return LIB


def build_host(force=False):
    """gcc/g++ -> the LV2 plugin .so and the phase-rotate CLI (host C/C++ over the C ABI)."""
    os.makedirs(LIBDIR, exist_ok=True)
    built = []
    inc = "-I" + os.path.join(ROOT, "include")
    plug_src = os.path.join(HOST, "lv2_plugin.c")
    if os.path.exists(plug_src):
        plug = os.path.join(LIBDIR, "phaserotate.so")
        if force or _newer(plug, [plug_src, LIB]):
            _run(["gcc", "-O2", "-fPIC", "-shared", "-fvisibility=hidden", "-std=gnu99", "-Wall", inc,
                  "-I" + HOST, plug_src, "-o", plug, "-L" + LIBDIR, "-lprot_hip",
                  "-Wl,-rpath,$ORIGIN", "-lm", "-lpthread"])
        built.append(plug)
    cli_src = os.path.join(HOST, "phase_rotate_cli.cc")
    if os.path.exists(cli_src):
        cli = os.path.join(LIBDIR, "phase-rotate")
        deps = [cli_src, LIB] + [os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".h")]
        if force or _newer(cli, deps):
            _run(["g++", "-O2", "-std=c++11", "-Wall", inc, "-I" + HOST, cli_src, "-o", cli,
                  "-L" + LIBDIR, "-lprot_hip", "-Wl,-rpath,$ORIGIN", "-lm", "-lpthread"])
        built.append(cli)
    return built


def build_all(force=False):
    lib = build_lib(force)
    return [lib, LIB_HOOKS] + build_host(force)


if __name__ == "__main__":
    import sys
    for p in build_all(force="--force" in sys.argv):
        print("built", p)
