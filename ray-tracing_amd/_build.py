"""Compile the native pieces in-tree (no JIT cache): the HIP C-ABI library for gfx950 and the
C++ host library.  Run here on CPU (hipcc cross-compiles); the built .so files travel to the GPU
box with the tree."""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
HOST = os.path.join(PKG, "host")
LIB_HIP = os.path.join(PKG, "librt_hip.so")
LIB_HOST = os.path.join(PKG, "librt_host.so")
OBJ = os.path.join(PKG, "csrc", "_obj")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
COMMON = ["--offload-arch=" + ARCH, "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include")]
# (source, extra flags).  The f64 unit must round every operation separately (the reference's
# Rust f64 code has no fused multiply-add), so contraction is off there and only there.
UNITS = [
    ("rt_kernels_f64.hip", ["-O3", "-ffp-contract=off"]),
    ("rt_kernels_f32.hip", ["-O3", "-ffp-contract=fast"]),
    ("rt_hip.hip", ["-O2"]),
    ("rt_exchange.hip", ["-O2"]),  # RCCL all-reduce entry points; RCCL itself is bound at run time (dlopen)
]
HOST_SRC = ["scene_api.cpp", "scenes.cpp", "flatten.cpp", "jpeg.cpp", "capi.cpp"]
HOST_FLAGS = ["-std=c++17", "-O2", "-march=x86-64-v3", "-ffp-contract=off", "-fPIC", "-Wall", "-Wextra", "-shared"]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout)
        raise RuntimeError("build failed: " + " ".join(cmd))


def source_id(defines=(), f64_flags=()):
    """Identity of a build of librt_hip.so: a hash of every source the device code is made of (csrc/, include/rt_hip.h) and of the
    flags.  The library reports it (rt_describe: "build=<id>"), bench.py prints it, and tools/pmc_summary.py stores it with every
    profile summary, so a number looked up from profiles/ can be told from one measured on the build that ran."""
    import hashlib

    h = hashlib.sha256()
    files = sorted(f for f in os.listdir(CSRC) if f.endswith((".hpp", ".inc", ".hip")))
    for f in files:
        h.update(f.encode() + b"\0" + open(os.path.join(CSRC, f), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "rt_hip.h"), "rb").read())
    h.update(repr((UNITS, COMMON[1:], sorted(defines), list(f64_flags))).encode())
    return h.hexdigest()[:12]


def build_hip(force=False, defines=(), out=None, tag="", f64_flags=()):
    """defines/out/tag/f64_flags: build an A/B variant library (tools/build_ab.py) beside the product one."""
    global LIB_HIP
    lib_out = out or LIB_HIP
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".inc"))]
    headers.append(os.path.join(ROOT, "include", "rt_hip.h"))
    all_src = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip")]
    build_id = source_id(defines, f64_flags)
    objs = []
    for src, flags in UNITS:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", tag + ".o"))
        # rt_hip.hip carries the build id of ALL the sources: rebuilt when any of them changes
        deps = [s] + headers + (all_src if src == "rt_hip.hip" else [])
        if force or _newer(o, deps):
            extra = list(f64_flags) if src == "rt_kernels_f64.hip" else []
            if src == "rt_hip.hip":
                extra.append('-DRT_BUILD_ID="%s"' % build_id)
            _run([HIPCC] + COMMON + flags + extra + ["-D" + d for d in defines] + ["-c", s, "-o", o])
        objs.append(o)
    if force or _newer(lib_out, objs):
        _run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib_out] + objs + ["-ldl"])
    return lib_out


def build_host(force=False):
    srcs = [os.path.join(HOST, f) for f in HOST_SRC]
    deps = srcs + [os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".hpp")]
    deps += [os.path.join(ROOT, "include", "rt_hip.h"), os.path.join(ROOT, "include", "rt_host.h")]
    if force or _newer(LIB_HOST, deps):
        _run(["g++"] + HOST_FLAGS + ["-o", LIB_HOST] + srcs)
    return LIB_HOST


LAUNCHER = os.path.join(PKG, "rt_render")


def build_launcher(force=False):
    """The native launcher (host/rt_render_main.cpp): C++ against include/rt_host.h + include/rt_hip.h only, linked to the
    two libraries beside it (rpath $ORIGIN)."""
    src = os.path.join(HOST, "rt_render_main.cpp")
    deps = [src, LIB_HOST, LIB_HIP, os.path.join(ROOT, "include", "rt_hip.h"), os.path.join(ROOT, "include", "rt_host.h")]
    if force or _newer(LAUNCHER, deps):
        _run(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-pthread", "-o", LAUNCHER, src, "-L" + PKG, "-lrt_host", "-lrt_hip",
              "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link," + PKG + ":/opt/rocm/lib"])
    return LAUNCHER


def build_all(force=False):
    return build_hip(force), build_host(force), build_launcher(force)


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv))
