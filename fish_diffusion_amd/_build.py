"""Build libfishdx.so (HIP, gfx950 only) in-tree with hipcc.

    python -m fish_diffusion_amd._build [--force]

The shared library is a build artefact (git-ignored) that travels to the GPU box with the repo
snapshot; there is no JIT and no CPU fallback.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libfishdx.so")
SOURCES = ["core.hip", "wavenet.hip", "nsf.hip", "mel.hip", "features.hip", "refinegan.hip", "convnext.hip", "tfdec.hip"]
# every header under csrc/ (+ the C ABI): a change to any of them rebuilds every object (a hand-kept list went stale once: a header-only fix
# of the fused ResBlock kernel did not rebuild nsf.o)
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".h")) + ["../../include/fishdx.h"]
# -amdgpu-kernarg-preload-count: the first 14 scalar kernel-argument dwords arrive in SGPRs at wave launch (convgemm.hip.h)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
         "-mllvm", "-amdgpu-kernarg-preload-count=14"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build_trace(verbose: bool = True, bisect: int = 0) -> str:
    """Instrumented build (tools/ktrace.py): libfishdx_trace.so with -DFDX_KTRACE, one shot, no object cache.  `bisect` = 1 | 2 | 3:
    libfishdx_trace_b<N>.so with one component of the residual-block kernels removed (-DFDX_BISECT=N, convgemm16s.hip.h: 1 = the gate
    epilogue's exp / rcp replaced by an add, 2 = non-temporal stores replaced by plain ones, 3 = the split-K LDS reduction stubbed) -- timing
    experiments only, the results are wrong by construction."""
    out = os.path.join(CSRC, f"libfishdx_trace_b{bisect}.so" if bisect else "libfishdx_trace.so")
    extra = [f"-DFDX_BISECT={bisect}"] if bisect else []
    cmd = [_hipcc(), *FLAGS, "-DFDX_KTRACE", *extra, "-shared", *[os.path.join(CSRC, s) for s in SOURCES], "-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    return out


def asan_runtime() -> str:
    """The AddressSanitizer runtime of hipcc's clang (to LD_PRELOAD into an uninstrumented python)."""
    clang = os.path.join(os.path.dirname(os.path.realpath(_hipcc())), "..", "lib", "llvm", "bin", "clang")
    if not os.path.exists(clang):
        clang = "/opt/rocm/lib/llvm/bin/clang"
    return subprocess.run([clang, "-print-file-name=libclang_rt.asan-x86_64.so"], check=True, capture_output=True, text=True).stdout.strip()


def build_asan(out_dir: str, verbose: bool = False) -> str:
    """Sanitizer build of the C ABI's HOST side (sanitizers): every translation unit with `-Xarch_host -fsanitize=address,undefined`
    (the device code is compiled as usual and never runs in the CPU tests), linked into <out_dir>/libfishdx_asan.so.  tests/test_sanitizers.py
    runs the host-side entry points (packing of every model family, filterbank, frame counts, descriptors' validation) through it."""
    os.makedirs(out_dir, exist_ok=True)
    hipcc = _hipcc()
    flags = ["--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-function",
             "-Xarch_host", "-fsanitize=address,undefined", "-Xarch_host", "-fno-omit-frame-pointer",
             "-mllvm", "-amdgpu-kernarg-preload-count=14"]

    def compile_one(name):
        obj = os.path.join(out_dir, name[:-4] + ".o")
        cmd = [hipcc, *flags, "-c", os.path.join(CSRC, name), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)
        return obj

    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    lib = os.path.join(out_dir, "libfishdx_asan.so")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-fsanitize=address,undefined", "-shared", "-fPIC", "-o", lib, *objs], check=True, cwd=CSRC)
    return lib


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    objs = []

    def compile_one(src):
        obj = src[:-4] + ".o"
        if force or _stale(obj, [src] + hdrs):
            cmd = [hipcc, *FLAGS, "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True, cwd=CSRC)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, srcs))
    if force or _stale(LIB, objs):
        for name in os.listdir(CSRC):      # instrumented libraries of an older source state must not outlive it (tools/ktrace.py would load them)
            if name.startswith("libfishdx_trace") and name.endswith(".so"):
                os.remove(os.path.join(CSRC, name))
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    if "--trace" in sys.argv:
        b = [int(a.split("=")[1]) for a in sys.argv if a.startswith("--bisect=")]
        print(build_trace(bisect=b[0] if b else 0))
    else:
        print(build(force="--force" in sys.argv))
