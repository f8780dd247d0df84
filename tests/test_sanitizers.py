"""CPU: the C ABI's host side under AddressSanitizer + UndefinedBehaviorSanitizer (sanitizers).

`fish_diffusion_amd._build.build_asan` compiles every translation unit with `-Xarch_host -fsanitize=address,undefined` (device code as usual: it
never runs here) into a scratch directory under the system's temp dir (nothing lands in the tree); the host-side tests -- fragment-order packing of
every model family against its numpy emulation, descriptor validation, the mel filterbank and frame counts, the exported-symbol check -- then run
in a subprocess against that library with the sanitizer runtime preloaded.  A report (heap overflow in a pack loop, signed overflow in a layout
computation ...) aborts the subprocess: `halt_on_error=1`."""
import hashlib
import os
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _source_digest():
    h = hashlib.sha1()
    csrc = os.path.join(ROOT, "fish_diffusion_amd", "csrc")
    for name in sorted(os.listdir(csrc)) + ["../../include/fishdx.h"]:
        if name.endswith((".hip", ".h")):
            with open(os.path.join(csrc, name), "rb") as f:
                h.update(name.encode() + f.read())
    return h.hexdigest()[:16]


def test_host_side_of_the_c_abi_is_clean_under_asan_and_ubsan():
    from fish_diffusion_amd import _build
    try:
        runtime = _build.asan_runtime()
    except Exception as e:  # noqa: BLE001
        pytest.skip(f"no sanitizer runtime next to hipcc: {e}")
    if not os.path.exists(runtime):
        pytest.skip(f"sanitizer runtime not found: {runtime}")
    out_dir = os.path.join(tempfile.gettempdir(), f"fishdx_asan_{_source_digest()}")
    lib = os.path.join(out_dir, "libfishdx_asan.so")
    if not os.path.exists(lib):
        _build.build_asan(out_dir)
    env = dict(os.environ, LD_PRELOAD=runtime, FDX_LIB_PATH=lib,
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_host.py", "tests/test_round5_host.py", "-x", "-q", "-p", "no:cacheprovider",
                        "-k", "not load_checkpoint and not resblock2"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    out = r.stdout + r.stderr
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-3000:]
    assert r.returncode == 0 and " passed" in out, out[-3000:]
