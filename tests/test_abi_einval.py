"""Every entry point of the C ABI refuses bad arguments with VX_EINVAL before it reaches the HIP runtime -- checked on the CPU
under AddressSanitizer + UBSan (sanitizers on the host build; GPU-side sanitizers do not exist on the pool).

`make -C vipsy_amd/csrc asan` compiles the HOST side of vx_abi.hip (dispatch, argument checks, workspace arithmetic) with
-fsanitize=address,undefined around the device code of the library as built; tests/helpers/abi_fuzz.py then calls all 58
compute and size entry points with null pointers, host pointers, zero / negative / 2^40 sizes and shapes beyond every limit.
A read through a bad pointer, an out-of-bounds table access or a signed overflow in a size computation aborts the child."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vipsy_amd", "csrc")
ASAN_LIB = os.path.join(ROOT, "vipsy_amd", "_lib", "libvipsy_hip_asan.so")


def _asan_runtime():
    r = subprocess.run(["/opt/rocm/lib/llvm/bin/clang", "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True)
    path = r.stdout.strip()
    assert os.path.isabs(path) and os.path.exists(path), path
    return path


def test_every_entry_point_refuses_bad_arguments_under_asan_ubsan():
    import __graft_entry__ as g
    g.build()
    subprocess.check_call(["make", "-C", CSRC, "asan"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    env = dict(os.environ, LD_PRELOAD=_asan_runtime(), ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", VX_LIB=ASAN_LIB, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "abi_fuzz.py")], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    tail = r.stdout[-3000:] + r.stderr[-3000:]
    assert r.returncode == 0, tail
    assert "abi_fuzz: 0 calls that were not refused" in r.stdout, tail
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, tail
    assert r.stdout.count(" -> ") > 200, tail                     # every entry point, several argument patterns each


def test_binding_and_header_agree_on_the_abi_version():
    import re
    from vipsy_amd import _hip
    hdr = open(os.path.join(ROOT, "include", "vipsy_amd.h")).read()
    assert int(re.search(r"#define\s+VX_ABI_VERSION\s+(\d+)", hdr).group(1)) == _hip.ABI_VERSION
