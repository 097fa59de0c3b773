"""tools/hazard_lint.py over the device assembly of the library as built (no GPU needed: hipcc cross-compiles).

hipcc pads the wait states of its own instructions and nothing inside -- or around the outputs of -- an `asm` statement
(docs/HARDWARE.md rules 31, 36, 40).  The lint re-does the hazard recogniser's job on the final ISA with the asm instructions
included; this test makes a violation a red CPU suite instead of a wrong gradient on some waves of some builds."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINT = os.path.join(ROOT, "tools", "hazard_lint.py")
LIBDIR = os.path.join(ROOT, "vipsy_amd", "_lib")


def _run(*args):
    return subprocess.run([sys.executable, LINT] + list(args), capture_output=True, text=True, timeout=900)


def test_lint_finds_the_planted_hazards_and_passes_compiler_padded_code():
    # builds three probe kernels: an MFMA result read inside asm right behind the MFMA (must be flagged), an asm scratch
    # output that the register allocator is free to put inside an accumulator in flight (the k_hodina_m fault), and a kernel
    # without asm whose hazards the compiler pads itself (must be clean: the table is no stricter than the compiler)
    r = _run("--selftest")
    assert r.returncode == 0, r.stdout + r.stderr
    assert "selftest ok" in r.stdout


@pytest.mark.parametrize("name", ["libvipsy_hip.gfx950.s", "libvipsy_hip_sched2.gfx950.s"])
def test_library_assembly_has_no_asm_hazard(name):
    import __graft_entry__ as g
    g.build()
    path = os.path.join(LIBDIR, name)
    assert os.path.exists(path), "the build keeps the device assembly beside the library (vipsy_amd/csrc/Makefile)"
    r = _run(path, "--quiet")
    last = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr
    assert r.returncode == 0, _run(path).stdout[-6000:]
    assert " 0 asm hazards" in last, last
    # every inline-asm site of the sources is in the assembly that was checked
    n_asm = int(last.split("(")[1].split()[0])
    assert n_asm > 2000, last
