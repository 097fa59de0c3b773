"""The oracle comparisons again, against the SAME sources compiled under another instruction schedule and register allocation.

docs/HARDWARE.md rule 40: `k_hodina_m` carried a data hazard for three rounds that showed in one schedule and hid in the next
(an inline-asm scratch register inside the destination of an MFMA still in flight).  A green test certifies a build, not the
source -- so the comparisons that pin the kernels with hand-written asm run against two builds: the shipped library and
`libvipsy_hip_sched2.so` (`make -C vipsy_amd/csrc sched2`: the AMDGPU machine scheduler's "max-ilp" strategy), selected through
`VX_LIB` in a child process.  tools/hazard_lint.py checks both builds' assembly on the CPU (tests/test_hazard_lint.py).
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCHED2 = os.path.join(ROOT, "vipsy_amd", "_lib", "libvipsy_hip_sched2.so")

pytestmark = pytest.mark.gpu

# the kernels with inline asm or hand-placed waits on the judged paths: HO-DINA (cfg5), the headline's guide and likelihood
# kernels at the judged shape, the 3PL / 4PL likelihood, the D = 1 kernels, the small-batch forms
SELECT = ("hodina_step_vs_oracle or headline_large_batch_kernels_vs_oracle or mvn_amortized_step_vs_oracle or "
          "irt1d_step_vs_oracle or irt1d_amortized_step_vs_oracle or mvn_bbvi_step_vs_oracle or ccdm_step_vs_oracle")


def test_oracle_comparisons_under_a_second_schedule():
    if not os.path.exists(SCHED2):                      # (the snapshot normally carries it; hipcc is on the GPU box too: ~3 minutes)
        subprocess.run(["make", "-C", os.path.join(ROOT, "vipsy_amd", "csrc"), "sched2"], capture_output=True, timeout=1500)
    assert os.path.exists(SCHED2), "build it: make -C vipsy_amd/csrc sched2 (or __graft_entry__.build())"
    env = dict(os.environ)
    env["VX_LIB"] = SCHED2
    cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-x", "-q", "-m", "gpu",
           "-k", SELECT, "-p", "no:cacheprovider"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout or "")[-3000:] + (r.stderr or "")[-2000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout.splitlines()[-1], tail
    # the child really ran on the second library
    probe = subprocess.run([sys.executable, "-c", "from vipsy_amd import _hip; print(_hip.LIB_PATH); _hip.lib()"], env=env,
                           cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert probe.returncode == 0 and probe.stdout.strip().endswith("libvipsy_hip_sched2.so"), probe.stdout + probe.stderr
