#!/usr/bin/env python3
"""ELBO-gradient steps/sec on the BASELINE.json headline workload:
2PL IRT, 1M persons x 500 items x 100 latent dims, amortized MvnEncoder guide (hidden 64),
full batch, one particle (reference: IrtMultiDimTestCase.test_ai_100_dim_2pl, test.py:336-361, scaled
to 1M persons).  One "step" = one svi.step: guide forward, likelihood + gradients, guide backward,
[all-reduce of the flat gradient buffer], Adam.  Persons are sharded over the ranks (strong scaling:
the 1M-person problem is fixed, every rank owns N / world persons).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python bench.py --gpus N --steps K --warmup W            # N > 1, no launcher around it: starts its N ranks itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W               # under a launcher: one rank per GPU, RCCL

Prints ONE JSON line on rank 0 (see the field notes in DESIGN.md section "Measurement").
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: dense fp32 matrix peak
PEAK_BF16_MFMA_TFLOPS = 2500.0    # same guide: dense bf16 / fp16 matrix peak (the fp16 forms take the same cycles)
# an fp32 product computed from two fp16 terms per operand costs three fp16 MFMA products:
PEAK_F16X2_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 3.0
PEAK_HBM_GBS = 8000.0
# What the 16-bit matrix pipe SUSTAINS on this board: a loop of nothing but dense v_mfma_f32_32x32x16_f16 on random operands, every
# CU busy, runs at the 1 400 W package power cap with the graphics clock at 1.52 GHz instead of 2.4 -- 1 590 TFLOP/s, 0.633 of
# the guide's dense peak (tools/power_ubench.hip, profiles/r06_power_ubench.txt; the judged step itself draws 1 375-1 399 W at
# 1.85-1.87 GHz: profiles/r06_power_headline.txt; docs/HARDWARE.md rule 41).  `roofline.peak` stays the guide's figure;
# `roofline.power_capped` prices the same kernel against this one.
SUSTAINED_MFMA16_FRACTION = 1590.0 / 2517.0


def _newest_profile_tag():
    """the newest committed PMC summary of the headline step: profiles/rNN[x]_headline_hbm_traffic.json -> "rNN[x]_headline"
    (a kernel change without a re-profile would otherwise keep quoting old bytes under a new round's name)"""
    import glob
    import re
    best = None
    for path in glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_headline_hbm_traffic.json")):
        m = re.match(r"r(\d+)([a-z]?)_headline_hbm_traffic\.json$", os.path.basename(path))
        if m:
            key = (int(m.group(1)), m.group(2))
            if best is None or key > best[0]:
                best = (key, os.path.basename(path)[:-len("_hbm_traffic.json")])
    return best[1] if best else "r05_headline"


PROFILE_TAG = _newest_profile_tag()
PAIR = "k_mvn_enc_bwd_h_b2 | k_mvn_enc_bwd_w_b side by side"      # the bracket name vx_mvn_enc_backward files the pair under

WORKLOADS = {
    # name: (model, N, J, D, H, amortized, missing)
    "irt2pl_100d_amortized_1Mx500": ("irt_2pl", 1000000, 500, 100, 64, True, 0.0),
    "irt4pl_1d_bbvi_100kx100": ("irt_4pl", 100000, 100, 1, 0, False, 0.0),
    "irt2pl_1d_bbvi_missing90_1Mx500": ("irt_2pl", 1000000, 500, 1, 0, False, 0.9),
    "irt2pl_1d_bbvi_dense_1Mx500": ("irt_2pl", 1000000, 500, 1, 0, False, 0.0),
    "irt2pl_1d_amortized_missing90_1Mx500": ("irt_2pl", 1000000, 500, 1, 64, True, 0.9),
    "hodina_1Mx30x8": ("hodina", 1000000, 30, 8, 0, False, 0.0),
}


def enc_fwd_flops_per_person(J, D, H):
    """Algorithmic flops of the guide-forward kernel per person:
    fc1 J*H + fc21 H*D + fc22 H*T + L.eps T multiply-adds, 2 flops each."""
    T = D * (D + 1) // 2
    return 2.0 * (J * H + H * D + H * T + T)


def kernel_model(name, J, D, H):
    """Algorithmic flops per person (S = 1) and the MFMA peak that bounds each large kernel of the amortized
    multidimensional step (2 flops per multiply-add)."""
    T = D * (D + 1) // 2
    heads = 2.0 * (H * D + H * T)
    f16x2 = "f32 via two fp16 terms per operand (f16x2, 2^-22), three products on the fp16 MFMA, fp32 accumulate"
    table = {
        "k_mvn_enc_fwd_b": (enc_fwd_flops_per_person(J, D, H), PEAK_F16X2_TFLOPS, f16x2),
        "k_mvn_enc_fwd_b2": (enc_fwd_flops_per_person(J, D, H), PEAK_F16X2_TFLOPS, f16x2),
        "k_mvn_enc_fwd_p": (enc_fwd_flops_per_person(J, D, H), PEAK_F32_MFMA_TFLOPS, "f32 MFMA"),
        "k_mvn_enc_bwd_h_t": (heads, PEAK_F32_MFMA_TFLOPS, "f32 MFMA"),
        "k_mvn_enc_bwd_h_b": (heads, PEAK_F16X2_TFLOPS, f16x2),
        "k_mvn_enc_bwd_w_t": (heads, PEAK_F32_MFMA_TFLOPS, "f32 MFMA"),
        "k_mvn_enc_bwd_w_b": (heads, PEAK_F16X2_TFLOPS, f16x2),
        "k_irt_lik_r": (2.0 * 3 * (D + 1) * J, PEAK_F32_MFMA_TFLOPS, "f32 MFMA"),        # Z, gx, GA
        "k_irt_lik_b": (2.0 * 3 * (D + 1) * J, PEAK_BF16_MFMA_TFLOPS * 3.0 / 16.0,
                        "f32 via bf16 terms on the bf16 MFMA (Z: six products, gx and GA: five)"),
        "k_irt_lik_h": (2.0 * 3 * (D + 1) * J, PEAK_F16X2_TFLOPS, f16x2),               # Z, gx, GA
        # the hidden gradient and the head weight gradient run SIDE BY SIDE on two streams: one bracket on the launch stream
        # from in front of the fork to behind the join, priced with the flops of both
        PAIR: (2.0 * heads, PEAK_F16X2_TFLOPS, f16x2),
        "k_fc1_bwd_c": (2.0 * J * H, PEAK_BF16_MFMA_TFLOPS / 2.0,
                        "response bytes exact in fp16 x two fp16 terms of ghpre: two products on the fp16 MFMA, fp32 accumulate"),
        # --estimator score: u = L^-T eps with the rows of L from the heads (the strictly lower triangle: T - D rows of H
        # multiply-adds) and the back-substitution itself (T - D multiply-adds)
        "k_mvn_score_b": (2.0 * (T - D) * (H + 1), PEAK_F16X2_TFLOPS, f16x2),
    }
    if D == 1:
        # the 1-D amortized guide (BASELINE config 4, amortized variant; SURVEY.md section 8d: "MFMA/FMA (encoder)"): fc1 and
        # its weight gradient, J H multiply-adds per person each, the response bytes exact in bf16 and the other operand in
        # three bf16 terms = three products per f32 product
        bf16x3 = "response bytes exact in bf16 x three bf16 terms of the f32 operand: three products on the bf16 MFMA, fp32 accumulate"
        table = {"k_norm_enc_fwd_b": (2.0 * J * H, PEAK_BF16_MFMA_TFLOPS / 3.0, bf16x3),
                 # round 5 (batches from 4 096 persons on): W1 as two fp16 terms against the exact response bytes
                 "k_norm_enc_fwd_h": (2.0 * J * H, PEAK_BF16_MFMA_TFLOPS / 2.0,
                                      "response bytes exact in fp16 x two fp16 terms of W1: two products on the fp16 MFMA, fp32 accumulate"),
                 "k_fc1_bwd_c": (2.0 * J * H, PEAK_BF16_MFMA_TFLOPS / 3.0, bf16x3)}
    return table.get(name)


def measured_traffic(kernel_prefix):
    """HBM bytes per launch of the dominant kernel on the headline workload, from the committed PMC summary
    (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 correction applied; see
    profiles/<tag>_hbm_traffic.json and tools/profile_round.sh).  None when the summary is not there."""
    path = os.path.join(ROOT, "profiles", PROFILE_TAG + "_hbm_traffic.json")
    try:
        with open(path) as f:
            doc = json.load(f)
            kernels = doc["kernels"]
    except (OSError, ValueError, KeyError):
        return None
    measured_traffic.git_head = doc.get("git_head")           # the sources the profile was taken from (tools/profile_round.sh)
    want = [w.strip().split(" ")[0] for w in kernel_prefix.split("|")]      # a side-by-side bracket: the sum of its kernels
    tot, found = 0.0, 0
    for name, v in kernels.items():
        if name.split("<")[0].split("(")[0] in want:                   # template arguments are part of the traced name
            tot += float(v["hbm_read_bytes_corrected"] + v["hbm_write_bytes"])
            found += 1
    return tot if found == len(want) else None


def physical_bytes(workload):
    """HBM bytes one launch of the D = 1 step kernel physically moves (PMC, the committed summaries of this round:
    profiles/r04_cfg2_hbm_traffic.json, profiles/r04_cfg4_hbm_traffic.json), or None."""
    tag = {"irt2pl_1d_bbvi_missing90_1Mx500": "r04_cfg4", "irt4pl_1d_bbvi_100kx100": "r05_cfg2"}.get(workload)
    if tag is None:
        return None
    try:
        with open(os.path.join(ROOT, "profiles", tag + "_hbm_traffic.json")) as f:
            kernels = json.load(f)["kernels"]
    except (OSError, ValueError, KeyError):
        return None
    best = None
    for name, v in kernels.items():
        if name.startswith("k_irt1d"):
            tot = float(v["hbm_read_bytes_corrected"] + v["hbm_write_bytes"])
            best = tot if best is None or tot > best else best
    return best


def _usable_cpus():
    """The cores this process may really use: the cgroup CPU quota when there is one (a GPU box hands out a share of a
    256-thread host; a torch pool of 256 threads on a 16-core share runs 70x slower than one thread), else the affinity
    mask, never more than 32."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:                                  # pragma: no cover
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            quota = int(parts[0]) if parts[0] != "max" else -1
            period = int(parts[1]) if len(parts) > 1 else 100000
            if path.endswith("cfs_quota_us"):
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    period = int(f.read())
            if quota > 0:
                n = min(n, max(1, -(-quota // period)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, min(n, 32))


def board_probe(run_steps, ms_per_step, seconds=1.6, settle=0.5):
    """Board power, graphics clock and energy WHILE further steps of the same workload run (after the timed region, never
    inside it): amdsmi's socket power and gfx clock every 20 ms from a second thread, and the board's energy accumulator
    around the part behind the first `settle` seconds.  None where amdsmi is not there or refuses (docs/HARDWARE.md rule 41:
    the judged step runs at the package power cap; this puts the evidence into the line itself)."""
    try:
        import threading
        import amdsmi
        amdsmi.amdsmi_init()
        handles = amdsmi.amdsmi_get_processor_handles()
        h = handles[0] if len(handles) == 1 else None
        if h is None:                                        # several boards in sight: the one whose PCI address is this device's
            pr = torch.cuda.get_device_properties(torch.cuda.current_device())
            want = "%04x:%02x:%02x" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            for cand in handles:
                if str(amdsmi.amdsmi_get_gpu_device_bdf(cand)).lower().startswith(want):
                    h = cand
        if h is None:                                        # no guessing which handle this process runs on
            amdsmi.amdsmi_shut_down()
            return None
        cap_w = float(amdsmi.amdsmi_get_power_cap_info(h)["power_cap"]) / 1e6
        samples, stop = [], threading.Event()

        def poll():
            while not stop.is_set():
                try:
                    pw = amdsmi.amdsmi_get_power_info(h)["current_socket_power"]
                    ck = amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX)["clk"]
                    samples.append((time.perf_counter(), float(pw), float(ck)))
                except Exception:
                    pass
                time.sleep(0.02)

        def energy_j():
            e = amdsmi.amdsmi_get_energy_count(h)
            return float(e["energy_accumulator"]) * float(e["counter_resolution"]) * 1e-6

        th = threading.Thread(target=poll, daemon=True)
        n_settle = max(1, int(settle * 1e3 / ms_per_step))
        n_run = max(4, int((seconds - settle) * 1e3 / ms_per_step))
        th.start()
        run_steps(n_settle)
        t0, e0 = time.perf_counter(), energy_j()
        run_steps(n_run)
        t1, e1 = time.perf_counter(), energy_j()
        stop.set()
        th.join(timeout=1.0)
        amdsmi.amdsmi_shut_down()
        inside = [(pw, ck) for (t, pw, ck) in samples if t0 <= t <= t1]
        if not inside:
            return None
        pws, cks = sorted(v[0] for v in inside), sorted(v[1] for v in inside)
        return {"power_w": pws[len(pws) // 2], "power_w_min": pws[0], "power_w_max": pws[-1], "power_cap_w": cap_w,
                "gfx_clock_mhz": cks[len(cks) // 2], "gfx_clock_peak_mhz": 2400.0, "samples": len(inside),
                "steps": n_run, "ms_per_step": 1e3 * (t1 - t0) / n_run,
                "joule_per_step": (e1 - e0) / n_run if e1 > e0 else None,
                "source": "amdsmi (socket power, gfx clock every 20 ms; energy accumulator) beside %d further steps of this "
                          "workload AFTER the timed region, the first %.1f s of them left out" % (n_run, settle)}
    except Exception:
        return None


def cpu_baseline(J, D, H, n_sample, min_seconds=6.0):
    """The CPU restatement of the reference step (oracle/vi_oracle.py, numpy float32) on a bounded sample of the same
    workload: full step (loss, all gradients, Adam).  Three figures: all host threads numpy's
    BLAS takes, ONE thread, and the reference's native minibatch B = 100 (test.py:338) on all threads.  Reported,
    never the thing measured by `value`."""
    from oracle import vi_oracle as vo
    try:
        from threadpoolctl import threadpool_limits, threadpool_info
    except ImportError:                                     # pragma: no cover
        threadpool_limits, threadpool_info = None, None
    rng = np.random.RandomState(0)
    enc = {"fc1.weight": rng.randn(H, J) / np.sqrt(J), "fc1.bias": np.zeros(H),
           "fc21.weight": rng.randn(D, H) / 8, "fc21.bias": np.zeros(D),
           "fc22.weight": 0.1 * rng.randn(D * (D + 1) // 2, H) / 8, "fc22.bias": np.zeros(D * (D + 1) // 2)}

    def timed(n, seconds):
        y = rng.randint(0, 2, size=(n, J)).astype(np.uint8)
        spec = {"family": "irt", "model": "irt_2pl", "D": D, "Dc": 1.0, "N": n, "amortized": True,
                "share_cov": False, "a_free": vo.default_a_free(D, J)}
        params = vo.init_irt_params(spec, J, np.float32, encoder=enc)
        adam = vo.Adam(1e-3)
        idx = np.arange(n)
        reps, t0 = 0, time.perf_counter()
        while True:
            eps = rng.randn(n, D).astype(np.float32)
            _, g = vo.loss_and_grads(spec, params, y, [idx], [eps])
            adam.step(params, g)
            reps += 1
            el = time.perf_counter() - t0
            if el >= seconds:
                return el / reps

    threads = os.cpu_count()
    if threadpool_info is not None:
        pools = [p.get("num_threads", 1) for p in threadpool_info()]
        threads = max(pools) if pools else 1
    # a plain PyTorch float32 restatement (autograd + torch.optim.Adam, the (B, D, D) scale matrix
    # materialised as the reference does) under torch.set_num_threads, on one thread and on all
    from oracle import torch_step
    n_torch = min(n_sample, 1000)
    n_cpu = _usable_cpus()
    torch_sec = {1: torch_step.time_step(J, D, H, n_torch, 1, min_seconds / 2),
                 n_cpu: torch_step.time_step(J, D, H, n_torch, n_cpu, min_seconds / 2)}
    torch_b100 = torch_step.time_step(J, D, H, 100, n_cpu, min_seconds / 3)
    if threadpool_limits is not None:                      # numpy's BLAS pool sized to the usable cores as well
        with threadpool_limits(limits=n_cpu):
            sec_all = timed(n_sample, min_seconds)
            sec_b100 = timed(100, min_seconds / 3)
        threads = min(threads, n_cpu)
    else:                                                   # pragma: no cover
        sec_all = timed(n_sample, min_seconds)
        sec_b100 = timed(100, min_seconds / 3)
    if threadpool_limits is not None:
        with threadpool_limits(limits=1):
            sec_one = timed(n_sample, min_seconds)
    else:                                                   # pragma: no cover
        sec_one = None
    return {"sec_all": sec_all, "sec_one": sec_one, "sec_b100": sec_b100, "threads": threads,
            "torch_sec": torch_sec, "torch_n": n_torch, "torch_b100": torch_b100}


def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks as CHILDREN (one process per
    GPU, torch.distributed.run on 127.0.0.1) before this process touches the GPU, relay rank 0's JSON line and exit
    with the children's code.  Never exec: a process that has initialised the GPU must not be replaced."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line)
    return p.returncode if p.returncode != 0 else (0 if line is not None else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default: 50 for the headline, 200 for the secondary workloads")
    ap.add_argument("--warmup", type=int, default=None, help="default: 5 for the headline, 50 for the secondary workloads")
    ap.add_argument("--workload", default="irt2pl_100d_amortized_1Mx500", choices=sorted(WORKLOADS))
    ap.add_argument("--persons", type=int, default=None, help="override N (debug only; makes the line non-headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", default="auto", choices=["auto", "0", "1"],
                    help="timed region replays the step's HIP graph (1) or launches kernel by kernel with HIP events around the "
                         "large kernels (0); auto = 0 for the judged headline run on one GPU (its roofline must be measured "
                         "inside the timed region), 1 wherever the engine can replay (shards, secondary workloads)")
    ap.add_argument("--estimator", default="pathwise", choices=["pathwise", "score"],
                    help="pathwise = what the reference's Normal / MultivariateNormal guides get from pyro's Trace_ELBO (the judged "
                         "line); score = the REINFORCE score-function gradient with a control-variate baseline that BASELINE.json's "
                         "north_star names (SURVEY.md F5, App. A.5): same forward, likelihood and item gradients, the guide's "
                         "gradient through d log q.  A line of its own, never the BASELINE metric")
    ap.add_argument("--baseline", default="avg", choices=["none", "avg"],
                    help="control variate of --estimator score: per-person decaying average (pyro's use_decaying_avg_baseline) or none")
    ap.add_argument("--no-board-probe", action="store_true",
                    help="skip the ~1.6 s of further steps behind the timed region during which board power / clock are sampled")
    ap.add_argument("--event-every", type=int, default=5,
                    help="kernel-by-kernel timed region: HIP events on every n-th step of it (1 = on all)")
    ap.add_argument("--dist-backend", default=os.environ.get("VX_DIST_BACKEND", "nccl"), choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the product path); gloo only to rehearse N ranks on fewer GPUs -- the line "
                         "such a run prints is marked a rehearsal (metric, config.collective) and is not an N-GPU result")
    args = ap.parse_args()
    # a secondary workload's step is 50 us - 1 ms: 5 + 2 of them end before the GPU has left its idle clock (551 MHz; measured:
    # HO-DINA 2 830 steps/s over 5 steps, 3 310 over 200)
    headline = args.workload == "irt2pl_100d_amortized_1Mx500"
    if args.steps is None:
        args.steps = 50 if headline else 200
    if args.warmup is None:
        args.warmup = 5 if headline else 50

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch one rank per GPU)" % (args.gpus, world))
    n_dev = torch.cuda.device_count()                      # counting devices does not initialise the GPU
    if n_dev < 1:
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if args.dist_backend == "nccl" and world > n_dev:
        raise SystemExit("bench.py: %d ranks need %d GPUs, %d visible (RCCL wants one device per rank)" % (world, world, n_dev))
    torch.cuda.set_device(local_rank % n_dev)
    dev = torch.device("cuda", local_rank % n_dev)
    group = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.dist_backend == "nccl":
            torch.distributed.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
        else:
            torch.distributed.init_process_group(backend="gloo", rank=rank, world_size=world)
        group = torch.distributed.group.WORLD              # the ranks SHARE the problem: persons are sharded over them
        # what the line is allowed to claim: an N-GPU figure needs N ranks on RCCL, one device each
        got_backend = torch.distributed.get_backend(group)
        got_world = torch.distributed.get_world_size(group)
        if got_world != world:
            raise SystemExit("bench.py: the process group has %d ranks, the launcher announced %d" % (got_world, world))
        if args.dist_backend == "nccl" and got_backend != "nccl":
            raise SystemExit("bench.py: asked for RCCL, the process group runs on %r -- no N-GPU line from this run" % got_backend)

    from vipsy_amd import synth
    from vipsy_amd.engine import IrtEngine, LrSpec

    model, N, J, D, H, amortized, missing = WORKLOADS[args.workload]
    if args.persons:
        N = args.persons
    # contiguous person shards
    per = (N + world - 1) // world
    gid0 = rank * per
    n_local = max(0, min(N, gid0 + per) - gid0)

    if model == "hodina":
        prm = synth.hodina_params(J, D, seed=20245)
        y = synth.simulate_hodina(n_local, gid0, prm, dev, seed=20240, missing=missing)
    elif D > 1:
        a, b = synth.mirt_item_params(J, D, seed=20243)
        items = {"a": a, "b": b}
        y = synth.simulate_responses(n_local, gid0, items, model, dev, seed=20240, missing=missing)
    else:
        items = synth.irt_item_params(J, model, seed=20242)
        y = synth.simulate_responses(n_local, gid0, items, model, dev, seed=20240, missing=missing)

    def lr_fn(module_name, param_name):                     # test.py:345-350
        return {"lr": 1e-2 if param_name in ("a", "b") else 1e-3}
    lrs = LrSpec(lr_fn, milestones=(), gamma=0.1)
    if model == "hodina":
        from vipsy_amd.engine import HoDinaEngine
        eng = HoDinaEngine(y, prm["q"], n_global=N, gid0=gid0, amortized=amortized, H=H, seed=1234, group=group)
    else:
        est_kw = {"estimator": "score", "baseline": args.baseline} if args.estimator == "score" else {}
        eng = IrtEngine(y, model=model, D=D, n_global=N, gid0=gid0, amortized=amortized, H=H, seed=1234, group=group, **est_kw)

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.step(lrs)
        lrs.scheduler_step()
    # per-phase HIP events on the launch stream (torch's current stream == the stream handed to the C ABI)
    ev = []
    from vipsy_amd import _hip
    # D = 1 per-person guides replay the whole step from one HIP graph (engine.py::_step_graph): no events inside the
    # timed region there; the per-phase figures come from an eager pass afterwards
    judged = headline and world == 1 and not args.persons and args.estimator == "pathwise"
    graphed = eng._graphable() and (args.graph == "1" or (args.graph == "auto" and not judged))
    if not graphed:
        eng.events = ev
        _hip.lib().vx_prof_enable(1)                        # HIP events on the launch stream around the large kernels
    else:
        eng.capture_steps(lrs)                              # (records the four-step graph; runs nothing)
    sync()
    t0 = time.perf_counter()
    loss_first = None
    if graphed:
        # as a fit loop runs them (vipsy_amd/vi.py::_loop -> IrtEngine.steps): graph_steps steps a replay where the form allows
        i = 0
        while i < args.steps:
            n = min(32, args.steps - i)
            losses = eng.steps(lrs, [None] * n, scheduler=True)
            if i == 0:                                      # a slot of the engine's loss ring: kept as it is while the timed
                loss_first = losses[0] if args.steps < 60 else losses[0].clone()    # region is shorter than the ring, copied otherwise
            loss = losses[-1]
            i += n
    else:
        # HIP events inside the timed region, on a SAMPLE of its steps (every `every`-th, from the first): a pair of events
        # around each large kernel and each phase is ~25 records a step, and each costs the stream a few microseconds -- 0.11 ms of
        # a 9.0 ms headline step when every step carries them (tools/step_events_cost.py); the sampled steps are ordinary steps
        # of the timed region, `roofline.avg_launch_ms` is the mean over their launches (`roofline.launches_sampled`)
        every = max(1, int(args.event_every))
        for i in range(args.steps):
            sampled = (i % every) == 0
            if every > 1:
                eng.events = ev if sampled else None
                _hip.lib().vx_prof_enable(2 if sampled else 3)
            loss = eng.step(lrs)
            if i == 0:
                loss_first = loss if args.steps < 60 else loss.clone()
            lrs.scheduler_step()
        if every > 1:
            eng.events = ev
            _hip.lib().vx_prof_enable(2)
    sync()
    dt = time.perf_counter() - t0
    loss_v = float(loss.item())
    loss_first_v = float(loss_first.item())
    if graphed:
        # the timed region replayed the graph: per-phase and per-kernel durations from an eager pass AFTER it
        eng.events = ev
        _hip.lib().vx_prof_enable(1)
        for _ in range(min(args.steps, 20)):
            eng.step(lrs)
            lrs.scheduler_step()
        sync()
    eng.events = None
    kernel_ms, kernel_units, kernel_launches = {}, {}, {}
    import ctypes
    for slot in range(_hip.lib().vx_prof_count()):
        nm, ms, cnt = ctypes.create_string_buffer(64), ctypes.c_float(0), ctypes.c_int(0)
        if _hip.lib().vx_prof_read(slot, nm, 64, ctypes.byref(ms), ctypes.byref(cnt)) == 0 and cnt.value:
            kernel_ms[nm.value.decode()] = float(ms.value)
            kernel_launches[nm.value.decode()] = int(cnt.value)
            un = ctypes.c_int64(0)                             # persons of a launch that takes only part of the batch
            if _hip.lib().vx_prof_units(slot, ctypes.byref(un)) == 0 and un.value:
                kernel_units[nm.value.decode()] = int(un.value)
    _hip.lib().vx_prof_enable(0)
    graph_ms = None
    if not graphed and world == 1 and eng._graphable():
        # the line above was timed kernel by kernel (HIP events inside the timed region); for comparison the same step replayed
        # from its HIP graph, after the timed region (the first call of the form runs eagerly, the second captures)
        ng = min(args.steps, 20)
        eng.steps(lrs, [None] * 8)                          # (eager, the captures, a replay)
        sync()
        tg = time.perf_counter()
        eng.steps(lrs, [None] * ng)
        sync()
        graph_ms = 1e3 * (time.perf_counter() - tg) / ng
    board = None
    if world == 1 and not args.no_board_probe:
        def run_more(n):
            for _ in range(n):
                eng.step(lrs)
                lrs.scheduler_step()
            sync()
        board = board_probe(run_more, 1e3 * dt / args.steps)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax.item())

    phases = {}
    for name, e0, e1 in ev:
        phases.setdefault(name, []).append(e0.elapsed_time(e1))
    phase_ms = {k: float(np.mean(v)) for k, v in phases.items()}

    if rank == 0:
        out = {
            "metric": "ELBO-grad steps/sec, 1M persons x 500 items x 100-dim 2PL IRT" if args.workload == "irt2pl_100d_amortized_1Mx500"
                      else "ELBO-grad steps/sec (%s; secondary workload, not the BASELINE metric)" % args.workload,
            "value": args.steps / dt, "unit": "steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "persons": N, "items": J, "dims": D, "hidden": H,
                       "arithmetic": ("results f32; every large GEMM (the guide's fc1, heads and their gradients; the likelihood's "
                                      "x.a, dlogp/dz.a, x.dlogp/dz) from two fp16 terms per operand (2^-22 relative, three "
                                      "products on the fp16 MFMA, fp32 accumulate)") if (D > 1 and amortized) else "f32",
                       "guide": "amortized MvnEncoder" if amortized else "BBVI per-person", "batch": "full (B=N)",
                       "particles": 1, "missing_rate": missing, "persons_per_rank": per,
                       "parallelism": "persons sharded x%d, 1 all-reduce/step" % world,
                       "collective": ("RCCL" if args.dist_backend == "nccl" else "gloo (rehearsal)") if world > 1 else None},
            "person_rows_per_s": N * args.steps / dt,
            "loss_first": loss_first_v, "loss_last": loss_v, "phase_ms": phase_ms,
        }
        if args.estimator == "score":
            out["metric"] += " [score-function (REINFORCE) estimator with the '%s' control variate: not the BASELINE metric]" % args.baseline
            out["config"]["estimator"] = "score-function gradient of the guide, baseline %s; item gradients pathwise" % args.baseline
        if world > 1:
            # the ONE exchange of a step: what ran it, over how many ranks, how many bytes, and what it
            # took on rank 0 between the last gradient kernel and the optimiser (HIP events on the launch stream; for a
            # replayed step from the eager pass behind the timed region).  Not overlapped with anything: it is this long.
            n_red = int(eng.n_params + 1)
            out["collective"] = {"backend": torch.distributed.get_backend(group), "ranks": torch.distributed.get_world_size(group),
                                 "devices_visible": n_dev, "op": "all_reduce(SUM, f32)", "floats": n_red, "bytes": 4 * n_red,
                                 "allreduce_ms": phase_ms.get("allreduce"),
                                 "calls_per_step": 1}
            if args.dist_backend != "nccl":
                out["metric"] += " [REHEARSAL over %s on %d visible device(s): not an %d-GPU result]" % (args.dist_backend, n_dev, world)
                out["rehearsal"] = True
        out["config"]["launch"] = ("whole step replayed from HIP graphs (one; two around the eager all-reduce when sharded); "
                                   "phase_ms / kernel_ms / roofline.avg_launch_ms from an eager pass with HIP events AFTER the "
                                   "timed region") if graphed else "kernel by kernel, HIP events around the large kernels inside the timed region"
        if getattr(eng, "graph_fallback", None):             # capture beside the live communicator failed: said in the line
            out["config"]["launch"] = ("kernel by kernel -- HIP graph capture of the sharded step failed on this rank (%s); same "
                                       "kernels, launched one by one" % eng.graph_fallback[:160])
        if graph_ms is not None:
            out["graph_replay_ms_per_step"] = graph_ms         # same step, same engine, replayed from its HIP graph afterwards
        if board is not None:
            out["board"] = board                               # power / clock / energy beside further steps (board_probe)
        if os.environ.get("VX_MFMA16"):                     # non-default kernel selection: say so in the line itself
            out["config"]["kernel_switch"] = "VX_MFMA16=" + os.environ["VX_MFMA16"]
        if kernel_ms:
            out["kernel_ms"] = kernel_ms
        priced = {k: v for k, v in kernel_ms.items() if kernel_model(k, J, D, H)}
        if priced and (D > 1 or amortized):
            # the dominant kernel = the one with the largest mean duration inside the timed steps (HIP events recorded
            # by the library on the launch stream, vx_prof_*); achieved = algorithmic flops per launch / that duration
            name = max(priced, key=priced.get)
            fl_pp, peak, arith = kernel_model(name, J, D, H)
            fl = fl_pp * kernel_units.get(name, n_local)
            ach = fl / (priced[name] * 1e-3) / 1e12
            headline = (args.workload == "irt2pl_100d_amortized_1Mx500" and world == 1 and not args.persons
                        and args.estimator == "pathwise")
            traffic = measured_traffic(name) if headline else None
            out["roofline"] = {"kernel": name, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                               "frac": ach / peak, "traffic": traffic,
                               "traffic_source": ("committed profile profiles/%s_hbm_traffic.json (rocprofv3 --pmc, bytes per "
                                                  "launch of this kernel on this workload; taken at git %s), NOT measured in "
                                                  "this run" % (PROFILE_TAG, getattr(measured_traffic, "git_head", None) or "?"))
                                                 if traffic is not None else None,
                               "arithmetic": arith,
                               "peak_basis": "dense fp16 MFMA peak 2500 / 3 products per f32 product"
                                             if peak == PEAK_F16X2_TFLOPS else
                                             "dense bf16 MFMA peak 2500 / (16 / 3) products per f32 product"
                                             if peak == PEAK_BF16_MFMA_TFLOPS * 3.0 / 16.0 else
                                             "dense bf16 MFMA peak 2500 / 3 products per f32 product"
                                             if peak == PEAK_BF16_MFMA_TFLOPS / 3.0 else
                                             "dense fp16 MFMA peak 2500 / 2 products per f32 x {-1,0,1} product"
                                             if peak == PEAK_BF16_MFMA_TFLOPS / 2.0 else "dense f32 MFMA peak",
                               "algorithmic_flops_per_launch": fl, "avg_launch_ms": priced[name],
                               "persons_per_launch": kernel_units.get(name, n_local),
                               "launches_sampled": kernel_launches.get(name)}
            if peak != PEAK_F32_MFMA_TFLOPS:
                cap = peak * SUSTAINED_MFMA16_FRACTION
                out["roofline"]["power_capped"] = {
                    "peak": cap, "frac": ach / cap,
                    "basis": "the 16-bit matrix pipe's SUSTAINED rate on this board: a loop of dense 32x32x16 fp16 MFMAs on random "
                             "operands and nothing else holds 1 590 TFLOP/s (0.633 of the guide's 2 517: the 1 400 W package cap "
                             "brings the clock to 1.52 GHz; tools/power_ubench.hip, profiles/r06_power_ubench.txt), and this step "
                             "runs at that cap too (1 375-1 399 W, 1.85-1.87 GHz: profiles/r06_power_headline.txt); not measured "
                             "in this run"}
            if graphed:
                out["roofline"]["avg_launch_ms_source"] = ("HIP events around the kernel in an eager pass after the "
                                                           "timed region (the timed region replays a graph)")
        elif "hodina" in phase_ms:
            # SURVEY.md section 8d, cfg 5: (2 K + J) C MACs forward, ~3x with the backward = compute-bound.  The pattern
            # contractions run on the bf16 MFMA with one operand exact (0/1) and the other split into bf16 terms (three
            # forward, two backward): priced at the dense bf16 peak / 3
            fl = 3.0 * 2.0 * (2 * D + J) * (1 << D) * n_local
            ach = fl / (phase_ms["hodina"] * 1e-3) / 1e12
            peak = PEAK_BF16_MFMA_TFLOPS / 3.0
            out["roofline"] = {"kernel": "k_hodina_m", "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                               "frac": ach / peak, "traffic": None, "algorithmic_flops_per_launch": fl,
                               "avg_launch_ms": phase_ms["hodina"],
                               "peak_basis": "dense bf16 MFMA peak 2500 / 3 products per f32 x {0,1} product",
                               "note": "the softmax over the 2^K patterns (VALU + exp) bounds this kernel, not the MFMA"}
        elif "irt1d" in phase_ms:
            key = "irt1d"
            by = (J + 48.0) * n_local                       # y row + 6 fp32 r/w per person
            ach = by / (phase_ms[key] * 1e-3) / 1e9
            out["roofline"] = {"kernel": "k_" + key, "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS,
                               "unit": "GB/s", "frac": ach / PEAK_HBM_GBS, "traffic": None,
                               "algorithmic_bytes_per_launch": by, "avg_launch_ms": phase_ms[key],
                               "achieved_note": "an ALGORITHMIC rate: SURVEY.md section 8d's dense J + 48 bytes per person over the "
                                                "launch time; the bytes a launch physically moves are physical_bytes_per_launch"}
            phys = physical_bytes(args.workload)
            if phys is not None:
                out["roofline"]["physical_bytes_per_launch"] = phys
                out["roofline"]["physical_GBs"] = phys / (phase_ms[key] * 1e-3) / 1e9
            if graphed:
                out["roofline"]["avg_launch_ms_source"] = ("HIP events around the kernel in an eager pass after the "
                                                           "timed region (the timed region replays a graph)")
        if world == 1 and not args.no_cpu_baseline and D > 1:
            # the reference's own usage (subsample_size = 100) on the same engine, outside the timed region
            rg = np.random.Generator(np.random.PCG64(7))
            def draw():                                     # host indices, as the fit loop draws them (vipsy_amd/vi.py::_subsample)
                return torch.from_numpy(rg.choice(n_local, size=100, replace=False).astype(np.int64))
            import gc
            eng.steps(lrs, [draw() for _ in range(12)], b_global=100)      # (eager, the captures, a replay)
            torch.cuda.synchronize()
            gc.collect()                                    # (a full collection is a 40 ms stall of the host, once: not the step's cost)
            tb = time.perf_counter()
            n_b100 = 400
            for _ in range(n_b100 // 4):                    # as the fit loop does: four draws, one replay (IrtEngine.steps)
                eng.steps(lrs, [draw() for _ in range(4)], b_global=100)
            torch.cuda.synchronize()
            gpu_b100 = n_b100 / (time.perf_counter() - tb)
            n_s = 4000
            cb = cpu_baseline(J, D, H, n_s)
            # `value` = the FASTEST of the restatements timed here (numpy port on one / all threads, the plain PyTorch float32
            # step on one / all threads), each scaled linearly from its own sample to the 1M persons of the workload
            cands = [("numpy float32 port (oracle/vi_oracle.py), all BLAS threads", cb["sec_all"] * N / n_s, cb["threads"])]
            if cb["sec_one"] is not None:
                cands.append(("numpy float32 port (oracle/vi_oracle.py), one thread", cb["sec_one"] * N / n_s, 1))
            for thr, sec in cb["torch_sec"].items():
                cands.append(("plain PyTorch float32 step (oracle/torch_step.py: autograd + torch.optim.Adam), %d threads" % thr,
                              sec * N / cb["torch_n"], int(thr)))
            best_name, best_sec_1m, best_cores = min(cands, key=lambda c: c[1])
            best_one = cb["sec_one"] is not None and cb["sec_one"] < cb["sec_all"]     # BLAS oversubscription happens
            sec_best = cb["sec_one"] if best_one else cb["sec_all"]
            out["cpu_baseline"] = {"value": 1.0 / best_sec_1m, "unit": "steps/s",
                                   "cores": best_cores, "kind": "port", "variant": best_name,
                                   "sample": "%d of %d persons, full step (loss + all grads + Adam) with the numpy "
                                             "oracle in float32, time scaled linearly to %d persons; a CPU restatement "
                                             "of vi.py semantics, not vi.py + pyro (not installable)" % (n_s, N, N),
                                   "sample_seconds_per_step": sec_best,
                                   "all_threads": {"value": 1.0 / (cb["sec_all"] * N / n_s), "unit": "steps/s",
                                                   "cores": cb["threads"], "sample_seconds_per_step": cb["sec_all"]},
                                   "single_thread": {"value": None if cb["sec_one"] is None else 1.0 / (cb["sec_one"] * N / n_s),
                                                     "unit": "steps/s", "cores": 1, "sample_seconds_per_step": cb["sec_one"]},
                                   "native_minibatch": {"B": 100, "steps_per_s": 1.0 / cb["sec_b100"],
                                                        "person_rows_per_s": 100.0 / cb["sec_b100"], "cores": cb["threads"],
                                                        "gpu_steps_per_s": gpu_b100,
                                                        "note": "the reference's own B = 100 step (test.py:338): CPU port on "
                                                                "all threads; gpu_steps_per_s = the same step on this GPU, "
                                                                "rows subsampled per step, after the timed region"},
                                   "torch_f32": {"sample": "%d persons, full step (autograd + torch.optim.Adam), float32, the (B, D, D) "
                                                           "scale matrix materialised as vi.py does; scaled linearly" % cb["torch_n"],
                                                 "steps_per_s_by_threads": {str(k): 1.0 / (v * N / cb["torch_n"])
                                                                            for k, v in cb["torch_sec"].items()},
                                                 "native_minibatch_steps_per_s": 1.0 / cb["torch_b100"]},
                                   "host_cpus": os.cpu_count()}
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
