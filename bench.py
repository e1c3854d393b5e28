#!/usr/bin/env python3
"""bench.py -- PEA-Diffusion KD training step on MI355X (BASELINE.json metric).

`python bench.py --gpus N --steps K --warmup W`.  For N>1 either torch.distributed.run starts one rank per GPU
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment), or -- when those are absent -- this script starts the
N ranks itself as fresh child processes BEFORE anything touches the GPU and relays rank 0's JSON line.  A step = one pass of the hot path over one synthetic batch already resident in HBM:
add_noise -> adapter (cond|uncond) -> student SDXL UNet fwd (taps) -> teacher SDXL UNet fwd -> fused KD
loss -> student data-gradient backward -> adapter backward [-> ONE RCCL all-reduce of the flat adapter
gradient when N>1, on the communicator's own HIP stream, joined before AdamW] -> fused AdamW.  Weak scaling: ranks
share nothing but the adapter-gradient all-reduce.  Per-GPU batch: 4 at N=1 (BASELINE configs[1]), 8 at N>1
(configs[2]: global 64 = 8 x 8); `--batch` overrides.
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# kernel arguments in device memory: the default of this ROCm stack (measured: unset = 1; 0 costs 3.3 ms per step over the ~2300
# launches, profiles/r05_ab_dev_kernarg.log) -- stated here so that a box with another default runs the same configuration
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC: RCCL across processes on this host driver (before any HIP call)

# exit codes of a rank (the launcher relays the first non-zero one): 2 usage, 3 the HBM plan does not fit, 4 the native RCCL
# communicator failed under --collective native, 5 its rendezvous timed out, 124 the launcher's own deadline
RC_COMM_FAILED, RC_COMM_TIMEOUT = 4, 5
# keys every N-rank result line carries (asserted on the CPU dry-run line and on the GPU line: tests/test_dp_cpu.py, test_dist_gpu.py)
NRANK_KEYS = ("n_gpus", "rccl_ranks", "collective", "control_plane", "allreduce_ms", "allreduce_exposed_ms", "allreduce_bytes",
              "single_gpu_equivalent", "hbm_plan_gb_per_rank", "cores_per_rank")

import torch  # noqa: E402

TFLOP_PER_IMAGE = {"sdxl": 20.31, "tiny": None}      # 3 x 6.765 + adapter
MFMA_PEAK_TFLOPS = 2500.0                            # MI355X dense bf16 (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0


def synthetic_batch(cfg, B, L, enc_dim, hw, device, seed):
    """post-encoder synthetic batch; generated on the device, resident before timing"""
    g = torch.Generator(device=device).manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g, device=device)
    zh = torch.zeros(B, dtype=torch.int64, device=device)
    zh[: max(1, B // 2)] = 1                                      # half native captions, half translated
    pm = torch.zeros(B, dtype=torch.uint8, device=device)
    pm[B - 1] = 1                                                 # one CFG-dropped sample (p = 0.1 in the reference)
    px = hw * 8
    return dict(latents=r(B, 4, hw, hw), noise=r(B, 4, hw, hw) + 0.5 * r(B, 4, 1, 1),
                timesteps=torch.randint(0, 1000, (B,), generator=g, device=device),
                enc=r(B, L, enc_dim), enc_uncond=r(1, L, enc_dim).repeat(B, 1, 1).contiguous(),
                prompt_mask=pm, zh_or_not=zh, teacher_ehs=r(B, 77, cfg.cross_attention_dim),
                teacher_neg=r(1, 77, cfg.cross_attention_dim).repeat(B, 1, 1).contiguous(),
                teacher_pooled=r(B, cfg.pooled_dim),
                time_ids=torch.tensor([[px, px, 0, 0, px, px]] * B, dtype=torch.float32, device=device))


def pmc_traffic():
    """HBM-side bytes per GEMM launch from the newest committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this
    same command (profiles/rNN_pmc_traffic.json; gfx950 FETCH_SIZE x2 correction applied); (None, None) if absent."""
    import glob
    try:
        f = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[-1]
        d = json.load(open(f))
        return round(d["gemm_family"]["traffic_MB_per_launch"] * 1e6), os.path.basename(f)      # bytes per launch
    except Exception:
        return None, None


def _fast_fill_(module):
    """cheap deterministic weights for the CPU timing model (torch's default init of 2.57 B parameters
    is single-threaded and takes about a minute; the values do not affect the timing)"""
    buf = (torch.rand(1 << 22) - 0.5)
    with torch.no_grad():
        for n, p in module.named_parameters():
            if p.dim() >= 2:
                fan = p[0].numel()
                flat = p.view(-1)
                for o in range(0, flat.numel(), buf.numel()):
                    k = min(buf.numel(), flat.numel() - o)
                    flat[o:o + k].copy_(buf[:k])
                flat.mul_(2.0 * fan ** -0.5)
            elif n.endswith("weight"):
                p.fill_(1.0)
            else:
                p.zero_()


def cpu_baseline_worker(model_name, budget_s, faithful=False):
    """The CPU oracle (oracle/*.py, the restatement of the reference's PyTorch path) timed on this box's
    host cores: one KD step at batch 1 (teacher fwd no_grad + student fwd + adapter-only backward).
    `faithful`: additionally the step with the student UNet left trainable, as the reference leaves it
    (train_sdxl_zh.py:138,166-168: only proj is optimised, but unet.requires_grad stays True, so autograd also
    produces 2.57 B weight gradients nobody reads)."""
    from oracle.step_ref import AdapterRef, synthetic_batch as sb, training_step_ref
    from oracle.unet_ref import UNet2DConditionRef, cast_hook_ref, sdxl_config, tiny_config
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    threads = min(cores, 64)                      # one socket's worth; more threads do not help these sizes
    torch.set_num_threads(threads)
    cfg = sdxl_config() if model_name == "sdxl" else tiny_config()
    t0 = time.time()
    with torch.device("meta"):
        unet = UNet2DConditionRef(cfg)            # teacher == student checkpoint (train_sdxl_zh.py:138,151)
    unet = unet.to_empty(device="cpu")
    _fast_fill_(unet)
    for p in unet.parameters():
        p.requires_grad_(False)
    enc_dim = 1024 if model_name == "sdxl" else 128
    ad = AdapterRef(enc_dim, cfg.pooled_dim, 1024 if model_name == "sdxl" else 192, cfg.cross_attention_dim, False)
    build_s = time.time() - t0
    runs = {}
    for hw in ([64, 128] if model_name == "sdxl" else [16]):
        batch = sb(cfg, 1, L=77, enc_dim=enc_dim, seed=0, latent_hw=hw)
        t1 = time.time()
        out = training_step_ref(ad, unet, unet, batch, cast_hook_ref)
        out["loss"].backward()
        runs[hw] = time.time() - t1
        for p in ad.parameters():
            p.grad = None
        del out
        if hw != 128 and runs[hw] * 5.5 > budget_s:    # the metric's resolution costs ~4.4-5x the 512 px step
            break
    faithful_s = None
    if faithful:
        import copy
        hwf = max(runs)
        stu = copy.deepcopy(unet)                 # the reference holds two copies (unet, unet_teacher)
        for p in stu.parameters():
            p.requires_grad_(True)
        batch = sb(cfg, 1, L=77, enc_dim=enc_dim, seed=0, latent_hw=hwf)
        t1 = time.time()
        out = training_step_ref(ad, stu, unet, batch, cast_hook_ref)
        out["loss"].backward()
        faithful_s = {"px": hwf * 8, "seconds": round(time.time() - t1, 2),
                      "unet_wgrad_populated": all(p.grad is not None for p in stu.parameters())}
        del out, stu
    hw = max(runs)
    dt = runs[hw]
    full = hw == 128 or model_name != "sdxl"
    # value: images/s of the METRIC's workload (1024 px), measured directly -- one whole KD step at batch 1.  Only when
    # the time budget forbids the 1024 px step is the 512 px step reported (value stays null then: no extrapolation).
    res = {"value": round(1.0 / dt, 5) if full else None, "unit": "images/s", "cores": threads,
           "box_cores": os.cpu_count(), "kind": "port",
           "cores_note": "64 of the box's cores are used: the oracle is torch CPU fp32 and its conv / GEMM / attention kernels at batch 1 "
                         "stop scaling beyond one socket's worth of threads (128 and 256 threads measured no faster and noisier); "
                         "`cores` is the thread count actually used, `box_cores` what the box has (BASELINE.md 3 asks for the count to be stated)",
           "sample": f"1 KD step (teacher fwd no_grad + student fwd + adapter-only backward), batch 1, {hw * 8}x{hw * 8} px "
                     f"(latent {hw}x{hw}), fp32 torch CPU oracle, teacher==student weights, {dt:.1f} s wall "
                     f"(model build {build_s:.0f} s not counted)" + ("" if full else "; 1024 px step skipped: over the CPU budget"),
           "value_512px": (round(1.0 / runs[64], 5) if 64 in runs else None),
           "seconds": {f"{k * 8}px": round(v, 2) for k, v in runs.items()}}
    if faithful_s is not None:
        res["reference_faithful"] = dict(faithful_s, value=round(1.0 / faithful_s["seconds"], 5),
                                         note="student UNet trainable as in the reference: adapter + 2.57 B unused UNet wgrads")
    print("CPU_BASELINE_JSON " + json.dumps(res), flush=True)


def cpu_baseline(model_name, budget_s, faithful=False):
    """runs the worker in a child process with a hard time limit, so the bench line is always printed"""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "--model", model_name,
           "--cpu-budget", str(budget_s)] + (["--cpu-reference-faithful"] if faithful else [])
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=budget_s * 1.5 + 240)
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "images/s", "cores": None, "kind": "port", "sample": "timed out"}
    for line in r.stdout.splitlines():
        if line.startswith("CPU_BASELINE_JSON "):
            return json.loads(line[len("CPU_BASELINE_JSON "):])
    return {"value": None, "unit": "images/s", "cores": None, "kind": "port",
            "sample": "worker failed: " + (r.stderr or "")[-300:]}


def _free_port():
    import socket
    so = socket.socket()
    so.bind(("127.0.0.1", 0))
    port = so.getsockname()[1]
    so.close()
    return port


def launch_ranks(n, argv, timeout_s):
    """Start `n` ranks of this script as fresh child processes (one per GPU; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*
    set as torch.distributed.run would), relay rank 0's single JSON line, return non-zero if any rank fails or the
    deadline passes (a rank stuck in a collective never exits by itself).  Every rank's stderr (and the stdout of
    ranks > 0) goes to a file of its own whose tail is printed when something goes wrong.  The parent never
    initialises the GPU and never re-executes itself (train_sdxl_zh.sh:108-114 is the reference's launcher)."""
    import subprocess
    import tempfile
    port = _free_port()
    logdir = tempfile.mkdtemp(prefix="pea_bench_ranks_")
    procs, logs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (RCCL across processes on this host driver)
        err = open(os.path.join(logdir, f"rank{r}.err"), "wb")
        out = subprocess.PIPE if r == 0 else open(os.path.join(logdir, f"rank{r}.out"), "wb")
        logs.append(err.name)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=out, stderr=err))
    import threading
    out = []
    rd = threading.Thread(target=lambda: out.append(procs[0].stdout.read()), daemon=True)
    rd.start()

    def tails(which):
        for r in which:
            try:
                data = open(logs[r], "rb").read()[-1500:].decode(errors="replace")
            except OSError:
                data = ""
            print(f"---- rank {r} stderr tail ({logs[r]}) ----\n{data}", file=sys.stderr)

    rc = 0
    deadline = time.time() + timeout_s
    while any(p.poll() is None for p in procs):
        failed = [i for i, p in enumerate(procs) if p.poll() not in (None, 0)]
        hung = time.time() > deadline
        if failed or hung:                  # a dead rank leaves the others waiting in a collective: stop exactly those
            alive = [i for i, p in enumerate(procs) if p.poll() is None]
            if hung and not failed:
                rc = 124
                print(f"bench.py: deadline of {timeout_s:.0f} s passed; ranks still running: {alive}", file=sys.stderr)
                tails(alive)
            else:
                rc = procs[failed[0]].returncode or 1
                print(f"bench.py: rank(s) {failed} exited with {rc}; stopping ranks {alive}", file=sys.stderr)
                tails(failed)
            for i in alive:
                procs[i].kill()             # exactly the children started above
            break
        time.sleep(0.2)
    reported = rc != 0
    for p in procs:
        p.wait()
        if p.returncode != 0 and rc == 0:
            rc = p.returncode
    if rc != 0 and not reported:            # every rank had exited before the loop above saw a failure
        failed = [i for i, p in enumerate(procs) if p.returncode != 0]
        print(f"bench.py: rank(s) {failed} exited with {rc}", file=sys.stderr)
        tails(failed)
    rd.join(timeout=10)
    line = (out[0] if out else b"").decode()
    if rc == 0 and line.strip():
        sys.stdout.write(line)
        sys.stdout.flush()
    elif rc == 0:
        rc = 1
        print("bench.py: rank 0 produced no result line", file=sys.stderr)
        tails([0])
    if rc == 0:
        import shutil
        shutil.rmtree(logdir, ignore_errors=True)
    return rc


class GpuSampler:
    """sclk / board power of this rank's GPU, sampled from sysfs while the timed steps run (explains box-to-box spread of
    the same build; informational: the clock an MFMA loop holds in-kernel is up to 10 % under pp_dpm_sclk)."""

    def __init__(self, index, pci=None):
        import glob
        import threading
        cards = sorted(d for d in glob.glob("/sys/class/drm/card[0-9]*/device") if os.path.exists(d + "/pp_dpm_sclk"))
        self.dev = None
        if pci:                              # the card whose PCI address is this rank's device (a box exposes all its cards)
            for d in cards:
                if os.path.basename(os.path.realpath(d)).lower().endswith(pci.lower()):
                    self.dev = d
        if self.dev is None:
            self.dev = cards[index] if index < len(cards) else None
        self.matched = bool(pci) and self.dev is not None and os.path.basename(os.path.realpath(self.dev)).lower().endswith(pci.lower())
        self.sclk, self.power, self.ptime = [], [], []
        self._stop = threading.Event()
        self._t = threading.Thread(target=self._run, daemon=True)

    def _read(self):
        try:
            for ln in open(self.dev + "/pp_dpm_sclk"):
                if "*" in ln:
                    self.sclk.append(int("".join(c for c in ln.split(":")[1] if c.isdigit())))
        except Exception:
            pass
        try:
            import glob
            for f in glob.glob(self.dev + "/hwmon/hwmon*/power1_average") + glob.glob(self.dev + "/hwmon/hwmon*/power1_input"):
                self.power.append(int(open(f).read()) / 1e6)
                self.ptime.append(time.perf_counter())
                break
        except Exception:
            pass

    def _run(self):
        while not self._stop.is_set():
            self._read()
            self._stop.wait(0.05)

    def __enter__(self):
        if self.dev:
            self._t.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        if self.dev:
            self._t.join(timeout=1)

    def mean_power_w(self):
        """time-weighted mean of the board-power samples over the sampled region (trapezoid rule); None without samples"""
        if len(self.power) < 2:
            return self.power[0] if self.power else None
        e = sum(0.5 * (self.power[i] + self.power[i + 1]) * (self.ptime[i + 1] - self.ptime[i]) for i in range(len(self.power) - 1))
        return e / (self.ptime[-1] - self.ptime[0])

    def summary(self):
        med = lambda v: sorted(v)[len(v) // 2] if v else None
        pw = self.mean_power_w()
        return {"sclk_mhz_median": med(self.sclk), "sclk_mhz_min": min(self.sclk) if self.sclk else None,
                "power_w_median": med(self.power), "power_w_max": max(self.power) if self.power else None,
                "power_w_mean": round(pw, 1) if pw is not None else None,
                "samples": len(self.sclk), "card_matched_by_pci": self.matched,
                "source": "sysfs pp_dpm_sclk / hwmon power1_average, 50 ms period, timed region only"}


def energy_fields(sampler, ms_mean, images_per_step, tflop_per_step):
    """The chip is power-bound under this workload (the clock follows the load), so energy per step is the currency an A/B
    should be judged in: J/step = mean board power over the timed region x mean step time (the region is K back-to-back
    steps; power1_average is the SMU's own running average, sampled every 50 ms)."""
    pw = sampler.mean_power_w() if sampler is not None else None
    if pw is None:
        return {"energy_j_per_step": None, "energy_note": "no hwmon power samples on this box"}
    j = pw * ms_mean * 1e-3
    return {"energy_j_per_step": round(j, 2), "energy_j_per_image": round(j / images_per_step, 2),
            "energy_j_per_tflop": round(j / tflop_per_step, 4) if tflop_per_step else None,
            "energy_note": "mean hwmon board power over the timed region x ms_per_step_mean (rank 0's card); J per algorithmic TFLOP "
                           "of the step (20.31 TFLOP per image)"}


def run_protocol(args, rank, world, step, barrier, allreduce_max, sampler=None):
    """The contract's timing protocol, shared by the GPU run and the CPU dry run: W untimed steps, barrier, EXACTLY K
    timed steps, barrier, MAX over ranks.  `step(i)` returns an object with .elapsed_time() semantics or None."""
    for i in range(args.warmup):
        step(None)
    barrier()
    marks = []
    t0 = time.perf_counter()
    if sampler is not None:
        sampler.__enter__()
    for i in range(args.steps):
        step(marks)
    barrier()
    dt = time.perf_counter() - t0
    if sampler is not None:
        sampler.__exit__()
    return allreduce_max(dt), marks


def after_timing(args, rank, world, step, instrument, barrier):
    """What follows the timed region on EVERY rank, in the same order on every rank (a step contains the gradient
    all-reduce, so a rank that skipped the instrumented replay would leave the others waiting in it): the per-kernel
    replay, then -- rank 0, one GPU only -- the CPU baseline, then the closing barrier."""
    roof = None
    if not args.no_roofline:
        roof = instrument(step)                  # all ranks replay; only rank 0 keeps the report
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.model, args.cpu_budget, args.cpu_reference_faithful)
    barrier()
    return (roof if rank == 0 else None), cpu


def pin_rank_to_cores(local_rank, local_world):
    """N ranks on one host: give each a disjoint, contiguous slice of the cores this process may use, so that the ranks'
    launch threads (and torch's intra-op pool) do not migrate onto each other's cores; returns the slice size (0: untouched).
    (train_sdxl_zh.sh:17-22 starts 8 ranks per node and leaves placement to the scheduler.)"""
    try:
        cores = sorted(os.sched_getaffinity(0))
        if local_world <= 1 or len(cores) < 2 * local_world:
            return 0
        per = len(cores) // local_world
        mine = cores[local_rank * per:(local_rank + 1) * per]
        os.sched_setaffinity(0, mine)
        torch.set_num_threads(max(1, min(8, per)))
        return per
    except (AttributeError, OSError):
        return 0


def hbm_plan_gb(cfg, B, hw, L, merged=True):
    """host-only planning pass of the C ABI (pea_unet_plan: no device needed): weights + activations + gradients of the step's
    graphs at this per-GPU batch, in GB -- checked against the device's memory BEFORE anything is allocated"""
    from pea_diffusion_amd import config as pc
    from pea_diffusion_amd._lib import check, lib
    c = pc.to_c(cfg)

    def plan(batch, flags, Lctx):
        w, a, g = ctypes.c_longlong(), ctypes.c_longlong(), ctypes.c_longlong()
        check(lib().pea_unet_plan(ctypes.byref(c), batch, hw, hw, Lctx, flags, None, None, None, ctypes.byref(w), ctypes.byref(a),
                                  ctypes.byref(g)))
        return w.value, a.value, g.value
    w, a, g = plan(2 * B if merged else B, 1, L)           # merged passes: ONE graph over 2B samples, gradients for the first B
    total = w + a + g
    if not merged:
        total += plan(B, 0, 77)[1]                          # + the teacher's activations (weights shared or counted once)
    return round(total / 1e9, 1)


def dry_run_collective(args, rank, world, json_fd):
    """Host-logic check of the N-rank path without a GPU (tests/test_dp_cpu.py): rendezvous over gloo, shard a global
    batch, then the SAME control flow as the GPU run -- run_protocol() (warm-up, barrier, K steps each ending in ONE
    all-reduce-mean of a flat buffer through pea_diffusion_amd.dist, barrier, max over ranks), after_timing() (the
    instrumented replay on every rank, closing barrier) -- and one JSON line from rank 0.  NOT a benchmark: no kernel of
    the product runs here."""
    import torch.distributed as dist
    from pea_diffusion_amd import dist as pdist
    if world > 1:
        assert pdist.init_control_plane(timeout_s=args.rendezvous_timeout) == world
    if args.inject_comm_failure:
        # the failure path of the GPU run without a GPU: what main() does when NativeComm.from_env() raises
        e = pdist.CommTimeout("injected: rendezvous not complete") if args.inject_comm_failure == "timeout" else RuntimeError("injected")
        comm_failure_exit(e, rank)
    B = args.batch or (4 if world == 1 else 8)
    cores_per_rank = pin_rank_to_cores(int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    plan_gb = None
    if args.model == "sdxl":
        from pea_diffusion_amd import config as pc
        plan_gb = hbm_plan_gb(pc.sdxl_config(), B, args.latent or 128, args.ctx)
        assert plan_gb < 288.0 * 0.92, f"per-rank plan of {plan_gb} GB does not fit 288 GB of HBM3E"
    g = torch.Generator().manual_seed(1234)
    glob = torch.randn(world * B, 16, generator=g)                         # every rank builds the same global batch
    mine = pdist.shard_batch({"x": glob}, rank, world)["x"]
    flat = mine.sum(0).repeat(64).contiguous()                             # this rank's "gradient"
    last = [None]
    count = [0]
    ar_s = [0.0]
    local_only = [False]

    def step(marks):
        f = flat.clone()
        if not local_only[0]:
            t0 = time.perf_counter()
            pdist.allreduce_mean_(f)
            ar_s[0] = time.perf_counter() - t0
            last[0] = f
        count[0] += 1

    def barrier():
        pdist.control_barrier()

    def instrument(step_fn):
        for _ in range(min(args.steps, 3)):
            step_fn(None)
        return {"replayed_steps": min(args.steps, 3)}

    args.no_cpu_baseline = True
    tmax, _ = run_protocol(args, rank, world, step, barrier, pdist.control_max)
    # the single-GPU-equivalent block of the GPU run: the same steps with the collective switched off, symmetric on all ranks
    barrier()
    local_only[0] = True
    t0 = time.perf_counter()
    for _ in range(3):
        step(None)
    solo_s = (time.perf_counter() - t0) / 3
    local_only[0] = False
    barrier()
    roof, _ = after_timing(args, rank, world, step, instrument, barrier)
    want = glob.view(world, B, 16).sum(1).mean(0).repeat(64)
    err = float((last[0] - want).abs().max())
    if world > 1:
        dist.destroy_process_group()
    if rank == 0:
        out = {"metric": "launcher dry run (no kernels; not a benchmark)", "value": None, "unit": None, "n_gpus": world,
               "ranks": world, "global_batch": world * B, "per_gpu_batch": B, "allreduce_max_abs_err": err,
               "seconds_max_over_ranks": tmax, "dry_run": True, "steps_run_per_rank": count[0],
               "replay": roof, "hbm_plan_gb_per_rank": plan_gb, "hbm_capacity_gb": 288.0, "cores_per_rank": cores_per_rank,
               # the N-rank schema of the GPU line (NRANK_KEYS), with what a CPU run can fill in
               "rccl_ranks": 0, "collective": "gloo all_reduce on CPU tensors (dry run: no RCCL communicator exists here)",
               "control_plane": "gloo", "allreduce_ms": round(ar_s[0] * 1e3, 4), "allreduce_exposed_ms": None,
               "allreduce_bytes": int(flat.numel() * 4),
               "single_gpu_equivalent": {"ms_per_step": round(solo_s * 1e3, 4), "what": "dry run: the step with the collective off"}}
        assert all(k in out for k in NRANK_KEYS)
        os.write(json_fd, (json.dumps(out) + "\n").encode())


def comm_failure_exit(e, rank):
    """--collective native and the library's communicator could not be built: that is the run's collective, so the rank
    stops with a code of its own (never a silent fall-back to another collective).  os._exit: after a rendezvous timeout a
    detached thread is still inside RCCL's bootstrap, and interpreter shutdown must not run library destructors beside it."""
    from pea_diffusion_amd import dist as pdist
    timeout = isinstance(e, pdist.CommTimeout)
    print(f"bench.py: rank {rank}: native RCCL communicator {'rendezvous timed out' if timeout else 'failed'}: {e}\n"
          "bench.py: --collective native makes this fatal (use --collective torch to run torch.distributed's collective instead)",
          file=sys.stderr, flush=True)
    os._exit(RC_COMM_TIMEOUT if timeout else RC_COMM_FAILED)


FAMILY_BOUND = {"gemm_lc[p]_kernel<plain>": "mfma", "gemm_lc[p]_kernel<conv3x3>": "mfma", "attn_fwd": "mfma",
                "attn_bwd": "mfma", "groupnorm": "hbm", "layernorm": "hbm", "elementwise": "hbm", "kd_loss": "hbm"}


def adapter_golden_rel_l2():
    """measured, in this process: rel-L2 of the HIP adapter forward against the reference MLP's own outputs (tests/golden/
    mlp_sdxl_6M.npz, generated by importing the reference: oracle/make_golden.py) -- printed beside the stated tolerances"""
    try:
        import numpy as np
        from pea_diffusion_amd.adapter import PEAAdapter
        g = np.load(os.path.join(ROOT, "tests", "golden", "mlp_sdxl_6M.npz"))
        a = [int(v) for v in g["args"]]
        rs = torch.random.get_rng_state()
        torch.manual_seed(int(g["seed"]))
        m = PEAAdapter(a[0], a[1], a[2], a[3], bool(a[4]))
        torch.random.set_rng_state(rs)
        wsum = float(sum(v.double().abs().sum().item() for v in m.state_dict().values()))
        if abs(wsum - float(g["wsum"])) > 1e-6 * float(g["wsum"]):
            return None
        m = m.cuda()
        with torch.no_grad():
            outs = m(torch.from_numpy(g["x"]).cuda())
        errs = []
        for i, o in enumerate(outs):
            ref = torch.from_numpy(g[f"out{i}"]).float()
            errs.append(float((o.float().cpu() - ref).norm() / ref.norm()))
        return [round(e, 5) for e in errs]
    except Exception as e:          # the golden file travels with the repo; never fail the bench line over this note
        return f"unavailable ({type(e).__name__})"


def sustained_mfma_peak(seconds=2.0):
    """bare-MFMA probe of THIS device (csrc/prof.hip:pea_probe_mfma_peak): the GEMM family's instruction from registers on
    random operands, back-to-back launches for `seconds`; TFLOP/s of the last launch + its in-kernel clock.  The attention
    kernels' 32x32x16 shape is probed for a quarter of that time (MI355X_MICROARCH.md DVFS give-back items 6, 7)."""
    from pea_diffusion_amd._lib import lib
    out = {}
    for key, shape32, sec in (("v_mfma_f32_16x16x32_bf16", 0, seconds), ("v_mfma_f32_32x32x16_bf16", 1, seconds / 4)):
        try:
            tf, mhz = ctypes.c_double(), ctypes.c_double()
            rc = lib().pea_probe_mfma_peak(ctypes.c_double(sec), shape32, ctypes.byref(tf), ctypes.byref(mhz), None)
            out[key] = {"tflops": round(tf.value, 1), "in_kernel_clock_mhz": round(mhz.value, 0)} if rc == 0 and tf.value > 0 else None
        except Exception:                                         # a note beside the headline: never fail the line over it
            out[key] = None
    return out


def family_rooflines(fams, nprof, sustained=None):
    """north_star: 'achieved fraction of MFMA and HBM roofline per kernel' -- one entry per kernel family from the
    HIP-event replay: algorithmic FLOPs (MFMA-bound families) or algorithmic bytes (HBM-bound) over the summed launch
    durations, against 2.5 PFLOP/s dense bf16 / 8 TB/s."""
    out = []
    for x in fams:
        if not x["launches"]:
            continue
        bound = FAMILY_BOUND.get(x["name"], "hbm")
        sec = x["ms"] * 1e-3
        if bound == "mfma":
            ach, peak, unit = x["flops"] / sec / 1e12, MFMA_PEAK_TFLOPS, "TFLOP/s"
        else:
            ach, peak, unit = x["bytes"] / sec / 1e9, HBM_PEAK_GBS, "GB/s"
        row = {"family": x["name"], "bound": bound, "ms_per_step": round(x["ms"] / nprof, 3),
               "launches_per_step": x["launches"] // nprof, "achieved": round(ach, 1), "peak": peak, "unit": unit,
               "frac": round(ach / peak, 4)}
        if bound == "mfma" and sustained:
            sp = sustained.get("v_mfma_f32_32x32x16_bf16" if x["name"].startswith("attn") else "v_mfma_f32_16x16x32_bf16")
            if sp and sp["tflops"] > 0:
                row["frac_of_sustained"] = round(ach / sp["tflops"], 4)
        out.append(row)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0,
                    help="per-GPU batch; default 4 at --gpus 1 (BASELINE configs[1]) and 8 at --gpus N>1 (configs[2]: 8 x 8 = 64)")
    ap.add_argument("--collective", default="native", choices=["native", "torch"],
                    help="native: RCCL communicator + comm stream inside libpea_hip.so (pea_comm_*); torch: torch.distributed")
    ap.add_argument("--dry-run-collective", action="store_true",
                    help="launcher / rendezvous / all-reduce path only, on CPU tensors over gloo (tests/test_dp_cpu.py); not a benchmark")
    ap.add_argument("--model", default="sdxl", choices=["sdxl", "tiny"])
    ap.add_argument("--latent", type=int, default=0, help="latent side (default: model sample_size, 128 = 1024 px)")
    ap.add_argument("--ctx", type=int, default=77, help="student context length (77; cn_clip default is 52)")
    ap.add_argument("--hidden", type=int, default=1024, help="adapter hidden dim (1024 = the 6M-param adapter)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=400.0,
                    help="seconds the CPU-oracle leg may take for the metric's 1024 px step (measured: 55-90 s on the box)")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-reference-faithful", action="store_true",
                    help="also time the CPU step with the student UNet left trainable as the reference leaves it "
                         "(train_sdxl_zh.py:166-168: 2.57 B unused weight gradients); one-off, recorded under profiles/")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-dead-row-line", action="store_true",
                    help="skip the supplementary dead-row-elimination measurement (10 extra steps after the timed region)")
    ap.add_argument("--force-collective", action="store_true", help="init RCCL and all-reduce even with one rank (path test)")
    ap.add_argument("--dump-prof", default="", help="write every profiled launch (shape-tagged) to this CSV")
    ap.add_argument("--breakdown", action="store_true", help="print the per-kernel-family table to stderr")
    ap.add_argument("--student", default="same", choices=["same", "ssd1b"],
                    help="BASELINE config 4: a smaller student UNet (SSD-1B-shaped, own weights) under the SDXL teacher")
    ap.add_argument("--with-vae", action="store_true",
                    help="also run a later batch's VAE encode (train_sdxl_zh.py:306-309, 1024x1024 pixels) on a side HIP "
                         "stream, enqueued between training_step and optimizer_step so that it overlaps the gradient "
                         "all-reduce; not the default: BASELINE's metric starts from latents (SURVEY 8d)")
    ap.add_argument("--single-stream", action="store_true", help="analysis only: teacher and student passes on ONE stream")
    ap.add_argument("--rendezvous-timeout", type=float, default=300.0,
                    help="seconds a rank waits for the others: in the gloo control plane's rendezvous and in the RCCL "
                         "communicator's (pea_comm_init_timeout); past it the rank exits non-zero instead of hanging")
    ap.add_argument("--inject-comm-failure", default="", choices=["", "error", "timeout"], help=argparse.SUPPRESS)
    ap.add_argument("--launch-timeout", type=float, default=0.0,
                    help="seconds the self-launched ranks (--gpus N without torch.distributed.run) may take in all; "
                         "default 900 + 2 s per step")
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        cpu_baseline_worker(args.model, args.cpu_budget, args.cpu_reference_faithful)
        return

    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None:
        # `python bench.py --gpus N` without torch.distributed.run: start the N ranks here.  Nothing in this process has
        # touched the GPU (importing torch does not), and it never does: it only waits and relays rank 0's line.
        sys.exit(launch_ranks(args.gpus, sys.argv[1:],
                              args.launch_timeout or 900.0 + 2.0 * (args.steps + args.warmup)))
    world = int(world_env or "1")
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch `python -m torch.distributed.run --nproc-per-node "
              f"{args.gpus} bench.py --gpus {args.gpus} ...` or plain `python bench.py --gpus {args.gpus}`", file=sys.stderr)
        sys.exit(2)

    # stdout carries exactly ONE line (the JSON result); libraries that print to fd 1 (the RCCL banner) go to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    if args.dry_run_collective:
        dry_run_collective(args, rank, world, json_fd)
        return
    use_dist = world > 1 or args.force_collective        # --force-collective: exercise the RCCL path with one rank
    from pea_diffusion_amd import dist as pdist
    control_plane = None
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        if args.collective == "native":
            # control plane on gloo (CPU tensors): ncclUniqueId broadcast, barriers, max of wall times.  The library's
            # communicator (below) is then the ONLY RCCL communicator of this process: one bootstrap, one set of rings.
            assert pdist.init_control_plane(timeout_s=args.rendezvous_timeout) == world
            control_plane = "gloo"
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            control_plane = "nccl (torch.distributed)"
        assert dist.get_world_size() == world
    dev = torch.device("cuda", local_rank if use_dist else 0)
    torch.cuda.set_device(dev)

    from pea_diffusion_amd import config as pc
    from pea_diffusion_amd._lib import lib
    from pea_diffusion_amd.adapter import PEAAdapter
    from pea_diffusion_amd.train import PEATrainer
    from pea_diffusion_amd.unet import HipUNet

    cfg = pc.sdxl_config() if args.model == "sdxl" else pc.tiny_config()
    hw = args.latent or cfg.sample_size
    B = args.batch or (4 if world == 1 else 8)            # BASELINE configs[1]: 4 on one GPU; configs[2]: 8 x 8 = 64
    cores_per_rank = pin_rank_to_cores(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))) if world > 1 else 0
    plan_gb = hbm_plan_gb(cfg if args.student == "same" else pc.ssd1b_config(), B, hw, args.ctx, merged=args.student == "same")
    free_b, total_b = torch.cuda.mem_get_info(dev)
    if plan_gb * 1e9 > 0.97 * total_b:
        print(f"bench.py: the step at per-GPU batch {B} plans {plan_gb} GB of HBM; this device has {total_b / 1e9:.0f} GB", file=sys.stderr)
        sys.exit(3)
    enc_dim = 1024 if args.model == "sdxl" else 128
    hidden = args.hidden if args.model == "sdxl" else 192
    if args.student == "ssd1b" and args.model == "sdxl":
        student = HipUNet(pc.ssd1b_config(), B, hw, hw, args.ctx, needs_grad=True)
        student.init_random(seed=7)
        teacher = HipUNet(cfg, B, hw, hw, 77, needs_grad=False)
        teacher.init_random(seed=8)
    else:
        student = HipUNet(cfg, B, hw, hw, args.ctx, needs_grad=True)
        student.init_random(seed=7)
        teacher = HipUNet(cfg, B, hw, hw, 77, needs_grad=False, share_weights_from=student)
    torch.manual_seed(7)
    adapter = PEAAdapter(enc_dim, cfg.pooled_dim, hidden, cfg.cross_attention_dim, False).to(dev)
    trainer = PEATrainer(adapter, student, teacher)
    batch = synthetic_batch(cfg, B, args.ctx, enc_dim, hw, dev, seed=100 + rank)
    # profiling only: how many samples the KD-loss kernel actually reads (zh_or_not == 0) -> its family's byte count
    lib().pea_trainer_set_option(trainer._h, b"kd_samples_hint", int((batch["zh_or_not"] == 0).sum().item()))

    comm, collective = None, None
    if use_dist:
        if args.collective == "native":
            try:
                comm = pdist.NativeComm.from_env(timeout_s=args.rendezvous_timeout)   # ncclUniqueId from rank 0 over gloo
            except Exception as e:                            # the run's collective is missing: fatal, own exit code
                comm_failure_exit(e, rank)
            trainer.attach_comm(comm)
            collective = "RCCL all-reduce on the communicator's own HIP stream (libpea_hip.so pea_allreduce_grads)"
            comm.broadcast_(adapter.flat_param, 0)                # identical replicas
        else:
            collective = "torch.distributed all_reduce (nccl backend = RCCL), async on ProcessGroupNCCL's stream"
            pdist.broadcast_params_(adapter.flat_param, src=0)
        adapter.mark_updated()

    vae = None
    if args.with_vae:
        from pea_diffusion_amd.vae import HipVAEEncoder
        vae = HipVAEEncoder(pc.sdxl_vae_config() if args.model == "sdxl" else pc.tiny_vae_config(), B, hw * 8, hw * 8)
        vae.init_random(seed=11)
        pixels = torch.randn(B, 3, hw * 8, hw * 8, device=dev).clamp_(-1, 1)
        side = torch.cuda.Stream(device=dev)
        encoded = []                         # (latents, event): encodes in flight, oldest first

    def step(marks):
        """one step in the order of INTEGRATION.md's data-parallel loop: KD step (ends by LAUNCHING the all-reduce on
        the comm stream) -> enqueue a later batch's VAE encode on the side stream (starts when the gradients are
        complete, i.e. beside the all-reduce, and runs on beside AdamW and the next step) -> join + fused AdamW."""
        if vae is not None and len(encoded) >= 2:      # latents encoded while the PREVIOUS step ran
            lat, ev = encoded.pop(0)
            torch.cuda.current_stream().wait_event(ev)
            batch["latents"] = lat
        trainer.training_step(batch, async_allreduce=True)
        if args.force_collective and world == 1 and comm is None:
            dist.all_reduce(adapter.flat_grad)
        if vae is not None:
            side.wait_stream(torch.cuda.current_stream())        # = "adapter gradients complete"
            with torch.cuda.stream(side):
                lat = vae.encode_latents(pixels)
                ev = torch.cuda.Event()
                ev.record(side)
            encoded.append((lat, ev))
        trainer.optimizer_step()              # joins the all-reduce, then fused AdamW
        if marks is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append(e)

    def barrier():
        # device work of this rank done, THEN the ranks meet (a rank must not leave the barrier with kernels of the timed
        # region still running), then once more after the meeting for the nccl control plane's own barrier kernel
        torch.cuda.synchronize()
        if use_dist:
            if control_plane == "gloo":
                pdist.control_barrier()
            else:
                dist.barrier()
                torch.cuda.synchronize()

    def allreduce_max(dt):
        if use_dist and control_plane == "gloo":
            return pdist.control_max(dt)
        if use_dist:
            tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            return float(tmax.item())
        return dt

    def instrument(step_fn):
        """per-launch HIP events on the launch stream over a single-stream replay of the step (every rank runs it: the
        step contains the collective)"""
        L = lib()
        nprof = min(args.steps, 3)
        L.pea_trainer_set_option(trainer._h, b"two_stream", 0)   # clean per-kernel durations (no cross-stream overlap)
        L.pea_prof_reset()
        L.pea_prof_enable(1)
        for _ in range(nprof):
            step_fn(None)
        torch.cuda.synchronize()
        L.pea_prof_enable(0)
        if args.dump_prof and rank == 0:
            L.pea_prof_dump(args.dump_prof.encode())
        L.pea_trainer_set_option(trainer._h, b"two_stream", 1)
        fams = []
        for f in range(8):
            t, fl, by, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_longlong()
            L.pea_prof_report(f, ctypes.byref(t), ctypes.byref(fl), ctypes.byref(by), ctypes.byref(n))
            fams.append(dict(name=L.pea_prof_family_name(f).decode(), ms=t.value, flops=fl.value, bytes=by.value,
                             launches=n.value))
        L.pea_prof_reset()
        sustained = sustained_mfma_peak(2.0) if rank == 0 or world == 1 else None
        gem = [fams[0], fams[1]]
        g_ms = sum(x["ms"] for x in gem)
        g_fl = sum(x["flops"] for x in gem)
        g_n = sum(x["launches"] for x in gem)
        ach = g_fl / (g_ms * 1e-3) / 1e12 if g_ms > 0 else 0.0
        roof = {"bound": "mfma", "kernel": "gemm_lc_kernel / gemm_lcp_kernel (plain + implicit-GEMM conv3x3)",
                "achieved": round(ach, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / MFMA_PEAK_TFLOPS, 4),
                "sustained_peak": (sustained or {}).get("v_mfma_f32_16x16x32_bf16"),
                "frac_of_sustained": (round(ach / sustained["v_mfma_f32_16x16x32_bf16"]["tflops"], 4)
                                      if sustained and sustained.get("v_mfma_f32_16x16x32_bf16") else None),
                "sustained_peak_note": "bare v_mfma_f32_16x16x32_bf16 from registers on random operands, two waves per SIMD on every "
                                       "CU, 2 s of back-to-back launches in this process after the timed region; in_kernel_clock = "
                                       "delta s_memtime / delta s_memrealtime x 100 MHz, median over workgroups; the primary "
                                       "denominator stays the 2.5 PFLOP/s spec peak (`frac`)",
                "sustained_peak_32x32x16": (sustained or {}).get("v_mfma_f32_32x32x16_bf16"),
                "traffic": pmc_traffic()[0],
                "traffic_unit": "bytes per launch, TCC FETCH_SIZE x2 (gfx950) + WRITE_SIZE over the family's launches, from "
                                f"two separate rocprofv3 --pmc passes of this command (profiles/{pmc_traffic()[1]}); not "
                                "re-measured inside this run (PMC needs the profiler)",
                "launches_per_step": g_n // nprof, "avg_launch_us": round(g_ms * 1e3 / max(g_n, 1), 2),
                "gflop_per_launch": round(g_fl / max(g_n, 1) / 1e9, 3),
                "ms_per_step_single_stream": round(g_ms / nprof, 2),
                "flops_counted": "2*M*N*K of each launch as EXECUTED: a transposed stride-2 dgrad counts its real quarter of the taps, the "
                                 "upsampler convs in sub-pixel form count 4 Cin (not 9 Cin) per output pixel -- 1.6 TFLOP per step fewer "
                                 "than SURVEY 8d's formulation, which `step_frac` keeps using",
                "method": "hip events around every launch on the launch stream; instrumented single-stream replay of the timed "
                          "steps (the timed region overlaps teacher and student passes on two streams)",
                "families": family_rooflines(fams, nprof, sustained),
                "tolerances": "per kernel: fp32-stored outputs rtol 1e-3 / atol 1e-4 vs the fp32 oracle; bf16-stored outputs 1 bf16 ulp "
                              "(2-4 ulp for multi-product attention gradients and folded LN->Linear) + rms-scaled atol "
                              "(tests/test_ops_gpu.py).  END TO END the path is ON the bf16-storage noise floor, not within rtol 1e-3: "
                              "relative L2 vs the fp32 oracle 6.8e-3 (eps) and 7.9e-3 (flat adapter gradient) at 1024x1024 = 0.97-1.06 x what "
                              "bf16 storage alone does to the oracle in the same run (limit 1.5 x that floor; fixed limits 1.5e-2 / 2e-2), "
                              "every one of the 140 cross-attention K / V projections and 17 time_emb_proj layers on its own (limit 2e-2).  "
                              "That oracle comparison runs at batch 2 (tests/test_model_gpu.py::test_sdxl_full_model_step_vs_oracle_1024); "
                              "THIS line's workload (batch 4) meets the oracle through properties only (idempotence, masks, per-sample "
                              "independence, bit-reproducibility: tests/test_configs_gpu.py, test_model_gpu.py::test_full_size_properties)",
                "adapter_golden_rel_l2": {"measured_in_this_run": adapter_golden_rel_l2(), "limit": 1e-2,
                                          "what": "HIP adapter forward (pooled, tokens) vs the reference MLP's own outputs, "
                                                  "tests/golden/mlp_sdxl_6M.npz (one bf16 rounding of weights and activations)"}}
        if args.breakdown and rank == 0:
            tot = sum(x["ms"] for x in fams)
            for x in fams:
                tf = x["flops"] / (x["ms"] * 1e-3) / 1e12 if x["ms"] > 0 else 0
                gb = x["bytes"] / (x["ms"] * 1e-3) / 1e9 if x["ms"] > 0 else 0
                print(f"  {x['name']:28s} {x['ms'] / nprof:9.2f} ms/step {x['launches'] // nprof:6d} launches "
                      f"{tf:8.1f} TFLOP/s {gb:8.0f} GB/s(alg)", file=sys.stderr)
            print(f"  instrumented families total {tot / nprof:.2f} ms/step (single-stream replay)", file=sys.stderr)
        return roof

    if args.single_stream:
        lib().pea_trainer_set_option(trainer._h, b"two_stream", 0)
    pci = None
    try:
        pr = torch.cuda.get_device_properties(dev)
        pci = f"{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
    except Exception:
        pass
    sampler = GpuSampler(local_rank, pci) if rank == 0 else None
    dt, marks = run_protocol(args, rank, world, step, barrier, allreduce_max, sampler)
    ms_mean = dt / args.steps * 1e3
    ips = world * B * args.steps / dt
    # per-step device times: distance between the events recorded behind each step's AdamW on the compute stream
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(len(marks) - 1))
    ms_median = per_step[len(per_step) // 2] if per_step else ms_mean
    loss = float(trainer.losses[0])
    allreduce_ms = allreduce_exposed_ms = None
    if comm is not None:
        allreduce_ms = round(comm.last_ms(), 4)      # device time of the last step's all-reduce + 1/world scale (comm stream)
        allreduce_exposed_ms = round(comm.last_exposed_ms(), 4)   # how long AdamW's stream stood still for it
    rccl_ranks = (comm.world if comm is not None else dist.get_world_size()) if use_dist else 1

    # N > 1: the same per-GPU batch with the collective switched off, on every rank (symmetric: no rank waits for another),
    # measured in this run -- the single-GPU number the N-rank value is weak scaling against (the driver's N = 1 line runs
    # BASELINE configs[1], batch 4: a different per-GPU workload)
    solo_ms = None
    if world > 1 or args.force_collective:                  # (--force-collective: the same code path with one rank)
        barrier()
        # the local-only steps update each replica from its own shard: parameters AND both AdamW moments (and the step counter)
        # are put back afterwards, so every rank leaves this block in the replicated state it entered with
        snap = (adapter.flat_param.clone(), trainer._m.clone(), trainer._v.clone(), trainer.global_step)
        trainer.local_only = True
        evs = []
        for _ in range(7):
            step(evs)
        torch.cuda.synchronize()
        ts = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(1, len(evs) - 1))
        solo_ms = ts[len(ts) // 2]
        trainer.local_only = False
        with torch.no_grad():
            adapter.flat_param.copy_(snap[0])
            trainer._m.copy_(snap[1])
            trainer._v.copy_(snap[2])
        trainer.global_step = snap[3]
        del snap
        adapter.mark_updated()
        barrier()

    # Supplementary (never `value`): the same steps with dead-row elimination -- teacher rows whose KD weight (1 - zh_or_not) is
    # zero are not computed (PEATrainer.skip_dead_teacher_rows; identical losses and gradients, tests/test_dead_rows_gpu.py).
    # The headline above computes every teacher row, as the reference does.
    dre = None
    if world == 1 and not args.no_dead_row_line and lib().pea_trainer_get_option(trainer._h, b"merge_state") == 1:
        zh_host = batch["zh_or_not"].cpu()
        batch_dre = dict(batch, zh_or_not=zh_host)               # the dataloader's mask is a host tensor: no synchronising copy
        trainer.skip_dead_teacher_rows = True
        try:
            def step_dre(marks):
                trainer.training_step(batch_dre, async_allreduce=True)
                trainer.optimizer_step()
                if marks is not None:
                    e = torch.cuda.Event(enable_timing=True)
                    e.record()
                    marks.append(e)
            for _ in range(3):
                step_dre(None)
            torch.cuda.synchronize()
            evs = []
            t0 = time.perf_counter()
            for _ in range(10):
                step_dre(evs)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            ts = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(len(evs) - 1))
            rows = lib().pea_trainer_get_option(trainer._h, b"merged_rows")
            dre = {"ms_per_step": round(ts[len(ts) // 2], 3), "images_per_s": round(B * 10 / wall, 3),
                   "merged_rows": rows, "of": 2 * B,
                   "what": f"{int((zh_host == 1).sum())} of {B} samples have zh_or_not = 1: their teacher rows carry KD weight 0 and are "
                           "skipped; supplementary, NOT the headline (10 steps after the timed region)"}
        except Exception as e:                                    # supplementary: never lose the headline over it
            dre = {"error": f"{type(e).__name__}: {e}"[:300]}
        finally:
            trainer.skip_dead_teacher_rows = False
            try:                                                  # back on the full 2B-row context before the instrumented replay
                trainer.training_step(batch, async_allreduce=True)
                trainer.optimizer_step()
                torch.cuda.synchronize()
            except Exception as e:                                # a device error in the supplementary block must not cost the headline line
                dre = dict(dre or {}, restore_error=f"{type(e).__name__}: {e}"[:300])

    roof, cpu = after_timing(args, rank, world, step, instrument, barrier)

    if use_dist:
        if comm is not None:
            comm.close()
        dist.destroy_process_group()
    if rank == 0:
        mem = student.memory()
        if roof is not None and TFLOP_PER_IMAGE.get(args.model) and args.student == "same":
            # the WHOLE step against the spec peak (all families, all gaps): algorithmic TFLOP of the step / median step time
            step_tf = B * TFLOP_PER_IMAGE[args.model] / (ms_median * 1e-3)
            roof["step_achieved"] = round(step_tf, 1)
            roof["step_frac"] = round(step_tf / MFMA_PEAK_TFLOPS, 4)
            roof["step_frac_note"] = ("whole training step: 20.31 TFLOP per image x per-GPU batch / median step time, against the same 2.5 PFLOP/s "
                                      "spec peak as `frac` (which is the GEMM + conv family alone, from its launch durations)")
        out = {
            "metric": "training images/sec (SDXL 1024px bf16, adapter-only bwd)", "value": round(ips, 4),
            "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_median, 3), "ms_per_step_mean": round(ms_mean, 3),
            "ms_per_step_min_max": [round(per_step[0], 3), round(per_step[-1], 3)] if per_step else None,
            "timing": "value = global batch x K / wall time of the K steps between barrier+synchronize pairs, max over "
                      "ranks (so value x ms_per_step_mean = 1000 x global batch); ms_per_step = median of the per-step "
                      "device times (events behind each step's AdamW, SURVEY 8d)",
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{args.model.upper()} {hw * 8}x{hw * 8} KD training step (teacher fwd + student "
                                   f"fwd + student dgrad bwd + adapter fwd/bwd + AdamW), per-GPU batch {B}",
                       "global_batch": world * B, "per_gpu_batch": B, "latent": hw, "ctx_len": args.ctx,
                       "adapter": f"MLP({enc_dim},{cfg.pooled_dim},{hidden},{cfg.cross_attention_dim})",
                       "parallelism": f"dp{world}",
                       "per_gpu_batch_rule": "4 at N=1 (BASELINE configs[1]); 8 at N>1 (configs[2]: global 64 = 8 x 8)",
                       "weights": ("random init (teacher == student checkpoint)" if args.student == "same" else
                                   "random init, SSD-1B student (per-position depths [2,2],[4,4] down / [4,4,10],[2,1,1] up, no mid "
                                   "block; 1 300 195 844 parameters) under the SDXL teacher"),
                       "loss": round(loss, 6),
                       **({"vae_encode": f"a later batch's {hw * 8}x{hw * 8} VAE encode on a side HIP stream, enqueued "
                                         "between training_step and optimizer_step (beside the all-reduce), consumed two steps later"}
                          if args.with_vae else {})},
            "tflop_per_image": TFLOP_PER_IMAGE.get(args.model) if args.student == "same" else None,
            "achieved_tflops_per_gpu": (round(ips / world * TFLOP_PER_IMAGE[args.model], 1)
                                        if TFLOP_PER_IMAGE.get(args.model) and args.student == "same" else None),
            "hbm_resident_gb": round((mem["weight_bytes"] + mem["activation_bytes"] + teacher.memory()["activation_bytes"]
                                      + mem["grad_bytes"]) / 2 ** 30
                                     + lib().pea_trainer_get_option(trainer._h, b"merged_mib") / 1024, 1),
            "passes": ("merged: teacher == student checkpoint, one forward over 2B samples + backward on the first B"
                       if lib().pea_trainer_get_option(trainer._h, b"merge_state") == 1 else
                       "teacher forward on a side HIP stream beside the student forward"),
            "rccl_ranks": rccl_ranks, "collective": collective, "control_plane": control_plane, "allreduce_ms": allreduce_ms,
            "allreduce_exposed_ms": allreduce_exposed_ms,
            "allreduce_bytes": int(adapter.flat_grad.numel() * 4) if use_dist else 0,
            "single_gpu_equivalent": ({"ms_per_step": round(solo_ms, 3), "images_per_s": round(B / solo_ms * 1e3, 3),
                                       "what": f"the same per-GPU batch {B} step with the all-reduce switched off, median of 5 steps on "
                                               "rank 0 right after the timed region of THIS run: the weak-scaling reference for this "
                                               "line (N x images_per_s = perfect scaling)"} if solo_ms else None),
            "dead_row_elimination": dre,
            "hbm_plan_gb_per_rank": plan_gb, "cores_per_rank": cores_per_rank or None,
            "gpu": sampler.summary() if sampler is not None else None,
            **energy_fields(sampler, ms_mean, B, (B * TFLOP_PER_IMAGE[args.model]) if TFLOP_PER_IMAGE.get(args.model) and args.student == "same" else None),
            "roofline": roof, "cpu_baseline": cpu,
        }
        assert all(k in out for k in NRANK_KEYS)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
