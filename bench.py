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


def cpu_baseline_worker(model_name, budget_s):
    """The CPU oracle (oracle/*.py, the restatement of the reference's PyTorch path) timed on this box's
    host cores: one KD step at batch 1 (teacher fwd no_grad + student fwd + adapter-only backward)."""
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
    hw = max(runs)
    dt = runs[hw]
    full = hw == 128 or model_name != "sdxl"
    # value: images/s of the METRIC's workload (1024 px), measured directly -- one whole KD step at batch 1.  Only when
    # the time budget forbids the 1024 px step is the 512 px step reported (value stays null then: no extrapolation).
    res = {"value": round(1.0 / dt, 5) if full else None, "unit": "images/s", "cores": threads, "kind": "port",
           "sample": f"1 KD step (teacher fwd no_grad + student fwd + adapter-only backward), batch 1, {hw * 8}x{hw * 8} px "
                     f"(latent {hw}x{hw}), fp32 torch CPU oracle, teacher==student weights, {dt:.1f} s wall "
                     f"(model build {build_s:.0f} s not counted)" + ("" if full else "; 1024 px step skipped: over the CPU budget"),
           "value_512px": (round(1.0 / runs[64], 5) if 64 in runs else None),
           "seconds": {f"{k * 8}px": round(v, 2) for k, v in runs.items()}}
    print("CPU_BASELINE_JSON " + json.dumps(res), flush=True)


def cpu_baseline(model_name, budget_s):
    """runs the worker in a child process with a hard time limit, so the bench line is always printed"""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "--model", model_name,
           "--cpu-budget", str(budget_s)]
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


def launch_ranks(n, argv):
    """Start `n` ranks of this script as fresh child processes (one per GPU; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*
    set as torch.distributed.run would), relay rank 0's single JSON line, return non-zero if any rank fails.  The
    parent never initialises the GPU and never re-executes itself (train_sdxl_zh.sh:108-114 is the reference's launcher)."""
    import subprocess
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (RCCL across processes on this host driver)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    import threading
    out = []
    rd = threading.Thread(target=lambda: out.append(procs[0].stdout.read()), daemon=True)
    rd.start()
    rc = 0
    while any(p.poll() is None for p in procs):
        failed = [p for p in procs if p.poll() not in (None, 0)]
        if failed:                          # a dead rank leaves the others waiting in a collective: stop exactly those
            rc = failed[0].returncode or 1
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.2)
    for p in procs:
        p.wait()
        if p.returncode != 0 and rc == 0:
            rc = p.returncode
    rd.join(timeout=10)
    line = (out[0] if out else b"").decode()
    if rc == 0 and line.strip():
        sys.stdout.write(line)
        sys.stdout.flush()
    elif rc == 0:
        rc = 1
        print("bench.py: rank 0 produced no result line", file=sys.stderr)
    return rc


def dry_run_collective(args, rank, world, json_fd):
    """Host-logic check of the N-rank path without a GPU (tests/test_dp_cpu.py): rendezvous over gloo, shard a global
    batch, ONE all-reduce-mean of a flat buffer through pea_diffusion_amd.dist, barrier, max-over-ranks timing, one
    JSON line from rank 0.  NOT a benchmark: no kernel of the product runs here."""
    import torch.distributed as dist
    from pea_diffusion_amd import dist as pdist
    if world > 1:
        assert pdist.init_from_env("gloo") == world
    B = args.batch or (4 if world == 1 else 8)
    g = torch.Generator().manual_seed(1234)
    glob = torch.randn(world * B, 16, generator=g)                         # every rank builds the same global batch
    mine = pdist.shard_batch({"x": glob}, rank, world)["x"]
    flat = mine.sum(0).repeat(64).contiguous()                             # this rank's "gradient"
    t0 = time.perf_counter()
    for _ in range(max(1, args.steps)):
        f = flat.clone()
        pdist.allreduce_mean_(f)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    want = glob.view(world, B, 16).sum(1).mean(0).repeat(64)
    err = float((f - want).abs().max())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        out = {"metric": "launcher dry run (no kernels; not a benchmark)", "value": None, "unit": None, "n_gpus": world,
               "ranks": world, "global_batch": world * B, "per_gpu_batch": B, "allreduce_max_abs_err": err,
               "seconds_max_over_ranks": float(tmax.item()), "dry_run": True}
        os.write(json_fd, (json.dumps(out) + "\n").encode())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
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
                    help="seconds the CPU-oracle leg may take for the metric's 1024 px step (measured: 70-90 s on the box)")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--force-collective", action="store_true", help="init RCCL and all-reduce even with one rank (path test)")
    ap.add_argument("--dump-prof", default="", help="write every profiled launch (shape-tagged) to this CSV")
    ap.add_argument("--breakdown", action="store_true", help="print the per-kernel-family table to stderr")
    ap.add_argument("--student", default="same", choices=["same", "ssd1b"],
                    help="BASELINE config 4: a smaller student UNet (SSD-1B-shaped, own weights) under the SDXL teacher")
    ap.add_argument("--with-vae", action="store_true",
                    help="also run the NEXT batch's VAE encode (train_sdxl_zh.py:306-309, 1024x1024 pixels) on a side "
                         "HIP stream inside every step; not the default: BASELINE's metric starts from latents (SURVEY 8d)")
    ap.add_argument("--single-stream", action="store_true", help="analysis only: teacher and student passes on ONE stream")
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        cpu_baseline_worker(args.model, args.cpu_budget)
        return

    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None:
        # `python bench.py --gpus N` without torch.distributed.run: start the N ranks here.  Nothing in this process has
        # touched the GPU (importing torch does not), and it never does: it only waits and relays rank 0's line.
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
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
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
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

    comm, collective = None, None
    if use_dist:
        from pea_diffusion_amd import dist as pdist
        if args.collective == "native":
            try:
                comm = pdist.NativeComm.from_env()            # ncclUniqueId from rank 0 through the torch group
                trainer.attach_comm(comm)
                collective = "RCCL all-reduce on the communicator's own HIP stream (libpea_hip.so pea_allreduce_grads)"
            except Exception as e:                            # still RCCL: torch.distributed's nccl backend
                print(f"bench.py: native communicator failed ({e}); using torch.distributed", file=sys.stderr)
        if comm is None:
            collective = "torch.distributed all_reduce (nccl backend = RCCL), async on ProcessGroupNCCL's stream"
        if comm is not None:
            comm.broadcast_(adapter.flat_param, 0)                # identical replicas
        else:
            pdist.broadcast_params_(adapter.flat_param, src=0)
        adapter.mark_updated()

    vae = None
    if args.with_vae:
        from pea_diffusion_amd.vae import HipVAEEncoder
        vae = HipVAEEncoder(pc.sdxl_vae_config() if args.model == "sdxl" else pc.tiny_vae_config(), B, hw * 8, hw * 8)
        vae.init_random(seed=11)
        pixels = torch.randn(B, 3, hw * 8, hw * 8, device=dev).clamp_(-1, 1)
        side = torch.cuda.Stream(device=dev)
        next_latents = [None]

    def step():
        if vae is not None:                   # next batch's latents on the side stream, overlapped with this step
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                next_latents[0] = vae.encode_latents(pixels)
        trainer.training_step(batch)          # ends by LAUNCHING the all-reduce of the flat adapter grad (comm stream)
        if args.force_collective and world == 1 and comm is None:
            dist.all_reduce(adapter.flat_grad)
        trainer.optimizer_step()              # joins the all-reduce, then fused AdamW
        if vae is not None:
            torch.cuda.current_stream().wait_stream(side)
            batch["latents"] = next_latents[0]

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if args.single_stream:
        lib().pea_trainer_set_option(trainer._h, b"two_stream", 0)
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ms = dt / args.steps * 1e3
    ips = world * B * args.steps / dt
    loss = float(trainer.losses[0])
    allreduce_ms = None
    if comm is not None:
        allreduce_ms = round(comm.last_ms(), 4)      # device time of the last step's all-reduce + 1/world scale (comm stream)
    rccl_ranks = dist.get_world_size() if use_dist else 1

    roof = None
    if not args.no_roofline and rank == 0:
        L = lib()
        L.pea_trainer_set_option(trainer._h, b"two_stream", 0)   # clean per-kernel durations (no cross-stream overlap)
        L.pea_prof_reset()
        L.pea_prof_enable(1)
        for _ in range(min(args.steps, 3)):
            step()
        torch.cuda.synchronize()
        L.pea_prof_enable(0)
        if args.dump_prof:
            L.pea_prof_dump(args.dump_prof.encode())
        L.pea_trainer_set_option(trainer._h, b"two_stream", 1)
        fams = []
        for f in range(8):
            t, fl, by, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_longlong()
            L.pea_prof_report(f, ctypes.byref(t), ctypes.byref(fl), ctypes.byref(by), ctypes.byref(n))
            fams.append(dict(name=L.pea_prof_family_name(f).decode(), ms=t.value, flops=fl.value, bytes=by.value,
                             launches=n.value))
        L.pea_prof_reset()
        nprof = min(args.steps, 3)
        gem = [fams[0], fams[1]]
        g_ms = sum(x["ms"] for x in gem)
        g_fl = sum(x["flops"] for x in gem)
        g_n = sum(x["launches"] for x in gem)
        ach = g_fl / (g_ms * 1e-3) / 1e12 if g_ms > 0 else 0.0
        roof = {"bound": "mfma", "kernel": "gemm_lc_kernel / gemm_lcp_kernel (plain + implicit-GEMM conv3x3)",
                "achieved": round(ach, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "traffic": pmc_traffic()[0],
                "traffic_unit": "bytes per launch, TCC FETCH_SIZE x2 (gfx950) + WRITE_SIZE over the family's launches, from "
                                f"two separate rocprofv3 --pmc passes of this command (profiles/{pmc_traffic()[1]})",
                "launches_per_step": g_n // nprof, "avg_launch_us": round(g_ms * 1e3 / max(g_n, 1), 2),
                "gflop_per_launch": round(g_fl / max(g_n, 1) / 1e9, 3),
                "ms_per_step_single_stream": round(g_ms / nprof, 2),
                "method": "hip events around every launch on the launch stream; instrumented single-stream replay of the timed "
                          "steps (the timed region overlaps teacher and student passes on two streams)"}
        if args.breakdown:
            tot = sum(x["ms"] for x in fams)
            for x in fams:
                tf = x["flops"] / (x["ms"] * 1e-3) / 1e12 if x["ms"] > 0 else 0
                gb = x["bytes"] / (x["ms"] * 1e-3) / 1e9 if x["ms"] > 0 else 0
                print(f"  {x['name']:28s} {x['ms'] / nprof:9.2f} ms/step {x['launches'] // nprof:6d} launches "
                      f"{tf:8.1f} TFLOP/s {gb:8.0f} GB/s(alg)", file=sys.stderr)
            print(f"  instrumented families total {tot / nprof:.2f} ms/step (single-stream replay) vs timed step {ms:.2f} ms",
                  file=sys.stderr)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.model, args.cpu_budget)

    if use_dist:
        dist.barrier()
        if comm is not None:
            comm.close()
        dist.destroy_process_group()
    if rank == 0:
        mem = student.memory()
        out = {
            "metric": "training images/sec (SDXL 1024px bf16, adapter-only bwd)", "value": round(ips, 4),
            "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
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
                       **({"vae_encode": f"next batch's {hw * 8}x{hw * 8} VAE encode on a side HIP stream inside every step"}
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
            "rccl_ranks": rccl_ranks, "collective": collective, "allreduce_ms": allreduce_ms,
            "allreduce_bytes": int(adapter.flat_grad.numel() * 4) if use_dist else 0,
            "roofline": roof, "cpu_baseline": cpu,
        }
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
