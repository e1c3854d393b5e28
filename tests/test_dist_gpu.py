"""-m gpu: the data-parallel collective on the real RCCL (row a11 / 8(e)).  The box has ONE GPU, so the communicators are
world-size 1 -- enough to run ncclCommInitRank, the all-reduce on the communicator's own HIP stream, the event hand-offs
between the compute and comm streams, the broadcast and torch.distributed's nccl backend; the 2-rank arithmetic is
covered over gloo in tests/test_dp_cpu.py and the N-rank run is the driver's SCALE bench."""
import ctypes
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu
from test_model_gpu import _train_pair, gpu  # noqa: E402,F401


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_native_comm_world1_allreduce_broadcast_streams(gpu):
    """C ABI pea_comm_unique_id / pea_comm_init / pea_allreduce_grads / pea_comm_join / pea_comm_broadcast through ctypes"""
    from pea_diffusion_amd.dist import NativeComm
    comm = NativeComm.from_env()
    assert comm.world == 1 and comm.rank == 0
    g = torch.randn(6_033_408, device="cuda")             # the 6M adapter's flat gradient (24 MB)
    ref = g.clone()
    side = torch.cuda.Stream()
    # the producer runs on a side stream: the comm stream must wait for it through the event, not through a device sync
    with torch.cuda.stream(side):
        g.mul_(3.0)
        comm.allreduce_mean_async(g, compute_stream=side)
    comm.join()                                            # current stream waits for the comm stream
    out = g.clone()
    torch.cuda.synchronize()
    assert torch.equal(out, ref * 3.0)                     # sum over one rank, x 1/1
    assert comm.last_ms() > 0.0
    p = torch.randn(1000, device="cuda")
    q = p.clone()
    comm.broadcast_(p, 0)
    torch.cuda.synchronize()
    assert torch.equal(p, q)
    comm.close()


def test_comm_rendezvous_timeout_returns_instead_of_hanging(gpu):
    """pea_comm_init_timeout with a rank that never arrives (world 2, only rank 0 present): PEA_E_TIMEOUT after the
    deadline, with a message naming the rank and the world -- ncclCommInitRank itself would block forever.  In a child
    process: the abandoned rendezvous thread stays inside RCCL's bootstrap, which is why the caller must exit."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import ctypes, os, sys, time; sys.path.insert(0, %r)\n"
            "import torch; torch.cuda.set_device(0); torch.zeros(1, device='cuda')\n"
            "from pea_diffusion_amd._lib import lib\n"
            "from pea_diffusion_amd.dist import NativeComm, CommTimeout\n"
            "uid = NativeComm.new_unique_id()\n"
            "t0 = time.time()\n"
            "try:\n"
            "    NativeComm(0, 2, uid, timeout_s=4.0)\n"
            "except CommTimeout as e:\n"
            "    print('TIMEOUT %%.1f %%s' %% (time.time() - t0, e), flush=True); os._exit(5)\n"
            "os._exit(0)\n" % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 5, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("TIMEOUT")][0]
    waited = float(line.split()[1])
    assert 3.5 <= waited < 30.0 and "rendezvous of 2 ranks not complete" in line and "rank 0" in line


def test_comm_argument_errors(gpu):
    from pea_diffusion_amd._lib import lib
    L = lib()
    h = ctypes.c_void_p()
    assert L.pea_comm_init(2, 2, b"\0" * 128, ctypes.byref(h)) == -3 and b"rank 2 of 2" in L.pea_last_error()
    assert L.pea_allreduce_grads(None, None, 4, None) == -1
    assert L.pea_comm_join(None, None) == -1


def test_trainer_step_with_rccl_world1_matches_plain_step(gpu):
    """the step with the all-reduce launched on the comm stream and joined by optimizer_step() == the step without it
    (world 1: the all-reduce is the identity), bit for bit; then the torch.distributed nccl path (all_reduce +
    broadcast_params_) on the same flat buffers."""
    import torch.distributed as dist
    from pea_diffusion_amd import dist as pdist
    cfg, us, ut, ad_ref, ad_hip, hs, ht, batch, tr = _train_pair(2, 12)
    tr.training_step(batch, 0, sync=True)
    g0 = ad_hip.flat_grad.clone()
    comm = pdist.NativeComm.from_env()
    tr.attach_comm(comm)
    tr.training_step(batch, 0, async_allreduce=True)       # only LAUNCHED on the comm stream ...
    assert tr._comm_inflight
    tr.join_grads()                                        # ... the current stream waits for it here
    torch.cuda.synchronize()
    assert torch.equal(g0, ad_hip.flat_grad) and not tr._comm_inflight
    tr.training_step(batch, 0)                             # default: gradient complete (joined) on return
    assert not tr._comm_inflight
    torch.cuda.synchronize()
    assert torch.equal(g0, ad_hip.flat_grad)
    tr.training_step(batch, 0, async_allreduce=True)
    tr.training_step(batch, 0, async_allreduce=True)       # a second step joins the first collective before it rewrites flat_grad
    tr.join_grads()
    torch.cuda.synchronize()
    assert torch.equal(g0, ad_hip.flat_grad)
    assert comm.last_ms() >= 0.0 and comm.last_exposed_ms() >= 0.0
    w0 = ad_hip.flat_param.clone()
    tr.lr, tr.warmup_steps = 1e-3, 0
    tr.training_step(batch, 0, async_allreduce=True)
    tr.optimizer_step()                                    # joins the comm stream before AdamW reads the gradient
    torch.cuda.synchronize()
    assert not torch.equal(w0, ad_hip.flat_param)
    tr.comm = None
    comm.close()
    # torch.distributed over RCCL (backend "nccl"), one rank
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    try:
        g = ad_hip.flat_grad.clone()
        dist.all_reduce(ad_hip.flat_grad)
        pdist.broadcast_params_(ad_hip.flat_param, src=0)
        c2 = pdist.NativeComm.from_env()                   # unique id shipped through the torch group's object broadcast
        c2.allreduce_mean_async(ad_hip.flat_grad)
        c2.join()
        torch.cuda.synchronize()
        assert torch.equal(g, ad_hip.flat_grad)
        c2.close()
    finally:
        dist.destroy_process_group()
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
            os.environ.pop(k, None)


def test_bench_subprocess_with_collective_and_side_stream_vae(gpu):
    """`bench.py --gpus 1 --force-collective --with-vae --model tiny` as a subprocess: the N > 1 code path on one GPU --
    RCCL communicator inside libpea_hip.so, all-reduce launched by training_step, a later batch's VAE encode enqueued
    on the side stream between training_step and optimizer_step, the instrumented replay, one JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["MASTER_PORT"] = str(_free_port())
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-collective", "--with-vae",
                        "--model", "tiny", "--steps", "6", "--warmup", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["rccl_ranks"] == 1 and "libpea_hip.so" in out["collective"]
    # round 6: the control plane is gloo -- the library's communicator is the only RCCL communicator of the process -- and the
    # N-rank schema is the one the CPU dry run asserts (bench.NRANK_KEYS)
    sys.path.insert(0, root)
    import bench
    assert out["control_plane"] == "gloo" and all(k in out for k in bench.NRANK_KEYS)
    assert out["config"]["loss"] == out["config"]["loss"] and abs(out["config"]["loss"]) < 1e4      # finite
    assert out["allreduce_ms"] is not None and out["allreduce_exposed_ms"] is not None
    assert out["allreduce_bytes"] > 0 and "vae_encode" in out["config"]
    fams = {f["family"]: f for f in out["roofline"]["families"]}
    assert any(k.startswith("gemm") for k in fams) and "attn_bwd" in fams and "kd_loss" in fams
    assert out["ms_per_step"] > 0 and out["ms_per_step_mean"] > 0
    # round 4: the single-GPU-equivalent step (collective switched off) measured in the same run, the HBM plan, the MFMA probe
    assert out["single_gpu_equivalent"]["ms_per_step"] > 0 and out["hbm_plan_gb_per_rank"] > 0
    sp = out["roofline"]["sustained_peak"]
    assert sp["tflops"] > 500 and 800 < sp["in_kernel_clock_mhz"] <= 2500 and out["roofline"]["frac_of_sustained"] > 0
    assert isinstance(out["roofline"]["adapter_golden_rel_l2"]["measured_in_this_run"], list)
