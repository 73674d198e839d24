#!/usr/bin/env python3
"""Headline benchmark: PM-VAE MNIST training images/sec (BASELINE.json metric), per-GPU batch 256.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one full optimizer step of train_pm_vae.py on configs/pm_vae_mnist.py: eps sampling,
masked-encoder + encoder + decoder forward, ELBO + posterior-matching loss, full backward, Adam.
Inputs are synthetic batches resident in HBM.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: peak FP32 (matrix), v_mfma_f32_32x32x2_f32
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (not the 2:1-sparsity figure)
HBM_PEAK_GBS = 8000.0
# SURVEY.md 8(d) algorithmic figures per training example of configs/pm_vae_mnist.py
F_ALG_EXECUTED = 563.8e6        # 6 x forward MACs with only the 30 consumed AR-GMM head columns per scan step computed
F_ALG_AS_WRITTEN = 609.5e6      # the same with all 960 head columns, as the reference states the scan
B_ALG_F32 = 2.25e6              # bytes per example: f32 activations written once + re-read once, weights, gradients, Adam


def _short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    if ">(" in name:
        return name.split(">(")[0] + ">"
    return name.split("(")[0]


def pmc_mfma_child(timeout_s=240):
    """One more PMC pass of the same serial run: SQ_VALU_MFMA_BUSY_CYCLES (cycles a SIMD's matrix pipe is busy, summed over
    the chip's 1024 SIMDs) and GRBM_GUI_ACTIVE (active cycles, summed over the 8 XCDs) per kernel ->
    {kernel: busy / (active / 8 * 1024)}: the share of the matrix pipes' cycles the kernel keeps busy while it runs."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3")
    if exe is None:
        return None
    tmp = tempfile.mkdtemp(prefix="pm_pmc_", dir="/tmp")
    try:
        env = dict(os.environ, TMPDIR="/tmp")
        cmd = [exe, "--pmc", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "--output-format", "csv", "-d", tmp, "--",
               sys.executable, os.path.join(ROOT, "bench.py"), "--serial", "--no-cpu-baseline", "--no-pmc", "--steps", "3",
               "--warmup", "2", "--profile-steps", "0", "--no-f32-aux", "--no-secondary", "--spread-steps", "0"]
        subprocess.run(cmd, cwd="/tmp", env=env, timeout=timeout_s, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                       check=True)
        files = glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            return None
        busy, act = {}, {}
        with open(files[-1]) as fp:
            for r in csv.DictReader(fp):
                n = _short(r["Kernel_Name"])
                if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
                    busy[n] = busy.get(n, 0.0) + float(r["Counter_Value"])
                elif r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                    act[n] = act.get(n, 0.0) + float(r["Counter_Value"])
        return {n: round(busy[n] / (act[n] / 8.0 * 1024.0), 4) for n in busy if act.get(n)}
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def pmc_traffic_child(counter, timeout_s=240):
    """One rocprofv3 PMC pass (its own run, one counter - MI355X_MICROARCH.md HBM section) over a short serial run of THIS
    script in a child process; must be called before this process touches the GPU.  -> {kernel: (sum KB, launches)}"""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3")
    if exe is None:
        return None
    tmp = tempfile.mkdtemp(prefix="pm_pmc_", dir="/tmp")
    try:
        env = dict(os.environ, TMPDIR="/tmp")
        cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", tmp, "--", sys.executable,
               os.path.join(ROOT, "bench.py"), "--serial", "--no-cpu-baseline", "--no-pmc", "--steps", "3", "--warmup", "2",
               "--profile-steps", "0", "--no-f32-aux", "--no-secondary", "--spread-steps", "0"]
        subprocess.run(cmd, cwd="/tmp", env=env, timeout=timeout_s, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                       check=True)
        files = glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            return None
        tot, disp = {}, {}
        with open(files[-1]) as fp:
            for r in csv.DictReader(fp):
                if r["Counter_Name"] != counter:
                    continue
                n = _short(r["Kernel_Name"])
                tot[n] = tot.get(n, 0.0) + float(r["Counter_Value"])
                disp.setdefault(n, set()).add(r["Dispatch_Id"])
        return {n: (tot[n], len(disp[n])) for n in tot}
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def measured_traffic():
    """HBM-side bytes per launch and kernel from two PMC passes of this run: 2 x FETCH_SIZE + WRITE_SIZE (counters in
    KB; FETCH_SIZE tallies the 128-B requests of 16-B-per-lane loads at 64 B on gfx950: doubled, as the guide prescribes)"""
    fetch = pmc_traffic_child("FETCH_SIZE")
    if fetch is None:
        return None
    write = pmc_traffic_child("WRITE_SIZE")
    if write is None:
        return None
    out = {}
    for n, (kb, launches) in fetch.items():
        wkb, wl = write.get(n, (0.0, 1))
        out[n] = int((2.0 * kb / max(launches, 1) + wkb / max(wl, 1)) * 1024)
    return out


def cpu_baseline(cfg, xs, B, budget_s=25.0):
    """The oracle's train step (torch CPU, float32) on the host cores: a reported baseline."""
    import torch

    from oracle import pm_vae_oracle as O

    p = O.init_params(cfg["model"], xs, seed=1, dtype=torch.float32)
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v = {k: torch.zeros_like(t) for k, t in p.items()}
    gen = torch.Generator().manual_seed(0)
    x = torch.rand((B,) + xs, generator=gen) * (torch.rand((B,) + xs, generator=gen) < 0.19)
    b = (torch.rand((B,) + xs, generator=gen) < 0.5).float()
    eps = torch.randn((B, cfg["model"]["latent_dim"]), generator=gen)
    O.train_step(p, m, v, cfg, x, b, eps, 0)                       # warm-up (two steps: thread pools, allocator)
    O.train_step(p, m, v, cfg, x, b, eps, 0)
    n, t0 = 0, time.perf_counter()
    while True:
        O.train_step(p, m, v, cfg, x, b, eps, n + 1)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or n >= 20:
            break
    return {"value": round(n * B / dt, 1), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} optimizer steps of the torch-CPU float32 oracle at batch {B} ({dt:.1f} s); the JAX "
                      "reference itself cannot run here (jax/haiku/tfp absent)"}


# BASELINE.json's global batches (configs 2, 4, 5; the rest: the config's per-device batch on one GPU) and per-GPU weak batches
GLOBAL_BATCH = {"pm_vae_mnist": 256, "pm_vae_gas": 128, "vqvae_mnist": 256, "pm_vqvae_mnist": 256, "pm_vdvae_mnist": 64,
                "pm_vqvae_celeb_a": 128}
WEAK_BATCH = {"pm_vae_mnist": 256, "pm_vae_gas": 128, "vqvae_mnist": 256, "pm_vqvae_mnist": 256, "pm_vdvae_mnist": 8,
              "pm_vqvae_celeb_a": 16}


def dist_info(world, dev):
    """what a reader needs to see that the collective library really ran on `world` ranks: communicator size, backend string and
    the device every rank sits on (all-gathered through the communicator itself)"""
    import torch
    import torch.distributed as dist

    if world <= 1 or not dist.is_initialized():
        return {"rccl_ranks": 1, "backend": None, "rank_devices": [str(dev)]}
    ids = [None] * dist.get_world_size()
    dist.all_gather_object(ids, f"{dev}:{torch.cuda.get_device_properties(dev).name}")
    return {"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(), "rank_devices": ids}


def run_workload(args, world, rank, dev, json_fd):
    """--workload / --scaling other than the headline's defaults: the same contract (W warm-up steps, K timed steps between
    barrier + synchronize on both sides, MAX over ranks, one JSON line from rank 0) on tools/workloads.build, the step the train
    script of that config runs.  Data parallel exactly as the headline: per-rank batch, bucketed all-reduce (parallel.GradReducer)."""
    import torch
    import torch.distributed as dist

    from tools.workloads import WORKLOADS, build

    name = args.workload
    if args.scaling == "strong":
        if GLOBAL_BATCH[name] % world:
            raise SystemExit(f"global batch {GLOBAL_BATCH[name]} of {name} is not divisible by {world} ranks")
        B = GLOBAL_BATCH[name] // world
    else:
        B = args.batch if args.batch != 256 or name == "pm_vae_mnist" else WEAK_BATCH[name]
    kw = {"overlap_allreduce": not args.no_overlap} if (world > 1 or os.environ.get("PM_FORCE_DP") == "1") else {}
    w = build(name, B, device=str(dev), f32=args.f32, world_size=world, rank=rank, **kw)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    from posterior_matching_amd.utils import steady_state_gc

    with steady_state_gc():
        w.feed()
        for _ in range(args.warmup):
            w.step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            w.step()
        fence()
        dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    met = w.ts.read_metrics()
    info = dist_info(world, dev)
    red = getattr(w.ts, "reducer", None)
    if rank == 0:
        value = world * B * args.steps / dt
        line = {"metric": f"training images/sec, {name} (configs/{WORKLOADS[name][0]}), full train step",
                "value": round(value, 1), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": args.scaling,
                "vs_baseline": None, "dtype": "f32" if args.f32 else "bf16x3", "data": "synthetic",
                "config": {"workload": f"configs/{WORKLOADS[name][0]}", "per_gpu_batch": B, "global_batch": B * world,
                           "parallelism": f"dp{world}", "params": w.params,
                           "allreduce": None if red is None else ("single" if args.no_overlap else "bucketed_overlapped"),
                           "bucket_bytes": getattr(red, "bucket_bytes", None),
                           "gradient_bytes": 4 * w.params, **info},
                "aux": {"loss": round(float(met["loss"]), 4),
                        "collectives_per_step": getattr(red, "calls_last_step", None),
                        "collectives_before_backward_end": getattr(red, "calls_before_finish_last_step", None),
                        "whole_step_tflops_per_gpu": round(WORKLOADS[name][2] * B / (dt / args.steps) / 1e12, 2)},
                "roofline": None, "cpu_baseline": None}
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (BASELINE: 256)")
    ap.add_argument("--graph", action="store_true", help="replay one captured HIP graph per step (single stream) "
                                                        "instead of eager launches on two overlapping streams")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=5, help="eager steps with per-kernel HIP events")
    ap.add_argument("--f32", action="store_true", help="every GEMM on the f32 MFMA (strict-parity mode) instead of the "
                                                      "default bf16x3 split on the bf16 matrix cores")
    ap.add_argument("--wgrad-streams", action="store_true", help="weight-gradient kernels on companion streams")
    ap.add_argument("--serial", action="store_true", help="one stream (clean per-kernel durations for rocprofv3)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse "
                                                      "the N>1 path with several ranks on ONE GPU)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 PMC child passes (roofline.traffic)")
    ap.add_argument("--no-f32-aux", action="store_true", help="skip the strict-f32 throughput sample (aux.f32_images_per_sec)")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: one all-reduce after the backward pass instead of "
                                                              "reverse-order buckets overlapped with it")
    ap.add_argument("--no-secondary", action="store_true", help="skip aux.secondary (a few timed steps of BASELINE.json's "
                                                                "other configs, outside the headline's timed region)")
    ap.add_argument("--spread-steps", type=int, default=200, help="extra untimed-for-the-headline steps whose per-step HIP-event "
                                                                  "times give aux.step_ms quartiles (0 = off)")
    ap.add_argument("--workload", default="pm_vae_mnist", choices=sorted(GLOBAL_BATCH),
                    help="which BASELINE.json config to time (default: the headline, configs/pm_vae_mnist.py)")
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"),
                    help="weak: the config's per-GPU batch on every rank; strong: BASELINE.json's fixed GLOBAL batch split over "
                         "the ranks (pm_vdvae_mnist 64, pm_vqvae_celeb_a 128: per-GPU 8 / 16 at N = 8)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` unaided: this parent never touches the GPU; it starts the N ranks as CHILD processes
        # through torch.distributed.run (never an exec of a process that has initialised the GPU) and passes their exit code
        # on.  Rank 0 of the children writes the JSON line to the inherited stdout.
        import socket
        import subprocess

        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd, env=env).returncode)

    # The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL prints a version banner when its first
    # communicator is created), so file descriptor 1 is pointed at stderr for the whole run and the line goes to the saved one.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    # HBM-side traffic per kernel: two PMC passes over a short run of this same workload, each in a child process
    # BEFORE this process initialises the GPU (counters cannot be read from inside the measured process)
    traffic, traffic_src = None, None
    headline = args.workload == "pm_vae_mnist" and args.scaling == "weak"
    mfma_busy = None
    if world == 1 and not args.no_pmc and args.profile_steps > 0 and headline:
        mfma_busy = pmc_mfma_child()
    if world == 1 and not args.no_pmc and args.profile_steps > 0 and headline:
        traffic = measured_traffic()
        traffic_src = "this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE child passes, 2*FETCH_SIZE + WRITE_SIZE per launch"
    if traffic is None and world == 1 and args.profile_steps > 0:
        for name in ("r03_final_pmc_traffic.json", "r02_final_pmc_traffic.json", "r01_final_pmc_traffic.json"):
            try:
                with open(os.path.join(ROOT, "profiles", name)) as fp:
                    traffic = {k: v.get("traffic_bytes") for k, v in json.load(fp).items()}
                traffic_src = f"profiles/{name} (committed PMC passes of the same command; rocprofv3 not runnable here)"
                break
            except OSError:
                pass

    import torch
    import torch.distributed as dist

    from posterior_matching_amd import ops, optim
    from posterior_matching_amd.config_dict import load_config_file
    from posterior_matching_amd.data import SyntheticDataset, data_shape
    from posterior_matching_amd.engine import PMVAETrainStep
    from posterior_matching_amd.models import PosteriorMatchingVAE

    ndev = torch.cuda.device_count()
    local_rank = local_rank % max(ndev, 1)        # gloo rehearsal: several ranks may share one GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
    elif os.environ.get("PM_FORCE_DP") == "1":
        # rehearsal of the N > 1 code path on ONE GPU: a 1-rank RCCL communicator, bucketed asynchronous all-reduce inside the
        # replayed launch plan (no multi-GPU node is available to the builder)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(args.backend, rank=0, world_size=1, **({"device_id": dev} if args.backend == "nccl" else {}))

    if args.workload != "pm_vae_mnist" or args.scaling == "strong":
        run_workload(args, world, rank, dev, json_fd)
        return

    cfg = load_config_file(os.path.join(ROOT, "configs", "pm_vae_mnist.py")).to_dict()   # the CLI's own config file
    xs, B = data_shape(cfg["data"]["dataset"]), args.batch

    def make_opt():
        return optim.chain(optim.scale_by_adam(), optim.add_decayed_weights(cfg.get("weight_decay", 0.0)),
                           optim.scale_by_schedule(optim.exponential_decay(**cfg["lr_schedule"])), optim.scale(-1.0))

    model = PosteriorMatchingVAE.from_config(cfg["model"], device=dev, seed=1)    # same init on every rank
    model.init(xs)
    model.store.use_bf16 = not args.f32
    model.concurrent = not args.serial
    model.ws.overlap_wgrad = args.wgrad_streams
    opt = make_opt()
    ts = PMVAETrainStep(model, cfg, opt, B, xs, seed=1234, world_size=world, rank=rank, use_graph=args.graph,
                        **({"overlap_allreduce": not args.no_overlap} if (world > 1 or os.environ.get("PM_FORCE_DP") == "1") else {}))
    pool = SyntheticDataset(cfg["data"], B, num_batches=16, seed=100 + rank, device=dev)
    batches = pool.batches

    def run(step_obj, n, offset=0, marks=None):
        for i in range(n):
            bt = batches[(offset + i) % len(batches)]
            step_obj.set_batch(bt["image"], bt["mask"])          # device-to-device copy of the resident batch
            step_obj.step()
            if marks is not None:
                with torch.cuda.stream(step_obj.stream):
                    ev = ops.Event()
                    ev.record()
                marks.append(ev)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # the engine's training loop (Trainer.fit) pauses the cyclic garbage collector in its steady state: the same here, from the
    # warm-up on (utils.steady_state_gc; PM_NO_GC_FREEZE=1 for A/B runs)
    from posterior_matching_amd.utils import steady_state_gc

    gc_guard = steady_state_gc()
    gc_guard.__enter__()
    run(ts, args.warmup)
    fence()
    marks = []
    with torch.cuda.stream(ts.stream):
        e0 = ops.Event()
        e0.record()
    t0 = time.perf_counter()
    run(ts, args.steps, args.warmup, marks)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    metrics = ts.read_metrics()
    # spread of the K timed steps (HIP events on the step's stream, one per step; the headline stays K steps / wall time)
    per_step = sorted(a.elapsed_ms(b) for a, b in zip([e0] + marks[:-1], marks))
    spread = {"min": round(per_step[0], 4), "median": round(per_step[len(per_step) // 2], 4),
              "max": round(per_step[-1], 4)} if per_step else None
    # The driver fixes --steps (20 steps = a 29 ms region): quartiles over `--spread-steps` further steps of the same loop,
    # one HIP event per step on the step's stream, back the A/B claims in DESIGN.md (outside the headline's timed region)
    if spread is not None and args.spread_steps > 0:
        marks2 = []
        with torch.cuda.stream(ts.stream):
            s0 = ops.Event()
            s0.record()
        run(ts, args.spread_steps, args.warmup + args.steps, marks2)
        fence()
        ps = sorted(a.elapsed_ms(b) for a, b in zip([s0] + marks2[:-1], marks2))
        q = lambda f: round(ps[min(len(ps) - 1, int(f * len(ps)))], 4)   # noqa: E731
        spread["extra_steps"] = {"n": len(ps), "min": q(0.0), "q1": q(0.25), "median": q(0.5), "q3": q(0.75), "max": round(ps[-1], 4),
                                 "mean": round(sum(ps) / len(ps), 4)}

    gc_guard.__exit__(None, None, None)
    # strict arithmetic (every GEMM on the f32 MFMA) on the same batches: a few steps OUTSIDE the timed region
    f32_ips = None
    if rank == 0 and world == 1 and not args.f32 and not args.no_f32_aux:
        m32 = PosteriorMatchingVAE.from_config(cfg["model"], device=dev, seed=1)
        m32.init(xs)
        m32.store.use_bf16 = False
        m32.concurrent = model.concurrent
        t32 = PMVAETrainStep(m32, cfg, make_opt(), B, xs, seed=1234, use_graph=False)
        run(t32, 5)
        torch.cuda.synchronize()
        n32, t1 = 20, time.perf_counter()
        run(t32, n32, 5)
        torch.cuda.synchronize()
        f32_ips = round(B * n32 / (time.perf_counter() - t1), 1)
        del t32, m32

    # live per-kernel timing with HIP events on the launching stream (eager, outside the timed region)
    roofline = None
    if rank == 0 and args.profile_steps > 0:
        model.concurrent = False        # per-kernel times: one stream, so durations do not overlap
        prof = PMVAETrainStep(model, cfg, opt, B, xs, seed=1234, world_size=1, rank=0, use_graph=False)
        prof.set_batch(batches[0]["image"], batches[0]["mask"])
        prof.step()
        prof.synchronize()
        timer = ops.KernelTimer()
        ops.set_timer(timer)
        for _ in range(args.profile_steps):
            prof.step()
        prof.synchronize()
        ops.set_timer(None)
        summ = timer.summary()
        total_ms = sum(r["ms"] for r in summ.values())

        def entry(name, r):
            avg_s = r["ms"] / r["calls"] * 1e-3
            e = {"kernel": name, "avg_launch_us": round(avg_s * 1e6, 2),
                 "launches_per_step": r["calls"] // args.profile_steps,
                 "share_of_kernel_time": round(r["ms"] / total_ms, 3),
                 "algorithmic_bytes_per_launch": round(r["bytes"] / r["calls"]),
                 "traffic": (traffic or {}).get(name),
                 # share of the chip's matrix-pipe cycles busy while the kernel runs (its own PMC pass; None: not collected)
                 "mfma_busy": (mfma_busy or {}).get(name)}
            if r["flops"] > 0:          # GEMM class: priced against the matrix-core peak of its arithmetic
                fl = r["flops"] / r["calls"]
                peak = BF16_MFMA_PEAK_TFLOPS if "bf16" in name else F32_MFMA_PEAK_TFLOPS
                e.update({"bound": "mfma", "achieved": round(fl / avg_s / 1e12, 2), "peak": peak, "unit": "TFLOP/s",
                          "frac": round(fl / avg_s / 1e12 / peak, 4), "algorithmic_gflop_per_launch": round(fl / 1e9, 3)})
            else:                       # row-wise / optimizer class: operand bytes against HBM
                gbs = r["bytes"] / r["calls"] / avg_s / 1e9
                e.update({"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": round(gbs / HBM_PEAK_GBS, 4)})
            return e

        ranked = sorted(summ.items(), key=lambda kv: -kv[1]["ms"])
        roofline = entry(*ranked[0])                       # the dominant kernel of the step, by time
        roofline["note"] = ("achieved = ALGORITHMIC (float32-equivalent) flops of valid (position, tap) MACs / HIP-event "
                            "time on the launching stream; bf16x3 kernels issue 3 bf16 MFMA products per algorithmic MAC, "
                            "so their matrix pipe runs at 3x this rate")
        roofline["traffic_source"] = traffic_src
        roofline["top3"] = [entry(n, r) for n, r in ranked[:3]]
        roofline["hbm_class"] = [entry(n, r) for n, r in ranked if r["flops"] == 0 and r["bytes"] >= 4e6][:4]
        roofline["kernel_time_per_step_ms"] = round(total_ms / args.profile_steps, 4)
        roofline["total_launches_per_step"] = sum(r["calls"] for r in summ.values()) // args.profile_steps
        if os.environ.get("PM_BENCH_KERNEL_TABLE"):
            with open(os.environ["PM_BENCH_KERNEL_TABLE"], "w") as fp:
                for k_, v_ in ranked:
                    fp.write(f"{k_:45s} calls/step {v_['calls'] // args.profile_steps:3d}  "
                             f"ms/step {v_['ms'] / args.profile_steps:8.4f}  "
                             f"TFLOP/s {(v_['flops'] / (v_['ms'] * 1e-3) / 1e12) if v_['flops'] else 0:7.2f}  "
                             f"GB/s {(v_['bytes'] / (v_['ms'] * 1e-3) / 1e9):8.1f}  "
                             f"traffic MB/launch {((traffic or {}).get(k_) or 0) / 1e6:8.2f}\n")
                fp.write(f"sum of kernel time per step: {total_ms / args.profile_steps:.4f} ms\n\n")
                calls = timer.per_call()
                per_step_n = len(calls) // args.profile_steps
                fp.write("launch order of the last profiled step:\n")
                for tag, detail, ms, fl in calls[-per_step_n:]:
                    fp.write(f"{tag:42s} {detail:52s} {ms * 1e3:9.1f} us  {fl / (ms * 1e-3) / 1e12 if fl else 0:6.2f} TFLOP/s\n")

    # BASELINE.json's other configs on the driver's clock: a few timed steps each, after (outside) the headline's region
    secondary = None
    if rank == 0 and world == 1 and not args.no_secondary and not args.f32:
        from tools.workloads import build, measure

        secondary = []
        for name, bsz, n_steps, n_warm in (("pm_vae_gas", 128, 200, 20), ("vqvae_mnist", 256, 200, 20),
                                           ("pm_vqvae_mnist", 256, 20, 4), ("pm_vdvae_mnist", 8, 20, 4),
                                           ("pm_vdvae_mnist", 16, 20, 4), ("pm_vqvae_celeb_a", 16, 20, 4)):
            try:
                w = build(name, bsz, device=str(dev))
                secondary.append(measure(w, n_steps, n_warm))
                del w
            except Exception as exc:       # a secondary number must never take the headline down
                secondary.append({"workload": name, "per_gpu_batch": bsz, "error": repr(exc)[:200]})
            torch.cuda.empty_cache()

    info = dist_info(world, dev)               # a collective: every rank
    if rank == 0:
        value = world * B * args.steps / dt
        per_gpu = value / world
        line = {
            "metric": "training images/sec, PM-VAE MNIST bs=256 per GPU (ELBO / PM matching-LL in `aux`)",
            "value": round(value, 1), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16x3" if model.store.use_bf16 else "f32", "data": "synthetic",
            "config": {"workload": "configs/pm_vae_mnist.py: conv PM-VAE 28x28x1, latent 32, TriL posterior, "
                                   "AR-GMM partial posterior, Bernoulli decoder; full train step (fwd+loss+bwd+Adam)",
                       "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}",
                       "gemm_arithmetic": ("bf16x3 on the bf16 matrix cores for forward, data- and weight-gradient GEMMs: "
                                           "f32 operands split hi+lo bf16, 3 bf16-MFMA products, f32 accumulate; 1- and "
                                           "2-channel layers on the VALU in f32") if model.store.use_bf16 else "f32 MFMA",
                       "launch": "hip_graph_1stream" if args.graph else ("eager_1stream" if args.serial else "launch_plan_2streams"),
                       "allreduce": (None if world == 1 else ("single" if args.no_overlap else "bucketed_overlapped")),
                       "params": model.num_params, **info},
            "aux": {"elbo": round(metrics["reconstruction_ll"] - metrics["beta"] * metrics["kl"], 4),
                    "matching_ll": round(metrics["matching_ll"], 4), "kl": round(metrics["kl"], 4),
                    "loss": round(metrics["loss"], 4), "step_ms": spread, "f32_images_per_sec": f32_ips,
                    "secondary": secondary,
                    # whole-step fractions SURVEY.md 8(d) asks for, per GPU
                    "mfma_frac": round(per_gpu * F_ALG_EXECUTED / (BF16_MFMA_PEAK_TFLOPS * 1e12), 4),
                    "mfma_frac_as_reference_states_ar_gmm": round(per_gpu * F_ALG_AS_WRITTEN / (BF16_MFMA_PEAK_TFLOPS * 1e12), 4),
                    "hbm_frac": round(per_gpu * B_ALG_F32 / (HBM_PEAK_GBS * 1e9), 4)},
            "roofline": roofline,
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(cfg, xs, B)
        else:
            line["cpu_baseline"] = None
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
