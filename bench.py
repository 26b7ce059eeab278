#!/usr/bin/env python3
"""SHMGAN train_step benchmark on MI355X (BASELINE.json metric: images/sec of one full
generator+discriminator step at 256x256, batch 8 per GPU, fp32).

    python bench.py --gpus N --steps K --warmup W

N>1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py
--gpus N ...`: one process per GPU, RCCL all-reduce of the D and G gradients (weak scaling: the
per-GPU batch is fixed).  Rank 0 prints ONE JSON line.

A "step" = pre-processing + 6 generator forwards + 12 discriminator forwards + both backward
passes + clip + 2 Adam updates (+ all-reduce) on one synthetic batch already resident in HBM.
`roofline` is measured live with HIP events bracketing every MFMA conv launch on its stream; the
dominant kernel symbol's algorithmic FLOPs / its summed duration is `achieved`.  Because the step
runs weight gradients on a second stream (kernel lifetimes overlap, so a per-kernel duration is not
a property of the kernel any more), the event pass is a serialized replay of the same K steps in
the same process right after the timed region (`roofline.region`); `--serialize` runs the timed
region itself without the second stream, which is the command the committed rocprofv3 summaries
(profiles/) were taken with.
`cpu_baseline` times the CPU oracle (PyTorch-CPU/oneDNN restatement of the same step, fp32, with clip+Adam) on
this box's host cores on a bounded sample: BASELINE configs[0]'s batch of 2, median of 3 steps after one warm-up.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

MFMA_F32_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: dense f32 MFMA peak
MFMA_BF16_PEAK_TFLOPS = 2500.0    # same guide: dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16, 1024 FLOP/clk/SIMD)
HBM_PEAK_GBS = 8000.0             # same guide: HBM3E 8 TB/s (6.29 TB/s is what a float4 copy reaches)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8, help="per-GPU batch (BASELINE: 8)")
    ap.add_argument("--image-size", type=int, default=256)
    ap.add_argument("--filter-size", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the bf16 configs[3] / configs[4] runs behind the fp32 headline")
    ap.add_argument("--serialize", action="store_true", help="no second stream for weight gradients (profiling)")
    ap.add_argument("--cpu-steps", type=int, default=3, help="timed CPU-baseline steps (median), after one warm-up")
    ap.add_argument("--per-shape", type=str, default="", help="write per-(kernel,shape) timings to this JSON file")
    ap.add_argument("--backend", type=str, default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsal)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal only: all ranks share GPU 0")
    ap.add_argument("--grad-dtype", type=str, default=None, choices=["float32", "bfloat16"],
                    help="bf16 mode only: element type of the gradient-signal tensors (float32 = SHM_BF16_GF32)")
    ap.add_argument("--dtype", type=str, default="f32", choices=["f32", "bf16", "f32x3"],
                    help="f32 = BASELINE configs[1] (default, the headline line: exact-fp32 MFMA); bf16 = configs[3]/[4] (bf16 MFMA path); f32x3 = fp32 "
                         "tensors with the 3x3 unit-stride convolutions as six bf16 MFMA products of exact three-plane splits (opt-in: wgrad.f32_split, conv.f32_split)")
    args = ap.parse_args()

    # Before torch is imported or any torch.cuda function runs (device_count() may already bring HSA up, and HSA reads this at
    # initialisation): the host driver only supports dmabuf IPC, without which RCCL fails with hipIpcGetMemHandle: invalid argument
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")

    # --gpus N is the contract: N ranks, one per GPU.  Launched by the driver through torch.distributed.run the environment
    # carries WORLD_SIZE = N; launched as plain `python bench.py --gpus N` this process only becomes the launcher: it starts
    # the N ranks as children (before anything here touches the GPU: no exec from a GPU process) and relays their output.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a line for the wrong rank count",
              file=sys.stderr, flush=True)
        return 2

    import numpy as np
    import torch
    import torch.distributed as dist

    t_start = time.perf_counter()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not args.same_device and torch.cuda.device_count() < world:
        print(f"bench.py: {world} ranks but only {torch.cuda.device_count()} GPUs visible (use --same-device --backend gloo "
              "to rehearse on one GPU)", file=sys.stderr, flush=True)
        return 2
    if world > 1:
        from shmgan_amd.dist import init_process_group
        if args.same_device:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        init_process_group(args.backend, device=torch.device("cuda", local_rank))      # finite timeout: a dead rank ends the job
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    from shmgan_amd import ShmGANwithSSpecSeg, ops

    if args.dtype == "f32x3":
        ops.set_tuning("wgrad.f32_split", 1)
        ops.set_tuning("conv.f32_split", 1)
    S, F, B = args.image_size, args.filter_size, args.batch
    model = ShmGANwithSSpecSeg(image_size=S, filter_size=F, batch_size=B, device=dev,
                               compute_dtype="bfloat16" if args.dtype == "bf16" else "float32",          # f32x3: fp32 tensors
                               grad_dtype=args.grad_dtype).build()
    peak = MFMA_BF16_PEAK_TFLOPS if args.dtype == "bf16" else MFMA_F32_PEAK_TFLOPS        # (f32x3: fp32-equivalent FLOPs against the f32 pipe's peak)

    # synthetic inputs, resident in HBM before the timed region
    rng = np.random.default_rng(1234 + rank)
    inputs = [torch.from_numpy(rng.random((B, S, S, 3), dtype=np.float32)).to(dev) for _ in range(5)]
    s = S // 32

    class Draws:
        pass

    noise_buf = torch.empty((2 * B, S, S, 3), device=dev)
    keep_buf = torch.empty((2 * B, s, s, 16 * F), device=dev)

    def draws_for(step):
        r = np.random.default_rng(7 + step)              # flags / TARGET_LABELS shared by all ranks
        d = Draws()
        d.flags = tuple(bool(u < 0.5) for u in r.random(5))
        d.target_label = float(r.uniform(0.8, 1.2))
        # GaussianNoise / Dropout draws: the library's Philox kernels, keyed by (step, rank) -- part of the timed step
        ops.randn(noise_buf, 0.1, 7 + step, 2 * rank)
        ops.keep_mask(keep_buf, 0.2, 7 + step, 2 * rank + 1)
        d.noise, d.keep_mask = noise_buf, keep_buf
        return d

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def note(msg):
        if rank == 0:
            print(f"[bench +{time.perf_counter() - t_start:.1f}s] {msg}", file=sys.stderr, flush=True)

    if world > 1:            # bring the RCCL communicator up outside any timed region, whatever --warmup is
        dist.all_reduce(torch.zeros(1, device=dev))
        torch.cuda.synchronize()
    lane = model._get_lane()
    lane_stream = lane.stream
    if args.serialize:
        lane.stream = None
    note(f"model built, arena {model.arena.nbytes() / 2**30:.2f} GiB")
    # every step hands the following batch (here: the same resident tensors) to train_step's look-ahead, as the training
    # loop does: its weight-independent prologue is issued under this step's last gradient collective
    for i in range(args.warmup):
        model.train_step(*inputs, draws=draws_for(i), next_batch=inputs)
        torch.cuda.synchronize()
        note(f"warm-up step {i} done")
    sync()
    if world > 1:            # N > 1: bracket every collective (reducer stream) and every wait on one (main stream) with timing events
        model._reducer.probe = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        model.train_step(*inputs, draws=draws_for(args.warmup + i), next_batch=inputs)
    sync()
    dt = time.perf_counter() - t0
    note(f"timed {args.steps} steps in {dt:.3f}s")
    comm = None
    if world > 1:
        comm = model._reducer.comm_summary(args.steps)
        model._reducer.probe = None
    # per-kernel HIP-event pass: serialized replay (single stream) of the same number of steps
    timer = None
    dt_serial = 0.0
    if not args.no_kernel_timer:               # every rank replays (the steps contain collectives)
        lane.stream = None
        if rank == 0:
            timer = ops.KernelTimer()
            ops.TIMER = timer
        t1 = time.perf_counter()
        for i in range(args.steps):
            model.train_step(*inputs, draws=draws_for(args.warmup + args.steps + i), next_batch=inputs)
        torch.cuda.synchronize()
        dt_serial = time.perf_counter() - t1
        ops.TIMER = None
        lane.stream = lane_stream
        note(f"serialized event replay: {args.steps} steps in {dt_serial:.3f}s")
    sync()
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    losses = model.losses()
    finite = all(np.isfinite(v) for k, v in losses.items() if k != "ssim")

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * B * args.steps / dt
        out = {
            "metric": "images/sec (gen+disc train_step)", "value": round(value, 3), "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"SHMGAN train_step {S}x{S} 5-view, batch {B}/GPU, filter_size {F}, "
                                   + {"f32": "fp32", "bf16": "bf16 operands / fp32 accumulate",
                                      "f32x3": "fp32 tensors, 3x3 unit-stride convolutions (forward, input gradient, weight gradient) from six bf16 MFMA products of three-plane splits"}[args.dtype]
                                   + (" (BASELINE configs[1])" if (S, B, F, args.dtype) == (256, 8, 64, "f32") else "")
                                   + (" (BASELINE configs[3])" if (S, B, F, args.dtype) == (512, 4, 64, "bf16") else "")
                                   + (" (BASELINE configs[4], per GPU)" if (S, B, F, args.dtype) == (256, 32, 64, "bf16") else ""),
                       "global_batch": world * B, "image_size": S, "parallelism": f"dp{world}",
                       "losses_finite": bool(finite)},
            # what actually ran: ranks in the process group, its backend, GPUs visible to rank 0
            "rccl_ranks": {"world_size": dist.get_world_size() if world > 1 else 1,
                           "backend": dist.get_backend() if world > 1 else None,
                           "device_count": torch.cuda.device_count(), "same_device": bool(args.same_device)},
        }
        if comm is not None:
            # why the N-GPU number is what it is: per step on rank 0, the time its collectives took on the reducer stream (D bucket /
            # the G buckets), the bytes they reduced, and the part of it the main stream actually waited for (`exposed_ms`: timing
            # events either side of the waits in front of clip+Adam) -- everything else ran under the generator backward
            comm["exposed_frac_of_step"] = round(comm["exposed_ms"] / ms, 5)
            out["comm"] = comm
        if timer is not None:
            summ = timer.summary()
            # the dominant MFMA kernel by time (entries without FLOPs -- the slab reduce -- are bandwidth kernels; under a host-staged
            # rehearsal backend their event brackets also absorb collective stalls)
            dom = max((k for k in summ if summ[k]["flops"] > 0), key=lambda k: summ[k]["ms"])
            d = summ[dom]
            ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
            traffic = None
            # HBM-side bytes are PMC counters: they need rocprofv3 --pmc passes of this same command, which cannot run inside the
            # timed process -- the figure is read from the committed distillate of those passes (tools/profile_round.sh ->
            # tools/pmc_traffic.py, profiles/README.md); `traffic_source` names the file
            tags = ((["r06_f32x3", "r05_f32x3"] if args.dtype == "f32x3" else []) + ["r06_f32", "r05_f32", "r04_f32", "r03_f32", "r02_f32"] if args.dtype != "bf16" else
                    (["r06_s512_b4_bf16", "r05_s512_b4_bf16", "r04_s512_b4_bf16", "r03_s512_b4_bf16", "r02_s512_b4_bf16"] if (S, B) == (512, 4) else
                     ["r06_b32_bf16", "r05_b32_bf16"] if (S, B) == (256, 32) else ["r06_bf16", "r05_bf16", "r04_bf16", "r03_bf16", "r02_bf16"]))
            tpath = next((ROOT / "profiles" / f"{t}_traffic_pmc.json" for t in tags if (ROOT / "profiles" / f"{t}_traffic_pmc.json").exists()),
                         ROOT / "profiles" / "none")
            if tpath.exists():
                tj = json.loads(tpath.read_text())
                key = dom.split(" (+")[0].replace(" ", "")

                def same_symbol(prof):
                    """shm_last_kernel() spells a symbol without its trailing default template arguments, rocprofv3 with all of them"""
                    b = prof.replace(" ", "")
                    if b == key:
                        return True
                    base = key[:-1] + "," if key.endswith(">") else key + "<"
                    return b.startswith(base) and all(t in ("false", "0", "float") for t in b[len(base):].rstrip(">").split(","))
                for name, rec in tj.items():
                    if same_symbol(name):
                        traffic = rec["hbm_bytes_per_launch"]
            # f32x3: a six-product kernel issues six bf16 MFMA FLOPs per algorithmic fp32 FLOP -- priced on the pipe it runs on
            x3_dom = args.dtype == "f32x3" and "x3" in dom
            if x3_dom:
                out_extra_x3 = {"fp32_equivalent_tflops": round(ach, 2), "mfma_flops_per_fp32_flop": 6}
                ach, peak = ach * 6.0, MFMA_BF16_PEAK_TFLOPS
            out["roofline"] = {
                "bound": "mfma", "kernel": dom, "achieved": round(ach, 2), "peak": peak,
                "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic,
                "traffic_source": (f"profiles/{tpath.name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, not measured "
                                   "in this run)") if traffic is not None else None,
                "region": "serialized replay of the timed steps (single stream), same process",
                "replay_ms_per_step": round(dt_serial / args.steps * 1e3, 3),
                "whole_step_conv_tflops": round(sum(v["flops"] for v in summ.values()) / args.steps / (ms * 1e-3) / 1e12, 2),
                "launches": d["launches"], "avg_launch_ms": round(d["ms"] / d["launches"], 4),
                "flops_per_launch": d["flops"] / d["launches"],
                "kernels": {k: {"launches": v["launches"], "ms_per_step": round(v["ms"] / args.steps, 3),
                                "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2)} for k, v in summ.items()},
            }
            if x3_dom:
                out["roofline"].update(out_extra_x3)
                out["roofline"]["note"] = ("achieved / peak: bf16 MFMA FLOPs issued (6 per algorithmic fp32 FLOP) against the dense bf16 peak; "
                                           "kernels[*].tflops and whole_step_conv_tflops stay fp32-equivalent")
            # the HBM-bound side of the step (north star: ">= 70 % of the memory-bandwidth roofline"): the entry point with the most
            # time among the elementwise passes, algorithmic bytes (every tensor read or written once) over its HIP-event time
            bs = timer.bytes_summary()
            if bs:
                hdom = max(bs, key=lambda k: bs[k]["ms"])
                hb = bs[hdom]
                gbs = hb["bytes"] / (hb["ms"] * 1e-3) / 1e9
                out["roofline_hbm"] = {
                    "bound": "hbm", "kernel": hdom, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(gbs / HBM_PEAK_GBS, 4), "launches": hb["launches"], "ms_per_step": round(hb["ms"] / args.steps, 3),
                    "bytes_per_launch": hb["bytes"] / hb["launches"], "region": "serialized replay, HIP events around the entry point",
                    "passes": {k: {"launches": v["launches"], "ms_per_step": round(v["ms"] / args.steps, 3),
                                   "GBps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)} for k, v in bs.items()},
                }
            if args.per_shape:
                ps = timer.per_shape()
                rows = [dict(kernel=k[0], shape=k[1], launches=v["launches"], ms_per_step=v["ms"] / args.steps,
                             tflops=v["flops"] / (v["ms"] * 1e-3) / 1e12) for k, v in ps.items()]
                rows.sort(key=lambda r: -r["ms_per_step"])
                Path(args.per_shape).write_text(json.dumps(rows, indent=1))
        if world == 1 and args.dtype == "f32" and not args.no_extra_configs:
            # the opt-in fp32-from-bf16-planes weight gradient (csrc/conv_wgrad_x3.hip) on the same model, same process:
            # the headline above stays the exact-fp32 MFMA step
            ops.set_tuning("wgrad.f32_split", 1)
            ops.set_tuning("conv.f32_split", 1)
            for i in range(2):
                model.train_step(*inputs, draws=draws_for(1000 + i), next_batch=inputs)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for i in range(args.steps):
                model.train_step(*inputs, draws=draws_for(1002 + i), next_batch=inputs)
            torch.cuda.synchronize()
            dt3 = time.perf_counter() - t2
            ops.set_tuning("wgrad.f32_split", 0)
            ops.set_tuning("conv.f32_split", 0)
            out["extra"] = {"f32x3": {"ms_per_step": round(dt3 / args.steps * 1e3, 3), "value": round(B * args.steps / dt3, 3), "unit": "images/sec",
                                      "steps": args.steps, "warmup": 2,
                                      "arithmetic": "3xbf16 planes (exact truncation split of every fp32 operand), 6 products with i + j <= 2, fp32 accumulate; "
                                                    "3x3 unit-stride layers: weight gradients (wgrad_halo_x3_kernel) and the forward / input-gradient products (tapgemm_halo_x3_kernel: "
                                                    "128-channel blocks, and 64-channel blocks on maps whose height is a multiple of 32); stride-2 / transposed / 1x1 / first layers "
                                                    "exact-fp32 MFMA",
                                      "losses_finite": bool(all(np.isfinite(v) for k, v in model.losses().items() if k != "ssim"))}}
            note(f"f32x3 (opt-in): {dt3 / args.steps * 1e3:.2f} ms/step")
        if not args.no_kernel_timer and world == 1:      # a single-GPU property; at N > 1 the other ranks would wait behind it
            out["north_star_block"] = north_star_block(torch, ops, dev)
        if world == 1 and not args.no_extra_configs and (S, B, F, args.dtype) == (256, 8, 64, "f32"):
            # BASELINE configs[3] and [4] (per GPU) on the same clock as the headline: two more trainers in this process, one after the
            # other, each after the previous arena has been freed
            del inputs, noise_buf, keep_buf
            model.release()
            del model, lane
            torch.cuda.empty_cache()
            out["extra_configs"] = [extra_config(torch, np, ops, dev, ShmGANwithSSpecSeg, F, *cfg, note)
                                    for cfg in ((512, 4, "BASELINE configs[3]"), (256, 32, "BASELINE configs[4], per GPU"))]
        if not args.no_cpu_baseline and world == 1:         # reported at N=1 only: the other ranks would idle behind it
            out["cpu_baseline"] = cpu_baseline(S, F, args.cpu_steps, note)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


def extra_config(torch, np, ops, dev, Model, F, S, B, tag, note, steps=10, warmup=3, replay=3):
    """One of BASELINE.json's bf16 configurations, timed as the headline is: `warmup` untimed steps, `steps` timed steps of the production
    two-stream step between synchronisations, then a serialized replay of `replay` steps under the HIP-event kernel timer for the dominant
    MFMA kernel's roofline fraction."""
    model = Model(image_size=S, filter_size=F, batch_size=B, device=dev, compute_dtype="bfloat16").build()
    rng = np.random.default_rng(1234)
    inputs = [torch.from_numpy(rng.random((B, S, S, 3), dtype=np.float32)).to(dev) for _ in range(5)]
    s = S // 32
    noise_buf = torch.empty((2 * B, S, S, 3), device=dev)
    keep_buf = torch.empty((2 * B, s, s, 16 * F), device=dev)

    class Draws:
        pass

    def draws_for(step):
        r = np.random.default_rng(7 + step)
        d = Draws()
        d.flags = tuple(bool(u < 0.5) for u in r.random(5))
        d.target_label = float(r.uniform(0.8, 1.2))
        ops.randn(noise_buf, 0.1, 7 + step, 0)
        ops.keep_mask(keep_buf, 0.2, 7 + step, 1)
        d.noise, d.keep_mask = noise_buf, keep_buf
        return d
    for i in range(warmup):
        model.train_step(*inputs, draws=draws_for(i), next_batch=inputs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        model.train_step(*inputs, draws=draws_for(warmup + i), next_batch=inputs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms = dt / steps * 1e3
    lane = model._get_lane()
    lane_stream, lane.stream = lane.stream, None
    timer = ops.KernelTimer()
    ops.TIMER = timer
    for i in range(replay):
        model.train_step(*inputs, draws=draws_for(warmup + steps + i), next_batch=inputs)
    torch.cuda.synchronize()
    ops.TIMER = None
    lane.stream = lane_stream
    summ = timer.summary()
    dom = max((k for k in summ if summ[k]["flops"] > 0), key=lambda k: summ[k]["ms"])
    ach = summ[dom]["flops"] / (summ[dom]["ms"] * 1e-3) / 1e12
    finite = all(np.isfinite(v) for k, v in model.losses().items() if k != "ssim")
    res = {"workload": f"SHMGAN train_step {S}x{S} 5-view, batch {B}/GPU, filter_size {F}, bf16 operands / fp32 accumulate ({tag})",
           "dtype": "bf16", "steps": steps, "warmup": warmup, "ms_per_step": round(ms, 3), "value": round(B * steps / dt, 3), "unit": "images/sec",
           "roofline": {"bound": "mfma", "kernel": dom, "achieved": round(ach, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4), "ms_per_step": round(summ[dom]["ms"] / replay, 3),
                        "region": f"serialized replay of {replay} steps, same process"},
           "whole_step_conv_tflops": round(sum(v["flops"] for v in summ.values()) / replay / (ms * 1e-3) / 1e12, 1),
           "losses_finite": bool(finite)}
    note(f"extra config {tag}: {ms:.2f} ms/step")
    model.release()
    del model, lane, inputs, noise_buf, keep_buf
    torch.cuda.empty_cache()
    return res


def north_star_block(torch, ops, dev, n=40, h=256, c=64, reps=20):
    """The block BASELINE.json's north star names -- fused 3x3 conv + bias + LeakyReLU + InstanceNorm statistics, 64 -> 64 channels at
    256 x 256, in bf16, at the batch the step runs it with (n = 5 x 8 images of the cyclic generator pass) -- timed on its own after the
    timed region: `reps` launches of shm_conv2d_in_fwd between two HIP events on the launch stream.  bytes = algorithmic traffic of one
    launch (e * N*H*W*(Cin + Cout) + e * 9*Cin*Cout + 16 * N*Cout, e = 2), frac_hbm against the 8 TB/s HBM3E peak."""
    dt = torch.bfloat16
    x = torch.randn((n, h, h, c), device=dev).to(dt)
    w = (torch.randn((3, 3, c, c), device=dev) * 0.05)
    wk = torch.zeros(9 * c * c, device=dev, dtype=dt)
    ops.transpose_taps(w, wk, 9, c, c, c)
    b = torch.randn(c, device=dev)
    y = torch.empty((n, h, h, c), device=dev, dtype=dt)
    stats = torch.empty(n * c * 2, dtype=torch.float64, device=dev)
    scr = torch.zeros(ops.STATS_SLOTS * n * c * 2, dtype=torch.float64, device=dev)
    fn = lambda: ops.conv2d_in_fwd(x, None, 0, c, 0, wk, b, y, c, n, h, h, c, c, 3, 1, 0.2, stats, 1e-6, scratch=scr)
    for _ in range(3):
        fn()
    kernel = ops.last_kernel()
    clk = torch.zeros(2, dtype=torch.int64, device=dev)          # the clock the product kernel holds, from its own counters (shm_set_clock_probe)
    ops.set_clock_probe(clk)
    try:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
    finally:
        ops.set_clock_probe(None)
    us = e0.elapsed_time(e1) / reps * 1e3
    ck = clk.tolist()
    kernel_clock = ck[0] / ck[1] * 0.1 if ck[1] > 0 else None          # GHz: shader-clock ticks per 10 ns tick of the last launch's patch loop
    nbytes = 2 * n * h * h * (c + c) + 2 * 9 * c * c + 16 * n * c
    gbps = nbytes / us / 1e3
    flops = 2.0 * n * h * h * 9 * c * c
    out = {"block": f"conv3x3 {c}->{c} + bias + LeakyReLU + IN statistics, {h}x{h}, n={n}, bf16", "kernel": kernel, "bytes": nbytes,
           "us": round(us, 2), "GBps": round(gbps, 1), "frac_hbm": round(gbps / HBM_PEAK_GBS, 4), "launches": reps,
           "flops": flops, "tflops": round(flops / us / 1e6, 1)}
    out["kernel_clock_ghz"] = None if kernel_clock is None else round(kernel_clock, 3)
    out.update(north_star_ceiling(torch, dev, flops, 2 * n * h * h * c, us, reps, kernel_clock))
    return out


def north_star_ceiling(torch, dev, flops, tensor_bytes, us, reps, kernel_clock=None):
    """What the chip can do on the block's two resources taken one at a time, in this process, right after the product kernel
    (tools/probes/ceiling_ns_block.hip): the block's FLOPs as a bare v_mfma_f32_16x16x32_bf16 loop on random operands in registers (the
    product's launch geometry: 512 eight-wave blocks, four waves per SIMD) and the block's bytes as a 16-byte-per-lane streaming copy.
    clock_ghz: d(s_memtime) / d(s_memrealtime) x 100 MHz around the MFMA loop.  Round 6: the copy's shape is calibrated
    once on a 2 GB buffer (threads per block, loads in flight, blocks per CU, plain / non-temporal) and the best shape copies the block's bytes;
    `mfma_us_at_sustained_clock` = the bare loop's cycles at the clock the PRODUCT kernel held in this run (its own s_memtime / s_memrealtime,
    shm_set_clock_probe) -- a bare MFMA loop runs hotter than the chip allows the product's mix of HBM stream + LDS + MFMA;
    ceiling_us = the largest of the three; frac_of_ceiling = ceiling_us / us."""
    import ctypes as C
    lib_path = ROOT / "tools" / "probes" / "libceiling_ns_block.so"
    if not lib_path.exists():
        import __graft_entry__ as ge
        ge.build_probes()
    L = C.CDLL(str(lib_path))
    L.ceil_mfma_bf16.restype = C.c_int
    L.ceil_mfma_bf16.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.ceil_copy.restype = C.c_int
    L.ceil_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    ncu = torch.cuda.get_device_properties(dev).multi_processor_count
    blocks = 2 * ncu
    waves = blocks * 8
    per_wave = int(flops / 16384 / waves) // 8 * 8              # 16 x 16 x 32 x 2 FLOP per MFMA
    ops_ = torch.randn(4 * 64 * 8, device=dev).to(torch.bfloat16)
    sink = torch.empty(blocks * 512, device=dev)
    stamps = torch.zeros(2 * waves, dtype=torch.int64, device=dev)
    src = torch.randn(tensor_bytes // 4, device=dev)
    dst = torch.empty_like(src)
    st = torch.cuda.current_stream().cuda_stream

    def timed(fn):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    def mfma():
        if L.ceil_mfma_bf16(ops_.data_ptr(), sink.data_ptr(), stamps.data_ptr(), blocks, per_wave, st) < 0:
            raise RuntimeError("ceiling probe: MFMA launch failed")

    L.ceil_copy_cfg.restype = C.c_int
    L.ceil_copy_cfg.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]

    def copy_cfg(s_, d_, nbytes, cfg):
        bpc, threads, unroll, nt = cfg
        if L.ceil_copy_cfg(s_.data_ptr(), d_.data_ptr(), nbytes, bpc * ncu, threads, unroll, nt, st) != 0:
            raise RuntimeError(f"ceiling probe: copy launch failed {cfg}")
    # calibration: 1 GB read + 1 GB written per launch (far beyond the 256 MB Infinity Cache), every shape, best of two rounds
    cal_bytes = 1 << 30
    cal_src = torch.empty(cal_bytes // 4, device=dev).normal_()
    cal_dst = torch.empty_like(cal_src)
    cal = {}
    for cfg in [(bpc, th, un, nt) for th in (256, 512, 1024) for un in (4, 8) for bpc in (8, 16, 32) for nt in (1, 0)]:
        best = 0.0
        for _ in range(2):
            copy_cfg(cal_src, cal_dst, cal_bytes, cfg)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                copy_cfg(cal_src, cal_dst, cal_bytes, cfg)
            e1.record()
            torch.cuda.synchronize()
            best = max(best, 2 * cal_bytes * 3 / (e0.elapsed_time(e1) * 1e-3) / 1e9)
        cal[cfg] = best
    del cal_src, cal_dst
    best_cfg = max(cal, key=cal.get)

    def copy():
        copy_cfg(src, dst, tensor_bytes, best_cfg)

    def copy_r5():
        if L.ceil_copy(src.data_ptr(), dst.data_ptr(), tensor_bytes, 16 * ncu, st) != 0:
            raise RuntimeError("ceiling probe: copy launch failed")
    mfma_us = timed(mfma) * (flops / (per_wave * waves * 16384.0))          # scaled to the block's exact FLOPs (rounding of per_wave)
    sv = stamps.view(-1, 2).double()
    clock = float((sv[:, 0] / sv[:, 1].clamp(min=1)).median()) * 0.1            # cycles per 10 ns tick -> GHz
    # the same FLOPs in the ping-pong kernel's shape and geometry: v_mfma_f32_32x32x16_bf16, one eight-wave block per CU (two waves per SIMD)
    L.ceil_mfma32_bf16.restype = C.c_int
    L.ceil_mfma32_bf16.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    waves32 = ncu * 8
    per_wave32 = int(flops / 32768 / waves32) // 8 * 8

    def mfma32():
        if L.ceil_mfma32_bf16(ops_.data_ptr(), sink.data_ptr(), stamps.data_ptr(), ncu, per_wave32, st) < 0:
            raise RuntimeError("ceiling probe: MFMA (32x32x16) launch failed")
    mfma32_us = timed(mfma32) * (flops / (per_wave32 * waves32 * 32768.0))
    sv32 = stamps.view(-1, 2)[:waves32].double()
    clock32 = float((sv32[:, 0] / sv32[:, 1].clamp(min=1)).median()) * 0.1
    copy_us = timed(copy)
    copy_r5_us = timed(copy_r5)
    sustained_us = mfma_us * clock / kernel_clock if kernel_clock else None
    ceil_us = max(mfma_us, copy_us, sustained_us or 0.0)
    return {"ceiling": {"mfma_us": round(mfma_us, 2), "mfma_tflops": round(flops / mfma_us / 1e6, 1), "mfma_clock_ghz": round(clock, 3),
                        "mfma_us_at_sustained_clock": None if sustained_us is None else round(sustained_us, 2),
                        "mfma_32x32x16_us": round(mfma32_us, 2), "mfma_32x32x16_clock_ghz": round(clock32, 3),
                        "copy_us": round(copy_us, 2), "copy_GBps": round(2 * tensor_bytes / copy_us / 1e3, 1),
                        "copy_shape": {"blocks_per_cu": best_cfg[0], "threads": best_cfg[1], "loads_in_flight": best_cfg[2], "nontemporal": bool(best_cfg[3])},
                        "copy_calibration_GBps_2GB": {"best": round(cal[best_cfg], 1), "worst": round(min(cal.values()), 1),
                                                      "round5_shape_16x256x4_nt": round(cal[(16, 256, 4, 1)], 1)},
                        "copy_round5_shape_us": round(copy_r5_us, 2),
                        "what": "bare 16x16x32 bf16 MFMA loop with the block's FLOPs (random operands in registers, 4 waves per SIMD), the same "
                                "cycles at the clock the product kernel held, and a 16-byte streaming copy of the block's activation bytes in the "
                                "best of 36 calibrated shapes; same process, after the product kernel"},
            "ceiling_us": round(ceil_us, 2), "frac_of_ceiling": round(ceil_us / us, 4)}


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: run the N ranks as children of this (GPU-free) process through
    torch.distributed.run, exactly as the driver does, and pass rank 0's JSON line and the exit code through."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    print(f"[bench] launching {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def cpu_baseline(S, F, steps, note=lambda m: None):
    """The oracle (PyTorch-CPU / oneDNN restatement of the step) in fp32 on the host cores, as BASELINE.md section 3 /
    BASELINE.json configs[0] specify it: batch 2, forward + both gradient passes + clip + Adam on both models,
    median of `steps` (3) timed steps after one warm-up step."""
    import statistics
    import torch
    from oracle import step_torch as st
    # the GPU box gives one GPU's share of the host (16 cores); more threads than that only oversubscribe
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 16)))
    B = 2
    g, d, gb, db = st.init_params(F, S)
    gv = [torch.from_numpy(a).clone() for a in g]
    dv = [torch.from_numpy(a).clone() for a in d]
    sg = st.AdamState([torch.zeros_like(a) for a in gv], [torch.zeros_like(a) for a in gv])
    sd = st.AdamState([torch.zeros_like(a) for a in dv], [torch.zeros_like(a) for a in dv])
    inp = st.make_inputs(B, S)
    sf = st.style_factor_intended(S)
    times = []
    for i in range(steps + 1):
        t0 = time.perf_counter()
        r = st.train_step(gv, dv, gb, db, inp, st.make_draws(i, B, S, F), sf, F, dtype=torch.float32)
        st.adam_apply(dv, r["gD"], sd, 2e-5, 0.5, 0.99)
        st.adam_apply(gv, r["gG"], sg, 2e-5, 0.5, 0.99)
        dt = time.perf_counter() - t0
        if i > 0:
            times.append(dt)
        note(f"cpu baseline step {i} ({'warm-up' if i == 0 else 'timed'}): {dt:.2f}s")
    med = statistics.median(times)
    return {"value": round(B / med, 4), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"median of {steps} steps of B={B} at {S}x{S}, fp32, forward + both gradient passes + clip + Adam on G and D, "
                      "after 1 warm-up step"}


if __name__ == "__main__":
    sys.exit(main())
