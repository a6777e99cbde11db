#!/usr/bin/env python3
"""Benchmark of the hot path: images/s of the TT-small forward on MI355X.

    python bench.py [--gpus N --steps K --warmup W] [--batch B]

A step is one forward of ``--batch`` (default 256, BASELINE.json configs[1]) synthetic
224x224 images per GPU, already resident in HBM, plus -- for N > 1 -- the all-gather of the
logits (RCCL).  One process per GPU (torchrun env), batch sharded by image, weak scaling
(per-GPU batch fixed).  Rank 0 prints ONE JSON line with the driver's contract fields plus

  roofline          the dominant kernel by device time: algorithmic flops (or bytes) per
                    launch / its average launch duration, measured with HIP events on the
                    launch stream in a second pass of K steps (the event pairs would
                    perturb the throughput pass); peaks from MI355X_MICROARCH.md
  roofline_kernels  the same for every kernel, and for the gate (LUT) path as a whole
                    against HBM with SURVEY 8(d)'s algorithmic bytes 74,592*B + 14,155,776
  cpu_baseline      the oracle's float-mode restatement of the reference forward
                    (oracle/ttnet_float.py, torch-CPU, all host threads) timed on a bounded
                    sample on rank 0 at N = 1 -- a reported baseline, not the target.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from argparse import Namespace

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

from scale_imagenet_amd import synth, ttnet
from scale_imagenet_amd.dist import all_gather_logits, init_from_env
from scale_imagenet_amd.spec import make_spec

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
F32_PEAK_TFLOPS = 157.3      # fp32 matrix == fp32 vector peak; exact-f32 MFMA
F16_PEAK_TFLOPS = 2500.0     # dense fp16/bf16 MFMA; stem and lin1 issue 3 fp16 products per f32 product (fp16 x 2 split)

# algorithmic work per image; MACs -> 2 flops
STEM_MAC, LIN1_MAC, LIN2_MAC = 29_503_488, 16_384_000, 1_000_000
GATE_BYTES_PER_IMAGE, GATE_TABLE_BYTES = 74_592, 14_155_776


def kernel_models(batch: int):
    """name -> (bound, algorithmic units per launch).  Gate kernels: packed input + output
    bytes of that launch + its tables once (the unfused per-layer accounting of SURVEY 8(d))."""
    m = {
        "stem": ("mfma_f16x2", 2.0 * STEM_MAC * batch),
        "head.lin1": ("mfma_f16x2", 2.0 * LIN1_MAC * batch),
        "head.lin2": ("mfma_f16x2", 2.0 * LIN2_MAC * batch),
        "head.bn_poly": ("hbm", 8.0 * 1000 * batch),
        "head.bias": ("hbm", 8.0 * 1000 * batch),
    }
    c, h = 64, 56
    for i, tag in enumerate(("f4", "f5", "f6")):
        ho = h // 2 + 1
        plane_in, plane_out = c * h * h / 8.0, c * ho * ho / 8.0
        # stage 1 = Block_conv1/2 (rows in, 2 word planes out, 2*C tables of 8 KiB) +
        #           Block_conv3 and both majorities (words in, 2 word planes out, C/16 tables of 128 KiB)
        m[f"gate_stage1.{tag}"] = ("hbm", batch * (2 * plane_in + 4 * plane_out) + 2 * c * 8192 + (c // 16) * 131072)
        if i < 2:
            # 4 branch planes in, the next block's input out in both layouts, C/4 tables of 64 KiB
            m[f"gate_pf.{tag}"] = ("hbm", batch * (4 * plane_out + 2 * 2 * plane_out) + (c // 4) * 65536)
        else:
            # 4 branch tensors in, one 64-byte table row per (group, pixel), pooled floats out
            m["gate_last"] = ("hbm", batch * (4 * plane_out + (c // 4) * ho * ho * 64 + 4 * c * 16 * 4))
        c, h = 2 * c, ho
    return m


def measured_traffic(batch: int):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/*traffic_b<B>.json,
    collected with separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of this script and corrected
    as MI355X_MICROARCH.md prescribes).  bench.py cannot run the profiler on itself, so this is
    the last committed measurement for this batch size, or nothing."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*traffic_b{batch}.json")))
    if not files:
        return {}, None
    with open(files[-1]) as f:
        d = json.load(f)
    return {k: v["hbm_bytes"] for k, v in d.get("kernels", {}).items()}, os.path.basename(files[-1])


def host_cores() -> int:
    """CPU threads this process may really use: affinity mask capped by the cgroup quota
    (the GPU box exposes 256 logical CPUs but grants a 1-GPU job a 16-CPU share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("TTNET_CPU_THREADS", "16"))))


def cpu_baseline(spec, st, budget_s: float = 15.0):
    """The oracle's float-mode forward (the reference's op sequence, incl. its inert
    randint_like and clones) on all host threads, bounded sample."""
    from oracle import ttnet_float as OF
    cores = host_cores()
    torch.set_num_threads(cores)
    sd = OF.to_torch_state(st)
    bs = 32
    x = torch.from_numpy(synth.synth_images(bs))
    OF.forward(x, sd, spec)                       # warm-up
    t0 = time.perf_counter()
    done = 0
    while True:
        OF.forward(x, sd, spec)
        done += bs
        el = time.perf_counter() - t0
        if el >= budget_s or done >= 16 * bs:
            break
    return {"value": done / el, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{done} images as batches of {bs} (same synthetic generator and weights), "
                      f"oracle/ttnet_float.py = the reference's eval op sequence on torch-CPU "
                      f"{torch.__version__}, {cores} threads, {el:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU per step")
    ap.add_argument("--inflight", type=int, default=2,
                    help="batches kept in flight (model lanes on separate HIP streams, as evaluate.py runs the "
                         "eval loop); 1 = one batch at a time. The serial figure is reported alongside.")
    ap.add_argument("--input", default="f32", choices=["f32", "u8"],
                    help="f32 = the reference's contract (normalised float32 NCHW, the headline); u8 = uint8 HWC images "
                         "with ToTensor + Normalize fused into the stem (SURVEY 8f N1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--variant", default="small", choices=["small", "xsmall", "full", "valexnet"],
                    help="small = BASELINE.json configs[1] (the headline); the others are parity-test configs")
    args = ap.parse_args()

    rank, world, local_rank = init_from_env("nccl")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    assert torch.cuda.is_available(), "bench.py needs a HIP device (the product has no CPU path)"
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())   # (% only matters for the 1-GPU gloo rehearsal)
    torch.cuda.set_device(dev)

    vargs = dict(nfilter=6, tfilter=10) if args.variant == "full" else dict(nfilter=8, tfilter=8)
    if args.variant == "valexnet":
        from scale_imagenet_amd.spec import VAlexSpec
        spec = VAlexSpec()
    else:
        spec = make_spec(args.variant, **vargs)
    st = synth.synth_state_dict(spec)
    cls = {"small": ttnet.TT_vf_19lv3_imgnet_small, "xsmall": ttnet.TT_vf_19lv3_imgnet_xsmall,
           "full": ttnet.TT_vf_19lv3_imgnet, "valexnet": ttnet.TT_FHE_XSMALL_vAlexnet}[args.variant]
    model = cls(Namespace(layers=1, groups=[1, None, 4, None], **vargs))
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()}, strict=True)
    model = model.to(dev).eval().reserve(args.batch)

    B = args.batch
    n_total = B * world
    if args.input == "u8":
        if args.variant == "valexnet":
            raise SystemExit("--input u8 is not available for the vAlexnet variant")
        u8 = synth.synth_images_u8(B, first=rank * B, hw=spec.image_hw)
        x = torch.from_numpy(np.ascontiguousarray(u8.transpose(0, 2, 3, 1))).to(dev)             # uint8 [B,H,W,3], resident in HBM
        fwd = model.forward_u8
    else:
        x = torch.from_numpy(synth.synth_images(B, first=rank * B, hw=spec.image_hw)).to(dev)  # resident in HBM
        fwd = model.forward

    R = max(1, args.inflight)
    if R > 1:
        model.set_lanes(R)
    streams = [torch.cuda.Stream(dev) for _ in range(R)]

    def step(i=0, lanes=1):
        """One step = one forward of one batch (+ the logits all-gather when world > 1).  With
        lanes > 1, step i runs on lane i % lanes and its own stream, so consecutive steps overlap."""
        with torch.no_grad():
            if lanes == 1:
                y = fwd(x)
                return all_gather_logits(y, n_total) if world > 1 else y
            lane = i % lanes
            with torch.cuda.stream(streams[lane]):
                y = fwd(x, lane=lane)
                return all_gather_logits(y, n_total) if world > 1 else y

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed(lanes):
        for i in range(max(args.warmup, 3 * lanes)):     # (3 calls per lane: two plain, then the graph capture)
            step(i, lanes)
        fence()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i, lanes)
        fence()
        el = time.perf_counter() - t0
        if world > 1:
            gloo = dist.get_backend() == "gloo"
            t = torch.tensor([el], device="cpu" if gloo else dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    elapsed_serial = timed(1)
    elapsed = timed(R) if R > 1 else elapsed_serial

    # second pass: per-kernel device time with HIP events on the launch stream
    model.set_profiling(True)
    acc = {}
    prof_steps = min(args.steps, 30)
    for _ in range(prof_steps):
        step()
        for k, v in model.last_timings().items():
            acc[k] = acc.get(k, 0.0) + v
    model.set_profiling(False)
    avg_ms = {k: v / prof_steps for k, v in acc.items()}

    if rank == 0:
        models = kernel_models(B) if args.variant == "small" else {}
        traffic, traffic_src = measured_traffic(B) if args.variant == "small" else ({}, None)
        kernels = []
        for k, ms in avg_ms.items():
            bound, units = models.get(k, ("hbm", 0.0))
            if bound == "mfma_f16x2":
                # algorithmic f32 flops against the 16-bit dense peak; the kernel issues 3 fp16 MFMA
                # flops per algorithmic flop (operands split 2 x fp16), so frac <= 1/3
                bound, ach, peak, unit = "mfma", units / (ms * 1e-3) / 1e12, F16_PEAK_TFLOPS, "TFLOP/s"
            elif bound == "mfma":
                ach, peak, unit = units / (ms * 1e-3) / 1e12, F32_PEAK_TFLOPS, "TFLOP/s"
            else:
                ach, peak, unit = units / (ms * 1e-3) / 1e9, HBM_PEAK_GBS, "GB/s"
            rec = {"kernel": k, "ms": round(ms, 5), "bound": bound, "achieved": round(ach, 3),
                   "peak": peak, "unit": unit, "frac": round(ach / peak, 5), "traffic": traffic.get(k)}
            if peak == F16_PEAK_TFLOPS:
                # MFMA flops the kernel really issues per algorithmic flop: 3 products of the 2-way
                # fp16 split (stem: x 8/7 for the kw 7 -> 8 padding, x 22/21 for the k-row padding)
                issued = 3.0 * (8.0 / 7.0) * (22.0 / 21.0) if k == "stem" else 3.0
                rec["mfma_flops_issued_per_flop"] = round(issued, 3)
                rec["frac_issued"] = round(ach * issued / peak, 5)
            kernels.append(rec)
        gate_ms = sum(ms for k, ms in avg_ms.items() if k.startswith(("gate_stage1", "gate_pf"))) or 1e-9
        gate_bytes = GATE_BYTES_PER_IMAGE * B + GATE_TABLE_BYTES
        if args.variant != "small":
            gate_bytes = 0.0
        gate = {"kernel": "gate_path (all binarised LUT launches)", "ms": round(gate_ms, 5), "bound": "hbm",
                "achieved": round(gate_bytes / (gate_ms * 1e-3) / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(gate_bytes / (gate_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": None}
        dom = max(kernels, key=lambda r: r["ms"])
        roofline = {k: dom[k] for k in dom if k not in ("kernel", "ms")}
        roofline["kernel"] = dom["kernel"]
        roofline["ms"] = dom["ms"]
        out = {
            "metric": (f"images/sec ImageNet 224x224, TT-{args.variant}, MI355X; top-1 exact-match"
                       if args.variant != "valexnet" else "images/sec CIFAR 32x32, TT vAlexnet variant, MI355X"),
            "value": round(n_total * args.steps / elapsed, 2),
            "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u16/u64 packed bits (gate path); f32 as prescaled fp16x2 split operands on the 16-bit MFMA (stem, lin1, lin2)",
            "data": "synthetic",
            "config": {"workload": (f"TT_general_imagenet_v2_small forward, batch={B} 224x224 per GPU, "
                                    f"bit-packed HIP LUT kernels (BASELINE.json configs[1])") if args.variant == "small"
                       else f"TT {args.variant} variant forward, batch={B} 224x224 per GPU (parity-test configuration)",
                       "batch_per_gpu": B, "global_batch": n_total, "batches_in_flight": R,
                       "input": "float32 NCHW, normalised (the reference's contract)" if args.input == "f32"
                       else "uint8 HWC, ToTensor + Normalize fused into the stem",
                       "parallelism": f"batch shard x{world}" + (" + RCCL all-gather of logits" if world > 1 else "")},
            "inflight": R,
            "serial": {"value": round(n_total * args.steps / elapsed_serial, 2),
                       "ms_per_step": round(1e3 * elapsed_serial / args.steps, 4),
                       "note": "the same K steps with one batch in flight (every step waits for the previous one)"},
            "roofline": roofline,
            "roofline_kernels": kernels + [gate],
            "traffic_source": traffic_src,
            "kernel_ms_sum": round(sum(avg_ms.values()), 5),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(spec, st) if args.variant != "valexnet" else None
            if out["cpu_baseline"]:
                out["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
