#!/usr/bin/env python3
"""Benchmark of the hot path: images/s of the TT-small forward on MI355X.

    python bench.py [--gpus N --steps K --warmup W] [--batch B] [--variant small|xsmall|full|valexnet]

A step is one forward of ``--batch`` (default 256, BASELINE.json configs[1]) synthetic
224x224 images per GPU, already resident in HBM, plus -- for N > 1 -- the all-gather of the
logits (RCCL).  One process per GPU, batch sharded by image, weak scaling (per-GPU batch fixed).
``--gpus N`` with N > 1 works both under ``torch.distributed.run`` (RANK / WORLD_SIZE in the
environment) and as a plain ``python bench.py --gpus N``: in the second case this process -- before
it touches the GPU -- starts the N ranks itself (scale_imagenet_amd/launch.py) and relays rank 0's
line.  Rank 0 prints ONE JSON line with the driver's contract fields plus

  roofline          the dominant kernel by device time: algorithmic flops (or bytes) per
                    launch / its average launch duration, measured with HIP events on the
                    launch stream in a second pass of K steps (the event pairs would
                    perturb the throughput pass); peaks from MI355X_MICROARCH.md
  roofline_kernels  the same for every kernel, and for the gate (LUT) path as a whole
                    against HBM with SURVEY 8(d)'s algorithmic bytes 74,592*B + 14,155,776
  gate_path         that gate-path figure at B = 256 and at B = 2048 (TT-small, N = 1)
  parity            max |logit - reference| and top-1 agreement on the committed golden images
                    (tests/golden/ref_<variant>.npz: the imported reference's own outputs)
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

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
F32_PEAK_TFLOPS = 157.3      # fp32 matrix == fp32 vector peak; exact-f32 MFMA
F16_PEAK_TFLOPS = 2500.0     # dense fp16/bf16 MFMA; stem and lin1 issue 3 fp16 products per f32 product (fp16 x 2 split)
F64_PEAK_TFLOPS = 78.6       # AMD's MI355X datasheet figure for FP64 (vector = matrix); the guide lists none

# algorithmic work per image of TT-small; MACs -> 2 flops
GATE_BYTES_PER_IMAGE, GATE_TABLE_BYTES = 74_592, 14_155_776


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=None, help="images per GPU per step (default: the variant's BASELINE batch)")
    ap.add_argument("--inflight", type=int, default=2,
                    help="batches kept in flight (model lanes on separate HIP streams, as evaluate.py runs the "
                         "eval loop); 1 = one batch at a time. The serial figure is reported alongside.")
    ap.add_argument("--input", default="f32", choices=["f32", "u8"],
                    help="f32 = the reference's contract (normalised float32 NCHW, the headline); u8 = uint8 HWC images "
                         "with ToTensor + Normalize fused into the stem (SURVEY 8f N1)")
    ap.add_argument("--inputs", type=int, default=4,
                    help="distinct device-resident input batches rotated through the timed loop (default 4: 616 MB of "
                         "float32 at batch 256, more than the 256 MiB Infinity Cache, so the stem's input stream really "
                         "comes from HBM; 1 = the same buffer every step)")
    ap.add_argument("--windows", type=int, default=0,
                    help="timed windows of --steps steps each (0 = 9 when --steps <= 50 -- a 20-step window is 3 ms, and the clock is still "
                         "settling during the first ones --, else 3); value = the median window, min / max beside it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the B = 2048 gate-path pass and the golden-image parity check (profiling runs)")
    ap.add_argument("--verify-gather", action="store_true",
                    help="N > 1: rank 0 also runs every rank's shard itself and requires the gathered logits to be identical")
    ap.add_argument("--variant", default="small", choices=["small", "xsmall", "full", "valexnet"],
                    help="small = BASELINE.json configs[1] (the headline); full = configs[2] (batch 512); "
                         "valexnet = configs[4] (batch 256); xsmall = parity-test configuration")
    return ap.parse_args(argv)


def variant_spec(variant):
    from scale_imagenet_amd.spec import VAlexSpec, make_spec
    vargs = dict(nfilter=6, tfilter=10) if variant == "full" else dict(nfilter=8, tfilter=8)
    spec = VAlexSpec() if variant == "valexnet" else make_spec(variant, **vargs)
    return spec, vargs


def kernel_models(variant: str, spec, batch: int):
    """name -> (bound, algorithmic units per launch).  Float stages: 2 x MACs.  Gate kernels: packed
    input + output bytes of that launch + its tables once (the unfused per-layer accounting of
    SURVEY 8(d)); launches of the full variant: the flops of their two convolutions (2 x MACs; the GELU and
    BatchNorm, which set the time of the float32 evaluation, are not counted), priced against the pipe the fast
    path runs them on: the 16-bit matrix cores for the grouped 1x1 blocks, float32 vector for the depthwise ones."""
    m = {}
    if variant == "valexnet":
        m["va.stem"] = ("mfma", 2.0 * 64 * 27 * 30 * 30 * batch)              # f32 VALU conv 3x3 on the 30x30 pooled window
        m["head.lin1"] = ("mfma_f16x2", 2.0 * spec.fcsize * spec.inter * batch)
        m["head.lin2"] = ("mfma_f16x2", 2.0 * spec.inter * spec.n_classes * batch)
        m["head.bn"] = ("hbm", 8.0 * spec.inter * batch)
        # one block on 10x10 planes: 64 rows of 10 words in, 256 rows of 11 words out, 64- / 256-entry tables
        m["va.block"] = ("hbm", batch * (64 * 10 * 8 + 256 * 11 * 8) + 3 * 64 * 64)
        m["va.flatten"] = ("hbm", batch * (256 * 11 * 8 + spec.fcsize * 4))
        return m
    p = spec.p
    m["stem"] = ("mfma_f16x2", 2.0 * p * 147 * 56 * 56 * batch)
    m["head.lin1"] = ("mfma_f16x2", 2.0 * spec.fcsize * 1000 * batch)
    m["head.lin2"] = ("mfma_f16x2", 2.0 * 1000 * 1000 * batch)
    m["head.bn_poly"] = ("hbm", 8.0 * 1000 * batch)
    for i, b in enumerate(spec.blocks):
        tag = f"f{4 + i}"
        c, (h, w), (ho, wo) = b.in_planes, b.in_hw, b.out_hw
        plane_in, plane_out = c * h * w / 8.0, c * ho * wo / 8.0
        if variant == "full":
            n1, n2 = b.conv1.kh * b.conv1.kw, b.conv2.kh * b.conv2.kw
            px = ho * wo
            m[f"full.conv1.{tag}"] = ("full_dw", 2.0 * batch * c * px * (n1 * 8 + 8))
            m[f"full.conv2.{tag}"] = ("full_dw", 2.0 * batch * c * px * (n2 * 8 + 8))
            m[f"full.conv3.{tag}"] = ("full_pw", 2.0 * batch * h * w * b.conv3.groups * (30 * 240 + 240 * 30))
            cf = b.convf
            key = "full.convf_last" if b.last else f"full.convf.{tag}"
            m[key] = ("full_pw", 2.0 * batch * px * cf.groups * (30 * 240 + 240 * cf.cout_g))
            continue
        if variant == "xsmall":
            m[f"gate_stage1.{tag}"] = ("hbm", batch * (plane_in + 4 * plane_out))
            if not b.last:
                m[f"gate_pf.{tag}"] = ("hbm", batch * (4 * plane_out + 2 * plane_out))
            else:
                m["gate_last"] = ("hbm", batch * (4 * plane_out + b.convf.out_planes * (ho // 2) * (wo // 2) * 4))
            continue
        # small: a block-fused launch reads the block input once and writes the block output once
        tables = 2 * c * 8192 + (c // 16) * 131072
        m[f"gate_stage1.{tag}"] = ("hbm", batch * (2 * plane_in + 4 * plane_out) + tables)
        if not b.last:
            out_planes = b.convf.out_planes * ho * wo / 8.0
            m[f"gate_pf.{tag}"] = ("hbm", batch * (4 * plane_out + 2 * out_planes) + (c // 4) * 65536)
            m[f"gate_block.{tag}"] = ("hbm", batch * (plane_in + out_planes) + tables + (c // 4) * 65536)
        else:
            m[f"gate_block.{tag}"] = ("hbm", batch * (plane_in + 4 * plane_out) + tables)
            # 4 branch tensors in, pooled floats out (as two fp16 planes); the 64-byte table rows it gathers
            # come out of L2 / Infinity Cache (256 MiB table): they are not HBM bytes
            m["gate_last"] = ("l2_gather", batch * (4 * plane_out + b.convf.out_planes * (ho // 2) * (wo // 2) * 4))
    return m


def measured_traffic(variant: str, batch: int):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/*traffic_<variant>_b<B>.json,
    collected with separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of this script and corrected
    as MI355X_MICROARCH.md prescribes).  bench.py cannot run the profiler on itself, so this is
    the last committed measurement for this batch size, or nothing."""
    import glob
    pats = [f"*traffic_{variant}_b{batch}.json"] + ([f"*traffic_b{batch}.json"] if variant == "small" else [])
    files = sorted(f for p in pats for f in glob.glob(os.path.join(ROOT, "profiles", p)))
    files = sorted(files, key=os.path.basename)          # names start with the round tag (r01a .. r03e): the last is the latest
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


def cpu_baseline(variant, spec, st, batch: int, budget_s: float = 8.0):
    """The oracle's float-mode forward (the reference's op sequence, incl. its inert randint_like and clones) on
    all host threads: ONE batch of the configuration's size ("same synthetic batch"), timed after a small
    warm-up, is the reported value; a bounded run of batches of 32 (the size the CPU likes better) is kept beside it."""
    import torch
    from oracle import ttnet_float as OF
    from scale_imagenet_amd import synth
    cores = host_cores()
    torch.set_num_threads(cores)
    sd = OF.to_torch_state(st)
    fwd = OF.forward_valexnet if variant == "valexnet" else OF.forward
    small = 256 if variant == "valexnet" else 32
    xs = torch.from_numpy(synth.synth_images(small, hw=spec.image_hw))
    fwd(xs[:8], sd, spec)                  # warm-up
    nb = min(batch, 256)                   # (config 3's 512 would take over a minute on 16 threads: one 256-image batch)
    xb = xs if nb == small else torch.from_numpy(synth.synth_images(nb, hw=spec.image_hw))
    t0 = time.perf_counter()
    fwd(xb, sd, spec)
    el_b = time.perf_counter() - t0
    t0 = time.perf_counter()
    done = 0
    while True:
        fwd(xs, sd, spec)
        done += small
        el = time.perf_counter() - t0
        if el >= budget_s or done >= 16 * small:
            break
    return {"value": nb / el_b, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"one batch of {nb} images (same synthetic generator and weights), oracle/ttnet_float.py = the "
                      f"reference's eval op sequence on torch-CPU {torch.__version__}, {cores} threads, {el_b:.1f} s",
            "batches_of_%d" % small: {"value": done / el, "images": done, "seconds": round(el, 1)}}


def roofline_records(avg_ms, models, traffic, stem_products=3.0):
    kernels = []
    for k, ms in avg_ms.items():
        bound, units = models.get(k, ("hbm", 0.0))
        note = None
        if bound == "mfma_f16x2":
            # algorithmic f32 flops against the 16-bit dense peak; the kernel issues 3 fp16 MFMA
            # flops per algorithmic flop (operands split 2 x fp16), so frac <= 1/3
            bname, ach, peak, unit = "mfma", units / (ms * 1e-3) / 1e12, F16_PEAK_TFLOPS, "TFLOP/s"
        elif bound == "mfma":
            bname, ach, peak, unit = "mfma", units / (ms * 1e-3) / 1e12, F32_PEAK_TFLOPS, "TFLOP/s"
        elif bound == "full_pw":
            bname, ach, peak, unit = "mfma", units / (ms * 1e-3) / 1e12, F16_PEAK_TFLOPS, "TFLOP/s"
            note = ("flops of the two grouped convolutions against the 16-bit dense peak (split fp16: 2.5 MFMA flops issued per flop); "
                    "the launch is bound by the float32 GELU of its 240 hidden values per pixel and group (vector issue, "
                    "16 lanes per clock and SIMD) and includes the float64 pass over the listed pixels (DESIGN.md 4)")
            if os.environ.get("TTNET_FULL_EXACT") == "1":
                peak, note = F64_PEAK_TFLOPS, "float64 flops of the two grouped convolutions against the FP64 datasheet peak (TTNET_FULL_EXACT=1)"
        elif bound == "full_dw":
            bname, ach, peak, unit = "mfma", units / (ms * 1e-3) / 1e12, F32_PEAK_TFLOPS, "TFLOP/s"
            note = ("flops of the depthwise block against the float32 vector peak; the launch is bound by table lookups and "
                    "the float32 GELU of 8 hidden values per output")
        else:
            bname, ach, peak, unit = "hbm", units / (ms * 1e-3) / 1e9, HBM_PEAK_GBS, "GB/s"
            if bound == "l2_gather":
                note = ("bound: L2 / Infinity-Cache gather of 64-byte table rows; achieved = packed inputs + pooled "
                        "outputs only (the bytes that must cross HBM), the rows themselves are cache hits")
        rec = {"kernel": k, "ms": round(ms, 5), "bound": bname, "achieved": round(ach, 3),
               "peak": peak, "unit": unit, "frac": round(ach / peak, 5), "traffic": traffic.get(k)}
        if note:
            rec["note"] = note
        if traffic.get(k) and bname == "hbm":
            rec["frac_on_measured_traffic"] = round(traffic[k] / (ms * 1e-3) / 1e9 / peak, 5)
        if bound == "mfma_f16x2":
            # MFMA flops the kernel really issues per algorithmic flop: 3 products of the 2-way
            # fp16 split (stem: x 8/7 for the kw 7 -> 8 padding, x 22/21 for the k-row padding)
            # (uint8 input: two products, the image's byte sums are exact in fp16 -- csrc/stem.hip)
            issued = stem_products * (8.0 / 7.0) * (22.0 / 21.0) if k == "stem" else 3.0
            rec["mfma_flops_issued_per_flop"] = round(issued, 3)
            rec["frac_issued"] = round(ach * issued / peak, 5)
        kernels.append(rec)
    return kernels


def is_lut_kernel(name: str) -> bool:
    return name.startswith(("gate_stage1", "gate_pf", "gate_block"))


GATE_LOOKUPS_PER_IMAGE = 241_544      # table lookups of the 11 binarised Block_TTs of TT-small
VALU_LANES_PER_CLK, VALU_CLK_HZ = 256 * 4 * 32, 2.4e9      # MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32 (a wave64 instruction takes 2 cycles)


def measured_counters():
    """SQ counters per launch of the gate kernels from the committed rocprofv3 --pmc pass (profiles/*_counters_small_b256.json,
    tools/collect_counters.sh): bench.py cannot run the profiler on itself."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_counters_small_b256.json")), key=os.path.basename)
    if not files:
        return None, None
    with open(files[-1]) as f:
        return json.load(f).get("kernels", {}), os.path.basename(files[-1])


def gate_path_record(avg_ms, batch):
    gate_ms = sum(ms for k, ms in avg_ms.items() if is_lut_kernel(k)) or 1e-9
    gate_bytes = GATE_BYTES_PER_IMAGE * batch + GATE_TABLE_BYTES
    rec = {"kernel": "gate_path (all binarised LUT launches)", "batch": batch, "ms": round(gate_ms, 5), "bound": "hbm",
           "achieved": round(gate_bytes / (gate_ms * 1e-3) / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(gate_bytes / (gate_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": None,
           "launches": sum(1 for k in avg_ms if is_lut_kernel(k))}
    # the second, honest bound of a 16-bit-index table path: vector lane-operations.  Floor: 2 per lookup (form the index,
    # extract the bit / byte) at 32,768 lanes per clock; measured: SQ_INSTS_VALU x 64 lanes of the committed counter pass
    lookups = GATE_LOOKUPS_PER_IMAGE * batch
    floor_ms = 2.0 * lookups / (VALU_LANES_PER_CLK * VALU_CLK_HZ) * 1e3
    rec["valu_bound"] = {"lookups": lookups, "floor_lane_ops_per_lookup": 2, "floor_ms": round(floor_ms, 5),
                         "frac_of_floor": round(floor_ms / gate_ms, 5)}
    ctr, src = measured_counters()
    if ctr and batch == 256:
        insts = sum(v.get("SQ_INSTS_VALU", 0.0) for k, v in ctr.items() if k.startswith("gate_block"))
        if insts:
            rec["valu_bound"].update({"measured_lane_ops_per_lookup": round(insts * 64.0 / lookups, 2),
                                      "measured_valu_issue_ms_at_2_cycles": round(insts / 1024.0 * 2.0 / VALU_CLK_HZ * 1e3, 5),
                                      "counters_source": src})
    return rec


def main():
    args = parse_args()
    from scale_imagenet_amd.launch import spawn_ranks, under_launcher
    if args.gpus > 1 and not under_launcher():
        # plain `python bench.py --gpus N`: this process has not touched the GPU (nothing above
        # initialises HIP) -- start the N ranks as fresh interpreters and relay rank 0's line
        sys.exit(spawn_ranks([os.path.abspath(__file__), *sys.argv[1:]], args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist

    from scale_imagenet_amd import synth, ttnet
    from scale_imagenet_amd.dist import all_gather_logits, init_from_env

    rank, world, local_rank = init_from_env("nccl")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a HIP device (the product has no CPU path)"
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())   # (% only matters for the 1-GPU gloo rehearsal)
    torch.cuda.set_device(dev)

    spec, vargs = variant_spec(args.variant)
    B = args.batch or {"full": 512}.get(args.variant, 256)
    st = synth.synth_state_dict(spec)
    cls = {"small": ttnet.TT_vf_19lv3_imgnet_small, "xsmall": ttnet.TT_vf_19lv3_imgnet_xsmall,
           "full": ttnet.TT_vf_19lv3_imgnet, "valexnet": ttnet.TT_FHE_XSMALL_vAlexnet}[args.variant]
    model = cls(Namespace(layers=1, groups=[1, None, 4, None], **vargs))
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()}, strict=True)
    extras = args.variant == "small" and world == 1 and not args.no_extras
    model = model.to(dev).eval().reserve(max(B, 2048) if extras else B)

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

    # distinct input buffers rotated through the steps: the same images in another order (a roll along the batch), so
    # that nothing has to be synthesised again, but other addresses -- the Infinity Cache is physically indexed
    NX = max(1, args.inputs)
    xs = [x] + [x.roll(k, dims=0).contiguous() for k in range(1, NX)]

    R = max(1, args.inflight)
    if R > 1:
        model.set_lanes(R)
    streams = [torch.cuda.Stream(dev) for _ in range(R)]

    def step(i=0, lanes=1):
        """One step = one forward of one batch (+ the logits all-gather when world > 1).  With
        lanes > 1, step i runs on lane i % lanes with that lane's stream current: the forward is
        launched on it, and the all-gather is ordered behind the forward through it (torch's NCCL /
        RCCL work waits for the current stream and the current stream waits for the collective), so
        a lane's gather never overtakes its own forward while the other lane keeps computing."""
        with torch.no_grad():
            xi = xs[i % NX]
            if lanes == 1:
                y = fwd(xi)
                return all_gather_logits(y, n_total) if world > 1 else y
            lane = i % lanes
            with torch.cuda.stream(streams[lane]):
                y = fwd(xi, lane=lane)
                return all_gather_logits(y, n_total) if world > 1 else y

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    n_windows = args.windows if args.windows > 0 else (9 if args.steps <= 50 else 3)

    def timed(lanes):
        """W untimed warm-up steps, then n_windows windows of EXACTLY K steps, each bracketed by a barrier +
        device synchronisation on both sides, MAX over ranks; returns the windows' times."""
        for i in range(max(args.warmup, 3 * lanes * (NX if lanes > 1 else 1))):     # (3 calls per lane: two plain, then the graph capture)
            step(i, lanes)
        els = []
        for _ in range(n_windows):
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
            els.append(el)
        return els

    def median(v):
        v = sorted(v)
        return v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])

    serial_windows = timed(1)
    windows = timed(R) if R > 1 else serial_windows
    elapsed_serial, elapsed = median(serial_windows), median(windows)

    gather_ok = None
    if world > 1 and args.verify_gather:
        # every rank gathers; rank 0 recomputes all shards alone and compares bit for bit
        torch.cuda.synchronize(dev)
        got = step()
        torch.cuda.synchronize(dev)
        if rank == 0:
            with torch.no_grad():
                want = torch.cat([fwd(torch.from_numpy(synth.synth_images(B, first=r * B, hw=spec.image_hw)).to(dev)).clone()
                                  for r in range(world)])
            gather_ok = bool(torch.equal(got, want))
            if not gather_ok:
                raise SystemExit("gathered logits differ from the single-process logits")

    # second pass: per-kernel device time with HIP events on the launch stream
    def profile(xb, steps):
        model.set_profiling(True)
        acc = {}
        with torch.no_grad():
            for _ in range(steps):
                fwd(xb)
                for k, v in model.last_timings().items():
                    acc[k] = acc.get(k, 0.0) + v
        model.set_profiling(False)
        return {k: v / steps for k, v in acc.items()}

    prof_steps = min(args.steps, 30)
    avg_ms = profile(x, prof_steps)
    # With two or more lanes the plan runs its float32 stem on 128 of the 256 CUs (csrc/stem.hip: the other batch's kernels use the
    # rest).  The per-kernel figures of the roofline are one forward at a time on the whole chip, as in every earlier round: a twin
    # model with a single lane; the stem's time in the in-flight configuration is kept beside them.
    stem_ms_lanes = None
    if R > 1 and args.variant in ("small", "xsmall") and args.input == "f32":
        stem_ms_lanes = avg_ms.get("stem")
        twin = cls(Namespace(layers=1, groups=[1, None, 4, None], **vargs))
        twin.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()}, strict=True)
        twin = twin.to(dev).eval().reserve(B)
        with torch.no_grad():
            for _ in range(3):
                twin(x)
        keep_model, model = model, twin
        fwd_keep, fwd = fwd, twin.forward
        avg_ms = profile(x, prof_steps)
        model, fwd = keep_model, fwd_keep
        del twin

    if rank == 0:
        models = kernel_models(args.variant, spec, B)
        traffic, traffic_src = measured_traffic(args.variant, B)
        kernels = roofline_records(avg_ms, models, traffic, 2.0 if args.input == "u8" else 3.0)
        gate = gate_path_record(avg_ms, B) if args.variant == "small" else None
        dom = max(kernels, key=lambda r: r["ms"])
        roofline = {k: dom[k] for k in dom if k not in ("kernel", "ms")}
        roofline["kernel"] = dom["kernel"]
        roofline["ms"] = dom["ms"]
        if stem_ms_lanes is not None:
            roofline["note"] = ("kernel times: one forward at a time on the whole chip (a single-lane twin of the model); in the timed "
                                "configuration (two batches in flight) the stem runs on 128 of the 256 CUs while the other batch's "
                                "kernels use the rest: see stem_ms_128_cus")
            roofline["stem_ms_128_cus"] = round(stem_ms_lanes, 5)
        metric = {"small": "images/sec ImageNet 224x224, TT-small, MI355X; top-1 exact-match",
                  "xsmall": "images/sec ImageNet 224x224, TT-xsmall, MI355X; top-1 exact-match",
                  "full": "images/sec ImageNet 224x224, TT-full (p=60), MI355X; top-1 exact-match",
                  "valexnet": "images/sec CIFAR 32x32, TT vAlexnet variant, MI355X"}[args.variant]
        workload = {"small": f"TT_general_imagenet_v2_small forward, batch={B} 224x224 per GPU, bit-packed HIP LUT kernels "
                             f"(BASELINE.json configs[1]{'; configs[3] shape when run on 8 GPUs at batch 512' if world > 1 else ''})",
                    "full": f"TT_general_imagenet_v2 (full, p=60) forward, batch={B} 224x224 per GPU (BASELINE.json configs[2])",
                    "valexnet": f"TT_FHE_XSMALL_vAlexnet forward, batch={B} 32x32 per GPU (BASELINE.json configs[4])",
                    "xsmall": f"TT x-small variant forward, batch={B} 224x224 per GPU (parity-test configuration)"}[args.variant]
        out = {
            "metric": metric,
            "value": round(n_total * args.steps / elapsed, 2),
            "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("f32 as fp16x2 split operands on the 16-bit MFMA + float32 GELU, f64 for the outputs whose sign that cannot vouch for (gate path); "
                      if args.variant == "full" else "u16/u32/u64 packed bits (gate path); ") +
                     "f32 as prescaled fp16x2 split operands on the 16-bit MFMA (stem, lin1, lin2)",
            "data": "synthetic",
            "config": {"workload": workload,
                       "batch_per_gpu": B, "global_batch": n_total, "batches_in_flight": R,
                       "input": "float32 NCHW, normalised (the reference's contract)" if args.input == "f32"
                       else "uint8 HWC, ToTensor + Normalize fused into the stem",
                       "parallelism": f"batch shard x{world}" + (" + RCCL all-gather of logits" if world > 1 else "")},
            "inflight": R,
            "windows": {"n": n_windows, "steps_each": args.steps, "value": "median window",
                        "images_per_s": [round(n_total * args.steps / e, 1) for e in windows],
                        "min": round(n_total * args.steps / max(windows), 1), "max": round(n_total * args.steps / min(windows), 1)},
            "input_buffers": {"n": NX, "bytes_each": int(x.numel() * x.element_size()),
                              "note": "rotated step by step; more than the 256 MiB Infinity Cache in total when n >= 2 at "
                                      "batch 256 float32" if NX > 1 else "one buffer: the input may be served from the Infinity Cache"},
            "serial": {"value": round(n_total * args.steps / elapsed_serial, 2),
                       "ms_per_step": round(1e3 * elapsed_serial / args.steps, 4),
                       "note": "the same K steps with one batch in flight (every step waits for the previous one), on the same plan: with lanes >= 2 its float32 stem runs on 128 CUs, which costs a forward that runs alone ~4 %"},
            "roofline": roofline,
            "roofline_kernels": kernels + ([gate] if gate else []),
            "traffic_source": traffic_src,
            "kernel_ms_sum": round(sum(avg_ms.values()), 5),
        }
        if gather_ok is not None:
            out["gather_verified"] = gather_ok
        if world > 1:
            out["dist_backend"] = dist.get_backend()
        # the extra legs never cost the line: a failure in one of them is reported in it
        try:
            if extras:
                # the LUT path again at B = 2048 (fixed per-launch costs amortised), same event timing
                xb = torch.from_numpy(synth.synth_images(256)).to(dev).repeat(8, 1, 1, 1)
                big = profile(xb, 5)
                del xb
                out["gate_path"] = {"b256": gate if B == 256 else None, "b2048": gate_path_record(big, 2048)}
                out["kernel_us_per_256_at_b2048"] = {k: round(v * 1e3 / 8.0, 2) for k, v in big.items()}
            if extras and args.input == "f32" and R > 1:
                # the same forward from the decoder's uint8 images (ToTensor + Normalize fused into an integer stem
                # with two matrix products, csrc/stem.hip): what an input pipeline that hands over uint8 gets
                u8 = synth.synth_images_u8(B, first=rank * B, hw=spec.image_hw)
                xu0 = torch.from_numpy(np.ascontiguousarray(u8.transpose(0, 2, 3, 1))).to(dev)
                xu = [xu0] + [xu0.roll(k, dims=0).contiguous() for k in range(1, NX)]

                def ustep(i):
                    with torch.no_grad(), torch.cuda.stream(streams[i % R]):
                        model.forward_u8(xu[i % NX], lane=i % R)
                for i in range(3 * R * NX):
                    ustep(i)
                ks = 200
                rates = []
                for _ in range(3):
                    fence()
                    t0 = time.perf_counter()
                    for i in range(ks):
                        ustep(i)
                    fence()
                    rates.append(ks * B / (time.perf_counter() - t0))
                rates.sort()
                out["uint8_input"] = {"value": round(rates[1], 2), "unit": "images/s", "ms_per_step": round(1e3 * B / rates[1], 4),
                                      "inflight": R, "input": "uint8 NHWC (ttnet_forward_u8)",
                                      "note": "median of three windows of 200 steps; not the headline: the reference's API takes the "
                                              "normalised float32 tensor"}
                del xu, xu0
            if world == 1 and not args.no_extras:
                out["parity"] = golden_parity(args.variant, spec, model, dev)
            if args.variant == "full" and os.environ.get("TTNET_FULL_EXACT") != "1":
                # share of the blocks' outputs the fast evaluation could not vouch for and sent through float64;
                # counters of the plan's current lane since it was created
                plan = model._any_plan()
                out["full_fast_path"] = {"listed_pixel_groups_1x1": plan.query("full_listed_pw"),
                                         "listed_outputs_depthwise": plan.query("full_listed_dw"),
                                         "note": "running totals over every forward of this process on one lane; "
                                                 "about 1.1 % of the (pixel, group) pairs and 1e-4 of the depthwise outputs"}
        except Exception as e:                      # noqa: BLE001 - reported, not swallowed
            out["extras_error"] = f"{type(e).__name__}: {e}"
        try:
            if world == 1 and not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(args.variant, spec, st, B)
                out["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
        except Exception as e:                      # noqa: BLE001
            out["cpu_baseline"] = None
            out["cpu_baseline_error"] = f"{type(e).__name__}: {e}"
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def golden_parity(variant, spec, model, dev):
    """|hip - reference| on the committed golden images: tests/golden/ref_<variant>.npz holds the logits
    the imported reference itself produced for synth_images(n) (oracle/gen_golden.py)."""
    import numpy as np
    import torch
    from scale_imagenet_amd import synth
    path = os.path.join(ROOT, "tests", "golden", f"ref_{variant}.npz")
    if not os.path.exists(path):
        return None
    with np.load(path) as z:
        ref, n = z["logits"], int(z["n_images"])
    with torch.no_grad():
        y = model(torch.from_numpy(synth.synth_images(n, hw=spec.image_hw)).to(dev)).cpu().numpy()
    d = np.abs(y - ref).max(axis=1)
    return {"images": n, "max_abs_logit_diff_vs_reference": float(d.max()), "median_image": float(np.median(d)),
            "images_within_1e-5": int((d <= 1e-5).sum()), "top1_equal": int((y.argmax(1) == ref.argmax(1)).sum()),
            "note": "reference = the imported PyTorch-CPU model's own logits (fixture); an image that crosses a float32 "
                    "near tie of the reference's stem or tables differs by ~1e-3, see DESIGN.md 'Near ties'"}


if __name__ == "__main__":
    main()
