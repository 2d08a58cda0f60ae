#!/usr/bin/env python3
"""Throughput of ProbTalk's token hot path on MI355X: generated motion frames/sec.

Workload (BASELINE.json configs[3], the per-GPU share of the configuration the metric is quoted
on): the full holistic pipeline  audio features -> AudEnc -> motion-context encoder -> 8-pass
MaskGIT predictor -> PQ gather + conv decoder -> resize/blend -> refiner transformer -> blend,
one batch of 512 synthetic 300-frame clips per step per GPU (4096 clips over 8 GPUs), weights
random-initialised with the reference's architecture (probtalk_amd.weights), inputs resident in HBM.

    python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU over RCCL.  Under torch.distributed.run (WORLD_SIZE set) this process is one of the ranks;
without it `--gpus N` makes this process the parent of N fresh rank processes (probtalk_amd/launch.py: started before
anything here touches the GPU, watched with a wall-clock limit, all of them ended when one fails).

Prints ONE JSON line on rank 0 (see the field notes in DESIGN.md §Measurement).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: dense fp32 matrix peak
PEAK_BF16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 matrix peak (never the 2:1-sparsity figure)
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E peak (6.3 TB/s measured achievable)
# fp32 tap-GEMM with option split_bf16 (default): six bf16 MFMA products per fp32 multiply-add, so the roof of the
# ALGORITHMIC fp32 FLOP rate is the dense bf16 peak / 6
PEAK_F32_SPLIT_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0
MFLOP_PER_FRAME_VQVAE = 15.86    # PQ-VAE encode 8.11 + quantize 0.016 + decode 7.73
PQ_BYTES_PER_LATENT_STEP = 4128  # read z 2048 + write q 2048 + 4 int64 indices
MFLOP_PER_FRAME = 296.0          # full pipeline, algorithmic (2*MAC of the reference modules)


def cpu_baseline(B=32, T=300, runs=5):
    """The CPU oracle ("port") on a bounded sample of the same workload, on this box's host cores."""
    from oracle import probtalk_oracle as O
    from probtalk_amd.weights import make_input
    from tests.helpers import ref_state_dict
    sd_vq, sd_a, sd_p, sd_r = (ref_state_dict(k) for k in ("VQ", "AudEnc", "Predictor", "HFTransformer"))
    Tq = T
    for _ in range(3):
        Tq = (Tq - 2) // 2 + 1
    N = Tq * 4
    aud = make_input("cpu.aud", (B, 768, T))
    gt, mask = torch.zeros(B, 376, T), torch.zeros(B, 1, T)
    ids = torch.arange(B) % 4
    nq, ng = make_input("cpu.q", (8, B, N, 128), "exp"), make_input("cpu.g", (8, B, N), "gumbel")

    def run():
        body, _ = O.predictor_infer(sd_a, sd_p, sd_vq, aud, gt, mask, ids, nq, ng)
        return O.refiner_infer(sd_r, aud, gt, mask, ids, body)

    with torch.no_grad():
        run()
        ts = []
        for _ in range(runs):
            t0 = time.perf_counter()
            run()
            ts.append(time.perf_counter() - t0)
    med = sorted(ts)[len(ts) // 2]
    return {"value": B * T / med, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"full pipeline (oracle: AudEnc+predictor 8 passes+PQ decode+refiner), {B} clips x {T} frames, "
                      f"fp32, median of {runs} runs after 1 warm-up ({med:.2f} s/run)"}


def split_on():
    import ctypes
    from probtalk_amd import _lib
    v = ctypes.c_int()
    _lib.check(_lib.lib().pt_get_option(b"split_bf16", ctypes.byref(v)))
    return v.value == 1


def f32_peak():
    """(peak TFLOP/s, kernel name) of the fp32 tap-GEMM as configured (option split_bf16)."""
    if split_on():
        return PEAK_F32_SPLIT_TFLOPS, ("pt::gemm_f32s_kernel (fp32 tap-GEMM on the bf16 MFMA: 3-term operand splits, 6 products; "
                                       "peak = dense bf16 peak / 6)")
    return PEAK_F32_MFMA_TFLOPS, "pt::gemm_f32_kernel (tap-GEMM, fp32 MFMA 32x32x2)"


def vqvae_bench(a, rank, world, dev):
    """BASELINE.json configs[1]: PQ-VAE encode -> nearest-code search -> decode (VQVAE_SC.forward, vqvae_1d_sc.py:300-324),
    `--batch` (256) synthetic 88-frame clips per GPU per step, fp32, one launch sequence.  Rooflines: the fp32 tap-GEMM
    (encoder + decoder convs, MFMA) from the per-launch HIP events, and the code search kernel against HBM: its
    algorithmic bytes (4128 per latent step) over its own launch time, timed alone with HIP events."""
    import torch.distributed as dist
    from probtalk_amd import _lib
    from probtalk_amd.synthetic import build_pipeline
    L = _lib.lib()
    pred, _ = build_pipeline(seed=0)
    vq = pred.VQ.eval()
    B, T = (a.batch if a.batch != 512 else 256), 88
    g = torch.Generator(device=dev).manual_seed(4321 + rank)
    x = torch.randn(B, 376, T, device=dev, generator=g)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        with torch.no_grad():
            return vq(x, result_form="full")

    dt, (z, qo, rec) = time_steps(step, fence, a.steps, a.warmup, world, dev)
    assert bool(torch.isfinite(rec).all()) and tuple(rec.shape) == (B, 376, T)
    res = {"metric": "generated motion frames/sec (PQ-VAE encode+quantize+decode)", "value": world * B * T * a.steps / dt,
           "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "configs[1]: PQ-VAE (config/vq.json) encode -> 4 x 128-code product quantizer -> decode, fp32, "
                                  "bit-exact code indices", "clips_per_step_per_gpu": B, "frames_per_clip": T,
                      "parallelism": f"dp{world}: clips sharded over ranks"}}
    if rank == 0 and not a.no_roofline:
        torch.cuda.synchronize()
        _lib.check(L.pt_prof_enable(1))
        step()
        torch.cuda.synchronize()
        ms, fl, ln = (C.c_double * 8)(), (C.c_double * 8)(), (C.c_int * 8)()
        _lib.check(L.pt_prof_read(ms, fl, ln))
        _lib.check(L.pt_prof_enable(0))
        tot_ms, tot_fl, tot_n = sum(ms[i] for i in range(3)), sum(fl[i] for i in range(3)), sum(ln[i] for i in range(3))
        peak, kname = f32_peak()
        ach = tot_fl / (tot_ms * 1e-3) / 1e12
        res["roofline"] = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": None,
                           "kernel": kname + ", all launches of one step", "launches_per_step": tot_n,
                           "avg_launch_ms": tot_ms / max(tot_n, 1), "kernel_ms_per_step": tot_ms,
                           "whole_step": {"algorithmic_tflop": MFLOP_PER_FRAME_VQVAE * 1e6 * B * T / 1e12,
                                          "achieved_tflops": MFLOP_PER_FRAME_VQVAE * 1e6 * B * T / 1e12 / (1e-3 * res["ms_per_step"])}}
        # the quantizer alone: pq_search_kernel on this step's latent rows, and on the bench batch of the pipeline
        # (512 clips x 37 latent steps) for a launch long enough to read a rate from
        cb = vq.vq_layer._codebooks(dev)
        Gq, K, D = cb.shape
        quant = {}
        for tag, rows in (("this step", B * (T // 8)), ("512 clips x 300 frames", 512 * 37), ("262,144 rows", 262144)):
            zc = torch.randn(rows, Gq * D, device=dev, generator=g)
            qc = torch.empty_like(zc)
            idx = torch.empty(rows, Gq, dtype=torch.int64, device=dev)

            def run():
                _lib.check(L.pt_pq_search_cl(_lib.ptr(zc), Gq * D, _lib.ptr(cb), rows, Gq, D, K, _lib.ptr(idx), _lib.ptr(qc), Gq * D,
                                             _lib.stream()))
            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 20
            torch.cuda.synchronize()
            e0.record()
            for _ in range(n):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / n
            gbs = rows * PQ_BYTES_PER_LATENT_STEP / (us * 1e-6) / 1e9
            tfl = 2.0 * rows * Gq * K * D / (us * 1e-6) / 1e12
            quant[tag] = {"latent_rows": rows, "launch_us": us, "achieved_GBps": gbs, "frac_of_hbm_peak": gbs / PEAK_HBM_GBS,
                          "distance_tflops_fp32_mfma": tfl, "frac_of_fp32_mfma_peak": tfl / PEAK_F32_MFMA_TFLOPS}
        res["quantizer_roofline"] = {"kernel": "pt::pq_search_kernel (distances on fp32 MFMA, wave argmin, fused gather)",
                                     "bound": "hbm", "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                     "algorithmic_bytes_per_latent_step": PQ_BYTES_PER_LATENT_STEP,
                                     "timed_as": "torch.cuda.Event around 20 back-to-back launches on the launch stream, "
                                                 "per row count", "by_rows": quant}
    if rank == 0:
        print(json.dumps(res), flush=True)


def train_bench(a, rank, world, dev):
    """BASELINE.json configs[4]: predictor training step (frozen PQ-VAE tokenizer, AudEnc + PredictorNet forward and
    backward, gradient all-reduce over RCCL when N > 1, per-module clipping, AdamW), `--batch` clips of 180 frames
    (config Data.pose.generate_length) per GPU per step.  Secondary workload: the default run stays the metric's."""
    import torch.distributed as dist
    from probtalk_amd.synthetic import build_pipeline
    pred, _ = build_pipeline(seed=0)
    B, T = a.batch, 180
    g = torch.Generator(device=dev).manual_seed(99 + rank)
    bat = {"aud_feat": torch.randn(B, 768, T, device=dev, generator=g),
           "poses": 0.3 * torch.randn(B, 330, T, device=dev, generator=g),
           "expression": 0.3 * torch.randn(B, 100, T, device=dev, generator=g),
           "speaker": torch.randint(0, 4, (B,), device=dev, generator=g), "epoch": 50}

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        pred(bat)
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        _, ld = pred(bat)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    assert ld["loss"] == ld["loss"], "non-finite loss"
    res = {"metric": "predictor training frames/sec (fwd+bwd+all-reduce+clip+AdamW)", "value": world * B * T * a.steps / dt,
           "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "configs[4]: s2g_body_predictor TrainWrapper.__call__ (frozen PQ-VAE encode, AudEnc + "
                                  "PredictorNet train forward with dropout 0.1, cross-entropy, backward, clip, AdamW)",
                      "clips_per_step_per_gpu": B, "frames_per_clip": T, "last_loss": ld["loss"],
                      "parallelism": f"dp{world}: clips sharded over ranks, bucketed gradient all-reduce over RCCL"}}
    if rank == 0 and not a.no_roofline and (world == 1 or a.roofline):
        # dominant kernel of the step: the fp32 tap-GEMM (forward + input gradients), every launch of one extra, untimed step
        # bracketed by HIP events on its stream; the weight-gradient kernel's own figures: profiles/r3/wgrad_split_bf16.txt
        from probtalk_amd import _lib
        L = _lib.lib()
        torch.cuda.synchronize()
        _lib.check(L.pt_prof_enable(1))
        pred(bat)
        torch.cuda.synchronize()
        ms, fl, ln = (C.c_double * 8)(), (C.c_double * 8)(), (C.c_int * 8)()
        _lib.check(L.pt_prof_read(ms, fl, ln))
        _lib.check(L.pt_prof_enable(0))
        tot_ms, tot_fl, tot_n = sum(ms[i] for i in range(3)), sum(fl[i] for i in range(3)), sum(ln[i] for i in range(3))
        peak, kname = f32_peak()
        ach = tot_fl / (tot_ms * 1e-3) / 1e12
        res["roofline"] = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": None,
                           "kernel": kname + ": forward and input-gradient launches of one train step",
                           "launches_per_step": tot_n, "avg_launch_ms": tot_ms / max(tot_n, 1), "kernel_ms_per_step": tot_ms,
                           "algorithmic_gflop_per_launch": tot_fl / max(tot_n, 1) / 1e9,
                           "timed_as": "HIP events around every tap-GEMM launch of one untimed extra step; rocprofv3 summary: "
                                       "profiles/r3/train_kernel_stats.csv"}
    if world > 1:
        dist.barrier()
    if rank == 0:
        print(json.dumps(res), flush=True)


def time_steps(step, fence, steps, warmup, world, dev):
    """W untimed warm-up steps, then exactly K steps between two fences (barrier + device synchronize); max over ranks."""
    import torch.distributed as dist
    out = None
    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    return dt, out


def gemm_roofline(pred, ref, inputs, dtype, ms_per_step, B, T):
    """HIP-event timing of every GEMM launch of one extra, untimed step on one stream (pt_prof_enable/pt_prof_read)."""
    from probtalk_amd import _lib
    from probtalk_amd.synthetic import holistic_step
    L = _lib.lib()
    torch.cuda.synchronize()
    _lib.check(L.pt_prof_enable(1))
    holistic_step(pred, ref, *inputs, sync=False)
    torch.cuda.synchronize()
    ms, fl, ln = (C.c_double * 8)(), (C.c_double * 8)(), (C.c_int * 8)()
    _lib.check(L.pt_prof_read(ms, fl, ln))
    _lib.check(L.pt_prof_enable(0))
    names = ("f32 128x128", "f32 64x128", "f32 64x64", "bf16 conv stacks: 256x256 lds-dma tap-GEMM (conv8_kernel) + 4-wave 128x128", "bf16 64x128", "bf16 64x64",
             "bf16 256x256 8-phase lds-dma (gemm8_kernel)", "bf16 N=512 row-complete +LN epilogues (gemm_row_kernel, sublayer_kernel)")
    dom = range(3, 8) if dtype == "bf16" else range(0, 3)     # the dominant kernel family of this mode
    peak = PEAK_BF16_MFMA_TFLOPS
    kname = "pt::gemm_row_kernel + pt::gemm8_kernel + pt::conv8_kernel + pt::gemm_bf16_kernel (bf16 MFMA GEMMs of the transformers and conv stacks)"
    if dtype != "bf16":
        peak, kname = f32_peak()
    tot_ms, tot_fl, tot_n = sum(ms[i] for i in dom), sum(fl[i] for i in dom), sum(ln[i] for i in dom)
    ach = tot_fl / (tot_ms * 1e-3) / 1e12
    step_tflop = MFLOP_PER_FRAME * 1e6 * B * T / 1e12
    return {
        "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": None,
        "kernel": kname + ", all launches of one step",
        "launches_per_step": tot_n, "avg_launch_ms": tot_ms / max(tot_n, 1),
        "algorithmic_gflop_per_launch": tot_fl / max(tot_n, 1) / 1e9,
        "kernel_ms_per_step": tot_ms,
        "timed_as": "every launch of one untimed extra step on ONE stream at the full per-GPU batch, HIP events on the "
                    "launch stream; rocprofv3 --kernel-trace --stats summaries of the same launch shapes are under profiles/",
        "per_kernel_config": {names[i]: {"launches": ln[i], "ms": ms[i], "gflop": fl[i] / 1e9,
                                         "tflops": (fl[i] / (ms[i] * 1e-3) / 1e12) if ms[i] else 0.0}
                              for i in range(8) if ln[i]},
        "whole_step": {"algorithmic_tflop": step_tflop, "achieved_tflops": step_tflop / (1e-3 * ms_per_step)},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=("infer", "train", "vqvae"), default="infer",
                    help="infer (default): the metric's full pipeline; train: configs[4] predictor training step; "
                         "vqvae: configs[1] PQ-VAE encode -> quantize -> decode, 256 clips x 88 frames, fp32")
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks (one process per GPU). Without WORLD_SIZE in the environment and N > 1 this process spawns "
                         "the N ranks itself; under torch.distributed.run it must equal WORLD_SIZE")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=512,
                    help="clips per step per GPU: the per-GPU share of configs[3] (4096 clips over 8 GPUs), one batch")
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--dtype", choices=("bf16", "f32"), default="bf16",
                    help="arithmetic of the predictor/refiner transformers (PQ-VAE encode and code search always fp32)")
    ap.add_argument("--streams", type=int, default=None,
                    help="HIP streams the step's clips are split over (contiguous shares, joined before the step returns); "
                         "1 = the whole batch as one launch sequence.  Default: 2 above 256 clips per step (the streams fill each "
                         "other's partly filled rounds of tiles: 50.2 vs 53.4 ms at 384 clips), 1 up to there (measured equal at "
                         "256, 31.2 vs 29.6 ms at 192, 10.3 vs 6.2 ms at 2 clips: small launches are latency-bound)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step from one captured HIP graph (probtalk_amd/graphs.py): for small, launch-bound batches")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--roofline", action="store_true",
                    help="N > 1: run rank 0's per-launch roofline pass too (off by default there: the other ranks would sit in a "
                         "barrier for its duration)")
    ap.add_argument("--no-fp32", action="store_true", help="skip the fp32-arithmetic leg of the default bf16 run")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="library switch for A/B runs (pt_set_option: resid_bf16, fused_tail, small_rows, row_tile, ...)")
    a = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and (a.gpus or 1) > 1:
        # parent of the ranks: nothing in this process has touched (or will touch) the GPU
        from probtalk_amd.launch import spawn_ranks
        limit = float(os.environ.get("PT_BENCH_LAUNCH_LIMIT", "1500"))
        sys.exit(spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], a.gpus, limit))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus is not None and a.gpus != world:
        sys.exit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch as `python bench.py --gpus N` (spawns its own "
                 f"ranks) or under torch.distributed.run with --nproc-per-node equal to --gpus")
    if os.environ.get("PT_BENCH_FAIL_RANK") == str(rank) and world > 1:       # launcher failure-path test hook
        sys.exit(1)
    backend = os.environ.get("PT_BENCH_BACKEND", "nccl")      # "gloo": rehearsal of the N > 1 control flow on one GPU
    if backend == "gloo":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import datetime
        import torch.distributed as dist
        # a stuck collective ends the job instead of hanging it (RCCL's watchdog aborts after `timeout`)
        tmo = datetime.timedelta(seconds=float(os.environ.get("PT_BENCH_COLLECTIVE_TIMEOUT", "300")))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)      # RCCL over xGMI
        else:
            dist.init_process_group(backend, timeout=tmo)
    for kv in a.option:
        from probtalk_amd import _lib
        name, _, val = kv.partition("=")
        _lib.check(_lib.lib().pt_set_option(name.encode(), int(val)))
    if world > 1 and rank == 0:
        # first contact with RCCL over xGMI happens on the driver's node: say what it runs on
        try:
            print(f"bench.py: world {world}, backend {backend}, RCCL/NCCL {torch.cuda.nccl.version()}, rank {rank} -> "
                  f"{torch.cuda.get_device_name(local_rank)} (cuda:{local_rank})", file=sys.stderr, flush=True)
        except Exception as e:  # informational only
            print(f"bench.py: world {world}, backend {backend} ({e})", file=sys.stderr, flush=True)
    if a.workload in ("train", "vqvae"):
        (train_bench if a.workload == "train" else vqvae_bench)(a, rank, world, dev)
        if world > 1:
            dist.destroy_process_group()
        return
    from probtalk_amd.dist import OutputGatherer
    from probtalk_amd.synthetic import build_pipeline, holistic_step

    pred, ref = build_pipeline(seed=0)
    if a.dtype == "bf16":
        for mod in (pred.Predictor, pred.AudEnc, pred.VQ, ref.RTransformer):
            mod.set_precision("bf16")
    B, T = a.batch, a.frames
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    aud = torch.randn(B, 768, T, device=dev, generator=g)          # wav2vec2-base feature shape (768-d, 30 fps)
    gt = torch.zeros(B, 376, T, device=dev)
    mask = torch.zeros(B, 1, T, device=dev)
    ids = torch.randint(0, 4, (B,), device=dev, generator=g)
    gather = OutputGatherer(world, (B, 376, T), dev) if world > 1 else None

    if a.streams is None:
        a.streams = 2 if B > 256 else 1
    graphed = None
    if a.graph:
        from probtalk_amd.graphs import GraphedHolisticStep
        graphed = GraphedHolisticStep(pred, ref, B, T)

    streams = [torch.cuda.Stream(device=dev) for _ in range(a.streams)] if a.streams > 1 and not a.graph else None

    def step():
        if graphed:
            if gather is not None:
                gather.wait()       # the replay overwrites the graph's static output: the previous gather must have read it
            out = graphed(aud, gt, mask, ids)
        else:
            out = holistic_step(pred, ref, aud, gt, mask, ids, sync=False, streams=streams)
        if gather is not None:
            gather.submit(out)          # RCCL gather of the outputs to rank 0 on a side stream, overlapped with the next step
        return out

    def fence():
        if gather is not None:
            gather.wait()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    dt, out = time_steps(step, fence, a.steps, a.warmup, world, dev)
    assert bool(torch.isfinite(out).all()), "non-finite output"
    frames = world * B * T * a.steps
    arith = {"bf16": "bf16 MFMA operands / fp32 accumulate in the transformers, condition encoders and PQ-VAE decoder", "f32": "fp32"}
    res = {
        "metric": "generated motion frames/sec (PQ-VAE+predictor+refiner)",
        "value": frames / dt, "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": "configs[3] per-GPU share: full holistic pipeline (AudEnc + motion-context encoder + "
                               "8-pass MaskGIT predictor + PQ gather/decode + refiner); " + arith[a.dtype],
                   "clips_per_step_per_gpu": B, "frames_per_clip": T, "launch": "hip graph replay" if a.graph else (f"eager, clips split over {a.streams} HIP streams" if streams else "eager, one stream"),
                   "parallelism": f"dp{world}: clips sharded over ranks, outputs gathered to rank 0 over RCCL"},
    }

    if rank == 0 and not a.no_roofline and (world == 1 or a.roofline):
        res["roofline"] = gemm_roofline(pred, ref, (aud, gt, mask, ids), a.dtype, res["ms_per_step"], B, T)
        # HBM traffic of the same kernels from the PMC passes committed under profiles/ (rocprofv3 cannot run inside
        # this process): launch-weighted mean bytes per launch, FETCH_SIZE doubled as the microarch guide prescribes
        pmc_path = next((q for q in (os.path.join(ROOT, "profiles", r, "bf16_pmc_summary.json") for r in ("r3", "r2", "r1"))
                         if os.path.exists(q)), None)
        if a.dtype == "bf16" and pmc_path:
            with open(pmc_path) as f:
                pk = json.load(f)["kernels"]
            fam = {k: v for k, v in pk.items() if any(t in k for t in ("gemm_row_kernel", "gemm8_kernel", "gemm_big_kernel", "gemm_bf16_kernel", "sublayer_kernel", "conv8_kernel"))}
            n = sum(v["launches"] for v in fam.values())
            if n:
                res["roofline"]["traffic"] = sum(v["launches"] * v["hbm_bytes_per_launch"] for v in fam.values()) / n
                res["roofline"]["traffic_unit"] = f"bytes per launch (PMC: 2*FETCH_SIZE + WRITE_SIZE, {os.path.relpath(pmc_path, ROOT)})"
                res["roofline"]["mfma_util_pmc"] = {k.split("pt::")[-1]: round(v["mfma_util"], 4) for k, v in fam.items()
                                                    if v.get("mfma_util") is not None}
    if rank == 0 and world == 1 and a.dtype == "bf16" and not a.no_fp32 and not a.graph:
        # the reference's own arithmetic (fp32, the 1e-4 parity path) on the same workload, a few steps
        del pred, ref
        torch.cuda.empty_cache()
        pred32, ref32 = build_pipeline(seed=0)
        k32 = max(2, min(3, a.steps))
        dt32, out32 = time_steps(lambda: holistic_step(pred32, ref32, aud, gt, mask, ids, sync=False, streams=streams),
                                 torch.cuda.synchronize, k32, 1, 1, dev)
        assert bool(torch.isfinite(out32).all()), "non-finite fp32 output"
        res["fp32"] = {"value": B * T * k32 / dt32, "unit": "frames/s", "steps": k32, "warmup": 1, "ms_per_step": 1e3 * dt32 / k32,
                       "dtype": "f32", "note": "same workload in the reference's arithmetic (pt_model_set_precision 0): fp32 "
                                               "operands and accumulation, tap-GEMMs on " + f32_peak()[1]}
        if not a.no_roofline:
            r32 = gemm_roofline(pred32, ref32, (aud, gt, mask, ids), "f32", res["fp32"]["ms_per_step"], B, T)
            res["fp32"]["roofline"] = {k: r32[k] for k in ("bound", "achieved", "peak", "unit", "frac", "kernel", "launches_per_step",
                                                           "kernel_ms_per_step", "whole_step")}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()          # rank 0 is still in its roofline pass: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
