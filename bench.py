#!/usr/bin/env python3
"""Headline benchmark: IQ MSamples/s analysed by the signal-analysis path.

Workload (BASELINE.json configs[1]): per GPU, 256 synthetic complex64 streams at
2.048 MSPS, one second each (B = 2 048 000, T = 8000), nperseg 256 hamming,
4-8 sparse 15 ms pulses per stream, resident in HBM.  A step = one pass of the
whole path (fused STFT/scan kernel + detect kernels + records copied to the
host) over that batch; consecutive steps are pipelined two deep inside the
library (scan of step i+1 overlaps detect/fetch of step i).  With N > 1 every rank analyses its own 256 streams
(weak scaling, no data-path collective); value = samples of all ranks / max
time over ranks.

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel
(stft_scan) at 8 algorithmic bytes per IQ sample against the 8 TB/s HBM peak,
from HIP events recorded on the launch stream around every launch of the timed
region.  `cpu_baseline` is the oracle (port of the reference's SciPy/NumPy
path) on this node's host cores, N = 1 only.
"""
import argparse
import json
import os
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
BYTES_PER_SAMPLE = 8  # one complex64 read
# HBM bytes per scan-kernel launch on the default workload, from the PMC passes in
# profiles/r01_i_pmc_traffic.txt (FETCH_SIZE x 1024 / 0.51 [gfx950 half-count, calibrated on the
# kernel's own load stream] + WRITE_SIZE x 1024).  Only quoted for that exact workload.
PMC_TRAFFIC_DEFAULT = {"bytes_per_launch": 4456600000, "source": "profiles/r01_i_pmc_traffic.txt"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--streams", type=int, default=256, help="streams per GPU")
    ap.add_argument("--sample-rate", type=int, default=2048000)
    ap.add_argument("--seconds", type=float, default=1.0, help="buffer length per stream")
    ap.add_argument("--nperseg", type=int, default=256)
    ap.add_argument("--window", default="hamming")
    ap.add_argument("--mode", default="auto", choices=["auto", "dense", "sparse"])
    ap.add_argument("--segs-per-chunk", type=int, default=0)
    ap.add_argument("--input", default="c64", choices=["c64", "u8"],
                    help="IQ representation in HBM: complex64 (the BASELINE workload) or the RTL-SDR wire format "
                         "(interleaved uint8, converted inside the scan kernel; SURVEY 8(f) rank 1)")
    ap.add_argument("--threshold-dbw", type=float, default=None,
                    help="signal_threshold_dbw (default: the reference's -90, or -80 with --input u8)")
    ap.add_argument("--trains", action="store_true",
                    help="BASELINE config 5 style input: 8-16 tags per stream, pulse trains 10-38 ms, period 0.1-1 s")
    ap.add_argument("--settle", type=int, default=30,
                    help="untimed steps run once during set-up, before the W warm-up steps: the GPU's clocks need ~10 "
                         "launches after idle to settle (0.85 -> 0.77 ms per scan launch), whatever W the caller picks")
    ap.add_argument("--lanes", type=int, default=2,
                    help="stream groups per GPU, each on its own handle / HIP stream (detect of one group overlaps the "
                         "scan of the other); 1 = one launch sequence per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-streams", type=int, default=0, help="streams in the CPU-baseline sample (0 = auto)")
    ap.add_argument("--parity-streams", type=int, default=4)
    return ap.parse_args()


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hook (1-GPU box): all ranks on GPU 0, control plane over gloo -- exercises the N > 1 code path
    # (rendezvous, per-rank seeds, barrier, max-over-ranks, rank-0 print) where only one GPU exists
    share_gpu = os.environ.get("RT_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from pyradiotracking_amd import synth
    from pyradiotracking_amd.analyze import BatchSignalAnalyzer, window_coefficients

    fs, nperseg = args.sample_rate, args.nperseg
    blen = int(round(args.seconds * fs))
    n_seg = blen // nperseg
    S = args.streams
    win = window_coefficients(args.window, nperseg)
    kw = dict(sample_rate=fs, fft_nperseg=nperseg, fft_window=args.window)
    u8 = args.input == "u8"
    if u8:
        # 8-bit front end: noise ~1.5 LSB rms, pulses 18..32 dB above the -80 dBW threshold
        kw["signal_threshold_dbw"] = -80.0
        iq_c = synth.make_batch_device(S, blen, fs, win, seed=1000 + rank, device=f"cuda:{local_rank}",
                                       noise_sigma=0.012, peak_dbw=(-62.0, -48.0))
        iq = synth.quantize_u8_device(iq_c)
        del iq_c
    else:
        iq = synth.make_batch_device(S, blen, fs, win, seed=1000 + rank, device=f"cuda:{local_rank}", trains=args.trains)
    if args.threshold_dbw is not None:
        kw["signal_threshold_dbw"] = args.threshold_dbw
    stream = torch.cuda.current_stream()
    torch.cuda.synchronize()  # the IQ is complete before any lane's own stream reads it
    an = BatchSignalAnalyzer(
        [str(i) for i in range(S)],
        sdr_callback_length=blen,
        gpu=local_rank,
        mode=args.mode,
        timing=True,
        segs_per_chunk=args.segs_per_chunk,
        hip_stream=stream.cuda_stream if args.lanes <= 1 else None,
        lanes=args.lanes,
        **kw,
    )

    def barrier():
        if world > 1:
            dist.barrier()

    def run(n_steps):
        """n_steps full steps (enqueue + fetch each).  The handle keeps two calls in flight, so
        step i+1 is enqueued before step i is fetched: its scan overlaps step i's detect kernels,
        record copy and host-side fetch.  Every step's work completes inside this function."""
        acc = [0.0, 0.0, 0]
        rec = info = None
        enq = an.enqueue_bytes if u8 else an.enqueue
        if n_steps:
            enq(iq)
        for i in range(n_steps):
            if i + 1 < n_steps:
                enq(iq)
            rec = an.fetch_records()
            info = an.call_info()
            acc[0] += info.ms_stft
            acc[1] += info.ms_detect
            acc[2] += info.fell_back
        return rec, info, acc

    run(args.settle)  # set-up: clocks to steady state (not part of the W warm-up steps, never timed)
    rec, info, _ = run(args.warmup)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rec, info, (ms_stft, ms_detect, fell_back) = run(args.steps)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share_gpu else f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    samples_per_step_gpu = S * n_seg * nperseg  # samples actually transformed (T6)
    total_samples = samples_per_step_gpu * world * args.steps
    value = total_samples / elapsed / 1e6
    k_ms = ms_stft / max(1, args.steps)
    bytes_per_sample = 2 if u8 else BYTES_PER_SAMPLE
    achieved = samples_per_step_gpu * bytes_per_sample / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0

    # BASELINE.json configs by their geometry (per-GPU stream counts of configs 4 and 5 are the 8-GPU shares or more)
    config_name = {(2048000, 256, "hamming", 2048000): "config2", (2400000, 1024, "hann", 2400000): "config3",
                   (2048000, 256, "hamming", 524288): "config4 (B = 524288)",
                   (3200000, 4096, "hamming", 3200000): "config5"}.get((fs, nperseg, args.window, blen), "custom")
    lanes = max(1, args.lanes)
    default_workload = (S, fs, blen, nperseg, args.window, args.segs_per_chunk, args.mode, args.input) == (256, 2048000, 2048000, 256, "hamming", 0, "auto", "c64")
    traffic = PMC_TRAFFIC_DEFAULT["bytes_per_launch"] if default_workload else None
    out = {
        "metric": "IQ MSamples/s analysed (STFT + detect + records), detected-signal parity vs CPU",
        "value": round(value, 1),
        "unit": "MSamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32" if not u8 else "f32 (uint8 IQ converted in the load)",
        "data": "synthetic",
        "config": {
            "workload": f"{config_name}: {S} streams/GPU x {fs} SPS x {args.seconds:g} s {'uint8 I/Q' if u8 else 'complex64'}, nperseg {nperseg} {args.window}, {'tag trains, 8-16 tags/stream' if args.trains else '4-8 sparse 15 ms pulses/stream'}",
            "streams_per_gpu": S,
            "samples_per_stream": blen,
            "segments_per_stream": n_seg,
            "mode": {1: "dense", 2: "sparse"}.get(info.mode_used, "?"),
            "fallbacks": fell_back,
            "records_per_step": int(len(rec)),
            "candidate_cells_per_step": int(info.n_hot),
            "sharding": "streams sharded per GPU, no collective",
            "lanes_per_gpu": args.lanes,
            "pulse_recipe": "tag trains (config 5)" if args.trains else "4-8 pulses of 15 ms",
        },
        "roofline": {
            "bound": "hbm",
            "kernel": "stft_scan",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            # per launch like `achieved`: the step's bytes split evenly over its scan launches (one per lane)
            "traffic": traffic // lanes if traffic else None,
            "traffic_unit": "bytes/launch (PMC on one 256-stream launch, " + PMC_TRAFFIC_DEFAULT["source"] + ")" if traffic else None,
            "launches_per_step": lanes,
            "kernel_ms": round(k_ms / lanes, 4),
            "kernel_ms_note": "mean duration of one stft_scan launch (HIP events on its stream); with more than one lane the launches"
                              " of different lanes run concurrently with each other's scan and detect kernels, which stretches each of them",
            "detect_kernel_ms": round(ms_detect / max(1, args.steps) / lanes, 4),
            "algorithmic_bytes_per_launch": samples_per_step_gpu * bytes_per_sample // lanes,
            "whole_path_frac": round(value * 1e6 / world * bytes_per_sample / 1e9 / HBM_PEAK_GBS, 4),
        },
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline and not u8:
        out["cpu_baseline"], out["parity"] = cpu_baseline(args, iq, rec, kw, blen, n_seg, nperseg)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


def cpu_baseline(args, iq, rec, kw, blen, n_seg, nperseg):
    """Oracle on the host cores over a bounded sample of the same IQ bits."""
    import numpy as np

    from oracle import cpu_bench

    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    workers = max(1, min(cores, 64))
    n = args.cpu_streams or min(iq.shape[0], max(2 * workers, 32))
    host = iq[:n].cpu().numpy()
    tmpdir = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    path = os.path.join(tmpdir, f"rt_bench_iq_{os.getpid()}.npy")
    np.save(path, host)
    try:
        res = cpu_bench.run(path, n, kw, workers)
    finally:
        os.unlink(path)
    samples = n * n_seg * nperseg
    per_core = n_seg * nperseg / (sum(res["per_stream_s"]) / n) / 1e6
    base = {
        "value": round(samples / res["wall_s"] / 1e6, 2),
        "unit": "MSamples/s",
        "cores": workers,
        "kind": "port",
        "sample": f"{n} of the {iq.shape[0]} streams (same IQ bits, {blen} samples each), one oracle process per core; "
        f"single-core rate {per_core:.1f} MSamples/s",
    }
    # parity of the GPU records against the oracle on the sampled streams
    checked = mismatched = 0
    for s in range(min(n, max(args.parity_streams, 1))):
        mine = rec[rec["stream"] == s]
        got = [(int(r["fi"]), int(r["start"]), int(r["end"]), not bool(r["shadowed"])) for r in mine]
        checked += 1
        if got != res["results"][s]:
            mismatched += 1
    return base, {"streams_checked": checked, "streams_mismatched": mismatched, "fields": "count, bin, start, end, shadow verdict"}


if __name__ == "__main__":
    main()
