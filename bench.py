#!/usr/bin/env python3
"""Headline benchmark: IQ MSamples/s analysed by the signal-analysis path.

Default workload (BASELINE.json configs[1], "config2"): per GPU, 256 synthetic complex64
streams at 2.048 MSPS, one second each (B = 2 048 000, T = 8000), nperseg 256 hamming,
4-8 sparse 15 ms pulses per stream, resident in HBM; with N > 1 every rank analyses its
own 256 streams (weak scaling).  A step = one pass of the whole path (fused STFT/scan
kernel + detect kernels + records on the host) over the rank's batch; consecutive steps
are pipelined two deep inside the library.

``--workload config3|config4|config5`` run the other BASELINE configurations as ONE fixed
stream population sharded over the ranks with ``shard.stream_range`` (strong scaling:
config 4 = 32 768 streams x 524 288 samples, config 5 = 8 192 streams of tag trains at
nperseg 4096).  A stream's content depends on (seed, global stream number) only, so the
population -- and the total number of records, which is printed -- is the same at every N.

There is no collective on the data path.  The control plane (rendezvous, barrier, the
max-over-ranks of one double, record counts) runs over gloo at every N.

``--gpus N`` without a launcher (no RANK / WORLD_SIZE in the environment) starts the N
ranks itself: N fresh child processes of this script, before anything here has touched a
GPU, one per GPU, rank 0's JSON line relayed; a failing rank fails the run.  Launched by
``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`` it is a rank.

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel (stft_scan) at 8
algorithmic bytes per IQ sample against the 8 TB/s HBM peak: `kernel_ms` / `achieved` /
`frac` are the launch ALONE (one lane: one launch over all streams of the rank per step,
HIP events on its stream, a pass after the timed region unless the timed region itself
ran one lane) -- the figure a one-lane rocprofv3 --kernel-trace average reproduces;
`kernel_ms_concurrent` / `frac_concurrent` are the per-launch figures of the timed region,
where the lanes' launches run beside each other.  `host_sinks` (outside the timed region,
one core) puts the host-side consumers next to the record rate the GPU path produced.
`cpu_baseline` is the oracle (port of the reference's SciPy/NumPy path) on this node's
host cores, N = 1 only; `parity` compares sampled streams with it.
"""
import argparse
import hashlib
import json
import os
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
BYTES_PER_SAMPLE = 8  # one complex64 read
PMC_TRAFFIC_FILE = os.path.join(REPO, "profiles", "pmc_traffic.json")  # written by tools/profile_round.sh

# BASELINE.json configs.  "streams" = per GPU (weak scaling); "total" = one population sharded over the ranks (strong)
WORKLOADS = {
    "config2": dict(streams=256, sample_rate=2048000, samples=2048000, nperseg=256, window="hamming", trains=False),
    "config3": dict(total=4096, sample_rate=2400000, samples=2400000, nperseg=1024, window="hann", trains=False),
    "config4": dict(total=32768, sample_rate=2048000, samples=524288, nperseg=256, window="hamming", trains=False),
    "config5": dict(total=8192, sample_rate=3200000, samples=3200000, nperseg=4096, window="hamming", trains=True),
}

# What `other_configs` times after the headline at N = 1 (one GPU's share of every BASELINE configuration + the reference's
# defaults with a real receiver's noise floor).  Same generator, same seed as --workload configN; lanes as bench defaults.
OTHER_CONFIGS = [
    ("config3", dict(streams=4096, sample_rate=2400000, samples=2400000, nperseg=1024, window="hann", trains=False, lanes=1,
                     what="BASELINE config 3, the whole population on one GPU")),
    ("config4", dict(streams=32768, sample_rate=2048000, samples=524288, nperseg=256, window="hamming", trains=False, lanes=1,
                     what="BASELINE config 4, all 32 768 streams on one GPU at B = 524 288 (137 GB resident): the N = 1 point of north_star's sharded curve")),
    ("config5_share", dict(streams=1024, sample_rate=3200000, samples=3200000, nperseg=4096, window="hamming", trains=True, lanes=1,
                           what="BASELINE config 5, the 1 024-stream share one GPU of eight analyses (tag trains)")),
    ("default_geometry_noise_floor", dict(streams=4096, sample_rate=300000, samples=300000, nperseg=256, window="hamming", trains=False, lanes=3,
                                          noise_dbw=-88.0, steps=40, settle=20,
                                          what="the reference's defaults (300 kS/s, nperseg 256, -90 dBW, 8-40 ms) with the noise floor at -88 dBW, 2 dB OVER "
                                               "the threshold (a real RTL-SDR): AUTO reaches the exact run-length pre-filter")),
    ("config2_uint8", dict(streams=256, sample_rate=2048000, samples=2048000, nperseg=256, window="hamming", trains=False, lanes=3, input="u8",
                           steps=200, settle=30,  # (half-millisecond steps: ten of them end before the clocks have settled)
                           what="config 2's geometry from the RTL-SDR wire format (interleaved uint8 I/Q, converted in the scan's load): a quarter of the bytes, "
                                "the same arithmetic -- bound by the step's instructions, not by HBM")),
    ("default_geometry_uint8_noise_floor", dict(streams=4096, sample_rate=300000, samples=300000, nperseg=256, window="hamming", trains=False, lanes=3, input="u8",
                                                threshold_dbw=-91.0, steps=40, settle=20,
                                                what="the deployment case: the reference's defaults from the uint8 wire format, the threshold 0.8 dB UNDER the quantisation "
                                                     "noise (-90.2 dBW per bin at 300 kS/s): AUTO reaches the exact run-length pre-filter")),
    # half the default nperseg -- a plausible station setting: a fused scan since round 6 (lane groups of eight lanes, csrc/rt_kernels.h: stft_scan<.., QS>)
    ("nperseg128_defaults", dict(streams=4096, sample_rate=300000, samples=300000, nperseg=128, window="hamming", trains=False, lanes=3,
                                 steps=40, settle=10,  # (2-ms steps: ten of them are 20 ms, one hiccup of the box is a quarter of that)
                                 what="the reference's defaults at fft_nperseg 128: the fused scan with lane groups of eight lanes, sparse path")),
    # twice the largest size of the rounds before: a fused scan since round 6 as well (csrc/rt_scan_wg.h)
    ("nperseg8192", dict(streams=512, sample_rate=3200000, samples=3200000, nperseg=8192, window="hamming", trains=False, lanes=1,
                         what="fft_nperseg 8192 at 3.2 MS/s: stft_wg (one workgroup per segment, radix 32 x 16 x 16 in registers, two LDS exchanges), sparse path")),
]


# What `sharded_configs` times after the weak-scaled headline when the bare command runs on N > 1 GPUs: north_star's own curve --
# ONE population sharded one-subset-per-GPU (shard.stream_range), no collective on the data path (the reference: one analyzer
# process per SDR, radiotracking/__main__.py:118-140).  Same generator and seed as `--workload config4|5`: a stream's samples
# depend on (seed, global stream number) only, so the population is the same at every N.
SHARDED_CONFIGS = [
    ("config4", dict(total=32768, sample_rate=2048000, samples=524288, nperseg=256, window="hamming", trains=False,
                     what="BASELINE config 4: 32 768 streams x 524 288 samples (137 GB) sharded over the GPUs -- north_star's 1/2/4/8 curve")),
    ("config5", dict(total=8192, sample_rate=3200000, samples=3200000, nperseg=4096, window="hamming", trains=True,
                     what="BASELINE config 5: 8 192 streams x 3.2 MS (210 GB) of dense tag trains sharded over the GPUs")),
]
N1_REFERENCE_FILE = os.path.join(REPO, "profiles", "n1_reference.json")  # the N = 1 points of those curves, measured by this script on one GPU


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--workload", default="config2", choices=sorted(WORKLOADS),
                    help="BASELINE.json configuration; config2 (default) is weak-scaled, the others shard one fixed population")
    ap.add_argument("--streams", type=int, default=None, help="streams per GPU (makes the run weak-scaled)")
    ap.add_argument("--total-streams", type=int, default=None, help="population size of a strong-scaled workload")
    ap.add_argument("--sample-rate", type=int, default=None)
    ap.add_argument("--seconds", type=float, default=None, help="buffer length per stream")
    ap.add_argument("--nperseg", type=int, default=None)
    ap.add_argument("--window", default=None)
    ap.add_argument("--mode", default="auto", choices=["auto", "dense", "sparse", "prefilter", "runfilter"])
    ap.add_argument("--segs-per-chunk", type=int, default=0)
    ap.add_argument("--group-detect", choices=["auto", "on", "off"], default="auto",
                    help="sparse detection by groups of candidate lists (detect_group): the library's rule (from 1 024 streams per handle on), or forced for A/B runs")
    ap.add_argument("--hot-capacity", type=int, default=0,
                    help="candidate cells kept per (stream, bin mod 16 bucket) and call (rt_config.hot_capacity; 0 = default: one bin row "
                         "times max(1, nperseg / 1024), at most 8192; up to 16384 fits the detection's LDS sort)")
    ap.add_argument("--input", default="c64", choices=["c64", "u8"],
                    help="IQ representation in HBM: complex64 (the BASELINE workload) or the RTL-SDR wire format "
                         "(interleaved uint8, converted inside the scan kernel; SURVEY 8(f) rank 1)")
    ap.add_argument("--threshold-dbw", type=float, default=None,
                    help="signal_threshold_dbw (default: the reference's -90, or -80 with --input u8)")
    ap.add_argument("--noise-dbw", type=float, default=None,
                    help="noise floor of the synthetic streams as a PSD per bin (default: sigma 1e-5 per component = -160 dBW at "
                         "2.048 MS/s, far under the threshold); e.g. -89 puts it 1 dB OVER the reference's -90 dBW threshold -- the "
                         "regime of a real RTL-SDR, analysed through the run-length pre-filter")
    ap.add_argument("--noisy-streams", type=int, default=0,
                    help="with --noise-dbw: only the first K streams of every rank get that noise floor (a few noisy SDRs in a "
                         "batch: AUTO re-runs just those on the dense path)")
    ap.add_argument("--trains", action="store_true", default=None,
                    help="BASELINE config 5 style input: 8-16 tags per stream, pulse trains 10-38 ms, period 0.1-1 s")
    ap.add_argument("--settle", type=int, default=30,
                    help="untimed steps run once during set-up, before the W warm-up steps: the GPU's clocks need ~10 "
                         "launches after idle to settle (0.85 -> 0.77 ms per scan launch), whatever W the caller picks")
    ap.add_argument("--lanes", type=int, default=None,
                    help="stream groups per GPU, each on its own handle / HIP stream (detect of one group overlaps the "
                         "scan of the other); 1 = one launch sequence per step.  Default: 3 up to nperseg 512 (1 from 16 384 streams per GPU on), 1 from "
                         "nperseg 1024 on (those scans are chip-filling grids of persistent workgroups: a second lane's "
                         "kernels wait behind them -- config 3: 650 k with two lanes, 676 k with one)")
    ap.add_argument("--isolated-steps", type=int, default=50,
                    help="steps of the one-lane pass after the timed region that measures the scan launch alone (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-streams", type=int, default=0, help="streams in the CPU-baseline sample (0 = auto)")
    ap.add_argument("--parity-streams", type=int, default=16)
    ap.add_argument("--other-configs", default="auto", choices=["auto", "on", "off"],
                    help="after the headline, time the other BASELINE configurations (3, 4, the config-5 share) and the reference's default "
                         "geometry with a noise floor over its threshold in the same run -> `other_configs` in the JSON line.  auto: on for the bare "
                         "default command at N = 1")
    ap.add_argument("--sharded-configs", default="auto", choices=["auto", "on", "off"],
                    help="after the headline, time BASELINE configs 4 and 5 as ONE population each sharded over the ranks (strong scaling, "
                         "north_star's curve) -> `sharded_configs` in the JSON line.  auto: on for the bare default command at N > 1")
    ap.add_argument("--sharded-budget-s", type=float, default=300.0, help="wall-clock budget of the whole --sharded-configs block")
    ap.add_argument("--other-steps", type=int, default=10, help="timed steps per configuration of --other-configs")
    ap.add_argument("--other-budget-s", type=float, default=240.0,
                    help="wall-clock budget of the whole --other-configs block: a configuration is only started while it is not spent")
    return ap.parse_args()


def resolve_workload(args, world):
    """-> dict(name, scaling, total (population or None), sample_rate, samples, nperseg, window, trains)"""
    w = dict(WORKLOADS[args.workload])
    custom = False
    for key, val in (("sample_rate", args.sample_rate), ("nperseg", args.nperseg), ("window", args.window), ("trains", args.trains)):
        if val is not None and val != w[key]:
            w[key] = val
            custom = True
    if args.seconds is not None:
        samples = int(round(args.seconds * w["sample_rate"]))
        custom |= samples != w["samples"]
        w["samples"] = samples
    elif args.sample_rate is not None and args.workload == "config2":
        w["samples"] = w["sample_rate"]  # one second (the reference's default callback length)
    if args.streams is not None:  # per GPU: weak scaling
        w.pop("total", None)
        w["streams"] = args.streams
    if args.total_streams is not None:
        w.pop("streams", None)
        w["total"] = args.total_streams
    w["scaling"] = "strong" if "total" in w else "weak"
    # name by geometry (the flags of tools/run_configs.sh describe BASELINE configs too)
    by_geometry = {(v["sample_rate"], v["nperseg"], v["window"], v["samples"], v["trains"]): k for k, v in WORKLOADS.items()}
    w["name"] = by_geometry.get((w["sample_rate"], w["nperseg"], w["window"], w["samples"], w["trains"]), "custom")
    if getattr(args, "lanes", 0) is None:
        # three lanes up to nperseg 512 -- while a rank's launches are small enough to have ends worth filling: at config-4 geometry
        # two lanes +2 ... +3 % over one up to 8 192 streams, equal at 16 384, -3 % with all 32 768 on one GPU; three over two +1.5 %
        # at config 2, +3 ... 5 % at 1 024 streams, -0.7 ... +2 % at 4 096, +1 % at 8 192, and +3 ... 4 % on the reference's defaults
        # under a noise floor; four lanes -16 % at config 2 (profiles/r05_n_lanes_by_batch_size.txt)
        # (the rule itself: pyradiotracking_amd.analyze.default_lanes, what BatchSignalAnalyzer(lanes="auto") takes)
        from pyradiotracking_amd.analyze import default_lanes

        per_rank = w["streams"] if "streams" in w else -(-w["total"] // max(1, world))
        args.lanes = default_lanes(w["nperseg"], per_rank)
    return w


SCAN_KERNEL_SYMBOL = "_ZN2rt9stft_scanILi1ELi0ELb0ELb1ELi0EEEvNS_10StftParamsE"  # rt::stft_scan<1, 0, false, true, 0>: the default workload's scan


SCAN_KERNEL_SYMBOL_U8 = "_ZN2rt9stft_scanILi1ELi0ELb1ELb0ELi0EEEvNS_10StftParamsE"  # rt::stft_scan<1, 0, true, false, 0>: the sparse scan of uint8 input at nperseg 256
PMC_VALU_FILE = os.path.join(REPO, "profiles", "pmc_valu_u8.json")  # written by tools/r6/u8_valu.sh
# The roofline of the uint8 path: vector-instruction issue.  A wave64 v_fma_f32 takes a SIMD 2 cycles (MI355X_MICROARCH.md, "Per-instruction
# cycle constants"); 256 CUs x 4 SIMDs at the 2.4 GHz engine clock issue at most 1 024 x 2.4e9 / 2 wave-level vector instructions a second.
VALU_PEAK_GINST = 1024 * 2.4 / 2.0  # 1 228.8 G wave-instructions/s


def valu_roofline(block, kernel_ms):
    """`roofline`-style object for a uint8 line: the scan kernels' wave-level vector instructions per step (SQ_INSTS_VALU of the
    committed PMC pass over the same geometry, tools/r6/u8_valu.sh) / the step's scan time, against the chip's issue peak --
    quoted only while the uint8 scan kernel's machine code is the one that was counted."""
    try:
        with open(PMC_VALU_FILE) as f:
            doc = json.load(f).get(block)
    except (OSError, ValueError):
        doc = None
    if not doc or not doc.get("insts_valu_per_step"):
        return {"bound": "valu", "frac": None, "note": "profiles/pmc_valu_u8.json has no entry for this block (run tools/r6/u8_valu.sh)"}
    if doc.get("scan_kernel_sha256") != scan_kernel_sha256(symbol=SCAN_KERNEL_SYMBOL_U8):
        return {"bound": "valu", "frac": None, "note": "the uint8 scan kernel's machine code differs from the one profiles/pmc_valu_u8.json was counted on (run tools/r6/u8_valu.sh)"}
    insts = float(doc["insts_valu_per_step"])
    achieved = insts / (kernel_ms * 1e-3) / 1e9 if kernel_ms and kernel_ms > 0 else 0.0
    c = doc.get("counters_per_step", {})
    out = {"bound": "valu", "achieved": round(achieved, 1), "peak": round(VALU_PEAK_GINST, 1), "unit": "G wave-instructions/s", "frac": round(achieved / VALU_PEAK_GINST, 4),
           "insts_valu_per_step": int(insts), "insts_valu_per_sample": round(insts * 64 / doc["workload"]["samples_per_step"], 2),
           "peak_note": "1 024 SIMDs x 2.4 GHz / 2 cycles per wave64 vector instruction (MI355X_MICROARCH.md)",
           "source": "SQ_INSTS_VALU per step, " + doc.get("source", "profiles/pmc_valu_u8.json")}
    if c.get("SQ_WAVE_CYCLES"):
        # where a wave's time goes in the counted pass (quad-cycles over all waves): issuing, waiting on s_waitcnt / barriers, stalled at issue
        out["wave_time_shares"] = {k: round(c[n] / c["SQ_WAVE_CYCLES"], 3) for k, n in (("active_inst_any", "SQ_ACTIVE_INST_ANY"), ("active_inst_valu", "SQ_ACTIVE_INST_VALU"),
                                                                                         ("wait_any", "SQ_WAIT_ANY"), ("wait_inst_any", "SQ_WAIT_INST_ANY")) if c.get(n) is not None}
    return out


def scan_kernel_sha256(lib_path=None, symbol=SCAN_KERNEL_SYMBOL):
    """sha256 of the MACHINE CODE of the default workload's scan kernel inside the built library (the gfx950 code object of
    the offload bundle, the symbol's bytes in .text).  It ties profiles/pmc_traffic.json to the kernel that was measured:
    edits that cannot change that kernel's traffic (host code, other kernels, comments) leave it alone, anything that
    changes its instructions changes it.  None if the library or the symbol cannot be read."""
    import struct

    path = lib_path or os.environ.get("RT_ANALYZE_LIB") or os.path.join(REPO, "pyradiotracking_amd", "librt_analyze.so")
    try:
        with open(path, "rb") as f:
            data = f.read()
        i = data.find(b"__CLANG_OFFLOAD_BUNDLE__")
        if i < 0:
            return None
        n = struct.unpack_from("<Q", data, i + 24)[0]
        off, elf = i + 32, None
        for _ in range(n):
            o, sz, tl = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off:off + tl]
            off += tl
            if b"gfx950" in triple and sz:
                elf = data[i + o:i + o + sz]
        if elf is None or elf[:4] != b"\x7fELF":
            return None
        shoff, = struct.unpack_from("<Q", elf, 0x28)
        shentsize, shnum, _ = struct.unpack_from("<HHH", elf, 0x3A)
        secs = [struct.unpack_from("<IIQQQQIIQQ", elf, shoff + k * shentsize) for k in range(shnum)]  # name, type, flags, addr, offset, size, link, info, align, entsize
        for sec in secs:
            if sec[1] not in (2, 11):  # SHT_SYMTAB, SHT_DYNSYM
                continue
            strtab = secs[sec[6]]
            for k in range(sec[5] // 24):
                st_name, _info, _other, shndx, value, size = struct.unpack_from("<IBBHQQ", elf, sec[4] + 24 * k)
                end = elf.index(b"\0", strtab[4] + st_name)
                if elf[strtab[4] + st_name:end].decode(errors="replace") == symbol and size and 0 < shndx < shnum:
                    text = secs[shndx]
                    start = text[4] + (value - text[3])
                    return hashlib.sha256(elf[start:start + size]).hexdigest()
    except (OSError, struct.error, ValueError, IndexError):
        return None
    return None


def pmc_traffic(default_workload, lanes, segs_per_chunk=None):
    """HBM bytes per scan launch from the committed PMC passes -- only for the exact workload, the exact kernel
    machine code and the launch geometry (segments per chunk) they were measured on; otherwise null with the reason."""
    if not default_workload:
        return None, "PMC traffic is only on file for the default workload"
    try:
        with open(PMC_TRAFFIC_FILE) as f:
            doc = json.load(f)
    except (OSError, ValueError):
        return None, "profiles/pmc_traffic.json missing"
    have = scan_kernel_sha256()
    if have is None or doc.get("scan_kernel_sha256") != have:
        return None, "the scan kernel's machine code differs from the one profiles/pmc_traffic.json was measured on (run tools/profile_round.sh)"
    if doc.get("segs_per_chunk") is not None and segs_per_chunk is not None and int(doc["segs_per_chunk"]) != int(segs_per_chunk):
        return None, (f"this run's chunk length ({segs_per_chunk} segments) differs from the one profiles/pmc_traffic.json was measured with "
                      f"({doc['segs_per_chunk']}): host-side launch geometry changes bytes per launch (run tools/profile_round.sh)")
    return int(doc["bytes_per_launch_256_streams"]) // lanes, "bytes/launch: PMC FETCH_SIZE (calibrated on the kernel's own load stream) + WRITE_SIZE on one 256-stream launch; " + doc.get("source", "profiles/pmc_traffic.json")


_ORIG_AFFINITY = None  # the cores the job had before the rank pinned itself (the CPU baseline's workers get them back)


def affinity_summary(plan_entry):
    """A rank's core set for the JSON line: compact cpulist text, NUMA node, whether the mask is in force."""
    cpus = sorted(plan_entry.get("cpus") or [])
    runs, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        runs.append(str(cpus[i]) if i == j else f"{cpus[i]}-{cpus[j]}")
        i = j + 1
    return {"cpulist": ",".join(runs), "n": len(cpus), "numa_node": plan_entry.get("numa_node"), "pinned": bool(plan_entry.get("pinned")),
            "how": plan_entry.get("how")}


def _tail(path, n=25):
    try:
        with open(path, errors="replace") as f:
            return "".join(f.readlines()[-n:])
    except OSError:
        return ""


def spawn_ranks(args):
    """``--gpus N`` started by hand (no launcher): run the N ranks as N fresh child processes of this script, one per
    GPU, rendezvous on 127.0.0.1.  This process has not imported torch and never touches a GPU (every GPU question is
    asked in a child); it relays rank 0's output.  It cannot hang on a rank that dies: all children are polled, the
    first non-zero exit stops the others by their exact PIDs and is reported with that rank's stderr tail (the
    reference's supervisor does the same for its analyzer processes, __main__.py:152-190)."""
    import socket
    import subprocess
    import tempfile
    import threading
    import time

    share = os.environ.get("RT_BENCH_SHARE_GPU") == "1"
    # the lease has N GPUs?  Asked in a child (HIP runtime only, no torch), answered in a second or two
    try:
        probe = subprocess.run([sys.executable, "-c", "from pyradiotracking_amd import _native; print(_native.device_count())"],
                               env=dict(os.environ, RT_NO_TORCH="1"), cwd=REPO, capture_output=True, text=True, timeout=120)
        n_dev = int(probe.stdout.strip().splitlines()[-1]) if probe.returncode == 0 and probe.stdout.strip() else -1
    except (subprocess.TimeoutExpired, ValueError):
        n_dev, probe = -1, None
    if n_dev < 0:
        raise SystemExit(f"bench.py --gpus {args.gpus}: could not count the GPUs: {(probe.stderr if probe else 'timed out')[-500:]}")
    if n_dev < (1 if share else args.gpus):
        raise SystemExit(f"bench.py --gpus {args.gpus}: this box has {n_dev} GPU(s)")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs, errs = [], []
    tmp = tempfile.mkdtemp(prefix="rt_bench_ranks_")
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        errs.append(os.path.join(tmp, f"rank{r}.err"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=open(errs[-1], "w"), text=True))
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.monotonic() + float(os.environ.get("RT_BENCH_RANKS_TIMEOUT_S", "1500"))
    failed = None  # (rank, exit code) of the first rank that failed
    while failed is None and any(pr.poll() is None for pr in procs):
        for r, pr in enumerate(procs):
            rc = pr.poll()
            if rc is not None and rc != 0:
                failed = (r, rc)
                break
        if failed is None and time.monotonic() > deadline:
            failed = (-1, "timeout")
        if failed is None:
            time.sleep(0.2)
    if failed is None:
        bad = [(r, pr.returncode) for r, pr in enumerate(procs) if pr.returncode != 0]
        failed = bad[0] if bad else None
    if failed is not None:
        for pr in procs:  # the others: by their exact PIDs, politely, then not
            if pr.poll() is None:
                pr.terminate()
        t_end = time.monotonic() + 20
        for pr in procs:
            try:
                pr.wait(timeout=max(0.1, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                pr.kill()
                pr.wait()
    reader.join(timeout=10)
    sys.stdout.write("".join(out0))
    sys.stdout.flush()
    for r, path in enumerate(errs):  # rank 0's stderr is the run's own; the others' only matter when something failed
        if r == 0 or failed is not None:
            text = _tail(path, 40 if failed is not None else 1000)
            if text:
                sys.stderr.write(text if r == 0 and failed is None else f"---- rank {r} stderr (tail) ----\n{text}")
    if failed is not None:
        raise SystemExit(f"bench.py --gpus {args.gpus}: rank {failed[0]} failed (exit code {failed[1]}); the other ranks were stopped")


def device_identity(torch, local_rank):
    """What tells two ranks' GPUs apart in a SCALE record without any collective library: ordinal, PCI address, name."""
    props = torch.cuda.get_device_properties(local_rank)
    pci = None
    try:
        pci = f"{props.pci_domain_id:04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}.0"
    except AttributeError:
        pass
    uuid = getattr(props, "uuid", None)
    return {"ordinal": local_rank, "pci_bus_id": pci, "uuid": str(uuid) if uuid is not None else None, "name": props.name}


def main():
    args = parse()
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not launched:
        return spawn_ranks(args)  # before torch is imported: this process never initialises a GPU
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("RT_BENCH_FAIL_RANK") == str(rank) and launched:
        raise SystemExit(3)  # test hook: this rank dies before the rendezvous (tests/test_multigpu.py)
    # test hook (1-GPU box): all ranks on GPU 0 -- exercises the N > 1 code path (rendezvous, sharding, barrier,
    # max-over-ranks, rank-0 print) where only one GPU exists
    share_gpu = os.environ.get("RT_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    world = int(os.environ.get("WORLD_SIZE", "1")) if launched else 1
    # NUMA-local ranks, before this process has touched a GPU (the runtime's helper threads inherit the mask): the cores of the
    # NUMA node the rank's GPU hangs off, shared evenly with the other ranks of that node (pyradiotracking_amd/affinity.py;
    # the reference pins each analyzer with taskset, __main__.py:122-128).  RT_BENCH_NO_PIN=1 leaves the process where it is.
    from pyradiotracking_amd import affinity

    global _ORIG_AFFINITY
    try:
        _ORIG_AFFINITY = sorted(os.sched_getaffinity(0))
    except AttributeError:
        _ORIG_AFFINITY = None
    if os.environ.get("RT_BENCH_NO_PIN") == "1":
        cpu_plan = {"cpus": _ORIG_AFFINITY or [], "numa_node": None, "pci": None, "how": "not pinned (RT_BENCH_NO_PIN=1)", "pinned": False}
    else:
        # the plan is per NODE: this rank's place among the ranks of its own node (a second node's ranks 8 .. 15 drive its GPUs 0 .. 7)
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world))) if launched else 1
        local_index = int(os.environ.get("LOCAL_RANK", "0")) if launched else 0
        if not 0 <= local_index < local_world:
            local_index, local_world = rank, world
        cpu_plan = affinity.pin_rank(local_index, [0] * local_world if share_gpu else list(range(local_world)))
    if world != args.gpus and rank == 0:
        # a launcher's WORLD_SIZE is what actually runs; the JSON line reports it as n_gpus
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); running {world}", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        # control plane only (a barrier, one double, record counts): gloo at every N; the data path has no collective
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime

        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))  # a rank that never arrives: minutes, not half an hour

    from pyradiotracking_amd import shard, synth
    from pyradiotracking_amd.analyze import BatchSignalAnalyzer, window_coefficients

    wl = resolve_workload(args, world)
    fs, nperseg, blen = wl["sample_rate"], wl["nperseg"], wl["samples"]
    n_seg = blen // nperseg
    if wl["scaling"] == "strong":
        lo, hi = shard.stream_range(rank, world, wl["total"])
        total_streams = wl["total"]
    else:
        lo, hi = rank * wl["streams"], (rank + 1) * wl["streams"]
        total_streams = wl["streams"] * world
    S = hi - lo
    if S < 1:
        raise SystemExit(f"rank {rank} has no streams ({total_streams} streams over {world} ranks)")
    win = window_coefficients(wl["window"], nperseg)
    kw = dict(sample_rate=fs, fft_nperseg=nperseg, fft_window=wl["window"])
    u8 = args.input == "u8"
    dev = f"cuda:{local_rank}"
    seed = 1000
    if u8:
        # 8-bit front end: noise ~1.5 LSB rms, pulses 18..32 dB above the -80 dBW threshold
        kw["signal_threshold_dbw"] = -80.0
        iq_c = synth.make_batch_device(S, blen, fs, win, seed=seed, device=dev, noise_sigma=0.012, peak_dbw=(-62.0, -48.0),
                                       first_stream=lo)
        iq = synth.quantize_u8_device(iq_c)
        del iq_c
    else:
        extra = {}
        sigma = None
        if args.noise_dbw is not None:
            sigma = float(np.sqrt(10.0 ** (args.noise_dbw / 10.0) * fs / 2.0))  # PSD per bin = 2 sigma^2 / fs
            if not args.noisy_streams:
                extra["noise_sigma"] = sigma
        iq = synth.make_batch_device(S, blen, fs, win, seed=seed, device=dev, trains=wl["trains"], first_stream=lo, **extra)
        if sigma is not None and args.noisy_streams:
            k = min(S, args.noisy_streams)
            gen = torch.Generator(device=dev)
            gen.manual_seed(seed + 7)
            torch.view_as_real(iq)[:k].add_(torch.empty((k, blen, 2), dtype=torch.float32, device=dev).normal_(0.0, sigma, generator=gen))
    if args.threshold_dbw is not None:
        kw["signal_threshold_dbw"] = args.threshold_dbw
    stream = torch.cuda.current_stream()
    torch.cuda.synchronize()  # the IQ is complete before any lane's own stream reads it

    def analyzer(lanes):
        return BatchSignalAnalyzer(
            [str(i) for i in range(lo, hi)],
            sdr_callback_length=blen,
            gpu=local_rank,
            mode=args.mode,
            timing=True,
            segs_per_chunk=args.segs_per_chunk,
            hot_capacity=args.hot_capacity,
            hip_stream=stream.cuda_stream if lanes <= 1 else None,
            lanes=lanes,
            group_detect={"auto": None, "on": True, "off": False}[args.group_detect],
            **kw,
        )

    an = analyzer(args.lanes)

    def barrier():
        if world > 1:
            dist.barrier()

    def run(an, n_steps, serial=False):
        """n_steps full steps (enqueue + fetch each).  The handle keeps two calls in flight, so
        step i+1 is enqueued before step i is fetched: its scan overlaps step i's detect kernels,
        record copy and host-side fetch.  Every step's work completes inside this function.
        `serial`: one call in flight (a step is fetched before the next is enqueued): nothing runs beside a scan launch."""
        acc = [0.0, 0.0, 0]
        rec = info = None
        enq = an.enqueue_bytes if u8 else an.enqueue
        if n_steps and not serial:
            enq(iq)
        for i in range(n_steps):
            if serial or i + 1 < n_steps:
                enq(iq)
            rec = an.fetch_records()
            info = an.call_info()
            acc[0] += info.ms_stft
            acc[1] += info.ms_detect
            acc[2] += info.fell_back
        return rec, info, acc

    run(an, args.settle)  # set-up: clocks to steady state (not part of the W warm-up steps, never timed)
    rec, info, _ = run(an, args.warmup)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rec, info, (ms_stft, ms_detect, fell_back) = run(an, args.steps)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    n_records = int(len(rec))
    n_hot = int(info.n_hot)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([n_records, n_hot, fell_back], dtype=torch.int64)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        n_records_total, n_hot_total, fell_back = (int(v) for v in c)
    else:
        n_records_total, n_hot_total = n_records, n_hot

    # which GPU every rank ran on (ordinal, PCI address): a SCALE record shows N distinct devices without any collective library
    me = device_identity(torch, local_rank)
    me["rank"] = rank
    me["cpus"] = affinity_summary(cpu_plan)
    if world > 1:
        devices = [None] * world
        dist.all_gather_object(devices, me)
    else:
        devices = [me]

    samples_per_step_rank = S * n_seg * nperseg  # samples actually transformed (T6)
    total_samples = total_streams * n_seg * nperseg * args.steps
    value = total_samples / elapsed / 1e6
    lanes = max(1, args.lanes)
    k_ms = ms_stft / max(1, args.steps) / lanes  # mean duration of one launch (rank 0's launches)
    bytes_per_sample = 2 if u8 else BYTES_PER_SAMPLE
    bytes_per_launch = samples_per_step_rank * bytes_per_sample // lanes
    achieved = bytes_per_launch / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0

    default_workload = (wl["name"], S, args.segs_per_chunk, args.mode, args.input, args.threshold_dbw, args.noise_dbw, args.hot_capacity) == ("config2", 256, 0, "auto", "c64", None, None, 0)
    n_dense_streams = int(info.n_dense_streams)
    traffic, traffic_note = pmc_traffic(default_workload, 1, int(getattr(info, "segs_per_chunk", 0)) or None)  # per launch over all 256 streams, like kernel_ms

    # parity + CPU baseline (untimed).  N = 1: the oracle on the host cores over a bounded sample (the baseline) and the
    # records of >= 16 sampled streams against it; N > 1: every rank checks the first and last stream of its shard.
    parity = base = None
    if not args.no_cpu_baseline and u8:
        # the wire format: no CPU baseline (BASELINE's metric is quoted on complex64), but the sampled streams against the oracle
        if args.parity_streams > 0:
            _, parity = cpu_baseline(args, an, iq, rec, kw, blen, n_seg, nperseg, timed=False, spread=args.parity_streams)
    elif not args.no_cpu_baseline:
        if world == 1:
            base, parity = cpu_baseline(args, an, iq, rec, kw, blen, n_seg, nperseg, timed=True)
        else:
            _, parity = cpu_baseline(args, an, iq, rec, kw, blen, n_seg, nperseg, timed=False)
            c = torch.tensor([parity["streams_checked"], parity["streams_mismatched"]], dtype=torch.int64)
            dist.all_reduce(c, op=dist.ReduceOp.SUM)
            parity["streams_checked"], parity["streams_mismatched"] = int(c[0]), int(c[1])
            parity["note"] = "first and last stream of every rank's shard"

    an_decoder = an.decoder
    # the scan launch alone: one lane, one launch per step, nothing running beside it
    iso_ms = None
    if args.isolated_steps > 0:
        # (one lane and one call in flight: a handle without lanes runs its detection on a stream of its own, beside the next scan)
        an.close()
        del an
        an1 = analyzer(1)
        run(an1, max(5, args.settle))  # the CPU baseline left the GPU idle: clocks back to steady state first
        _, _, (ms1, _, _) = run(an1, args.isolated_steps, serial=True)
        iso_ms = ms1 / args.isolated_steps
        an1.close()

    # host sinks, outside the timed region (rank 0, one core): what the reference's path ends in -- Signal objects on the
    # queue (analyze.py:251, 280) -- and the vectorised record -> CSV route, next to the record rate the timed steps produced
    sinks = None
    if rank == 0:
        sinks = host_sinks(an_decoder, rec, [str(i) for i in range(lo, hi)], n_records_total * args.steps / elapsed)

    # (d) as SURVEY wrote it: the drop-in class on config 1 from host memory, as a latency (N = 1, default command)
    latency = None
    if rank == 0 and world == 1 and default_workload and not args.no_cpu_baseline:
        latency = single_stream_latency(local_rank)

    # the other BASELINE configurations and the reference's defaults under a real noise floor, timed in this same run
    others = None
    if world == 1 and (args.other_configs == "on" or (args.other_configs == "auto" and default_workload)):
        if args.isolated_steps <= 0:
            an.close()
            del an
        del iq, rec
        torch.cuda.empty_cache()
        others = other_configs(torch, args, local_rank)

    # north_star's curve in the driver's own multi-GPU run: configs 4 and 5 as one population each, sharded over these same ranks
    sharded = None
    if args.sharded_configs == "on" or (args.sharded_configs == "auto" and world > 1 and default_workload):
        if others is None:
            if args.isolated_steps <= 0:
                an.close()
                del an
            del iq, rec
            torch.cuda.empty_cache()
        try:
            sharded = sharded_configs(torch, dist if world > 1 else None, args, rank, world, local_rank)
        except Exception as e:  # the headline above has been measured: it is printed whatever happens to this block
            sharded = [{"name": "sharded_configs", "failed": f"{type(e).__name__}: {e}"[:500]}]

    conc_frac = achieved / HBM_PEAK_GBS
    if iso_ms:
        kernel_ms, kernel_frac = iso_ms, samples_per_step_rank * bytes_per_sample / (iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        kernel_note = (f"the stft_scan launch ALONE: one launch over all {S} streams of the rank per step, one lane, HIP events on its stream, "
                       + f"one call in flight, {args.isolated_steps} steps after the timed region"
                       + "; a one-lane rocprofv3 --kernel-trace average of the same command reproduces it")
    else:
        kernel_ms, kernel_frac = k_ms, conc_frac
        kernel_note = "isolated pass skipped (--isolated-steps 0): the concurrent per-launch figure stands in"
    roofline = {
        "bound": "hbm",
        "kernel": "stft_scan",
        "achieved": round(kernel_frac * HBM_PEAK_GBS, 1),
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": round(kernel_frac, 4),
        "traffic": traffic,
        "traffic_note": traffic_note,
        "kernel_ms": round(kernel_ms, 4),
        "kernel_ms_note": kernel_note,
        "algorithmic_bytes_per_launch": samples_per_step_rank * bytes_per_sample,
        "launches_per_step_timed_region": lanes,
        "kernel_ms_concurrent": round(k_ms, 4),
        "frac_concurrent": round(conc_frac, 4),
        "concurrent_note": "mean duration of one stft_scan launch over the timed region; with more than one lane each launch covers "
                           "1/lanes of the streams and runs beside the other lanes' scan and detect kernels, which stretches it",
        "detect_kernel_ms": round(ms_detect / max(1, args.steps) / lanes, 4),
        "whole_path_frac": round(value * 1e6 / world * bytes_per_sample / 1e9 / HBM_PEAK_GBS, 4),
        "frac_note": "`frac` / `achieved` / `kernel_ms`: the dominant kernel's launch ALONE (what a one-lane kernel trace reproduces); `whole_path_frac`: the "
                     "timed region itself -- `value` x 8 B per sample (per GPU) over the 8 TB/s peak, detection, fetch and launch gaps included",
    }

    if u8:
        # (2 bytes per sample: HBM does not bound this step -- the fraction of the roofline that does, where it has been counted)
        roofline["valu_roofline"] = valu_roofline("config2_uint8" if (wl["name"], S) == ("config2", 256) else "", kernel_ms)
    part = (f"{total_streams} streams sharded over {world} GPU(s) ({S} on rank 0)" if wl["scaling"] == "strong"
            else f"{S} streams/GPU")
    out = {
        "metric": "IQ MSamples/s analysed (STFT + detect + records), detected-signal parity vs CPU",
        "value": round(value, 1),
        "unit": "MSamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": wl["scaling"],
        "vs_baseline": None,
        "dtype": "f32" if not u8 else "f32 (uint8 IQ converted in the load)",
        "data": "synthetic",
        "config": {
            "workload": f"{wl['name']}: {part} x {fs} SPS x {blen} samples {'uint8 I/Q' if u8 else 'complex64'}, nperseg {nperseg} {wl['window']}, "
                        f"{'tag trains, 8-16 tags/stream' if wl['trains'] else '4-8 sparse 15 ms pulses/stream'}",
            "streams_total": total_streams,
            "streams_rank0": S,
            "samples_per_stream": blen,
            "segments_per_stream": n_seg,
            "segments_per_chunk": int(getattr(info, "segs_per_chunk", 0)),
            "mode": {1: "dense", 2: "sparse", 3: "prefilter", 4: "runfilter"}.get(info.mode_used, "?"),
            "fallbacks": fell_back,
            "records_per_step": n_records_total,
            "candidate_cells_per_step": n_hot_total,
            "sharding": "contiguous stream blocks per GPU (shard.stream_range), no collective; control plane on gloo",
            "scaling_curves": "this line: " + ("one population sharded over the GPUs (strong)" if wl["scaling"] == "strong" else "256 streams per GPU (weak: the driver's bare command)")
                              + "; north_star's sharded curve (config 4: 32 768 streams x 524 288 samples over N GPUs, strong): python bench.py --gpus N --workload config4; config 5: --workload config5",
            "lanes_per_gpu": args.lanes,
            "settle_steps": args.settle,
            "population_seed": seed,
            "noise_floor_dbw": args.noise_dbw,
            "noisy_streams_per_gpu": args.noisy_streams or (S if args.noise_dbw is not None else 0),
            "streams_rerun_dense_rank0": n_dense_streams,
            "threshold_dbw": kw.get("signal_threshold_dbw", -90.0),
            "hot_capacity": args.hot_capacity or "default",
            "devices": devices,
        },
        "roofline": roofline,
        "host_sinks": sinks,
    }
    if latency is not None:
        out.update(latency)
    if others is not None:
        out["other_configs"] = others
    if sharded is not None:
        out["sharded_configs"] = sharded
    if base is not None:
        out["cpu_baseline"] = base
    if parity is not None:
        out["parity"] = parity

    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        try:
            if not (sharded and any("failed" in c for c in sharded)):
                dist.barrier()
            dist.destroy_process_group()
        except Exception:  # (a rank that failed inside the sharded block has left the others in a collective: the line is out already)
            pass


def host_sinks(decoder, rec, device_names, records_per_s_produced):
    """Records/s through the host-side consumers, on one core, outside the timed region: (a) `decoder.signals` -- one
    `Signal` object per record, what the reference's path ends in (`signal_queue.put(Signal)`, analyze.py:251, 280);
    (b) `rows_from_analysis` + the native CSV formatter (pyradiotracking_amd/consume.py) -- no Python object per
    record.  `records_per_s_produced` is what the timed steps delivered as rt_record arrays."""
    import datetime

    import numpy as np

    from pyradiotracking_amd import consume
    from pyradiotracking_amd.match import datetime_to_us

    n = len(rec)
    out = {
        "timed_region_ends_at": "rt_record arrays in pinned host memory (rt_fetch); the sinks below run after it and are NOT part of `value`",
        "records_per_s_produced": round(records_per_s_produced, 1),
        "sample_records": int(n),
        "cores": 1,
    }
    if n == 0:
        return out
    ts0 = [datetime.datetime(2024, 1, 1)] * len(device_names)
    ts0_us = [datetime_to_us(ts0[0])] * len(device_names)
    kept = rec[rec["shadowed"] == 0]

    def rate(fn, count):
        reps, t = 0, 0.0
        while t < 0.5 and reps < 50:
            t0 = time.perf_counter()
            fn()
            t += time.perf_counter() - t0
            reps += 1
        return round(count * reps / t, 1)

    out["signal_objects_per_s"] = rate(lambda: decoder.signals(kept, device_names, ts0), len(kept))
    out["signal_batch_records_per_s"] = rate(lambda: decoder.signal_batch(kept, device_names, ts0), len(kept))
    # The native sinks work on arrays and, since round 6, on several cores (rt_host_set_threads; byte-identical output for any number).
    # A step's records are few thousand: the sample is repeated to a batch a station fleet would hand over (>= 1 048 576 records; for
    # the matcher every repetition a second later than the one before, or the copies would all fall into the first one's groups).
    # (the rank is pinned to its NUMA share: the sinks get the cores the job had, like the CPU baseline's workers)
    pinned_to = None
    try:
        if _ORIG_AFFINITY:
            pinned_to = sorted(os.sched_getaffinity(0))
            os.sched_setaffinity(0, _ORIG_AFFINITY)
    except (AttributeError, OSError):
        pinned_to = None
    try:
        reps = max(1, -(-1048576 // max(1, n)))
        big = np.concatenate([rec] * reps) if reps > 1 else rec
        n_big = int((big["shadowed"] == 0).sum())
        out["sink_batch_records"] = n_big
        out["sink_batch_records_in"] = int(len(big))  # (the shadow filter passes n_big of them: the rates below count rows OUT)
        consume.set_host_threads(1)
        out["csv_rows_per_s_one_thread"] = rate(lambda: consume.format_signals("csv", consume.rows_from_analysis(big, decoder, ts0_us), device_names), n_big)
        out["threads"] = consume.set_host_threads(0)
        out["csv_rows_per_s"] = rate(lambda: consume.format_signals("csv", consume.rows_from_analysis(big, decoder, ts0_us), device_names), n_big)
        out["csv_records_in_per_s"] = round(out["csv_rows_per_s"] * len(big) / max(1, n_big), 1)
        out["json_documents_per_s"] = rate(lambda: consume.format_signals("json", consume.rows_from_analysis(big, decoder, ts0_us), device_names), n_big)
        # the matcher: stations of four consecutive streams (the reference: four antennas per station, one SignalMatcher per station
        # process, match.py:21-50), every station's signals in time order, all stations in one native call
        from pyradiotracking_amd import match as rtm

        nd = 4
        n_st = -(-len(device_names) // nd)
        rows = consume.rows_from_analysis(big, decoder, ts0_us)
        rows["ts_us"] += (np.flatnonzero(big["shadowed"] == 0) // max(1, n)).astype(np.int64) * 1000000
        st = rows["device"] // nd
        order = np.lexsort((rows["ts_us"], st))
        msig = np.zeros(len(order), dtype=rtm.SIGNAL_DTYPE)
        msig["device"], msig["ts_us"], msig["duration_us"] = rows["device"][order] % nd, rows["ts_us"][order], rows["duration_us"][order]
        msig["frequency"], msig["avg"] = rows["frequency"][order], rows["avg_dbw"][order]
        offs = np.searchsorted(st[order], np.arange(n_st + 1))

        def fleet_rate(threads):
            consume.set_host_threads(threads)
            best = 0.0
            for _ in range(3):
                fl = rtm.MatcherFleet(n_st, nd, timeout_s=2.0, time_diff_s=0.0, bandwidth_hz=4000.0)
                t0 = time.perf_counter()
                fl.add(msig, offs)
                best = max(best, len(msig) / (time.perf_counter() - t0))
                fl.close()
            return round(best, 1)

        out["matched_signals_per_s_one_thread"] = fleet_rate(1)
        out["matched_signals_per_s"] = fleet_rate(0)
        out["matcher_stations"] = n_st
    finally:
        consume.set_host_threads(0)
        if pinned_to:
            try:
                os.sched_setaffinity(0, pinned_to)
            except OSError:
                pass
    out["note"] = ("signal_objects_per_s: Signal objects built from the records that pass the shadow filter (the reference's signal_queue.put payload); "
                   "signal_batch_records_per_s: the same nine fields per record as a lazy sequence (SignalBatch: columns at once, a Signal object when an element is "
                   "asked for -- what BatchSignalAnalyzer.process_batch returns by default); csv_rows_per_s / json_documents_per_s: records to `;`-separated CSV rows / "
                   "JSON documents through rows_from_analysis (rt_signal_rows_from_records) + rt_format_signals on `threads` host threads, counted in rows OUT (the batch is "
                   "sink_batch_records_in records of which the shadow filter passes sink_batch_records; csv_records_in_per_s: the same time per record IN); matched_signals_per_s: "
                   "the same signals through one SignalMatcher per station of four streams, all stations in one rt_match_add_many call")
    return out


def cpu_baseline(args, an, iq, rec, kw, blen, n_seg, nperseg, timed, spread=0):
    """The oracle on the host cores over a bounded sample of the same IQ bits (``timed``), and the GPU records of the
    sampled parity streams against it: count, bin, start, end, shadow verdict exact, the five dB figures within 0.1 dB.
    ``spread`` (untimed): that many streams, first, last and an even spread, instead of first and last only.
    The ONLY place of this file that touches ``oracle/`` (always as the checker / the reported baseline)."""
    import numpy as np

    from oracle import analyze_oracle, cpu_bench

    # the baseline is "the node's own host cores": the oracle's workers get the cores the job had before the rank pinned itself
    pinned_to = None
    try:
        if _ORIG_AFFINITY:
            pinned_to = sorted(os.sched_getaffinity(0))
            os.sched_setaffinity(0, _ORIG_AFFINITY)
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cores = os.cpu_count() or 1
    S = iq.shape[0]
    if timed:
        workers = max(1, min(cores, 64))
        n = min(S, args.cpu_streams or max(2 * workers, 32))
        # parity streams: first and last of the batch plus an even spread; the CPU sample = those + the next ones up to n
        k = min(S, max(args.parity_streams, 2))
        parity_ids = sorted({int(round(i * (S - 1) / max(1, k - 1))) for i in range(k)})
    elif spread:
        k = min(S, max(spread, 2))
        parity_ids = sorted({int(round(i * (S - 1) / max(1, k - 1))) for i in range(k)})
        workers, n = min(len(parity_ids), 8), len(parity_ids)
    else:
        workers = 2
        parity_ids = sorted({0, S - 1})
        n = len(parity_ids)
    rows = list(parity_ids) + [i for i in range(S) if i not in set(parity_ids)][: max(0, n - len(parity_ids))]
    host = iq[rows].cpu().numpy()
    if host.dtype == np.uint8:  # the RTL-SDR wire format: the oracle gets what the scan kernel makes of the bytes (fma(byte, 1/127.5, -1), float32)
        from pyradiotracking_amd import synth

        host = synth.u8_to_complex64_like_kernel(host)
    tmpdir = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    path = os.path.join(tmpdir, f"rt_bench_iq_{os.getpid()}.npy")
    np.save(path, host)
    try:
        res = cpu_bench.run(path, len(rows), kw, workers, parity=range(len(parity_ids)))
    finally:
        os.unlink(path)
        if pinned_to:
            try:
                os.sched_setaffinity(0, pinned_to)
            except OSError:
                pass
    base = None
    if timed:
        samples = len(rows) * n_seg * nperseg
        per_core = n_seg * nperseg / (sum(res["per_stream_s"]) / len(rows)) / 1e6
        base = {
            "value": round(samples / res["wall_s"] / 1e6, 2),
            "unit": "MSamples/s",
            "cores": workers,
            "kind": "port",
            "sample": f"{len(rows)} of the {S} streams (same IQ bits, {blen} samples each), one oracle process per core; "
            f"single-core rate {per_core:.1f} MSamples/s",
        }
    parity = compare_with_oracle(an.decoder, rec, parity_ids, res["results"])
    parity["streams"] = "first, last and an even spread of the batch" if (timed or spread) else "first and last of the shard"
    if timed:
        base.update(config1_single_core(analyze_oracle))
    return base, parity


def compare_with_oracle(dec, rec, parity_ids, results):
    """GPU records of the streams `parity_ids` against the oracle's rows (results[j] for parity_ids[j]).
    The bench analyses the same resident buffer in every step: its records are those of that buffer arriving after
    itself (look-back live), which is what the oracle's second pass over the parity streams returns."""
    import numpy as np

    checked = mismatched = 0
    worst_db = 0.0
    for j, s in enumerate(parity_ids):
        mine = rec[rec["stream"] == s]
        _, _, _, max_dbw, avg_dbw, std_db, noise_dbw, snr_db = dec.decode(mine)
        want = results[j]
        ok = [(int(r["fi"]), int(r["start"]), int(r["end"]), not bool(r["shadowed"])) for r in mine] == [w[:4] for w in want]
        if ok:
            for i, w in enumerate(want):
                for got, ref in zip((max_dbw[i], avg_dbw[i], std_db[i], noise_dbw[i], snr_db[i]), w[4:]):
                    d = 0.0 if (np.isnan(got) and np.isnan(ref)) else abs(float(got) - float(ref))
                    worst_db = max(worst_db, d)
                    ok &= d <= 0.1
        checked += 1
        mismatched += 0 if ok else 1
    return {"streams_checked": checked, "streams_mismatched": mismatched, "records_checked": int(sum(len(results[j]) for j in range(len(parity_ids)))),
            "fields": "count, bin, start, end, shadow verdict exact; max/avg/std/noise/snr within 0.1 dB", "worst_db_difference": round(worst_db, 6)}


def config1_stream():
    """BASELINE config 1: one 300 kS/s stream, one second, one 20 ms tone at +50 kHz from sample 90 000, A = 1e-2, seed 0."""
    from pyradiotracking_amd import synth

    return synth.make_stream(synth.StreamSpec(300000, 300000, [synth.Pulse(90000, 6000, 50000.0, 1e-2)]), seed=0)


def config1_single_core(oracle, reps=8):
    """the reference's own case -- ONE process on ONE core on config 1 -- as milliseconds of CPU per
    second of IQ (the reference is real-time while this stays under 1 000; its own log line is off by 10 x)."""
    import datetime

    iq = config1_stream()
    ts = datetime.datetime(2024, 1, 1)
    oa = oracle.OracleAnalyzer(device="0")
    oa.process(iq, ts)
    t0 = time.perf_counter()
    for _ in range(reps):
        every, kept = oa.process(iq, ts)
    ms = (time.perf_counter() - t0) / reps * 1e3
    return {"config1_single_core_ms_per_s": round(ms, 3), "config1_single_core_MSamples_per_s": round(0.3 / (ms * 1e-3), 2),
            "config1_signals": [len(every), len(kept)],
            "config1_note": "BASELINE config 1 (1 stream, 300 kS/s, 1-s buffer, one 20 ms tone at +50 kHz): the oracle in this process on one core, "
                            f"steady state (look-back live), mean of {reps} buffers; [signals before, after the shadow filter]"}


def single_stream_latency(gpu, reps=50):
    """The reference's real-time requirement (analyze.py:223-229: a callback must return within the buffer's own length)
    as a number: one config-1 buffer from HOST memory through SignalAnalyzer.process_samples -- clock book-keeping,
    upload, scan + detection, records -> Signal objects on the queue.  Median / worst of `reps` callbacks, milliseconds."""
    import queue

    import numpy as np

    from pyradiotracking_amd.analyze import SignalAnalyzer

    iq = config1_stream()
    q = queue.SimpleQueue()
    an = SignalAnalyzer("0", gpu=gpu, signal_queue=q)
    for _ in range(5):
        an.process_samples(iq)
    while not q.empty():
        q.get()
    lat = []
    for _ in range(reps):
        t0 = time.perf_counter()
        an.process_samples(iq)
        lat.append((time.perf_counter() - t0) * 1e3)
    n_msgs = 0
    while not q.empty():
        q.get()
        n_msgs += 1
    an._batch.close()
    return {"single_stream_latency_ms": round(float(np.median(lat)), 3), "single_stream_latency_worst_ms": round(float(max(lat)), 3),
            "single_stream_latency_note": f"BASELINE config 1 buffer (300 000 complex64 samples in host memory) through SignalAnalyzer.process_samples, "
                                          f"Signal objects on the queue; median and worst of {reps} callbacks ({n_msgs // reps} messages each); "
                                          "the reference needs < 1 000 ms per callback to keep up with the SDR"}


def measure_other(torch, name, spec, local_rank, steps, seed=1000, parity_streams=4):
    """One configuration of OTHER_CONFIGS on this GPU: its IQ generated in HBM, `steps` timed steps of the whole path
    (two calls in flight, the bench default lanes), sampled streams against the oracle, then the scan launch alone on one
    lane.  Everything it allocates is released before it returns."""
    import numpy as np

    from pyradiotracking_amd import synth
    from pyradiotracking_amd.analyze import BatchSignalAnalyzer, window_coefficients

    t_begin = time.perf_counter()
    S, fs, blen, nperseg, lanes = spec["streams"], spec["sample_rate"], spec["samples"], spec["nperseg"], spec["lanes"]
    n_seg = blen // nperseg
    dev = f"cuda:{local_rank}"
    win = window_coefficients(spec["window"], nperseg)
    extra = {}
    if spec.get("noise_dbw") is not None:
        extra["noise_sigma"] = float(np.sqrt(10.0 ** (spec["noise_dbw"] / 10.0) * fs / 2.0))
    kw = dict(sample_rate=fs, fft_nperseg=nperseg, fft_window=spec["window"])
    u8 = spec.get("input") == "u8"
    if u8:
        # 8-bit front end, as `--input u8`: noise ~1.5 LSB rms, pulses 18..32 dB above a -80 dBW threshold, converted inside the scan's load
        kw["signal_threshold_dbw"] = float(spec.get("threshold_dbw", -80.0))
        iq_c = synth.make_batch_device(S, blen, fs, win, seed=seed, device=dev, noise_sigma=0.012, peak_dbw=(-62.0, -48.0), first_stream=0)
        iq = synth.quantize_u8_device(iq_c)
        del iq_c
    else:
        iq = synth.make_batch_device(S, blen, fs, win, seed=seed, device=dev, trains=spec["trains"], first_stream=0, **extra)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_begin
    names = [str(i) for i in range(S)]
    stream = torch.cuda.current_stream()

    def analyzer(n_lanes):
        return BatchSignalAnalyzer(names, sdr_callback_length=blen, gpu=local_rank, mode="auto", timing=True,
                                   hip_stream=stream.cuda_stream if n_lanes <= 1 else None, lanes=n_lanes, **kw)

    def run(an, n_steps, serial=False):
        acc = [0.0, 0.0, 0]
        rec = info = None
        enq = an.enqueue_bytes if u8 else an.enqueue
        if n_steps and not serial:
            enq(iq)
        for i in range(n_steps):
            if serial or i + 1 < n_steps:
                enq(iq)
            rec = an.fetch_records()
            info = an.call_info()
            acc[0] += info.ms_stft
            acc[1] += info.ms_detect
            acc[2] += info.fell_back
        return rec, info, acc

    an = analyzer(lanes)
    run(an, spec.get("settle", 4))
    run(an, 2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rec, info, (ms_stft, ms_detect, fell_back) = run(an, steps)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    samples = S * n_seg * nperseg
    value = samples * steps / elapsed / 1e6
    mode_used = {1: "dense", 2: "sparse", 3: "prefilter", 4: "runfilter"}.get(info.mode_used, "?")
    # parity: sampled streams against the oracle (the same bits, copied back)
    _, parity = cpu_baseline(None, an, iq, rec, kw, blen, n_seg, nperseg, timed=False, spread=parity_streams)
    n_records = int(len(rec))
    an.close()
    del an
    # the scan launch alone: one lane, one call in flight
    an1 = analyzer(1)
    run(an1, 3)
    iso = 5
    _, info1, (ms1, _, _) = run(an1, iso, serial=True)
    an1.close()
    del an1, iq
    torch.cuda.empty_cache()
    kernel_ms = ms1 / iso
    two_scans = mode_used in ("prefilter", "runfilter")
    scan_name = ("stft_scan64" if nperseg == 4096 else "stft_scan" if nperseg in (32, 64, 128, 256, 512, 1024, 2048)
                 else "stft_wg" if nperseg in (8192, 16384)
                 else "general transform")
    return {
        "name": name,
        "workload": f"{S} streams x {fs} SPS x {blen} samples {'uint8 I/Q (2 B per sample)' if u8 else 'complex64'}, nperseg {nperseg} {spec['window']}, "
                    + ("tag trains, 8-16 tags/stream" if spec["trains"] else "4-8 sparse 15 ms pulses/stream")
                    + (f", noise floor {spec['noise_dbw']} dBW" if spec.get("noise_dbw") is not None else "")
                    + (f", threshold {spec['threshold_dbw']} dBW" if spec.get("threshold_dbw") is not None else "") + f" -- {spec['what']}",
        "value": round(value, 1),
        "unit": "MSamples/s",
        "steps": steps,
        "ms_per_step": round(elapsed / steps * 1e3, 4),
        "lanes": lanes,
        "mode": mode_used,
        "fallbacks": int(fell_back),
        "segments_per_chunk": int(getattr(info, "segs_per_chunk", 0)),
        "records_per_step": n_records,
        "kernel": (f"{scan_name}: threshold-bit scan + planning + listed scan of a step (first launch to scan event)" if two_scans else scan_name),
        "kernel_ms": round(kernel_ms, 4),
        "kernel_ms_note": f"one lane, one call in flight, HIP events on its stream, mean of {iso} steps after the timed ones",
        "algorithmic_bytes_per_launch": samples * (2 if u8 else BYTES_PER_SAMPLE),
        "frac": round(samples * (2 if u8 else BYTES_PER_SAMPLE) / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if kernel_ms > 0 else None,
        "whole_path_frac": round(value * 1e6 * (2 if u8 else BYTES_PER_SAMPLE) / 1e9 / HBM_PEAK_GBS, 4),
        "detect_kernel_ms": round(ms_detect / max(1, steps) / max(1, lanes), 4),
        "parity_streams_checked": parity["streams_checked"],
        "parity_streams_mismatched": parity["streams_mismatched"],
        "parity_records_checked": parity["records_checked"],
        "parity_worst_db_difference": parity["worst_db_difference"],
        "generate_s": round(t_gen, 1),
        "wall_s": round(time.perf_counter() - t_begin, 1),
        # uint8 input: HBM (2 B per sample) does not bound the step, vector-instruction issue does -- the fraction of THAT roofline
        **({"valu_roofline": valu_roofline(name, kernel_ms)} if u8 else {}),
    }


def measure_sharded(torch, dist, name, spec, rank, world, local_rank, steps, seed=1000):
    """One population of SHARDED_CONFIGS over the `world` ranks of this run: rank r generates and analyses the streams
    shard.stream_range(r, world, total) in its own HBM; a barrier and a device synchronisation either side of the timed steps,
    the MAX over ranks of the elapsed time, value = all ranks' samples / that.  Every rank checks the first and the last stream
    of its shard against the oracle.  Collective calls (gloo, control plane only) are made by every rank in the same order."""
    import numpy as np

    from pyradiotracking_amd import shard, synth
    from pyradiotracking_amd.analyze import BatchSignalAnalyzer, default_lanes, window_coefficients

    t_begin = time.perf_counter()
    total, fs, blen, nperseg = spec["total"], spec["sample_rate"], spec["samples"], spec["nperseg"]
    n_seg = blen // nperseg
    lo, hi = shard.stream_range(rank, world, total)
    S = hi - lo
    dev = f"cuda:{local_rank}"
    win = window_coefficients(spec["window"], nperseg)
    kw = dict(sample_rate=fs, fft_nperseg=nperseg, fft_window=spec["window"])
    lanes = int(spec.get("lanes") or default_lanes(nperseg, S))
    iq = synth.make_batch_device(S, blen, fs, win, seed=seed, device=dev, trains=spec["trains"], first_stream=lo)
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream()
    an = BatchSignalAnalyzer([str(i) for i in range(lo, hi)], sdr_callback_length=blen, gpu=local_rank, mode="auto", timing=True,
                             hip_stream=stream.cuda_stream if lanes <= 1 else None, lanes=lanes, **kw)

    def run(n_steps):
        rec = info = None
        fell = 0
        if n_steps:
            an.enqueue(iq)
        for i in range(n_steps):
            if i + 1 < n_steps:
                an.enqueue(iq)
            rec = an.fetch_records()
            info = an.call_info()
            fell += info.fell_back
        return rec, info, fell

    def barrier():
        if dist is not None:
            dist.barrier()

    run(int(spec.get("settle", 4)))
    run(2)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rec, info, fell_back = run(steps)
    torch.cuda.synchronize()
    mine_s = time.perf_counter() - t0  # this rank's own steps
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    _, parity = cpu_baseline(None, an, iq, rec, kw, blen, n_seg, nperseg, timed=False)  # first and last stream of the shard
    counts = [int(len(rec)), int(info.n_hot), int(fell_back), parity["streams_checked"], parity["streams_mismatched"], parity["records_checked"], S]
    per_rank_ms = [mine_s / steps * 1e3]
    worst_db = parity["worst_db_difference"]
    if dist is not None:
        t = torch.tensor([elapsed, worst_db], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, worst_db = float(t[0]), float(t[1])
        c = torch.tensor(counts[:6], dtype=torch.int64)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        counts[:6] = [int(v) for v in c]
        gathered = [None] * world
        dist.all_gather_object(gathered, (per_rank_ms[0], S))
        per_rank_ms = [round(g[0], 4) for g in gathered]
        streams_per_rank = [g[1] for g in gathered]
    else:
        per_rank_ms = [round(per_rank_ms[0], 4)]
        streams_per_rank = [S]
    mode_used = {1: "dense", 2: "sparse", 3: "prefilter", 4: "runfilter"}.get(info.mode_used, "?")
    an.close()
    del an, iq, rec
    torch.cuda.empty_cache()
    value = total * n_seg * nperseg * steps / elapsed / 1e6
    ref = None
    try:
        with open(N1_REFERENCE_FILE) as f:
            ref = json.load(f).get(name)
    except (OSError, ValueError):
        pass
    same_population = ref is not None and (spec["total"], fs, blen, nperseg) == (ref.get("total"), ref.get("sample_rate"), ref.get("samples"), ref.get("nperseg"))
    return {
        "name": name,
        "workload": f"{total} streams x {fs} SPS x {blen} samples complex64, nperseg {nperseg} {spec['window']}, "
                    + ("tag trains, 8-16 tags/stream" if spec["trains"] else "4-8 sparse 15 ms pulses/stream") + f" -- {spec['what']}",
        "scaling": "strong",
        "n_gpus": world,
        "value": round(value, 1),
        "unit": "MSamples/s",
        "steps": steps,
        "ms_per_step": round(elapsed / steps * 1e3, 4),
        "per_rank_ms": per_rank_ms,
        "streams_per_rank": streams_per_rank,
        "lanes_per_gpu": lanes,
        "mode": mode_used,
        "fallbacks": counts[2],
        "records_per_step": counts[0],
        "candidate_cells_per_step": counts[1],
        "whole_path_frac_per_gpu": round(value * 1e6 / world * BYTES_PER_SAMPLE / 1e9 / HBM_PEAK_GBS, 4),
        "parity_streams_checked": counts[3],
        "parity_streams_mismatched": counts[4],
        "parity_records_checked": counts[5],
        "parity_worst_db_difference": round(worst_db, 6),
        "parity_note": "first and last stream of every rank's shard against the oracle on the same bits",
        "speedup_vs_n1_reference": round(value / ref["value"], 3) if same_population and ref.get("value") else None,
        "n1_reference": ({"value": ref.get("value"), "source": ref.get("source")} if same_population else None),
        "wall_s": round(time.perf_counter() - t_begin, 1),
    }


def sharded_configs(torch, dist, args, rank, world, local_rank):
    """The `sharded_configs` block: every population of SHARDED_CONFIGS in order while the block's budget lasts.  Whether a
    configuration runs is decided on rank 0 and told to the others (every rank must make the same collective calls)."""
    out = []
    t0 = time.perf_counter()
    configs = SHARDED_CONFIGS
    if os.environ.get("RT_BENCH_SHARDED_CONFIGS_JSON"):  # (tests: a scaled-down table, same code path)
        configs = [(n, dict(sp)) for n, sp in json.loads(os.environ["RT_BENCH_SHARDED_CONFIGS_JSON"])]
    for name, spec in configs:
        go = [time.perf_counter() - t0 <= args.sharded_budget_s]
        if dist is not None:
            dist.broadcast_object_list(go, src=0)
        if not go[0]:
            out.append({"name": name, "skipped": f"the block's budget of {args.sharded_budget_s:.0f} s was spent before this configuration"})
            continue
        try:
            res = measure_sharded(torch, dist, name, spec, rank, world, local_rank, max(args.other_steps, int(spec.get("steps", 0))))
            err = None
        except Exception as e:  # (a rank that fails leaves the others in a collective: the rendezvous time-out ends the run, named below)
            res, err = None, f"{type(e).__name__}: {e}"[:500]
        if err is not None:
            # (the other ranks are inside this configuration's collectives: they leave them by the rendezvous time-out and land here too;
            # nothing further is attempted)
            out.append({"name": name, "failed": f"rank {rank}: {err}"})
            torch.cuda.empty_cache()
            break
        out.append(res)
    return out


def other_configs(torch, args, local_rank):
    """The `other_configs` block of the N = 1 line: every configuration of OTHER_CONFIGS, in order, while the block's
    wall-clock budget lasts (a configuration that does not run is listed as skipped, never silently dropped)."""
    out = []
    t0 = time.perf_counter()
    configs = OTHER_CONFIGS
    if os.environ.get("RT_BENCH_OTHER_CONFIGS_JSON"):  # (tests: a scaled-down table, same code path)
        configs = [(n, dict(sp)) for n, sp in json.loads(os.environ["RT_BENCH_OTHER_CONFIGS_JSON"])]
    for name, spec in configs:
        spent = time.perf_counter() - t0
        if spent > args.other_budget_s:
            out.append({"name": name, "skipped": f"the block's budget of {args.other_budget_s:.0f} s was spent ({spent:.0f} s) before this configuration"})
            continue
        try:
            out.append(measure_other(torch, name, spec, local_rank, max(args.other_steps, int(spec.get("steps", 0)))))
        except Exception as e:  # a configuration that fails (e.g. no memory for 137 GB on a shared GPU) must not take the headline with it
            out.append({"name": name, "failed": f"{type(e).__name__}: {e}"[:500]})
            torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    main()
