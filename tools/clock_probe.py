#!/usr/bin/env python3
"""Shader clock while the scan kernel runs (is the gap between the nperseg-256 kernel and its
load-only floor the clock the chip holds under the combined load?).

Samples the GPU's current sclk from sysfs (hwmon freq1_input, else pp_dpm_sclk) every ~10 ms on a thread while one of
    full   the product scan + detect, one lane, back to back
    load   the scan's load stream only (rt_calibrate_read)
runs for --seconds; prints mean / min / max MHz and the kernel's mean duration.  Run it once per library build
(RT_ANALYZE_LIB=...librt_var_alias.so for the arithmetic-only build of tools/variant.sh)."""
import argparse, glob, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def clock_reader():
    """-> function returning {card: sclk MHz}.  A box shows every GPU of its host in sysfs; the one that runs this
    process is the one whose clock goes up (the others idle at 0.1 .. 0.5 GHz) -- main() reports the busiest."""
    paths = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input"))
    if not paths:
        return None

    def rd():
        out = {}
        for p in paths:
            try:
                out[p.split("/")[4]] = int(open(p).read()) / 1e6
            except Exception:
                pass
        return out
    return rd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="full", choices=["full", "load"])
    ap.add_argument("--seconds", type=float, default=2.0)
    args = ap.parse_args()
    import torch
    from pyradiotracking_amd import synth
    from pyradiotracking_amd.analyze import BatchSignalAnalyzer, window_coefficients
    fs, nperseg, S = 2048000, 256, 256
    win = window_coefficients("hamming", nperseg)
    iq = synth.make_batch_device(S, fs, fs, win, seed=1000, device="cuda:0")
    an = BatchSignalAnalyzer([str(i) for i in range(S)], sdr_callback_length=fs, sample_rate=fs, mode="sparse", timing=True,
                             hip_stream=torch.cuda.current_stream().cuda_stream)
    rd = clock_reader()
    samples, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            try:
                samples.append(rd())
            except Exception:
                pass
            time.sleep(0.01)

    def step():
        if args.what == "full":
            an.enqueue(iq); an.fetch_records()
            return an.call_info().ms_stft
        t0 = time.perf_counter()
        an.native.calibrate_read(iq.data_ptr(), fs, fs)
        return (time.perf_counter() - t0) * 1e3

    for _ in range(40):
        step()  # clocks to steady state
    th = threading.Thread(target=poll) if rd else None
    if th:
        th.start()
    t_end, ms, n = time.perf_counter() + args.seconds, 0.0, 0
    while time.perf_counter() < t_end:
        ms += step(); n += 1
    stop.set()
    if th:
        th.join()
    lib = os.path.basename(os.environ.get("RT_ANALYZE_LIB", "librt_analyze.so"))
    clk = "sclk not readable on this box"
    if samples:
        cards = sorted({c for smp in samples for c in smp})
        mean = {c: sum(smp.get(c, 0.0) for smp in samples) / len(samples) for c in cards}
        # this process's GPU by its PCI address (the box shows every GPU of its host in sysfs)
        mine, how = None, "busiest"
        try:
            pr = torch.cuda.get_device_properties(0)
            addr = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            for c in cards:
                if os.path.basename(os.path.realpath(f"/sys/class/drm/{c}/device")) == addr:
                    mine, how = c, f"PCI {addr}"
        except Exception:
            pass
        busy = mine or max(cards, key=lambda c: mean[c])
        vals = [smp[busy] for smp in samples if busy in smp]
        clk = f"sclk of {busy} ({how}; {len(cards)} cards visible): mean {mean[busy]:.0f} min {min(vals):.0f} max {max(vals):.0f} MHz ({len(vals)} samples)"
    print(f"{lib:28s} {args.what:5s} kernel {ms/n:.4f} ms over {n} launches   {clk}")


if __name__ == "__main__":
    main()
