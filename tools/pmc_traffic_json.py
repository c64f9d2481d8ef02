#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes of tools/profile_round.sh (FETCH_SIZE, WRITE_SIZE over tools/profile_traffic.py)
into profiles/pmc_traffic.json: corrected HBM bytes per scan launch of the default workload, tied to the scan kernel's
machine code by its sha256 (bench.py quotes `roofline.traffic` only while that hash matches the library it runs).

usage: pmc_traffic_json.py <fetch-dir> <write-dir> <traffic.json printed by profile_traffic.py> <out.json> [source note]

Correction (MI355X_MICROARCH.md, HBM / rocprofv3): the counters are in KiB; FETCH_SIZE under-counts wide reads on
gfx950, so it is calibrated on the kernel's own load stream (stft_scan<1,3>), whose byte count is known exactly.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def means(d, counter):
    acc = defaultdict(list)
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] == counter and "rt::stft_scan" in row.get("Kernel_Name", ""):
                    acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    fetch_dir, write_dir, traffic_json, out = sys.argv[1:5]
    note = sys.argv[5] if len(sys.argv) > 5 else ""
    import bench  # scan_kernel_sha256 (same function the bench checks with)

    with open(traffic_json) as f:
        exact = json.loads(f.read().strip().splitlines()[-1])
    fetch, write = means(fetch_dir, "FETCH_SIZE"), means(write_dir, "WRITE_SIZE")
    cal = [v for k, v in fetch.items() if "<1, 3" in k]
    scan_f = [v for k, v in fetch.items() if "<1, 0" in k]
    scan_w = [v for k, v in write.items() if "<1, 0" in k]
    if not (cal and scan_f and scan_w):
        raise SystemExit(f"counter rows missing: fetch {list(fetch)} write {list(write)}")
    ratio = cal[0] * 1024 / exact["scan_read_bytes_exact"]
    read_b = scan_f[0] * 1024 / ratio
    write_b = scan_w[0] * 1024
    doc = {
        "workload": f"config2: {exact['streams']} streams x {exact['segments'] * exact['nperseg']} samples, nperseg {exact['nperseg']}, one launch",
        "scan_kernel": "rt::stft_scan<1, 0, false, true>",
        "scan_kernel_sha256": bench.scan_kernel_sha256(),
        # host-side launch geometry changes bytes per launch too (chunk length -> halo segments and workgroups): the bench compares
        # these with what its own handle reports (rt_call_info.segs_per_chunk) besides the machine-code hash
        "segs_per_chunk": exact.get("segs_per_chunk"),
        "streams": exact["streams"],
        "segments": exact["segments"],
        "bytes_per_launch_256_streams": int(round(read_b + write_b)),
        "read_bytes": int(round(read_b)),
        "write_bytes": int(round(write_b)),
        "algorithmic_bytes": exact["algorithmic_bytes"],
        "traffic_over_algorithmic": round((read_b + write_b) / exact["algorithmic_bytes"], 4),
        "fetch_size_kib_scan": scan_f[0],
        "write_size_kib_scan": scan_w[0],
        "fetch_size_kib_load_only": cal[0],
        "load_only_exact_bytes": exact["scan_read_bytes_exact"],
        "fetch_counter_over_exact": round(ratio, 4),
        "source": note or "tools/profile_round.sh",
    }
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
        f.write("\n")
    print(json.dumps(doc))


if __name__ == "__main__":
    main()
