#!/usr/bin/env python3
"""one line per bench.py JSON line of a file: the figures that matter (roofline.frac / kernel_ms = the scan launch alone,
one lane, a pass after the timed region; frac_concurrent / kernel_ms_concurrent = per launch inside the timed region)"""
import json, sys
for ln in open(sys.argv[1]):
    ln = ln.strip()
    if not ln.startswith("{"):
        continue
    d = json.loads(ln); r = d["roofline"]
    print(d["config"]["workload"][:44], "| n_gpus", d["n_gpus"], "value", d["value"], "ms/step", d["ms_per_step"], "frac (launch alone)", r["frac"], "kernel_ms", r.get("kernel_ms"),
          "frac_concurrent", r.get("frac_concurrent"), "kernel_ms_concurrent", r.get("kernel_ms_concurrent"), "whole", r["whole_path_frac"],
          "fallbacks", d["config"]["fallbacks"], "records", d["config"]["records_per_step"],
          "parity_bad", d.get("parity", {}).get("streams_mismatched"), "cpu", d.get("cpu_baseline", {}).get("value"))
    for blk in ("other_configs", "sharded_configs"):
        for o in d.get(blk) or []:
            print("   ", blk, o.get("name"), {k: o.get(k) for k in ("value", "ms_per_step", "mode", "kernel", "kernel_ms", "frac", "whole_path_frac", "parity_streams_mismatched",
                                                                     "per_rank_ms", "speedup_vs_n1_reference", "skipped", "failed") if o.get(k) is not None})
