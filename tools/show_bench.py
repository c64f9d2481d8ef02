#!/usr/bin/env python3
"""one line per bench.py JSON line of a file: the figures that matter"""
import json, sys
for ln in open(sys.argv[1]):
    ln = ln.strip()
    if not ln.startswith("{"):
        continue
    d = json.loads(ln); r = d["roofline"]
    print(d["config"]["workload"][:44], "| n_gpus", d["n_gpus"], "value", d["value"], "ms/step", d["ms_per_step"], "frac", r["frac"], "iso_ms", r.get("kernel_ms_isolated"),
          "frac_iso", r.get("frac_isolated"), "whole", r["whole_path_frac"], "fallbacks", d["config"]["fallbacks"], "records", d["config"]["records_per_step"],
          "parity_bad", d.get("parity", {}).get("streams_mismatched"), "cpu", d.get("cpu_baseline", {}).get("value"))
