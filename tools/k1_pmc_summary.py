#!/usr/bin/env python3
"""rocprofv3 --pmc pass over tools/prof_k1.py -> profiles/rNN_pmc_k1.json (what bench.py's vectorise.roofline reads).
Usage: python3 tools/k1_pmc_summary.py <counter_collection.csv> <claims per launch> <ms per launch> [source]"""
import collections
import csv
import json
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
claims, ms = int(sys.argv[2]), float(sys.argv[3])
c = collections.defaultdict(list)
for r in rows:
    if "embed_kernel" in r["Kernel_Name"]:
        c[r["Counter_Name"]].append(float(r["Counter_Value"]))
med = {k: statistics.median(v) for k, v in c.items()}  # with the token memo on: launch 1 fills the table (cold), the median launch reads it (warm)
first = {k: v[0] for k, v in c.items()}
simds = 256 * 4
cycles = med["GRBM_GUI_ACTIVE"] / 8  # the counter is summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back)
out = {"source": sys.argv[4] if len(sys.argv) > 4 else "",
       "launches": len(c["SQ_INSTS_VALU"]), "claims_per_launch": claims, "ms_per_launch_under_the_profiler": ms,
       "counters_median_per_launch": med,
       "first_launch_cold_memo": {"valu_wave_instructions_per_claim": first["SQ_INSTS_VALU"] / claims,
                                  "valu_busy_fraction": first["SQ_ACTIVE_INST_VALU"] * 4 / (simds * first["GRBM_GUI_ACTIVE"] / 8)},
       "embed_kernel": {
           "valu_wave_instructions_per_claim": med["SQ_INSTS_VALU"] / claims,
           "salu_instructions_per_claim": med["SQ_INSTS_SALU"] / claims,
           "lds_instructions_per_claim": med["SQ_INSTS_LDS"] / claims,
           "gpu_cycles": cycles, "clock_ghz": cycles / (ms * 1e6),
           # SQ_ACTIVE_INST_VALU counts quad-cycles, summed over the SIMDs: the share of all SIMD-cycles with a vector instruction executing
           "valu_busy_fraction": med["SQ_ACTIVE_INST_VALU"] * 4 / (simds * cycles),
           "cycles_per_valu_instruction": med["SQ_ACTIVE_INST_VALU"] * 4 / med["SQ_INSTS_VALU"],
           "issue_stall_fraction_of_wave_cycles": med["SQ_WAIT_INST_ANY"] / med["SQ_WAVE_CYCLES"]}}
print(json.dumps(out, indent=1))
