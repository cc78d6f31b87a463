#!/usr/bin/env python3
"""profiles/rNN_pmc_batched.json from the three PMC passes of tools/scripts/r02_k3_profiles.sh (launches of the last search call)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pmc_summary as P
G = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
rows = P.launches([f"{G}/pmc_k3_sq/sq_counter_collection.csv", f"{G}/pmc_k3_fetch/f_counter_collection.csv",
                   f"{G}/pmc_k3_write/w_counter_collection.csv"], "batched")
by = {}
for r in rows:
    by.setdefault(r["kernel"], []).append(r)
ncalls = len(by["select_batched_kernel"])
out = []
for k, rs in by.items():
    for r in rs[-(len(rs) // ncalls):]:
        d = dict(r)
        cyc = d["GRBM_GUI_ACTIVE"] / 8.0  # summed over 8 XCDs
        d["gpu_cycles"] = cyc
        d["mfma_busy_fraction"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024) if cyc else 0.0  # 1024 SIMDs
        d["hbm_read_bytes (FETCH_SIZE KiB x2)"] = d["FETCH_SIZE"] * 2048
        d["hbm_write_bytes (WRITE_SIZE KiB)"] = d["WRITE_SIZE"] * 1024
        out.append(d)
print(json.dumps({
    "source": "three rocprofv3 --pmc passes (tools/scripts/r02_k3_profiles.sh: SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES "
              "SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY | FETCH_SIZE | WRITE_SIZE) -- python3 tools/bench_batched.py 10000000 256 8 2",
    "workload": "10,000,000 claims x 256 batched queries, top_k=8 (config 3), bq_mode 2: prep -> scan_batched_max_kernel (sample pass, group maxima) "
                "-> seed -> scan_batched_tiled_kernel (full pass, B fragments in registers) -> select_batched_kernel; the launches of the last search call",
    "notes": "GRBM_GUI_ACTIVE is summed over 8 XCDs; 1024 SIMDs; mfma_busy_fraction = SQ_VALU_MFMA_BUSY_CYCLES / (gpu_cycles x 1024)",
    "launches": out}, indent=1))
