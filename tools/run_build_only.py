#!/usr/bin/env python3
"""Build an index of N synthetic claims (texts -> K1 on the GPU) and exit: workload for profiling K1.
Usage: python3 tools/run_build_only.py [claims]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.run_configs import build_index
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
ix, t_add = build_index(n)
print("built", ix.size(), "claims; add_texts incl. pack + H2D:", round(t_add, 3), "s")
