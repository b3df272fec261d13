#!/usr/bin/env python
"""C5 (2500x2500x50, 5000 obs, max_points 30) once or a few times: for rocprofv3 runs (tools/bench_paths.py ensi runs a smaller case too)."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_paths import ensi_case  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
ensi_case(n, n, 50, 5000, 30, converged=False)
