#!/usr/bin/env python
"""C4 quantile_fast only (timing experiments / rocprofv3 runs)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_cases import nb_cases
for r in nb_cases(4000, 4000, 100, 15, two_d=False):
    if "quantile" in r["case"]:
        print(json.dumps({"case": r["case"], "ms": r["ms"]}))
