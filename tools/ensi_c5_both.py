"""config 5 in both modes (default / sweeps to convergence): python tools/ensi_c5_both.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_cases import ensi_case
r = ensi_case(2500, 2500, 50, 5000, 30, converged=True)
print(json.dumps({"ms": r["ms"], "ms_converged": r["ms_converged"]}))
