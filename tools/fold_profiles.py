#!/usr/bin/env python
"""Folds the CSV summaries of tools/profile_r04.sh (gpurun_out/prof_r04/) into the JSON bundle bench.py reads
(profiles/hbm_traffic.json): per dominant kernel and per launch, HBM traffic (FETCH_SIZE x 2 + WRITE_SIZE, KiB counters,
the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md) and the SQ instruction counters, FP64 classes included.
usage: fold_profiles.py <dir with *_pmc_*.csv> > hbm_traffic.json"""
import csv
import json
import os
import sys

d = sys.argv[1]
TAG = os.environ.get("PROFILE_TAG") or os.path.basename(os.path.normpath(d)).replace("prof_", "") or "r04"   # gpurun_out/prof_r05 -> r05


def rows(name):
    p = os.path.join(d, name + ".csv")
    return list(csv.DictReader(open(p))) if os.path.exists(p) else []


def pick(name, kernel_part, counter, grid=None):
    """mean per launch of `counter` for kernels whose name contains kernel_part (largest grid if several)"""
    best = None
    for r in rows(name):
        if kernel_part in r["kernel"] and r["counter"] == counter and (grid is None or r["grid_size"] == grid):
            v = float(r["mean_per_launch"])
            if best is None or int(r["grid_size"]) > best[0]:
                best = (int(r["grid_size"]), v)
    return best[1] if best else None


def total(name, kernel_parts, counter):
    """sum over the kernels of a call: per-launch mean x launches per call (launches / calls of the driving script)"""
    s = 0.0
    for r in rows(name):
        if any(k in r["kernel"] for k in kernel_parts) and r["counter"] == counter:
            s += float(r["mean_per_launch"]) * int(r["launches"])
    return s


out = {"_comment": "per-launch counters of the dominant kernels from the rocprofv3 PMC passes of tools/profile_r04.sh (separate --pmc runs; summaries in "
                   "profiles/" + TAG + "_*_pmc_*.csv).  FETCH_SIZE / WRITE_SIZE in KiB; traffic = FETCH_SIZE x 2 + WRITE_SIZE (gfx950 correction of MI355X_MICROARCH.md, HBM section). "
                   "Folded by tools/fold_profiles.py."}
U = "k_oi_union<true, false, 32"     # (the name as rocprofv3 prints it carries every template argument: <true, false, 32, false> since round 6; the largest grid = the launch over all tiles)
fs, wsz = pick("oi_pmc_fetch", U, "FETCH_SIZE"), pick("oi_pmc_write", U, "WRITE_SIZE")
oi = {"kernel": "k_oi_union<true, false, 32> (first pass, all tiles)", "workload": "optimal_interpolation 4000x4000 grid, 10000 obs, BarnesStructure(10000), max_points=30", "n_gpus": 1,
      "_source": "profiles/" + TAG + "_oi_pmc_*.csv"}
if fs is not None and wsz is not None:
    oi.update({"FETCH_SIZE_KiB": fs, "WRITE_SIZE_KiB": wsz, "traffic_bytes": int((2 * fs + wsz) * 1024), "algorithmic_bytes": 16000000 * 28})
for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"):
    v = pick("oi_pmc_sq", U, c)
    if v is not None:
        oi[c] = int(v)
f64 = [pick("oi_pmc_fp64", U, c) for c in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64")]
if all(v is not None for v in f64):
    oi["SQ_INSTS_VALU_FP64"] = int(sum(f64))
    oi["_fp64_classes"] = dict(zip(("ADD_F64", "MUL_F64", "FMA_F64", "TRANS_F64"), [int(v) for v in f64]))
# where the kernel's wave cycles go (round-3 verdict, item 3): SQ_WAVE_CYCLES = cycles waves spent resident, SQ_ACTIVE_INST_ANY = of those, cycles
# with an instruction of the wave in flight, SQ_WAIT_INST_ANY = waiting for an instruction to issue, SQ_WAIT_ANY = waiting on s_waitcnt
for name, cs in (("oi_pmc_busy", ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY")),
                 ("oi_pmc_wait", ("SQ_WAIT_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS")),
                 ("oi_pmc_act", ("SQ_ACTIVE_INST_SCA", "SQ_INST_CYCLES_SALU", "SQ_THREAD_CYCLES_VALU"))):
    for c in cs:
        v = pick(name, U, c)
        if v is not None:
            oi[c] = int(v)
if "SQ_WAVE_CYCLES" in oi and oi["SQ_WAVE_CYCLES"] > 0:
    w = float(oi["SQ_WAVE_CYCLES"])
    oi["wave_cycle_shares"] = {"active_inst_any": oi.get("SQ_ACTIVE_INST_ANY", 0) / w, "wait_inst_any": oi.get("SQ_WAIT_INST_ANY", 0) / w,
                               "wait_any": oi.get("SQ_WAIT_ANY", 0) / w, "active_inst_valu": oi.get("SQ_ACTIVE_INST_VALU", 0) / w,
                               "active_inst_lds": oi.get("SQ_ACTIVE_INST_LDS", 0) / w, "wait_inst_lds": oi.get("SQ_WAIT_INST_LDS", 0) / w,
                               "_note": "fractions of SQ_WAVE_CYCLES (the cycles waves are resident); with 3 waves per SIMD the SIMD's own VALU utilisation is "
                                        "about three times active_inst_valu"}
out["k_oi_union"] = oi

calls = 3.0   # tools/ensi_c5.py and tools/prof_nb.py make three calls each
en = {"workload": "optimal_interpolation_ensi 2500x2500x50, 5000 obs, max_points=30", "_source": "profiles/" + TAG + "_ensi_pmc_*.csv",
      "_note": "wave-instructions per CALL, summed over k_ensi_scan / k_ensi_pair / k_ensi_members (all batches)"}
for c in ("SQ_INSTS_VALU", "SQ_INSTS_VALU_MFMA_MOPS_F64"):
    en[c] = total("ensi_pmc_sq", ["k_ensi"], c) / calls
for c in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"):
    en[c] = total("ensi_pmc_fp64", ["k_ensi"], c) / calls
try:    # (round 4: the perturbation series runs on the FP32 matrix path)
    for c in ("SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES"):
        en[c] = total("ensi_pmc_mfma", ["k_ensi"], c) / calls
except (OSError, KeyError):
    pass
out["ensi_C5"] = en

qf = {"workload": "neighbourhood_quantile_fast 4000x4000x100, halfwidth 15, 11 thresholds", "_source": "profiles/" + TAG + "_nbh_pmc_*.csv"}
tr = 0.0
for k in ("k_qf_count", "k_qf_box"):
    f_, w_ = pick("nbh_pmc_fetch", k, "FETCH_SIZE"), pick("nbh_pmc_write", k, "WRITE_SIZE")
    if f_ is not None and w_ is not None:
        qf[k] = {"FETCH_SIZE_KiB": f_, "WRITE_SIZE_KiB": w_}
        tr += (2 * f_ + w_) * 1024
qf["traffic_bytes"] = int(tr)
qf["algorithmic_bytes"] = 16000000 * 404
out["quantile_fast_C4"] = qf
mp = pick("nbh_pmc_fetch", "k_member_pass<0>", "FETCH_SIZE")
if mp is not None:
    out["k_member_pass<0>"] = {"workload": "neighbourhood Mean 4000x4000x100 (member pass)", "FETCH_SIZE_KiB": mp, "traffic_bytes_read": int(2 * mp * 1024),
                               "algorithmic_bytes_read": 6400000000}
print(json.dumps(out, indent=1))
