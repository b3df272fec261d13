"""CPU-only: the structs of include/gridpp_hip.h as a C compiler lays them out == the ctypes mirrors of gridpp_amd/_capi.py ==
the ctypes stub printed in INTEGRATION.md (a caller who copies a short struct makes the library read past its end)."""
import ctypes as C
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SNIPPET = r"""
#include <stdio.h>
#include <stddef.h>
#include "gridpp_hip.h"
#define F(s, f) printf(#s "." #f " %zu\n", offsetof(s, f))
int main(void) {
    printf("gpp_structure %zu\n", sizeof(gpp_structure));
    F(gpp_structure, kind); F(gpp_structure, h); F(gpp_structure, v); F(gpp_structure, w); F(gpp_structure, min_rho);
    F(gpp_structure, kind_v); F(gpp_structure, kind_w); F(gpp_structure, loc); F(gpp_structure, cv_dist); F(gpp_structure, flags);
    F(gpp_structure, field); F(gpp_structure, field_v); F(gpp_structure, field_w);
    printf("gpp_oi_stats %zu\n", sizeof(gpp_oi_stats));
    printf("gpp_ensi_stats %zu\n", sizeof(gpp_ensi_stats));
    F(gpp_ensi_stats, cells); F(gpp_ensi_stats, condition_passthrough); F(gpp_ensi_stats, real_part_passthrough); F(gpp_ensi_stats, kernel_ms);
    return 0;
}
"""


def c_layout(tmp_path):
    src = tmp_path / "layout.c"
    src.write_text(SNIPPET)
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True)
    return {l.split()[0]: int(l.split()[1]) for l in out.splitlines()}


def test_ctypes_mirror_matches_the_header(tmp_path):
    from gridpp_amd import _capi
    lay = c_layout(tmp_path)
    assert C.sizeof(_capi.gpp_structure) == lay["gpp_structure"]
    assert len(_capi.gpp_structure._fields_) == 13
    for name, _ in _capi.gpp_structure._fields_:
        assert getattr(_capi.gpp_structure, name).offset == lay["gpp_structure." + name], name
    assert C.sizeof(_capi.gpp_oi_stats) == lay["gpp_oi_stats"]
    assert C.sizeof(_capi.gpp_ensi_stats) == lay["gpp_ensi_stats"]
    for name, _ in _capi.gpp_ensi_stats._fields_:
        assert getattr(_capi.gpp_ensi_stats, name).offset == lay["gpp_ensi_stats." + name], name


def test_integration_md_stub_matches_the_header(tmp_path):
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"class gpp_structure\(C\.Structure\):\n(.*?)\n\n", text, re.S)
    assert m, "INTEGRATION.md no longer shows the ctypes stub of gpp_structure"
    ns = {"C": C}
    exec("class gpp_structure(C.Structure):\n" + m.group(1), ns)
    lay = c_layout(tmp_path)
    assert C.sizeof(ns["gpp_structure"]) == lay["gpp_structure"]
    assert [f[0] for f in ns["gpp_structure"]._fields_] == [k.split(".")[1] for k in lay if k.startswith("gpp_structure.")]


def test_memory_flags_of_the_header_match_the_mirror(tmp_path):
    """GPP_MEM_* / GPP_ASYNC / GPP_HOST_F64 / GPP_Q_HOST (round 6) as the C preprocessor sees them == gridpp_amd._capi's constants."""
    from gridpp_amd import _capi
    src = tmp_path / "flags.c"
    src.write_text('#include <stdio.h>\n#include "gridpp_hip.h"\nint main(void) { printf("%d %d %d %d %d\\n", GPP_MEM_HOST, GPP_MEM_DEVICE, GPP_ASYNC, GPP_HOST_F64, GPP_Q_HOST); return 0; }\n')
    exe = tmp_path / "flags"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    vals = [int(x) for x in subprocess.check_output([str(exe)], text=True).split()]
    assert vals == [_capi.MEM_HOST, _capi.MEM_DEVICE, _capi.ASYNC, _capi.HOST_F64, _capi.Q_HOST]
    assert len(set(vals[1:])) == 4 and all(v & (v - 1) == 0 for v in vals[1:])      # distinct single bits
