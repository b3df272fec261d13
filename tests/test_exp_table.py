"""The table behind d_exp_core (gridpp_amd/csrc/oi_common.h): 2^(j/128) correctly rounded, and the CPU restatement of the kernel's
arithmetic (tools/ubench/exp_table.c) against glibc -- no float32-rounded result may differ, the double error stays below 1 ulp."""
import os
import re
import subprocess
from decimal import Decimal, getcontext

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _table_of(path, start):
    text = open(path).read()
    body = text[text.index(start):]
    body = body[:body.index("};")]
    return [float.fromhex(h) for h in re.findall(r"0x1\.[0-9a-f]+p[+-]\d+", body)]


def test_table_is_correctly_rounded():
    tab = _table_of(os.path.join(ROOT, "gridpp_amd", "csrc", "oi_common.h"), "c_exp2_tab[128]")
    assert len(tab) == 128
    getcontext().prec = 60
    for j, t in enumerate(tab):
        assert t == float(Decimal(2) ** (Decimal(j) / Decimal(128))), j   # float(Decimal) rounds correctly
    inc = [float.fromhex(h) for h in re.findall(r"0x1\.[0-9a-f]+p[+-]\d+", open(os.path.join(ROOT, "tools", "ubench", "exp_table_tab.inc")).read())]
    assert inc == tab      # the CPU restatement uses the same table


def test_cpu_restatement_matches_glibc(tmp_path):
    exe = str(tmp_path / "exp_table")
    subprocess.run(["gcc", "-O2", "-o", exe, os.path.join(ROOT, "tools", "ubench", "exp_table.c"), "-lm"], check=True, cwd=os.path.join(ROOT, "tools", "ubench"))
    out = subprocess.run([exe, "4000000"], check=True, capture_output=True, text=True).stdout
    m = re.search(r"float mismatches vs libm: old (\d+) new (\d+); max ulp\(double\) old ([0-9.]+) new ([0-9.]+)", out)
    assert m, out
    assert int(m.group(2)) == 0 and float(m.group(4)) < 1.0, out
