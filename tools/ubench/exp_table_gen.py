from decimal import Decimal, getcontext
from fractions import Fraction
import struct
getcontext().prec = 60
two = Decimal(2)
vals = []
for j in range(128):
    d = two ** (Decimal(j) / Decimal(128))
    # correctly rounded double: float(Decimal) rounds correctly (Python converts via string -> correctly rounded)
    f = float(d)
    vals.append(f)
lines = []
for i in range(0, 128, 4):
    lines.append("    " + ", ".join(f.hex() for f in vals[i:i+4]) + ",")
open("exp_table_tab.inc", "w").write("\n".join(lines) + "\n")
print(lines[0]); print(lines[-1])
