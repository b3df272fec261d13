"""Inventory of the accumulation-register (AGPR) traffic of the matrix-core kernels, from the device assembly.

    python tools/isa_agpr_check.py [unit=ensi] [kernel-name filter=k_ensi] [-v]

Why (round-4 verdict, Weak 3): the kernels that read registers ACROSS lanes (v_readlane, DPP, ds_bpermute of register rows, MFMA
operands) must never hold a register row that is valid in some lanes only.  `__graft_entry__.build()` enforces the simple rule: NO
kernel of the library may use AGPRs or scratch except the two on its allow-list, k_ensi_pair (256 VGPRs + ~245 AGPRs) and
k_ensi_big_ns, which use the matrix cores and in which the compiler parks whole register rows in AGPRs
(`v_accvgpr_write_b32 aN, vM` / `v_accvgpr_read_b32 vM, aN` / `v_accvgpr_mov_b32`).  This tool looks into those two.

The rule the verdict proposed -- every such copy runs under the full EXEC mask -- does not hold and need not: a copy moves the ACTIVE
lanes only, and inside a divergent region that is what the program means (a per-lane value parked and fetched by the same lanes; the
inactive lanes of both registers keep what they had, as if the value had stayed in its VGPR).  About 40 % of the AGPR moves of
k_ensi_pair sit in such regions (first column of the report).  The hazard is narrower -- the one that made a scratch-spilling build of
k_oi_union return wrong values: a row PARKED under a partial mask and RELOADED under the full one, or fed to an MFMA
(which reads every lane whatever EXEC says), by lanes that were switched off at the park.  The second column lists the CANDIDATES for
that: reloads at a provably full mask / MFMA operands whose reaching definition may be a copy under a mask the analysis could not prove
full.  A candidate is not a defect: `x = a; if(c) x = b;` on a value that lives in an AGPR has exactly this shape (full definition,
partial overlay, full read), and the EXEC analysis is conservative inside loops (a restore `s_or_b64 exec, exec, saved` only counts as
full when `saved` is provably the launch mask on every path).  The listing is for reading next to the source, the dynamic evidence is
the poisoned suite (tools/hostile: a0..a249 of every lane hold NaN patterns before every call; tools/*_hostile_soak.py).

How: two abstract interpretations over the control-flow graph of each kernel, both iterated to their fixed points.
 1. EXEC: exec is FULL (provably the mask the wave was launched with; all kernels are launched with whole waves) or PARTIAL; every
    SGPR pair is FULL (a copy of the launch mask / all ones) or OTHER.  Transfer functions for what the compiler does to exec
    (s_mov, s_and / s_andn2 / s_or _saveexec, s_or exec, exec, saved, s_and / s_andn2 / s_xor exec, v_cmpx); any other write to a
    tracked register forgets it; joins take the weaker state.  Unknown writers of exec make it PARTIAL.
 2. AGPR taint: an AGPR is tainted while its reaching definition may be a copy or load executed under a mask that (1) could not prove
    full; a definition under the full mask and an MFMA result (written in all lanes) clear it; joins take the union.
Exit code 0 unless an instruction the tool does not know writes an AGPR."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FULL, PART = 1, 0
RE_RANGE = re.compile(r"^s\[(\d+):(\d+)\]$")
RE_SINGLE = re.compile(r"^s(\d+)$")


def sregs(op):
    """the SGPR numbers an operand names (vcc = 106/107 on gfx9)"""
    op = op.strip()
    m = RE_RANGE.match(op)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    m = RE_SINGLE.match(op)
    if m:
        return [int(m.group(1))]
    if op == "vcc":
        return [106, 107]
    if op in ("vcc_lo",):
        return [106]
    if op in ("vcc_hi",):
        return [107]
    return []


def pair_key(op):
    r = sregs(op)
    return (r[0], r[-1]) if len(r) == 2 else None


def assemble(unit):
    out = "/tmp/isa/%s.s" % unit
    os.makedirs("/tmp/isa", exist_ok=True)
    src = os.path.join(ROOT, "gridpp_amd", "csrc", unit + ".hip")
    extra = []
    with open(src) as f:
        for i, l in enumerate(f):
            if i < 40 and l.startswith("// hipcc-flags:"):
                extra += l.split(":", 1)[1].split()
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
           "-fno-fast-math", "-fno-strict-aliasing", "--cuda-device-only", "-S", src, "-o", out] + extra + os.environ.get("GPP_HIP_DEFS", "").split()
    deps = [src] + [os.path.join(ROOT, "gridpp_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "gridpp_amd", "csrc")) if f.endswith(".h")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.check_call(cmd)
    return out


def functions(path):
    """name -> list of (label or None, mnemonic, operands) in program order"""
    funcs, cur, name = {}, None, None
    with open(path) as f:
        for line in f:
            s = line.split(";")[0].rstrip()
            if not s:
                continue
            if re.match(r"^[A-Za-z_.$][\w.$]*:\s*$", s):
                lab = s.strip()[:-1]
                if not lab.startswith(".L") and not lab.startswith("$"):
                    name, cur = lab, []
                    funcs[name] = cur
                elif cur is not None:
                    cur.append((lab, None, None))
                continue
            if cur is None or not s.startswith("\t") or s.strip().startswith("."):
                if s.strip().startswith(".end_amdhsa_kernel") or s.strip().startswith(".section"):
                    pass
                continue
            t = s.strip()
            parts = t.split(None, 1)
            ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
            cur.append((None, parts[0], ops))
            if parts[0] == "s_endpgm":
                pass
    return funcs


def analyse(instrs):
    # basic blocks
    leaders = {0}
    labels = {}
    for i, (lab, mn, ops) in enumerate(instrs):
        if lab is not None:
            labels[lab] = i
            leaders.add(i)
        elif mn and (mn.startswith("s_cbranch") or mn == "s_branch" or mn == "s_endpgm" or mn.startswith("s_setpc")):
            leaders.add(i + 1)
    leaders = sorted(x for x in leaders if x < len(instrs))
    block_of = {}
    blocks = []
    for b, start in enumerate(leaders):
        end = leaders[b + 1] if b + 1 < len(leaders) else len(instrs)
        blocks.append((start, end))
        block_of[start] = b
    succ = [[] for _ in blocks]
    for b, (start, end) in enumerate(blocks):
        last = None
        for i in range(end - 1, start - 1, -1):
            if instrs[i][1]:
                last = instrs[i]
                break
        fall = b + 1 if b + 1 < len(blocks) else None
        if last is None:
            if fall is not None:
                succ[b].append(fall)
            continue
        mn, ops = last[1], last[2]
        if mn == "s_branch":
            succ[b].append(block_of[labels[ops[0]]])
        elif mn.startswith("s_cbranch"):
            succ[b].append(block_of[labels[ops[0]]])
            if fall is not None:
                succ[b].append(fall)
        elif mn == "s_endpgm" or mn.startswith("s_setpc"):
            pass
        elif fall is not None:
            succ[b].append(fall)

    def step(state, mn, ops, at=None):
        """state = [exec, {pair: FULL}] (pairs not in the dict are OTHER); `at`: called with (instr, exec) for AGPR moves"""
        ex, full = state
        if mn.startswith("v_accvgpr") and at is not None:
            at(ex)
        if not ops:
            return
        dst = ops[0]
        if mn in ("s_mov_b64",):
            if dst == "exec":
                src = ops[1]
                state[0] = FULL if (src == "-1" or full.get(pair_key(src))) else PART
                return
            k = pair_key(dst)
            for r in sregs(dst):
                for kk in [q for q in full if q[0] <= r <= q[1]]:
                    del full[kk]
            if k and (ops[1] == "exec" and ex == FULL or ops[1] == "-1" or full.get(pair_key(ops[1]))):
                full[k] = FULL
            return
        if mn.endswith("_saveexec_b64"):
            k = pair_key(dst)
            src_full = ops[1] == "-1" or bool(full.get(pair_key(ops[1])))
            for r in sregs(dst):
                for kk in [q for q in full if q[0] <= r <= q[1]]:
                    del full[kk]
            if k and ex == FULL:
                full[k] = FULL
            if mn.startswith("s_or_saveexec"):
                state[0] = FULL if (ex == FULL or src_full) else PART
            elif mn.startswith("s_and_saveexec"):
                state[0] = FULL if (ex == FULL and src_full) else PART
            else:
                state[0] = PART
            return
        if dst == "exec":
            if mn == "s_or_b64" and "exec" in ops[1:]:
                other = [o for o in ops[1:] if o != "exec"]
                state[0] = FULL if (ex == FULL or (other and (other[0] == "-1" or full.get(pair_key(other[0]))))) else PART
            else:
                state[0] = PART
            return
        if mn.startswith("v_cmpx"):
            state[0] = PART
            return
        # any other instruction: forget every tracked pair its (first) destination overlaps.  v_cmp / v_readlane / s_* all name the
        # destination first; s_load_* and friends too.  Two-destination scalar forms (v_add_co: vdst, sdst) name the SGPR second.
        dsts = [dst]
        if mn.startswith("v_") and len(ops) > 1 and ("_co_" in mn or mn.startswith("v_div_scale") or mn.startswith("v_mad_u64") or mn.startswith("v_mad_i64")):
            dsts.append(ops[1])
        for d in dsts:
            for r in sregs(d):
                for kk in [q for q in full if q[0] <= r <= q[1]]:
                    del full[kk]

    # fixed point
    IN = [None] * len(blocks)
    IN[0] = [FULL, {}]
    work = [0]
    while work:
        b = work.pop()
        st = [IN[b][0], dict(IN[b][1])]
        for i in range(*blocks[b]):
            lab, mn, ops = instrs[i]
            if mn:
                step(st, mn, ops)
        for s_ in succ[b]:
            if IN[s_] is None:
                IN[s_] = [st[0], dict(st[1])]
                work.append(s_)
            else:
                ne = min(IN[s_][0], st[0])
                nf = {k: FULL for k in IN[s_][1] if k in st[1]}
                if ne != IN[s_][0] or nf != IN[s_][1]:
                    IN[s_] = [ne, nf]
                    work.append(s_)
    # exec state in front of every instruction
    ex_at = [None] * len(instrs)
    for b, (start, end) in enumerate(blocks):
        if IN[b] is None:
            continue
        st = [IN[b][0], dict(IN[b][1])]
        for i in range(start, end):
            lab, mn, ops = instrs[i]
            if not mn:
                continue
            ex_at[i] = st[0]
            step(st, mn, ops)

    # AGPR taint (second fixed point)
    def aregs(op):
        op = op.split(" ")[0]
        m = re.match(r"^a(\d+)$", op)
        if m:
            return [int(m.group(1))]
        m = re.match(r"^a\[(\d+):(\d+)\]$", op)
        return list(range(int(m.group(1)), int(m.group(2)) + 1)) if m else []

    def tstep(taint, i, report=None):
        lab, mn, ops = instrs[i]
        if not mn or not ops:
            return
        full = ex_at[i] == FULL
        if mn.startswith("v_mfma") or mn.startswith("v_smfma"):
            if report is not None:
                for o in ops[1:]:
                    t = [r for r in aregs(o) if r in taint]
                    if t:
                        report.append((i, mn, ops, "matrix-core operand a%d was parked under a partial mask" % t[0]))
            for r in aregs(ops[0]):
                taint.discard(r)
            return
        if mn in ("v_accvgpr_read_b32", "v_accvgpr_mov_b32") and report is not None and full:
            t = [r for r in aregs(ops[1]) if r in taint]
            if t:
                report.append((i, mn, ops, "a%d was parked under a partial mask and comes back under the full one" % t[0]))
        d = aregs(ops[0])
        if d and (mn.startswith("v_accvgpr_write") or mn.startswith("v_accvgpr_mov") or mn.startswith("global_load") or mn.startswith("ds_read") or mn.startswith("buffer_load") or mn.startswith("scratch_load") or mn.startswith("flat_load")):
            for r in d:
                if full:
                    taint.discard(r)
                else:
                    taint.add(r)
        elif d and not (mn.startswith("global_store") or mn.startswith("ds_write") or mn.startswith("buffer_store") or mn.startswith("flat_store") or mn.startswith("scratch_store") or mn.startswith("v_accvgpr_read")):
            for r in d:      # an instruction this tool does not know writes an AGPR: treat it as a partial definition
                taint.add(r)
            if report is not None:
                report.append((i, mn, ops, "this tool does not know the instruction that writes a%d" % d[0]))

    TIN = [None] * len(blocks)
    TIN[0] = set()
    work = [0]
    while work:
        b = work.pop()
        if IN[b] is None:
            continue
        t = set(TIN[b])
        for i in range(*blocks[b]):
            tstep(t, i)
        for s_ in succ[b]:
            if TIN[s_] is None:
                TIN[s_] = set(t)
                work.append(s_)
            elif not t <= TIN[s_]:
                TIN[s_] |= t
                work.append(s_)
    total, partial, bad = 0, 0, []
    for b, (start, end) in enumerate(blocks):
        if TIN[b] is None:
            continue
        t = set(TIN[b])
        for i in range(start, end):
            mn = instrs[i][1]
            if mn and mn.startswith("v_accvgpr"):
                total += 1
                partial += ex_at[i] != FULL
            tstep(t, i, bad)
    return total, partial, bad, len(blocks)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("-")]
    unit = args[0] if args else "ensi"
    filt = args[1] if len(args) > 1 else "k_ensi"
    verbose = "-v" in sys.argv
    path = assemble(unit)
    funcs = functions(path)
    rc = 0
    for name, instrs in funcs.items():
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        if filt not in dem:
            continue
        total, partial, bad, nb = analyse(instrs)
        if total == 0 and not verbose:
            continue
        short = dem.split("(")[0]
        print("%-44s %4d AGPR moves in %4d blocks, %4d inside divergent regions; candidates (partial definition -> full-mask reload / MFMA operand): %d"
              % (short[:44], total, nb, partial, len(bad)))
        for i, mn, ops, why in bad[:12 if not verbose else None]:
            print("    instruction %d: %s %s  (%s)" % (i, mn, ", ".join(ops), why))
        if any("does not know" in why for _, _, _, why in bad):
            rc = 1
    print("AGPR INVENTORY", "INCOMPLETE (unknown AGPR writers)" if rc else "DONE")
    return rc


if __name__ == "__main__":
    sys.exit(main())
