#!/usr/bin/env python3
"""Audit of hipcc's gfx950 ISA for the miscompile of profiles/r2_hipcc_switch_miscompile.md: a VGPR (pair) the compiler itself
marks `; implicit-def:` on some path and that is READ on that path before anything writes it.

    hipcc ... -S --cuda-device-only -o x.s x.hip ;  python tools/isa_undef_audit.py x.s [more.s ...]

For every `implicit-def` comment the script walks the control-flow graph of the kernel forward from that point (labels, s_branch,
s_cbranch_*) and reports the first instruction on any path that reads one of the marked registers while none has written it.
Conservative on purpose: a report is a lead to read, not a verdict (a read may be dead under EXEC masking); no report means no
path from an implicit-def reaches a reader."""
import re
import sys
from collections import defaultdict

REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
NO_VDST = ("global_store", "flat_store", "scratch_store", "buffer_store", "ds_write", "ds_add_u", "ds_add_f", "ds_min_", "ds_max_", "ds_or_", "ds_and_",
           "ds_sub_", "ds_inc_", "ds_dec_", "ds_xor_", "ds_cmpst_b", "global_atomic", "flat_atomic", "buffer_atomic", "s_", "v_cmp", "v_readlane", "v_readfirstlane",
           "ds_gws", "ds_nop", "buffer_wbl2", "buffer_inv", "v_nop", "ds_pk_add_f", "ds_pk_add_bf")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def parse_kernel(lines):
    """-> list of (label or None, mnemonic, defs, uses, branch target or None, kind) per instruction; implicit-def markers inline"""
    insts = []
    for ln in lines:
        s = ln.split(";")[0].strip() if "implicit-def" not in ln else ln.strip()
        if not s:
            continue
        if "implicit-def" in s:
            rr = set()
            for m in re.finditer(r"\$vgpr(\d+)", s):
                rr.add(int(m.group(1)))
            if rr:
                insts.append(("undef", None, rr, set(), None))
            continue
        if s.endswith(":") and not s.startswith("."):
            continue
        if re.match(r"^\.?[A-Za-z_][\w.$]*:$", s):
            insts.append(("label", s[:-1], set(), set(), None))
            continue
        if s.startswith("."):
            continue
        parts = s.split(None, 1)
        mn, ops = parts[0], (parts[1] if len(parts) > 1 else "")
        oplist = [o.strip() for o in ops.split(",")] if ops else []
        tgt = None
        if mn.startswith("s_cbranch") or mn == "s_branch":
            tgt = oplist[0] if oplist else None
        has_dst = not mn.startswith(NO_VDST) or "_rtn" in mn or mn.startswith("global_atomic") and "_rtn" in mn
        if mn.startswith(("global_atomic", "flat_atomic", "buffer_atomic")):
            has_dst = "sc0" in ops or " glc" in ops          # returning form
        defs = regs(oplist[0]) if (has_dst and oplist) else set()
        uses = regs(", ".join(oplist[1:] if (has_dst and oplist) else oplist))
        if mn.startswith("v_writelane") or mn.startswith("v_mac") or mn.startswith("v_fmac") or mn.startswith("v_dot") or "sdwa" in mn and "dst_unused:UNUSED_PRESERVE" in ops:
            uses |= defs                                  # read-modify-write destinations
        insts.append(("inst", mn, defs, uses, tgt))
    return insts


def audit(path):
    text = open(path).read().splitlines()
    # split into kernels: from "<name>:" followed somewhere by s_endpgm, up to ".Lfunc_end"
    kernels, cur, name = [], None, None
    for ln in text:
        m = re.match(r"^([A-Za-z_][\w$.]*):\s*(;.*)?$", ln)
        if m and not ln.startswith(".L"):
            name, cur = m.group(1), []
            continue
        if cur is not None:
            if ln.startswith(".Lfunc_end"):
                kernels.append((name, cur)); cur = None
            else:
                cur.append(ln)
    findings = []
    for name, lines in kernels:
        ins = parse_kernel(lines)
        label_at = {i[1]: k for k, i in enumerate(ins) if i[0] == "label"}
        nxt = {}                                             # successors of instruction k
        for k, (kind, mn, defs, uses, tgt) in enumerate(ins):
            if kind == "inst" and mn == "s_endpgm":
                nxt[k] = []
            elif kind == "inst" and mn == "s_branch":
                nxt[k] = [label_at[tgt]] if tgt in label_at else []
            elif kind == "inst" and mn.startswith("s_cbranch") and tgt in label_at:
                nxt[k] = [k + 1, label_at[tgt]]
            else:
                nxt[k] = [k + 1]
        for start, it in enumerate(ins):
            if it[0] != "undef":
                continue
            for r in it[2]:                                  # one register at a time: a plain reachability walk
                seen, stack = set(), [start + 1]
                while stack:
                    k = stack.pop()
                    if k >= len(ins) or k in seen:
                        continue
                    seen.add(k)
                    kind, mn, defs, uses, tgt = ins[k]
                    if kind == "inst":
                        if r in uses:
                            findings.append((name, [r], mn, k))
                            continue
                        if r in defs:
                            continue
                    elif kind == "undef" and r in ins[k][2]:
                        continue                             # the same register marked again further down: that walk covers it
                    stack.extend(nxt[k])
    return findings


if __name__ == "__main__":
    total = 0
    for p in sys.argv[1:]:
        f = audit(p)
        uniq = sorted(set((n, tuple(r), m) for n, r, m, _ in f))
        for n, r, m in uniq:
            print(f"{p}: {n}: v{list(r)} marked implicit-def is read by `{m}` before any write on some path")
        total += len(uniq)
    print(f"{total} lead(s)")
