#!/usr/bin/env python3
"""Static check of the hand-issued multiplier reads of mpe_pairexp.h (cios1q, MPE_BQ) in the EMITTED ISA.

A `ds_read_b64 vD, ... ; BQ_ISSUE vD` lands in its destination some hundred cycles later; hipcc does not know (the read is an
asm statement) and the program is only correct if NOTHING touches vD until the matching `s_waitcnt ... ; BQ_WAIT vD`: no read,
no write, no copy, no spill.  This script walks every pair_modexp_kernel in a -save-temps .s file as a control-flow graph,
propagates the set of pending destination registers along every edge (may-analysis to a fixed point) and reports
  * any instruction that names a pending register,
  * a BQ_WAIT on a register that is not pending on some path (then the issue went to another register: a copy),
  * registers still pending at s_endpgm.
Usage: tools/check_bq_isa.py build/mpe_pair2048-hip-amdgcn-amd-amdhsa-gfx950.s [more .s files]   (exit 1 on a violation)."""
import re
import sys


def regs_of(text):
    """every VGPR index named in an operand string: v12, v[12:15]"""
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", text):
        out.add(int(m.group(1)))
    return out


def parse_functions(path):
    lines = open(path).read().split("\n")
    funcs, i = {}, 0
    while i < len(lines):
        l = lines[i]
        if l.startswith("_ZN3mpe") and "pair_modexp_kernel" in l and re.match(r"^\S+:", l):
            name = l.split(":")[0]
            blocks, order, cur = {}, [], "entry"
            blocks[cur] = []
            order.append(cur)
            i += 1
            while i < len(lines) and not lines[i].strip().startswith("s_endpgm"):
                t = lines[i]
                m = re.match(r"^(\.LBB\d+_\d+):", t)
                if m:
                    cur = m.group(1)
                    blocks[cur] = []
                    order.append(cur)
                elif t.startswith("\t") and not t.strip().startswith((".", ";")):
                    blocks[cur].append(t.strip())
                i += 1
            blocks[cur].append("s_endpgm")
            funcs[name] = (blocks, order)
        i += 1
    return funcs


def fallthrough(blocks, order, name):
    """the block control falls into after `name` (None after an unconditional branch / s_endpgm)"""
    for x in blocks[name]:
        op = x.split()[0]
        if op in ("s_branch", "s_endpgm", "s_setpc_b64"):
            return None
    k = order.index(name)
    return order[k + 1] if k + 1 < len(order) else None


def check_function(fname, blocks, order):
    issued_anywhere = set()
    for b in order:
        for x in blocks[b]:
            for r in re.findall(r"BQ_ISSUE (v\[\d+:\d+\]|v\d+)", x):
                issued_anywhere |= regs_of(r)
    problems, n_issue, n_wait = [], 0, 0
    pend_in = {b: set() for b in order}
    work = list(order)
    reported = set()

    def flow(target, pend):
        if target in blocks and not pend <= pend_in[target]:
            pend_in[target] |= pend
            if target not in work:
                work.append(target)

    def report(key, msg):
        if key not in reported:
            reported.add(key)
            problems.append(msg)

    while work:
        b = work.pop(0)
        pend = set(pend_in[b])
        for k, x in enumerate(blocks[b]):
            code = x.partition(";")[0]
            op = code.split()[0] if code.split() else ""
            issue = re.findall(r"BQ_ISSUE (v\[\d+:\d+\]|v\d+)", x)
            waits = re.findall(r"BQ_WAIT (v\[\d+:\d+\]|v\d+)", x)
            if issue:
                dst = regs_of(issue[0])
                used = regs_of(code.split(None, 1)[1]) - dst
                if used & pend:
                    report((b, k, "addr"), f"{b}+{k}: issue reads a pending register: {x}")
                if dst & pend:
                    report((b, k, "re"), f"{b}+{k}: issue into a register whose read is still in flight: {x}")
                pend |= dst
                n_issue += 1
            elif waits:
                for w in waits:
                    r = regs_of(w)
                    if not r <= issued_anywhere:        # waits on a register no ds_read ever targets: the value was COPIED while in flight
                        report((b, k, w), f"{b}+{k}: wait on {w}, which is never an issue destination (a copy of an in-flight register): {x}")
                    pend -= r                            # (a wait on a register that already landed is harmless)
                n_wait += 1
            elif op.startswith("s_cbranch") or op == "s_branch":
                flow(code.split()[1], pend)              # the pending set AT the branch travels along this edge
            elif op == "s_endpgm":
                if pend:
                    report((b, "end"), f"{b}: registers still pending at s_endpgm: {sorted(pend)}")
            elif " " in code.strip():
                touched = regs_of(code.split(None, 1)[1])
                if touched & pend:
                    report((b, k), f"{b}+{k}: touches in-flight {sorted(touched & pend)}: {x}")
        ft = fallthrough(blocks, order, b)
        if ft:
            flow(ft, pend)
    return problems, n_issue, n_wait


def main():
    bad = 0
    for path in sys.argv[1:]:
        for fname, (blocks, order) in parse_functions(path).items():
            problems, ni, nw = check_function(fname, blocks, order)
            if ni == 0:
                continue
            tag = re.search(r"CfgILi(\d+)ELi\d+ELi(\d+)ELi(\d+)EEELb(\d)", fname)
            label = "Cfg<%s,.,%s,%s> SLIDE=%s" % tag.groups() if tag else fname[:60]
            print(f"{path.split('/')[-1]}: {label}: {ni} issues visited, {nw} waits visited, {len(problems)} problems")
            for p in problems[:12]:
                print("   ", p)
            bad += len(problems)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
