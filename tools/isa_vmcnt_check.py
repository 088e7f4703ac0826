#!/usr/bin/env python3
"""Build-time check of the hand-counted `s_waitcnt vmcnt(N)` contract of the z-slide convolution's helper waves
(calodiffusion_amd/csrc/kernels_conv_zs.hip).

The helper waves await a plane's global loads with `s_waitcnt vmcnt(N)`, N = the number of vector-memory operations the wave
issues between those loads and the wait (the 8 row stores of one epilogue).  vmcnt retires in order, so the count is only right
if, on EVERY path of the compiled code, exactly N stores -- and no other vector-memory instruction -- lie between a plane-load
group and the wait.  Commit 2271921 is the failure this guards against: the compiler merged the eight sink stores of a peeled
interval into one, the count was short and a plane was converted before it had landed.

The three kinds of instructions carry markers in their inline asm (`; zs_plane_load`, `; zs_row_store`, `; zs_landed`).  For every
marked wait with N > 0 the checker walks the control-flow graph of the kernel BACKWARDS from the wait, counting vector-memory
instructions, until it meets a plane load on each path, and asserts

  * every instruction counted is a marked row store (no compiler-generated load/store sneaks into the window),
  * the count at the first plane load met is a multiple of N on every path (statically there are paths on which an interval
    issues no loads -- the wait is skipped on those at run time -- so 2N, 3N are legitimate; anything else, 0 included, means
    an interval does not issue exactly N stores), and N itself occurs,
  * plane loads come in whole groups (ZS_NSL per plane).
A path that meets an `s_waitcnt vmcnt(0)` first is trivially fine (everything older has landed; the prologue's planes).

Usage:  isa_vmcnt_check.py <file.s> [kernel-symbol-substring]      (exit status 0 = contract holds)
        hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o file.s kernels_conv_zs.hip
"""
from __future__ import annotations

import re
import sys
from typing import Dict, List, Tuple

VMEM = re.compile(r"^(global_|buffer_|flat_|scratch_)(load|store|atomic)")


def kernels(text: str) -> Dict[str, List[str]]:
    """{symbol: instruction/label lines} of every kernel body in an AMDGPU .s file."""
    out: Dict[str, List[str]] = {}
    lines = text.split("\n")
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z\w+):", lines[i])
        if m:
            j = i + 1
            while j < len(lines) and not lines[j].strip().startswith(".Lfunc_end") and ".end_amdhsa_kernel" not in lines[j]:
                j += 1
            out[m.group(1)] = lines[i + 1:j]
            i = j
        i += 1
    return out


class Block:
    def __init__(self, label):
        self.label = label
        self.ins: List[str] = []
        self.succ: List[str] = []
        self.pred: List[str] = []


def build_cfg(body: List[str]) -> Dict[str, Block]:
    blocks: Dict[str, Block] = {}
    cur = Block("<entry>")
    blocks[cur.label] = cur
    order = [cur]
    n_anon = 0
    for raw in body:
        t = raw.strip()
        if not t or t.startswith((";", "//", ".p2align", ".loc", ".file", ".cfi", "#")):
            continue
        m = re.match(r"^(\.LBB\w+):", t)
        if m:
            nb = Block(m.group(1))
            blocks[nb.label] = nb
            order.append(nb)
            cur = nb
            continue
        if t.startswith("."):
            continue
        cur.ins.append(t)
        op = t.split()[0]
        if op.startswith("s_cbranch") or op in ("s_branch", "s_endpgm", "s_setpc_b64"):
            # the next instruction starts a new (anonymous) block
            n_anon += 1
            nb = Block(f"<anon{n_anon}>")
            blocks[nb.label] = nb
            order.append(nb)
            cur = nb
    for k, b in enumerate(order):
        last = b.ins[-1].split() if b.ins else []
        op = last[0] if last else ""
        fall = True
        if op.startswith("s_cbranch"):
            b.succ.append(last[1])
        elif op == "s_branch":
            b.succ.append(last[1])
            fall = False
        elif op in ("s_endpgm", "s_setpc_b64"):
            fall = False
        if fall and k + 1 < len(order):
            b.succ.append(order[k + 1].label)
    for b in blocks.values():
        for s in b.succ:
            if s not in blocks:
                raise SystemExit(f"branch to unknown label {s}")
            blocks[s].pred.append(b.label)
    return blocks


def check_kernel(sym: str, body: List[str], verbose=True) -> Tuple[int, List[str]]:
    """Returns (number of counted waits checked, list of violations)."""
    blocks = build_cfg(body)
    errors: List[str] = []
    checked = 0
    # plane loads come in whole groups: consecutive marked loads, uninterrupted by other vector-memory instructions.  A group has
    # one load per staging piece: 5 (ZS_NSL), or the last template argument of the one-wave-per-SIMD kernel's instances
    # (conv_zslide_sw_f16x2_kernel<ACC, MODE, DBG, NSL>: ...ELi<DBG>ELi<NSL>EEEv...)
    m = re.search(r"conv_zslide_sw_f16x2_kernelILb[01]ELi\d+ELi\d+ELi(\d+)EEE", sym)
    group = int(m.group(1)) if m else 5
    for b in blocks.values():
        run = 0
        for t in b.ins + ["<end>"]:
            if "zs_plane_load" in t:
                run += 1
            elif VMEM.match(t) or t == "<end>":
                if run and run % group:
                    errors.append(f"{sym}: plane-load group of {run} in block {b.label} (expected multiples of {group})")
                run = 0
    for b in blocks.values():
        for pos, t in enumerate(b.ins):
            if "zs_landed" not in t:
                continue
            n = int(re.search(r"vmcnt\((\d+)\)", t).group(1))
            if n == 0:
                continue
            checked += 1
            counts = set()
            drained = False
            # backward walk: state = (block, index of the next instruction to look at going up, count so far)
            stack = [(b.label, pos - 1, 0)]
            seen = set()
            while stack:
                lab, i, c = stack.pop()
                blk = blocks[lab]
                done = False
                while i >= 0:
                    ins = blk.ins[i]
                    if "zs_plane_load" in ins:
                        counts.add(c)
                        done = True
                        break
                    if ins.startswith("s_waitcnt") and re.search(r"vmcnt\(0\)", ins):
                        done = drained = True  # everything older has landed: nothing left to count on this path
                        break
                    if VMEM.match(ins):
                        if "zs_row_store" not in ins:
                            errors.append(f"{sym}: unmarked vector-memory instruction inside the counted window of "
                                          f"`{t}` ({lab}): {ins}")
                            done = True
                            break
                        c += 1
                        if c > 3 * n:  # three intervals without a load: statically possible, never awaited at run time
                            done = True
                            break
                    i -= 1
                if done:
                    continue
                if not blk.pred:
                    if lab == "<entry>":  # (other predecessor-less blocks are dead fall-throughs behind an s_branch)
                        errors.append(f"{sym}: a path from the kernel entry reaches `{t}` without a plane load or a full drain")
                    continue
                for p in blk.pred:
                    key = (p, c)
                    if key not in seen:
                        seen.add(key)
                        stack.append((p, len(blocks[p].ins) - 1, c))
            bad = sorted(c for c in counts if c % n or c < n)
            if bad:
                errors.append(f"{sym}: `{t}` in {b.label}: paths with {bad} vector-memory operations between the plane loads and "
                              f"the wait (every interval must issue exactly {n})")
            if n not in counts and not (drained and not counts):
                errors.append(f"{sym}: `{t}` in {b.label}: no path with exactly {n} stores after the plane loads (counts {sorted(counts)})")
            if verbose:
                print(f"{sym[:60]}: vmcnt({n}) wait in {b.label}: path counts {sorted(counts)}")
    return checked, errors


def check_file(path: str, pattern: str = "conv_zslide", verbose=True, require=1) -> None:
    ks = {k: v for k, v in kernels(open(path).read()).items() if pattern in k}
    if not ks:
        raise SystemExit(f"no kernel matching {pattern!r} in {path}")
    total, errors = 0, []
    for sym, body in ks.items():
        n, e = check_kernel(sym, body, verbose)
        total += n
        errors += e
    if total < require:
        errors.append(f"only {total} counted waits found (expected >= {require}): markers missing from the inline asm?")
    if errors:
        raise SystemExit("vmcnt contract violated:\n  " + "\n  ".join(sorted(set(errors))))
    if verbose:
        print(f"ok: {total} counted waits in {len(ks)} kernels hold their contract")


if __name__ == "__main__":
    check_file(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "conv_zslide")
