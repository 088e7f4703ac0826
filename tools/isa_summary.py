#!/usr/bin/env python3
"""Summarise the gfx950 ISA of one kernel: instruction mix and the order of MFMA / LDS / global / wait instructions.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o /tmp/k.s calodiffusion_amd/csrc/kernels_conv.hip
    python tools/isa_summary.py /tmp/k.s _ZN2cd17conv3_flat_kernelILi2ELi1EEEvNS_12ConvFlatArgsE
"""
import collections
import sys

path, sym = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(sym + ":"))
end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i] or lines[i].strip().startswith(".Lfunc_end"))
body = lines[start:end]
c = collections.Counter()
seq = []
for l in body:
    t = l.strip()
    op = t.split(" ")[0]
    if op.startswith(("v_mfma", "s_waitcnt", "ds_read", "ds_write", "global_load", "s_barrier", "scratch_", "global_store", "v_cndmask")):
        c[op] += 1
    if op.startswith("v_mfma"):
        seq.append("M")
    elif op.startswith("s_waitcnt"):
        seq.append("[" + t.replace("s_waitcnt ", "").replace("vmcnt", "v").replace("lgkmcnt", "l") + "]")
    elif op.startswith("ds_read"):
        seq.append("L")
    elif op.startswith("global_load"):
        seq.append("G")
    elif op.startswith("s_barrier"):
        seq.append("|B|")
    elif op.startswith("ds_write"):
        seq.append("W")
    elif op.startswith("global_store"):
        seq.append("S")
    elif op.startswith(("s_cbranch", "s_branch")):
        seq.append("~")
print(len(body), "lines")
print(dict(c))
print("".join(seq)[: int(sys.argv[3]) if len(sys.argv) > 3 else 2500])
for l in lines[end:end + 60]:
    if any(k in l for k in ("vgpr_count", "sgpr_count", "lds_size", "scratch", "Occupancy", "NumVgprs", "ScratchSize")):
        print(l.strip())
