#!/bin/bash
# bash tools/step_list.sh <tag> [bench args]: every kernel launch of ONE sampling step (the last of the run) with its duration, grid,
# workgroup size, LDS and register allocation, from a rocprofv3 kernel trace of the bench loop
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/${1:-steplist}; shift
mkdir -p $out
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $out/tr -o bench -- python3 bench.py --steps 1 --warmup 0 --no-extra --no-clocks "$@" > $out/bench.json 2> $out/tr.err; echo "trace rc=$?"
python3 - "$(find $out/tr -name '*kernel_trace.csv' | head -1)" > $out/step.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "load_step_kernel" in r["Kernel_Name"]]
a, b = marks[-2], marks[-1]
tot = 0
for r in rows[a:b]:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); tot += d
    print(f"{d/1e3:8.1f} us  grid {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):6d}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']} wg {r['Workgroup_Size_X']:>4} lds {r.get('LDS_Block_Size','?'):>6} vgpr {r.get('VGPR_Count','?'):>3}/{r.get('Accum_VGPR_Count','?'):>3}  {r['Kernel_Name'][:90]}")
print(f"sum {tot/1e3:.1f} us, wall {(int(rows[b]['Start_Timestamp'])-int(rows[a]['Start_Timestamp']))/1e3:.1f} us, {b-a} launches")
PY
rm -rf $out/tr; cat $out/step.txt
