#!/bin/bash
# usage (GPU box): tools/trace_kernels.sh <out.txt> <bench args ...> - rocprofv3 kernel trace of one bench run, per-kernel average durations
export TMPDIR=/tmp
OUT=$1; shift
D=gpurun_out/trace_tmp; rm -rf $D
rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" > $D.log 2>&1 || { tail -5 $D.log; exit 1; }
python3 - "$OUT" "$*" <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/trace_tmp/*/*_kernel_stats.csv")[0]
with open(sys.argv[1], "a") as o:
    o.write(f"== bench.py {sys.argv[2]}\n")
    for r in list(csv.DictReader(open(f)))[:6]:
        o.write(f"  {r['Name'][:100]:<100} calls {r['Calls']:>5} avg {float(r['AverageNs'])/1e3:9.1f} us min {float(r['MinNs'])/1e3:9.1f} max {float(r['MaxNs'])/1e3:9.1f}\n")
PY
