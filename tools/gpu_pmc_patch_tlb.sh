#!/bin/bash
# PMC of merge_patch_hot on a sparse queue: address-translation counters (GPU box). usage: tools/gpu_pmc_patch_tlb.sh "<bench args>" <out>
export TMPDIR=/tmp
D=gpurun_out/pmc_tlb; rm -rf $D
for C in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCP_UTCL1_STALL_ON_TRANSLATION_sum TCP_PENDING_STALL_CYCLES_sum SQ_WAVE_CYCLES SQ_WAIT_ANY" "TCC_MISS_sum TCC_HIT_sum SQ_INSTS_VMEM_RD"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-30)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D/$N -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline $1 > $D.log 2>&1 || { echo "pmc $C failed"; tail -3 $D.log; }
done
python3 - "$2" "$1" <<'PY'
import csv, glob, statistics, sys
csv.field_size_limit(1 << 30)
tot = {}
for f in glob.glob("gpurun_out/pmc_tlb/*/*/*_counter_collection.csv"):
    per = {}
    for r in csv.DictReader(open(f)):
        if "merge_patch_hot" in r["Kernel_Name"]:
            per.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in per.items():
        tot[k] = statistics.mean(v)
with open(sys.argv[1], "a") as o:
    o.write(f"== merge_patch_hot, bench.py {sys.argv[2]}\n")
    for k, v in sorted(tot.items()):
        o.write(f"  {k}: {v:.6g}\n")
print(open(sys.argv[1]).read())
PY
rm -rf $D
