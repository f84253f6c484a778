# usage (GPU box): bash tools/lab/icache_counters.sh <outdir>   -- instruction-cache and instruction-fetch counters of k_paths, C2 frame
set -e
O=${1:-gpurun_out/ic}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $O
P="bench.py --scene full_bsdf --spp 256 --no-cpu-baseline --no-extras --steps 1 --warmup 0 --no-kernel-timing"
n=0
for group in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQC_ICACHE_BUSY_CYCLES SQC_DCACHE_REQ SQC_DCACHE_MISSES"; do
  n=$((n+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $group --output-format csv -d $O/k$n -- python3 $P > $O/k$n.log 2>&1
done
python3 tools/pmc_summary.py $(ls -d $O/k*/) > $O/summary.json
rm -rf $O/k[0-9]*
python3 - <<PY
import json
d=json.load(open("$O/summary.json"))
for k,v in d.items():
    if "k_paths" in k: print({c: f"{x['mean_per_dispatch']:.4g}" for c,x in v.items()})
PY
