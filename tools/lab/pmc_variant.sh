# usage (GPU box): bash tools/lab/pmc_variant.sh <outdir> <lib-name> [scene]
# one SQ counter pass of the C2 frame with the given library build (RT_LIB_NAME); lab
set -e
O=$1; L=$2; SCENE=${3:-full_bsdf}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $O
export RT_LIB_NAME=$L
P="bench.py --scene $SCENE --spp 256 --no-cpu-baseline --no-extras --steps 1 --warmup 0 --no-kernel-timing"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/p1 -- python3 $P > $O/p1.log 2>&1
python3 tools/pmc_summary.py $O/p1 > $O/summary.json
python3 - <<PY
import json
d=json.load(open("$O/summary.json"))
for k,v in d.items():
    if "k_paths" in k: print("$L", {c: f"{x['mean_per_dispatch']:.4g}" for c,x in v.items()})
PY
rm -rf $O/p1
