# usage (GPU box): bash tools/variant_sweep.sh "<lib suffixes>" "<ENV=val ...>;<ENV=val ...>;..."
# Msamples/s of the bench workload for experiment builds (make -C rtcuda_amd/csrc variant NAME=.. DEFS=..) and knob sets
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/variants; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout -k 10 100 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/b_$tag.json 2> $O/b_$tag.err; python3 -c "
import json; d=json.load(open('$O/b_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['per_frame']['shade_events'])"; }
run base RT_NONE=0
for v in $1; do run $v RT_LIB_NAME=librtcuda_amd_$v.so; done
IFS=';' read -ra SETS <<< "$2"
i=0
for s in "${SETS[@]}"; do i=$((i+1)); run knob$i $s; echo "   knob$i = $s"; done
