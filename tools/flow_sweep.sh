# usage (GPU box): bash tools/flow_sweep.sh  -- knob sweep of the experimental k_flow kernel on the bench workload
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/flow_sweep; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout -k 10 100 python bench.py --no-cpu-baseline --steps 2 --warmup 1 > $O/b_$tag.json 2> $O/b_$tag.err; python3 -c "
import json; d=json.load(open('$O/b_$tag.json')); print('$tag', d['value'], d['ms_per_step'])"; }
run paths RT_FLOW=0
for t in 24 36 48 56; do for a in 32 48; do run t${t}a${a} RT_FLOW=96 RT_FLOW_TURN=$t RT_FLOW_ADV=$a; done; done
