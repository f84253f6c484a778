# VALU / SALU instruction counts and busy cycles of k_paths for one frame of the bench workload
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/valu
rm -rf $O && mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing > $O/log 2>&1
python3 tools/pmc_summary.py $O/p > $O/s.json
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/valu/s.json')); k=[x for x in d if 'k_paths' in x][0]
c={n:v['mean_per_dispatch'] for n,v in d[k].items()}
gui=c['GRBM_GUI_ACTIVE']/8
print("VALU %.4g SALU %.4g | %.1f ms | VALU busy %.1f %% lane util %.1f %%"%(c['SQ_INSTS_VALU'],c['SQ_INSTS_SALU'],gui/2.4e6,400*c['SQ_ACTIVE_INST_VALU']/(1024*gui),100*c['SQ_THREAD_CYCLES_VALU']/(64*c['SQ_ACTIVE_INST_VALU'])))
PY
