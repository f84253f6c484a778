set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r8
rm -rf $O && mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq1 -- python3 tools/shard_breakdown.py 8 > $O/pmc_sq1.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS --output-format csv -d $O/pmc_sq2 -- python3 tools/shard_breakdown.py 8 > $O/pmc_sq2.log 2>&1
python3 tools/pmc_summary.py $O/pmc_sq1 $O/pmc_sq2 > $O/pmc_summary.json
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r8/pmc_summary.json'))
k=[x for x in d if 'k_paths' in x][0]
c={n:v['mean_per_dispatch'] for n,v in d[k].items()}
print(k[:60])
for n,v in c.items(): print(n, "%.4g"%v)
gui=c['GRBM_GUI_ACTIVE']/8
print("kernel ms %.1f VALU busy %.1f%% lane util %.1f%%"%(gui/2.4e6, 400*c['SQ_ACTIVE_INST_VALU']/(1024*gui), 100*c['SQ_THREAD_CYCLES_VALU']/(64*c['SQ_ACTIVE_INST_VALU'])))
print("avg VMEM latency (cycles) ~ INST_LEVEL_VMEM/INSTS_VMEM = %.0f ; LDS %.0f"%(c['SQ_INST_LEVEL_VMEM']/c['SQ_INSTS_VMEM'], c.get('SQ_INST_LEVEL_LDS',0)/max(c['SQ_INSTS_LDS'],1)))
print("wave cycles per wave %.3g ; wait_inst_any/wave_cycles %.2f ; wait_any/wave_cycles %.2f"%(c['SQ_WAVE_CYCLES']/c['SQ_WAVES'], c['SQ_WAIT_INST_ANY']/c['SQ_WAVE_CYCLES'], c['SQ_WAIT_ANY']/c['SQ_WAVE_CYCLES']))
PY
