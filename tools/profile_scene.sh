# usage (on the GPU box): bash tools/profile_scene.sh <tag> <scene> <spp> [extra bench.py args]
# rocprofv3 kernel statistics + the PMC set of the dominant kernel for one BASELINE scene, each counter group in its
# own pass (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; no --pmc together with trace domains
# other than --kernel-trace).  Results under gpurun_out/<tag>/<scene>/; tools/update_profiles.py copies the summaries
# into profiles/.
set -e
TAG=$1; SCENE=$2; SPP=$3; shift 3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG/$SCENE
rm -rf $O && mkdir -p $O
python3 -c "from rtcuda_amd import api; print(api.build_id())" > $O/build_id.txt   # the build these counters belong to
B="bench.py --scene $SCENE --spp $SPP --no-cpu-baseline --no-extras $*"   # (--no-extras: every k_paths dispatch of the process is this frame's)
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $B --steps 3 --warmup 1 > $O/stats.log 2>&1
echo "$SCENE stats done"
P="$B --steps 1 --warmup 0 --no-kernel-timing"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $P > $O/pmc_fetch.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_write -- python3 $P > $O/pmc_write.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq1 -- python3 $P > $O/pmc_sq1.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS --output-format csv -d $O/pmc_sq2 -- python3 $P > $O/pmc_sq2.log 2>&1
# the texture addresser (vector-memory path): busy cycles, averaged over the units and of the busiest one (own pass)
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max --output-format csv -d $O/pmc_ta -- python3 $P > $O/pmc_ta.log 2>&1
echo "$SCENE pmc done"
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_sq1 $O/pmc_sq2 $O/pmc_ta > $O/pmc_summary.json
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
head -4 $O/kernel_stats.csv
rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/pmc_sq1 $O/pmc_sq2 $O/pmc_ta   # raw traces are large; the summaries are what is kept
