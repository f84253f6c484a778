# usage (on the GPU box): bash tools/refresh_profiles.sh <tag>      e.g. r02
# Everything profiles/ holds for a round: the default bench line, kernel statistics + PMC of the dominant kernel for
# every BASELINE scene that is quoted, the 1/8-shard counters, the shard-rate curve and the VALU calibration.
set -e
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG
mkdir -p $O
bash tools/profile_scene.sh $TAG full_bsdf 256
bash tools/profile_scene.sh $TAG four_bunnies 256
bash tools/profile_scene.sh $TAG sixteen_lights 256
bash tools/profile_scene.sh $TAG matte 256
# one rank's shard of a 2-, 4- and 8-GPU run, on one GPU (the entries an N-GPU bench line prices rank 0's k_paths with)
for R in 2 4 8; do
  S=$O/shard$R
  mkdir -p $S
  python3 -c "from rtcuda_amd import api; print(api.build_id())" > $S/build_id.txt
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $S/pmc_sq1 -- python3 tools/shard_breakdown.py $R > $S/pmc_sq1.log 2>&1
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS --output-format csv -d $S/pmc_sq2 -- python3 tools/shard_breakdown.py $R > $S/pmc_sq2.log 2>&1
  if [ $R = 8 ]; then
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_SENDMSG SQ_INST_LEVEL_SMEM SQ_INSTS_BRANCH --output-format csv -d $S/pmc_sq3 -- python3 tools/shard_breakdown.py $R > $S/pmc_sq3.log 2>&1 || echo "sq3 pass failed (counter names)"
  fi
  python3 tools/pmc_summary.py $S/pmc_sq1 $S/pmc_sq2 $( [ $R = 8 ] && echo $S/pmc_sq3 ) > $S/pmc_summary.json
  rm -rf $S/pmc_sq1 $S/pmc_sq2 $S/pmc_sq3
  echo "shard$R done"
done
timeout -k 10 300 python tools/shard_rate.py > $O/shard_rate.txt 2>&1; grep shards $O/shard_rate.txt
# VALU calibration, plain and under the counters
timeout -k 10 120 python tools/valu_calibrate.py > $O/valu_calibration.json 2> $O/valu_calibration.err; cat $O/valu_calibration.json
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU SQ_WAVES --output-format csv -d $O/valu_pmc -- python3 tools/valu_calibrate.py > $O/valu_pmc.log 2>&1
python3 tools/pmc_summary.py $O/valu_pmc > $O/valu_calibration_pmc.json
rm -rf $O/valu_pmc
echo "calibration done"
# the default bench line LAST, after the counters of this binary are in profiles/pmc_k_paths.json (bench.py's roofline
# divides them by the launch time it measures itself): tools/update_profiles.py works on the box's copy of the repo too
python3 tools/update_profiles.py $TAG > $O/update_profiles.log 2>&1 || echo "update_profiles failed on the box"
timeout -k 10 400 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
echo "bench done"; cut -c1-240 $O/bench_n1.json
