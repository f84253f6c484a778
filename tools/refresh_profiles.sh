set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r01
rm -rf $O && mkdir -p $O
timeout -k 10 300 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
echo "bench done"; cat $O/bench_n1.json | cut -c1-200
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/stats.log 2>&1
echo "stats done"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing > $O/pmc_fetch.log 2>&1
echo "fetch done"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing > $O/pmc_write.log 2>&1
echo "write done"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq1 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing > $O/pmc_sq1.log 2>&1
echo "sq1 done"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_sq2 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing > $O/pmc_sq2.log 2>&1
echo "sq2 done"
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_sq1 $O/pmc_sq2 > $O/pmc_summary.json
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
cat $O/kernel_stats.csv | head -8
timeout -k 10 300 python tools/shard_rate.py > $O/shard_rate.txt 2>&1; grep shards $O/shard_rate.txt
for s in matte four_bunnies sixteen_lights; do timeout -k 10 100 python bench.py --no-cpu-baseline --scene $s 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['scene'], d['value'], d['ms_per_frame'])"; done | tee $O/scenes.txt
