# usage (on the GPU box): bash tools/lab/ta_counters.sh <outdir>
# Texture-addresser / vector-L1 counters of k_paths on the C2 frame and of the gather microbenchmark (tools/lab/gather_rate),
# one small counter group per pass, only counters this rocprofv3 lists (a group the hardware cannot collect in one pass makes
# rocprofv3 abort and then hang in its finalisation until the timeout: three TA stall counters together did).  Lab, not product.
# The TA_BUSY pass is part of tools/profile_scene.sh since; this script is for looking further.
set -e
O=${1:-gpurun_out/ta}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $O
rocprofv3 --list-avail > $O/avail.txt 2>&1 || rocprofv3 -L > $O/avail.txt 2>&1 || true
grep -o "\bTA_[A-Za-z0-9_]*\|\bTCP_[A-Za-z0-9_]*\|\bTD_[A-Za-z0-9_]*" $O/avail.txt | sort -u > $O/avail_ta_tcp.txt || true
wc -l $O/avail_ta_tcp.txt
have() { grep -qx "$1" $O/avail_ta_tcp.txt; }
P="bench.py --scene full_bsdf --spp 256 --no-cpu-baseline --no-extras --steps 1 --warmup 0 --no-kernel-timing"
n=0
for group in "TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max" \
             "TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum" \
             "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
             "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  names=""
  for c in $group; do if have $c || echo $c | grep -q "^SQ_\|^GRBM"; then names="$names $c"; fi; done
  [ -z "$names" ] && continue
  n=$((n+1))
  echo "pass $n:$names"
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $names --output-format csv -d $O/k$n -- python3 $P > $O/k$n.log 2>&1
  timeout -k 10 100 rocprofv3 --kernel-trace --pmc $names --output-format csv -d $O/g$n -- ./tools/lab/gather_rate 46 8 > $O/g$n.log 2>&1
done
python3 tools/pmc_summary.py $(ls -d $O/k*/) > $O/k_paths_ta.json
python3 tools/pmc_summary.py $(ls -d $O/g*/) > $O/gather_ta.json
rm -rf $O/k[0-9]* $O/g[0-9]*.log $O/g[0-9]*/
echo done
