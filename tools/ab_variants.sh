# usage (GPU box): bash tools/ab_variants.sh <out-file> <variant> [<variant> ...]   ("base" = the product library)
# One process per library build (RT_LIB_NAME selects it: `make -C rtcuda_amd/csrc variant NAME=x DEFS=...`), same box.
OUT=$1; shift
for V in "$@"; do
  if [ "$V" = base ]; then L=librtcuda_amd.so; else L=librtcuda_amd_$V.so; fi
  RT_LIB_NAME=$L timeout -k 10 120 python tools/ab_bench.py ${AB_ARGS:-} "" 2>&1 | grep -v amdgpu.ids | sed "s/(defaults)/$V/" | tee -a $OUT
done
