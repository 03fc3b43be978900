#!/bin/bash
# The 7-block dual class at two waves per SIMD (round 3: wrong rows at C5 scale; then blamed on 24 bytes of scratch).
# Round 4, first pass (gpurun_out/dual7_*.log): it fails WITHOUT scratch too (ratings loaded behind the Gramian loop:
# 256 registers, 0 bytes), with and without the other queues running beside it.  Second pass: what does it depend on?
#   libycnr_d7w2.so        two waves per SIMD, no scratch
#   libycnr_d7w2_nodpp.so  the same with the v_readlane pivots instead of the inline-asm DPP pivots
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
AB=you-can-not-recommend_amd/csrc/devtest/ablibs
T="tests/test_gpu_hazard.py::test_every_dual_class_at_full_occupancy"
run() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 600 python -m pytest "$T" -m gpu -q -p no:cacheprovider -k 256 > gpurun_out/dual7_$name.log 2>&1
  echo "$name: $(tail -n 1 gpurun_out/dual7_$name.log)"; grep -h "rows differ\|sampled rows off" gpurun_out/dual7_$name.log | cut -c1-220 | grep AssertionError | head -3
}
run shipped YCNR_X=1
run w2 YCNR_ALS_LIB=$PWD/$AB/libycnr_d7w2.so
run w2_f32gram YCNR_ALS_LIB=$PWD/$AB/libycnr_d7w2.so YCNR_NO_DUAL_X6=1
run w2_ldspad40k YCNR_ALS_LIB=$PWD/$AB/libycnr_d7w2.so YCNR_DUAL_LDSPAD=38400
run w2_ldspad20k YCNR_ALS_LIB=$PWD/$AB/libycnr_d7w2.so YCNR_DUAL_LDSPAD=17920
run w2_nodpp YCNR_ALS_LIB=$PWD/$AB/libycnr_d7w2_nodpp.so
run w2_nodpp_f32gram YCNR_ALS_LIB=$PWD/$AB/libycnr_d7w2_nodpp.so YCNR_NO_DUAL_X6=1
