#!/bin/bash
# tests/manual/gpu_wg_ring_stress.sh -- run on the GPU box (gpurun) after building abtest/wgstress.so HERE with
#   bash tests/manual/ab_build.sh wgstress ltr_k_wg.hip -DLTR_WG_STRESS=100
#   bash tests/manual/ab_build.sh wgxstress ltr_k_exact.hip -DLTR_WG_STRESS=100     (the exact workgroup kernels)
#   bash tests/manual/ab_build.sh wgtstress ltr_k_wgt.hip -DLTR_WG_STRESS=100       (round 6: the threshold kernels as a first pass)
# The stress build makes the waves of a workgroup alternately slow (s_sleep per block of 8 steps, the slow side flipping
# every 512 steps), so that consumers lag their producers by whole rings and producers run into full rings: the
# progress-word protocol of ltr_dp_wg.hpp has to hold at any relative speed.  The workgroup-kernel parity tests
# (bit-exact against the oracle) then run on that library.
cd "$(dirname "$0")/../.."
for so in abtest/wgstress.so abtest/wgxstress.so abtest/wgtstress.so; do
  [ -f $so ] || { echo "missing $so (build it first, see the header of this script)"; continue; }
  echo "== $so"
  LTR_GPU_LIB=$PWD/$so timeout 900 python -m pytest tests/test_gpu_wg.py tests/test_gpu_align.py -m gpu -q -x 2>&1 | tail -4
done
