#!/bin/bash
# A/B battery to run FIRST next round (one gpurun call, ~8 GPU-minutes): settles what this round left open.
#   1. eager step with / without the plan cache on the non-ResNet GEMMs, same box (the 55.9 vs 45.3 ms question)
#   2. graph step with candidate timing off / on / on incl. fp32 plans (validated to 1e-5)
#   3. weight gradients as ONE GEMM each (OCPG_SPLIT_K=0): with candidate timing hipBLASLt's own split-K kernels may beat the
#      row-split batched GEMM + sum (82 reductions per step)
# Output: gpurun_out/ab/*.json (bench lines) and a one-line summary per run.
O=gpurun_out/ab; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout -k 10 240 python bench.py --no-cpu-baseline --no-kernel-timing --steps 12 --warmup 4 ${EXTRA} > $O/$tag.json 2> $O/$tag.err || { tail -3 $O/$tag.err; return 1; }
        python3 -c "import json,sys; d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['ms_per_step'],2), 'ms', d.get('gemm_plans'))"; }
EXTRA=--eager run eager_planned OCPG_PLANNED_GEMM=1 && EXTRA=--eager run eager_atmm OCPG_PLANNED_GEMM=0 && \
EXTRA= run graph_tune_off OCPG_GEMM_TUNE=0 && EXTRA= run graph_tune_on OCPG_GEMM_TUNE=1 && EXTRA= run graph_tune_fp32 OCPG_GEMM_TUNE_FP32=1 && EXTRA= run graph_no_manual_splitk OCPG_SPLIT_K=0
# Round 4, second half -- what to measure first next round (section 9 of DESIGN.md, method notes):
#   4. the MSDeformAttn forward and row-gather kernels on COLD operands (in the step they take 133 / 187 us against 116-122 / 180 back to back):
#        MSDA_COLD=1 MSDA_MODES=ring MSDA_COLS=1 python tools/bench_msda.py        vs        MSDA_MODES=ring MSDA_COLS=1 python tools/bench_msda.py
#   5. kernels that hold scratch or sit at a register cliff (compile with -S and read .amdhsa_private_segment_fixed_size / next_free_vgpr):
#      conv3x3_mfma 164-172 registers + 32 B (three waves per SIMD; 128 would be four), k_conv_n16 168 + 60 B, dal_bwd<*, 8> 241-255 + 144 B
#      (config #5's wide LayerNorms) -- try `#pragma unroll 1` on their instantiated inner loops first (the scatter's sum loop: no scratch, -9 %)
(MSDA_COLD=1 MSDA_MODES=ring MSDA_COLS=1 timeout -k 10 120 python tools/bench_msda.py 2>&1 | grep "Lq=" | sed "s/^/cold /"; MSDA_MODES=ring MSDA_COLS=1 timeout -k 10 120 python tools/bench_msda.py 2>&1 | grep "Lq=" | sed "s/^/warm /") > $O/msda_cold_vs_warm.txt; cat $O/msda_cold_vs_warm.txt
#   6. the exact-for-non-finite-weights form of the scatter's sum loop: tools/build_variant_one.sh zi ocpg_amd/csrc/msda_col.hip -DEXP_ZERO_ITEM=1 &&
#      tools/ab_gv_variants.sh "default zi" && OCPG_HIP_LIB=$PWD/ocpg_amd/lib/libocpg_hip_zi.so python -m pytest tests/test_msda_gpu.py -q -m gpu
