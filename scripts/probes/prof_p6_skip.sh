#!/bin/bash
# PMC comparison of the six-plane kernel with and without zero-plane skipping on SiLU-gated data (why is skipping slower there?)
# NOTE (round 3): -DMDG_I8_SKIP_ZERO6 lived in round 1; the later knob set is in scripts/probes/cov_i8_variants.patch.
export TMPDIR=/tmp
R=$PWD
for skip in 0 1; do
  (cd modegpt_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w -DMDG_I8_SKIP_ZERO6=$skip -c cov_i8.hip -o build/cov_i8.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmodegpt_hip.so build/*.o) || exit 1
  for pass in mfma fetch; do
    OUT=$R/gpurun_out/p6_${skip}_$pass
    rm -rf $OUT
    case $pass in
      fetch) ARGS="--kernel-trace --pmc FETCH_SIZE" ;;
      mfma)  ARGS="--kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT" ;;
    esac
    timeout -k 10 200 rocprofv3 $ARGS -d $OUT -o p -- python3 scripts/bench_kernels.py covi8p6 > $OUT.log 2>&1 || exit 1
    python3 scripts/rocpd_summary.py $(ls $OUT/*.db | head -1) bygrid > $R/gpurun_out/p6_${skip}_$pass.csv || exit 1
    rm -rf $OUT
    echo "== skip $skip, $pass"; grep -h "syrk_kernelILi6" $R/gpurun_out/p6_${skip}_$pass.csv | cut -c60-160
  done
done
