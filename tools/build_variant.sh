#!/bin/bash
# build the working tree's HIP library with extra compiler flags as wdpm_amd/csrc/alt_<name>_libwdpm_hip.so (A/B and timing builds)
# usage: tools/build_variant.sh <name> [flags, e.g. -DWDPM_WAVE_TIMES]
set -e
R=$(cd $(dirname $0)/.. && pwd); name=$1; shift; T=$(mktemp -d)
mkdir -p $T/wdpm_amd $T/tools; cp -r $R/wdpm_amd/csrc $T/wdpm_amd/; cp -r $R/include $T/; cp $R/tools/check_asm_loads.py $T/tools/
rm -rf $T/wdpm_amd/csrc/build $T/wdpm_amd/csrc/*.so
make -C $T/wdpm_amd/csrc lib check-asm EXTRA="$*" 2>&1 | grep -E "error|warning|asm prefetch" || true
cp $T/wdpm_amd/csrc/libwdpm_hip.so $R/wdpm_amd/csrc/alt_${name}_libwdpm_hip.so
grep -E "^\s+\.(vgpr_count|private_segment_fixed_size|name):" $T/wdpm_amd/csrc/build/wdpm_fused.s | paste - - - | grep "fused_iteration_kernelILi0ELb0ELb1ELb0ELb0ELb0E\|fused_iteration_kernelILi2ELb0ELb0ELb0ELb0ELb1E" | sed 's/_ZN12_GLOBAL__N_122//' | cut -c1-60,150-260
rm -rf $T; echo "built alt_${name}_libwdpm_hip.so with: $*"
