#!/bin/bash
# build the HIP library of a git revision (default HEAD) as wdpm_amd/csrc/alt_<name>_libwdpm_hip.so for A/B runs on one box
# usage: tools/build_alt.sh <name> [revision]
set -e
R=$(cd $(dirname $0)/.. && pwd); name=$1; rev=${2:-HEAD}; T=$(mktemp -d)
(cd $R && git archive $rev wdpm_amd/csrc include tools/check_asm_loads.py) | tar -x -C $T
make -C $T/wdpm_amd/csrc lib > /dev/null
cp $T/wdpm_amd/csrc/libwdpm_hip.so $R/wdpm_amd/csrc/alt_${name}_libwdpm_hip.so
rm -rf $T; echo "built alt_${name}_libwdpm_hip.so from $rev"
