#!/bin/bash
# A kernel-variant build next to the product library, for A/B runs inside ONE gpurun call:
#   scripts/build_variant.sh NAME "-DFD_BCS=8 -DFD_BCT=512"     ->  finitediff.jl_amd/lib/variants/libfdjac_NAME.so
# and on the GPU box:  scripts/with_variant.sh NAME <command>   (swaps the library in for the command, restores it after)
set -e
cd "$(dirname "$0")/../finitediff.jl_amd/csrc"
name=$1; shift
mkdir -p ../lib/variants
# only the translation units that see the kernels are rebuilt with the flags; the rest are copied from the product build
mkdir -p _obj_$name
for o in _obj/*.o; do b=$(basename $o); case $b in fdjac_builtin_f.o|fdjac_f32.o|fdjac_blocksolve.o|fdjac_bandsolve.o|fdjac_kernels.o|fdjac_api.o) ;; *) cp -u $o _obj_$name/$b;; esac; done
make -j4 OBJDIR=_obj_$name OUT=$(pwd)/../lib/variants/libfdjac_$name.so EXTRA="$*" 2>&1 | grep -E "error|warning" -A4 || true
ls -la ../lib/variants/libfdjac_$name.so
