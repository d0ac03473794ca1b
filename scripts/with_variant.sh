#!/bin/bash
# scripts/with_variant.sh NAME <command...>: run the command with finitediff.jl_amd/lib/variants/libfdjac_NAME.so in the product library's place
cd "$(dirname "$0")/.."
name=$1; shift
lib=finitediff.jl_amd/lib
cp $lib/libfdjac.so $lib/libfdjac.so.keep
cp $lib/variants/libfdjac_$name.so $lib/libfdjac.so
"$@"; rc=$?
mv $lib/libfdjac.so.keep $lib/libfdjac.so
exit $rc
