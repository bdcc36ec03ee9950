#!/bin/bash
# tools/build_igemm_variant.sh <name> <extra flags...>: experiment build of the three igemm translation units (fp32 / f16 / bf16) with
# the given flags, linked with the in-tree objects of everything else into build_var/lib_<name>.so
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd); cs=$root/dmmfods_amd/csrc
mkdir -p $root/build_var
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value"
hipcc $FLAGS "$@" -DIGEMM_PART=0 -c $cs/igemm.hip -o $root/build_var/${name}_igemm_f32.o &
hipcc $FLAGS "$@" -DIGEMM_PART=1 -c $cs/igemm.hip -o $root/build_var/${name}_igemm_f16.o &
hipcc $FLAGS "$@" -DIGEMM_PART=2 -c $cs/igemm.hip -o $root/build_var/${name}_igemm_bf16.o &
wait
objs=""
for o in $(sed -n 's/^OBJS = //p' $cs/Makefile); do
  case $o in igemm_f32.o|igemm_f16.o|igemm_bf16.o) objs="$objs $root/build_var/${name}_$o";; *) objs="$objs $cs/$o";; esac
done
hipcc --offload-arch=gfx950 -shared -o $root/build_var/lib_$name.so $objs
ls -la $root/build_var/lib_$name.so
