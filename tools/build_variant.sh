#!/bin/bash
# tools/build_variant.sh <name> <file.hip> <extra hipcc flags...>: an experiment build of ONE translation unit, linked with the in-tree
# objects of everything else into build_var/lib_<name>.so (git-ignored; travels to the GPU box; tools/sweep_lib.sh benches it via
# DMM_LIB_PATH).  Run `make -C dmmfods_amd/csrc` first so that the other objects are current.
set -e
name=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd); cs=$root/dmmfods_amd/csrc
mkdir -p $root/build_var
# <file.hip> may be given as <other.hip>:<file.o> - compile other.hip in place of the unit that builds file.o (e.g. an old revision
# written beside the sources: git show HEAD:dmmfods_amd/csrc/bw1.hip > dmmfods_amd/csrc/bw1_old.hip)
case $src in *:*) obj=${src#*:}; src=${src%%:*};; *) obj=${src%.hip}.o;; esac
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value"
hipcc $FLAGS "$@" -c $cs/$src -o $root/build_var/${name}_$obj
objs=""
for o in $(sed -n 's/^OBJS = //p' $cs/Makefile); do
  if [ "$o" = "$obj" ]; then objs="$objs $root/build_var/${name}_$obj"; else objs="$objs $cs/$o"; fi
done
hipcc --offload-arch=gfx950 -shared -o $root/build_var/lib_$name.so $objs
ls -la $root/build_var/lib_$name.so
