#!/bin/bash
# tools/build_lab.sh: the whole library with -DDMM_LAB=1 (every lab knob of common.h read from the environment again) as
# build_var/lib_lab.so (git-ignored; travels to the GPU box; used through DMM_LIB_PATH by the sweep scripts under tools/probes/).
set -e
root=$(cd "$(dirname "$0")/.." && pwd); cs=$root/dmmfods_amd/csrc; out=$root/build_var/lab; mkdir -p $out
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value -DDMM_LAB=1 $LAB_EXTRA"
pids=()
cc() { hipcc $FLAGS "${@:3}" -c $cs/$1 -o $out/$2 & pids+=($!); if [ ${#pids[@]} -ge 8 ]; then wait "${pids[0]}"; pids=("${pids[@]:1}"); fi; }
for f in bw1 cf conv3 cvp cvw halo hf pig pointwise thin wg3 wg5 wgp wgpw; do cc $f.hip $f.o; done
cc halo.hip halo32.o -DHALO_F32_PART
for p in 0 1 2; do n=(f32 f16 bf16); cc igemm.hip igemm_${n[$p]}.o -DIGEMM_PART=$p; cc wgrad.hip wgrad_${n[$p]}.o -DWGRAD_PART=$p; done
cc plan.cpp plan.o -x hip; cc capi.cpp capi.o -x hip
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -o $root/build_var/lib_lab.so $out/*.o
ls -la $root/build_var/lib_lab.so
