#!/bin/bash
# Builds the HOST code of libdmmfods_hip.so (plan.cpp, capi.cpp and the host side - launchers, eligibility tests - of every kernel
# file) with AddressSanitizer + UndefinedBehaviorSanitizer against the fake HIP runtime of this directory, plus the driver.
# No GPU, no device code: `-x hip --offload-host-only`.  Output: tools/hoststub/_build/drive   (about 20 s with 8 jobs)
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC="$HERE/../../dmmfods_amd/csrc"
OUT="${1:-$HERE/_build}"
mkdir -p "$OUT"
CLANG=/opt/rocm/lib/llvm/bin/clang++
FLAGS="-x hip --offload-host-only --offload-arch=gfx950 -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -w $DRIVE_EXTRA_FLAGS"
pids=()
cc() { $CLANG $FLAGS "${@:3}" -c "$1" -o "$OUT/$2" & pids+=($!); if [ ${#pids[@]} -ge 8 ]; then wait "${pids[0]}"; pids=("${pids[@]:1}"); fi; }
for f in bw1 cf conv3 cvp cvw halo hf pig pointwise thin wg3 wg5 wgp wgpw; do cc "$SRC/$f.hip" "$f.o"; done
cc "$SRC/halo.hip" halo32.o -DHALO_F32_PART
for p in 0 1 2; do n=(f32 f16 bf16); cc "$SRC/igemm.hip" "igemm_${n[$p]}.o" -DIGEMM_PART=$p; cc "$SRC/wgrad.hip" "wgrad_${n[$p]}.o" -DWGRAD_PART=$p; done
cc "$SRC/plan.cpp" plan.o
cc "$SRC/capi.cpp" capi.o
cc "$HERE/fake_hip.cpp" fake_hip.o
cc "$HERE/drive.cpp" drive.o
for p in "${pids[@]}"; do wait "$p"; done
# (the host-only objects reference their embedded device image, which does not exist here: leave those symbols unresolved)
$CLANG -fsanitize=address,undefined -o "$OUT/drive" "$OUT"/*.o -Wl,--unresolved-symbols=ignore-in-object-files -lpthread
echo "built $OUT/drive"
