# same-box A/B of this tree against the round-3 tree.  Prepare (CPU container, ~3 min):
#   mkdir -p build_var/r3tree && git archive 541ca33 | tar -x -C build_var/r3tree && make -C build_var/r3tree/dmmfods_amd/csrc -j6
# (build_var/ is git-ignored but travels to the GPU box)
B="--steps 16 --warmup 5 --no-cpu-baseline --no-profile"
r4() { echo "r04 $1: $(python3 bench.py --config $1 $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms']['median'])")"; }
r3() { echo "r03 $1: $(cd build_var/r3tree && python3 bench.py --config $1 $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms']['median'])")"; }
r3 c2; r4 c2; r3 c2; r4 c2; r3 c5; r4 c5; r3 c4; r4 c4; r3 c3; r4 c3
