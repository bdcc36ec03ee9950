O=gpurun_out/r04_cumask; mkdir -p $O
export CLASSES=wg3.n128,other
for m in "" "128:s" "64:s" "96:s" "64:f" "64:x" "32:s"; do
  if [ -z "$m" ]; then unset DMM_SIDE_CUS; else export DMM_SIDE_CUS=$m; fi
  echo "== mask '$m'"; bash tools/sweep_lib.sh r04_cumask_$(echo $m | tr ':' '_') main 2>&1 | tail -1
done | tee $O/sweep.txt
