python3 tools/grad_dump.py no fp32 /tmp/a.pt 2>/dev/null
for v in s32 s64 s128 slin1 slin0; do
  DMM_LIB_PATH=$PWD/build_var/lib_fb_$v.so python3 tools/grad_dump.py no fp32 /tmp/b.pt 2>/dev/null
  echo "== $v: $(python3 tools/grad_dump.py --diff /tmp/a.pt /tmp/b.pt | grep -v logits | awk '$NF+0 > 1e-4' | wc -l) tensors differ by > 1e-4; last: $(python3 tools/grad_dump.py --diff /tmp/a.pt /tmp/b.pt | grep -v logits | awk '$NF+0 > 1e-4' | tail -1)"
done
python3 tools/host_bound.py c2 c5 c4
DMM_NO_OVERLAP=1 python3 tools/host_bound.py c2
