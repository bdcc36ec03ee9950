python3 tools/grad_dump.py no fp32 /tmp/a.pt 2>/dev/null
for v in e0 b32 b64 b128 lin1 lin0; do
  DMM_LIB_PATH=$PWD/build_var/lib_fb_$v.so python3 tools/grad_dump.py no fp32 /tmp/b.pt 2>/dev/null
  echo "== $v: $(python3 tools/grad_dump.py --diff /tmp/a.pt /tmp/b.pt | grep -v logits | wc -l) tensors differ; last: $(python3 tools/grad_dump.py --diff /tmp/a.pt /tmp/b.pt | grep -v logits | tail -1)"
done
