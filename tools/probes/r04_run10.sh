python3 tools/grad_dump.py no fp32 /tmp/a.pt 2>/dev/null
DMM_LIB_PATH=$PWD/build_var/lib_fb_s32.so python3 tools/grad_dump.py no fp32 /tmp/b.pt 2>/dev/null
python3 tools/grad_dump.py --diff /tmp/a.pt /tmp/b.pt Sequence_4.norm0 | grep -A1 "buf/\|Sequence_4.norm0" | cut -c1-700
