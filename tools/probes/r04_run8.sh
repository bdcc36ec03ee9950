python3 tools/grad_dump.py no fp32 /tmp/a.pt 2>/dev/null
python3 tools/grad_dump.py no fp32 /tmp/a2.pt 2>/dev/null
echo "== main vs main"; python3 tools/grad_dump.py --diff /tmp/a.pt /tmp/a2.pt | tail -5
DMM_LIB_PATH=$PWD/build_var/lib_fb_e0.so python3 tools/grad_dump.py no fp32 /tmp/b.pt 2>/dev/null
DMM_LIB_PATH=$PWD/build_var/lib_fb_e0.so python3 tools/grad_dump.py no fp32 /tmp/b2.pt 2>/dev/null
echo "== e0 vs e0"; python3 tools/grad_dump.py --diff /tmp/b.pt /tmp/b2.pt | tail -5
echo "== main vs e0"; python3 tools/grad_dump.py --diff /tmp/a.pt /tmp/b.pt | tail -32
