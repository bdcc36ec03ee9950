"""sha256 (first 16 hex digits) over the library's sources (dmmfods_amd/csrc/*.hip *.cpp *.h + include/*.h, names and contents in
sorted order): the stamp that ties a committed profile summary (profiles/*/..._kernel_stats.json, ..._pmc_hbm_traffic.json) to the
build it was taken on.  bench.py quotes a figure from such a file only when the stamp equals that of the tree it runs from."""
import glob
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_hash(root=ROOT):
    files = []
    for pat in ("dmmfods_amd/csrc/*.hip", "dmmfods_amd/csrc/*.cpp", "dmmfods_amd/csrc/*.h", "dmmfods_amd/csrc/Makefile", "include/*.h"):
        files += glob.glob(os.path.join(root, pat))
    h = hashlib.sha256()
    for f in sorted(files):
        h.update(os.path.relpath(f, root).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    sys.stdout.write(source_hash() + "\n")
