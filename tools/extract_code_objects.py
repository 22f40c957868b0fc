"""Extract the gfx950 code objects of a HIP shared library / object (clang offload bundles in .hip_fatbin) into a directory:
tools/extract_code_objects.py LIB OUTDIR -> OUTDIR/co_<n>.co (one per translation unit), for llvm-objdump -d."""
import os
import struct
import sys

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
data = open(sys.argv[1], "rb").read()
out = sys.argv[2]
os.makedirs(out, exist_ok=True)
pos, n = 0, 0
while True:
    pos = data.find(MAGIC, pos)
    if pos < 0:
        break
    (cnt,) = struct.unpack_from("<Q", data, pos + len(MAGIC))
    p = pos + len(MAGIC) + 8
    for _ in range(cnt):
        off, size, tl = struct.unpack_from("<QQQ", data, p)
        triple = data[p + 24:p + 24 + tl].decode()
        p += 24 + tl
        if "gfx950" in triple and size:
            path = os.path.join(out, "co_%d.co" % n)
            open(path, "wb").write(data[pos + off:pos + off + size])
            print(path, triple, size)
            n += 1
    pos += len(MAGIC)
