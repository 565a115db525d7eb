#!/bin/bash
# scripts/disasm.sh <lib.so> <out.s> : gfx950 ISA of every kernel in a built library (llvm-objdump on the embedded code objects)
set -e
LIB="$1"; OUT="$2"; T=$(mktemp -d)
LL=/opt/rocm/lib/llvm/bin
# the fat binary sits in .hip_fatbin: unbundle the gfx950 code object
$LL/llvm-objcopy --dump-section .hip_fatbin="$T/fat.bin" "$LIB" 2>/dev/null || objcopy --dump-section .hip_fatbin="$T/fat.bin" "$LIB"
python3 - "$T/fat.bin" "$T" <<'PY'
import sys, struct
b = open(sys.argv[1], 'rb').read()
# concatenated clang offload bundles ("__CLANG_OFFLOAD_BUNDLE__"), one per translation unit
magic = b"__CLANG_OFFLOAD_BUNDLE__"
pos, n = 0, 0
while True:
    i = b.find(magic, pos)
    if i < 0: break
    cnt = struct.unpack_from("<Q", b, i + 24)[0]
    o = i + 32
    for _ in range(cnt):
        off, size, tl = struct.unpack_from("<QQQ", b, o); o += 24
        triple = b[o:o + tl].decode(); o += tl
        if "gfx950" in triple and size:
            open(f"{sys.argv[2]}/co{n}.o", "wb").write(b[i + off:i + off + size]); n += 1
    pos = i + 24
print(n, "code objects", file=sys.stderr)
PY
: > "$OUT"
for f in "$T"/co*.o; do $LL/llvm-objdump -d --mcpu=gfx950 "$f" >> "$OUT"; done
rm -rf "$T"
