#!/bin/bash
# scripts/kernel_regs.sh <lib.so> [filter]: VGPR / AGPR / SGPR / LDS / scratch of every kernel (code-object metadata)
LIB="$1"; T=$(mktemp -d); LL=/opt/rocm/lib/llvm/bin
$LL/llvm-objcopy --dump-section .hip_fatbin="$T/fat.bin" "$LIB" 2>/dev/null
python3 - "$T/fat.bin" "$T" <<'PY'
import sys, struct
b = open(sys.argv[1], 'rb').read(); magic = b"__CLANG_OFFLOAD_BUNDLE__"; pos = n = 0
while True:
    i = b.find(magic, pos)
    if i < 0: break
    cnt = struct.unpack_from("<Q", b, i + 24)[0]; o = i + 32
    for _ in range(cnt):
        off, size, tl = struct.unpack_from("<QQQ", b, o); o += 24
        triple = b[o:o + tl].decode(); o += tl
        if "gfx950" in triple and size:
            open(f"{sys.argv[2]}/co{n}.o", "wb").write(b[i + off:i + off + size]); n += 1
    pos = i + 24
PY
for f in "$T"/co*.o; do $LL/llvm-readelf --notes "$f"; done | python3 -c "
import sys, re, subprocess
txt = sys.stdin.read()
for blk in txt.split('- .agpr_count:')[1:]:
    g = lambda k: (re.search(r'\.' + k + r':\s*(\S+)', blk) or [None, '?'])[1]
    name = g('name')
    try: name = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    except Exception: pass
    agpr = blk.split()[0]
    print(f\"v{g('vgpr_count'):>4} a{agpr:>4} s{g('sgpr_count'):>4} lds{g('group_segment_fixed_size'):>7} scratch{g('private_segment_fixed_size'):>5} spill{g('vgpr_spill_count'):>4}  {name[:110]}\")
" | grep -i "${2:-.}"
rm -rf "$T"
