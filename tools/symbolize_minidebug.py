"""python tools/symbolize_minidebug.py <stripped .so> <hex offset> [...] : function names for offsets into a stripped library from its
.gnu_debugdata section (an xz-compressed ELF carrying the symbol table; the torch wheel's libamdhip64.so / libhsa-runtime64.so have one) plus
its dynamic symbols.  Used to read the heap guard's backtraces of round 6 (profiles/r06_heap_uaf.txt)."""
import bisect, lzma, re, subprocess, sys
lib = sys.argv[1]
sec = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-S", lib], capture_output=True, text=True).stdout
m = re.search(r"\.gnu_debugdata\s+PROGBITS\s+\S+\s+(\S+)\s+(\S+)", sec)
syms = []
def take(cmd):
    for line in subprocess.run(cmd, capture_output=True, text=True).stdout.splitlines():
        p = line.split(None, 2)
        if len(p) == 3:
            try:
                syms.append((int(p[0], 16), p[2]))
            except ValueError:
                pass
if m:
    off, size = int(m.group(1), 16), int(m.group(2), 16)
    open("/tmp/minidebug.elf", "wb").write(lzma.decompress(open(lib, "rb").read()[off:off + size]))
    take(["nm", "-C", "/tmp/minidebug.elf"])
take(["nm", "-D", "-C", "--defined-only", lib])
syms.sort()
keys = [s[0] for s in syms]
for a in sys.argv[2:]:
    v = int(a, 16)
    i = bisect.bisect_right(keys, v) - 1
    print(f"{a}  {syms[i][1]}+{v - syms[i][0]:#x}" if i >= 0 else f"{a}  ?")
