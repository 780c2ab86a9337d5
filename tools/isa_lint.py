#!/usr/bin/env python3
"""ISA lint: no instruction may touch the destination registers of a load that has not been waited for.

The kernels of this package hide some loads from hipcc's waitcnt bookkeeping (inline asm `ds_read_b128`, `global_load_dwordx4` whose
completion is counted by hand, chain.h) so that a counted `s_waitcnt` can leave the LDS-DMA ring in flight.  The price: the compiler
does not know the destination registers are "in flight" between the request and the wait and is free to copy them (register
coalescing around a tied asm operand, live-range splitting): a copy taken before the wait captures whatever the register held.
Round 4 found exactly that in the field kernels (a `v_mov_b64` of the carried weight fragments placed in front of the last
`s_waitcnt lgkmcnt(0)` of a product: one wave's next 32-feature tile wrong in roughly one launch of six at 33k points).

This tool disassembles the device code of the objects under build/ and checks the rule on the final ISA, for hidden and
compiler-visible loads alike, by a forward dataflow over the control-flow graph of every kernel:

  state   = {load instruction -> least number of younger operations of its counter on any path}, for the two in-order counters:
            vmcnt (global / buffer loads, stores, atomics, LDS-DMA) and lgkmcnt (LDS operations; scalar loads also count in the
            hardware but return out of order: leaving them out only makes the rule stricter)
  issue   : every entry of that counter gets one more younger operation; a load with a VGPR destination enters with 0
  wait(N) : entries with >= N younger operations are retired (s_waitcnt cnt(N) leaves at most the N youngest outstanding)
  join    : union, least count
  check   : an instruction that names a VGPR of an unretired load's destination (other than a younger load of the same counter
            writing it, which the in-order return makes safe) is a violation.

Usage: tools/isa_lint.py [objects or code objects ...]   (default: build/*.o); exit status 1 when a violation is found.
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
CAP = 64

RE_FUNC = re.compile(r"^([0-9a-f]+) <([^>]+)>:$")
RE_INS = re.compile(r"^\s+(\S+)(?:\s+(.*?))?\s*//\s*([0-9A-Fa-f]+):")
RE_VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
RE_WAIT = re.compile(r"(vmcnt|lgkmcnt|expcnt)\((\d+)\)")


def disassemble(path):
    """device disassembly of a host object with a .hip_fatbin section, or of a bare code object"""
    with tempfile.TemporaryDirectory() as d:
        co = path
        sections = subprocess.run([f"{LLVM}/llvm-readelf", "-S", path], capture_output=True, text=True).stdout
        if ".hip_fatbin" in sections:
            fat, co = os.path.join(d, "f.bin"), os.path.join(d, "f.co")
            subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", path], check=True)
            subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--targets={TARGET}", f"--input={fat}", f"--output={co}"], check=True)
        return subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], capture_output=True, text=True, check=True).stdout


def vregs(text):
    out = set()
    for m in RE_VREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


class Ins:
    __slots__ = ("addr", "op", "args", "counter", "dest", "touch", "waits", "target", "ends", "cond")

    def __init__(self, addr, op, args):
        self.addr, self.op, self.args = addr, op, args or ""
        self.counter, self.dest, self.waits, self.target, self.ends, self.cond = None, set(), {}, None, False, False
        first = self.args.split(",")[0] if self.args else ""
        if op.startswith(("global_", "buffer_", "scratch_", "flat_")):
            self.counter = "vm"
            returns = ("_load_" in op and "_lds_" not in op and " lds" not in self.args) or ("_atomic_" in op and " sc0" in self.args)
            if returns:
                self.dest = vregs(first)
        elif op.startswith("ds_"):
            self.counter = "lgkm"
            if any(k in op for k in ("read", "permute", "swizzle", "_rtn", "consume", "append")):
                self.dest = vregs(first)
        elif op == "s_waitcnt":
            self.waits = {k: int(v) for k, v in RE_WAIT.findall(self.args)}
        self.touch = vregs(self.args) if op != "s_waitcnt" else set()
        if op in ("s_endpgm", "s_branch", "s_setpc_b64"):
            self.ends = True
        if op.startswith("s_cbranch") or op == "s_branch":
            self.cond = op != "s_branch"


def parse(dis):
    funcs, cur, base = {}, None, 0
    for line in dis.splitlines():
        m = RE_FUNC.match(line)
        if m:
            cur, base = [], int(m.group(1), 16)
            funcs[m.group(2)] = cur
            continue
        if cur is None:
            continue
        m = RE_INS.match(line)
        if not m:
            continue
        ins = Ins(int(m.group(3), 16), m.group(1), m.group(2))
        if ins.op.startswith("s_cbranch") or ins.op == "s_branch":
            t = re.search(r"<[^>]*\+0x([0-9a-f]+)>", line)
            if t:
                ins.target = base + int(t.group(1), 16)
            elif re.search(r"<[^>+]*>\s*$", line):
                ins.target = base
        cur.append(ins)
    return funcs


def step(state, ins):
    """state: {(counter, addr): (younger, dest frozenset)} -> state after ins"""
    if ins.waits:
        lim = {"vm": ins.waits.get("vmcnt"), "lgkm": ins.waits.get("lgkmcnt")}
        state = {k: v for k, v in state.items() if lim[k[0]] is None or v[0] < lim[k[0]]}
    if ins.counter:
        state = {k: ((min(v[0] + 1, CAP), v[1]) if k[0] == ins.counter else v) for k, v in state.items()}
        if ins.dest:
            state[(ins.counter, ins.addr)] = (0, frozenset(ins.dest))
    return state


def join(a, b):
    if a is None:
        return dict(b)
    out = dict(a)
    for k, v in b.items():
        out[k] = (min(v[0], out[k][0]), v[1]) if k in out else v
    return out


def lint_function(name, code):
    index = {ins.addr: i for i, ins in enumerate(code)}
    n = len(code)
    succ = []
    for i, ins in enumerate(code):
        s = []
        if not ins.ends and i + 1 < n:
            s.append(i + 1)
        if ins.target is not None and ins.target in index:
            s.append(index[ins.target])
        succ.append(s)
    state_in = [None] * n
    state_in[0] = {}
    work = [0]
    while work:
        i = work.pop()
        out = step(state_in[i], code[i])
        for j in succ[i]:
            merged = join(state_in[j], out)
            if merged != state_in[j]:
                state_in[j] = merged
                work.append(j)
    problems = []
    for i, ins in enumerate(code):
        st = state_in[i]
        if not st or not ins.touch:
            continue
        for (counter, addr), (younger, dest) in st.items():
            hit = dest & ins.touch
            if not hit:
                continue
            if ins.counter == counter and ins.dest and hit <= ins.dest:
                # destination reuse by a younger load of the same counter is safe; a use as address / data is not
                others = vregs(",".join(ins.args.split(",")[1:]))
                if not (hit & others):
                    continue
            load = code[index[addr]]
            problems.append(f"{name}: {ins.addr:#x} `{ins.op} {ins.args}` touches v{sorted(hit)} of the {counter} load at {addr:#x} "
                            f"`{load.op} {load.args}` (at least {younger} younger operations; no covering s_waitcnt)")
    flat = [ins for ins in code if ins.op.startswith("flat_")]
    if flat:
        problems.append(f"{name}: {len(flat)} flat_* instructions (both counters, out of order): the in-order model does not hold")
    return problems


def lint(paths):
    problems, kernels = [], 0
    for p in paths:
        for name, code in parse(disassemble(p)).items():
            if not code:
                continue
            kernels += 1
            problems += [f"{os.path.basename(p)}: {x}" for x in lint_function(name, code)]
    return problems, kernels


def main():
    paths = sys.argv[1:] or sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "build", "*.hip.o")))
    problems, kernels = lint(paths)
    for x in problems:
        print(x)
    print(f"isa_lint: {kernels} kernels in {len(paths)} objects, {len(problems)} violations")
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
