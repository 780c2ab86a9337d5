"""tools/isa_lint.py: the waitcnt rule on the final ISA of every kernel (no instruction touches the destination of a load that has
not been waited for), and the lint's own check on the pattern it was written for.

CPU test: hipcc cross-compiles, llvm-objdump disassembles; nothing runs on a GPU."""
import glob
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_lint  # noqa: E402


def _fake(body):
    lines = ["0000000000001000 <k>:"]
    for i, text in enumerate(body):
        lines.append(f"\t{text}    // {0x1000 + 4 * i:012X}: 00000000")
    return "\n".join(lines) + "\n"


def test_lint_flags_a_copy_in_front_of_the_wait():
    """the round-3 build of the field kernels, reduced: the carried fragments copied before the last lgkmcnt(0)"""
    bad = _fake(["ds_read_b128 v[168:171], v211", "ds_read_b128 v[172:175], v211 offset:1024", "v_mfma_f32_32x32x16_f16 v[0:15], v[16:19], v[148:151], v[0:15]",
                 "v_mov_b64_e32 v[16:17], v[168:169]", "s_waitcnt lgkmcnt(0)", "s_endpgm"])
    problems = isa_lint.lint_function("k", isa_lint.parse(bad)["k"])
    assert len(problems) == 1 and "v[168, 169]" in problems[0], problems
    good = _fake(["ds_read_b128 v[168:171], v211", "ds_read_b128 v[172:175], v211 offset:1024", "s_waitcnt lgkmcnt(0)", "v_mov_b64_e32 v[16:17], v[168:169]", "s_endpgm"])
    assert isa_lint.lint_function("k", isa_lint.parse(good)["k"]) == []


def test_lint_counts_in_order_and_joins_paths():
    # lgkmcnt(2) retires everything but the two youngest LDS operations
    code = _fake(["ds_read_b128 v[0:3], v9", "ds_read_b128 v[4:7], v9", "ds_read_b128 v[10:13], v9", "s_waitcnt lgkmcnt(2)", "v_mov_b32_e32 v20, v0",
                  "v_mov_b32_e32 v21, v4", "s_endpgm"])
    problems = isa_lint.lint_function("k", isa_lint.parse(code)["k"])
    assert len(problems) == 1 and "v[4]" in problems[0], problems
    # a hidden global load is retired by vmcnt(N) only when N operations were issued after it on EVERY path into the wait
    code = _fake(["global_load_dwordx4 v[0:3], v[8:9], off", "s_cbranch_scc1 2", "global_load_lds_dwordx4 v[8:9], off", "global_load_lds_dwordx4 v[8:9], off",
                  "s_waitcnt vmcnt(2)", "v_mov_b32_e32 v20, v0", "s_endpgm"])
    funcs = isa_lint.parse(code)
    funcs["k"][1].target = funcs["k"][3].addr  # the branch skips one of the two younger operations
    problems = isa_lint.lint_function("k", funcs["k"])
    assert len(problems) == 1 and "at least 1 younger" in problems[0], problems
    funcs["k"][1].target = None
    assert isa_lint.lint_function("k", funcs["k"]) == []
    # stores count on vmcnt; a younger load of the same counter may reuse a destination, a younger VALU write may not
    code = _fake(["global_load_dwordx4 v[0:3], v[8:9], off", "global_load_dwordx4 v[0:3], v[10:11], off", "s_waitcnt vmcnt(0)", "v_mov_b32_e32 v20, v0", "s_endpgm"])
    assert isa_lint.lint_function("k", isa_lint.parse(code)["k"]) == []
    code = _fake(["global_load_dwordx4 v[0:3], v[8:9], off", "v_mov_b32_e32 v0, v20", "s_waitcnt vmcnt(0)", "s_endpgm"])
    assert len(isa_lint.lint_function("k", isa_lint.parse(code)["k"])) == 1


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_every_kernel_of_the_library_passes():
    objs = sorted(glob.glob(os.path.join(ROOT, "build", "*.hip.o")))
    srcs = sorted(glob.glob(os.path.join(ROOT, "neusky_amd", "csrc", "*.hip")))
    stale = len(objs) != len(srcs) or any(os.path.getmtime(o) < os.path.getmtime(s) for o, s in zip(objs, srcs))
    if stale:
        subprocess.run([os.path.join(ROOT, "build.sh")], check=True, capture_output=True)
        objs = sorted(glob.glob(os.path.join(ROOT, "build", "*.hip.o")))
    problems, kernels = isa_lint.lint(objs)
    assert kernels > 50, kernels
    assert problems == [], "\n".join(problems)
