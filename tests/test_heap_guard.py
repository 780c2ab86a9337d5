"""tools/heap_guard.c (the LD_PRELOAD allocator of the flake hunt): its self-test -- six deliberate heap errors reported, a clean
program and python + torch quiet.  Host code only."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc builds the guard")
def test_heap_guard_selftest():
    out = subprocess.run(["bash", os.path.join(ROOT, "tools", "heap_guard_selftest.sh")], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    for mode in range(6):
        assert f"mode {mode}: ok" in out.stdout, out.stdout
    assert "python + torch: quiet" in out.stdout
