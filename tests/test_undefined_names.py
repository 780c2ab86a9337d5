"""tools/undefined_names.py over the product, bench, oracle, tools and tests: no name is read that is bound nowhere (the image has no pyflakes;
round 5 deleted every CPU / torch alternative of the product -- a reference left behind in an untested branch would show here)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_undefined_names():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "undefined_names.py")], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-3000:]


def test_the_checker_sees_an_undefined_name(tmp_path):
    f = tmp_path / "bad.py"
    f.write_text("import os\ndef f(a):\n    return a + missing_one + os.sep\n")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "undefined_names.py"), str(f)], cwd=ROOT, capture_output=True, text=True, timeout=60)
    assert out.returncode == 1 and "missing_one" in out.stdout
