"""python tools/pytest_lab.py <pytest args...> : pytest in this process with the NSKY_* lab switches of tools/lab.py applied first"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
import lab
lab.apply()
import pytest
sys.exit(pytest.main(sys.argv[1:]))
