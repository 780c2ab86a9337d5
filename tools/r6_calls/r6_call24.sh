mkdir -p gpurun_out/r6
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r6/final_pytest.log 2>&1
echo "pytest rc=$? $(tail -1 gpurun_out/r6/final_pytest.log)"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
