mkdir -p gpurun_out
python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "edm" > gpurun_out/r2g_pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r2g_pytest.log
