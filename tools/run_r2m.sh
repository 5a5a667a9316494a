mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r2m_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2m_pytest.log
python bench.py > gpurun_out/r2m_bench_default.json 2> gpurun_out/r2m_bench_default.err; tail -c 2500 gpurun_out/r2m_bench_default.json
