set -x
O=gpurun_out/r2c; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "mxfp8" > $O/pytest_k.log 2>&1; echo "rc=$?" >> $O/pytest_k.log; tail -25 $O/pytest_k.log
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "fp8" > $O/pytest_e.log 2>&1; echo "rc=$?" >> $O/pytest_e.log; tail -25 $O/pytest_e.log
python bench.py --steps 5 --warmup 5 --no_cpu_baseline --precision fp8 > $O/bench_fp8_headlinecfg.json 2> $O/bench_fp8.err; tail -c 1500 $O/bench_fp8_headlinecfg.json
python bench.py --steps 5 --warmup 5 --no_cpu_baseline --precision fp8 --ddpm_steps 100 --class_cond_scale 2.0 > $O/bench_fp8_config5.json 2> $O/bench_fp8c5.err; tail -c 1500 $O/bench_fp8_config5.json
python bench.py --steps 5 --warmup 5 --no_cpu_baseline --precision bf16 --ddpm_steps 100 --class_cond_scale 2.0 > $O/bench_bf16_config5.json 2> $O/bench_bf16c5.err; tail -c 600 $O/bench_bf16_config5.json
grep fp8 gpurun_out/parity_report.jsonl | tail -8
