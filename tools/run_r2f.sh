# round 2, call f: fp8_mixed mode - full GPU suite, configs[4] bench in the three precisions (one box: A/B comparable)
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r2r_pytest.log 2>&1; echo "pytest rc=$?" ; tail -3 gpurun_out/r2r_pytest.log
for p in bf16 fp8 fp8_mixed; do
  python bench.py --precision $p --ddpm_steps 100 --class_cond_scale 2.0 --steps 2 --warmup 1 > gpurun_out/r2r_bench_config5_$p.json 2> gpurun_out/r2r_bench_config5_$p.err
  tail -c 1500 gpurun_out/r2r_bench_config5_$p.json
done
