mkdir -p gpurun_out/r01m
timeout 60 ./profiles/mfma_rates > gpurun_out/r01m/mfma_rates.txt 2>&1
timeout 150 python profiles/conv1_skip_stats.py > gpurun_out/r01m/conv1_skip_stats.txt 2>&1
timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r01m/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" >> gpurun_out/r01m/pytest_gpu.txt 2>&1
timeout 200 bash profiles/run_profile.sh r01m > /dev/null 2>&1
timeout 200 python bench.py --steps 20 --warmup 3 > gpurun_out/r01m/bench.json 2> gpurun_out/r01m/bench.err
cat gpurun_out/r01m/pytest_gpu.txt; tail -c 600 gpurun_out/r01m/bench.json
