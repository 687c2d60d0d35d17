# the round's closing run: the whole GPU suite, then the default bench (its line and details copied for profiles/)
timeout 5400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r06_final_pytest_gpu.txt
cat gpurun_out/r06_final_pytest_gpu.txt
python bench.py > gpurun_out/r06_final_bench_line.json 2> gpurun_out/r06_final_bench.err
tail -c 3500 gpurun_out/r06_final_bench_line.json
cp bench_details.json gpurun_out/r06_final_bench_details.json
