set -u
mkdir -p gpurun_out
( time python bench.py ) > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -c 600 gpurun_out/bench_default.err
bash tools/gpu_profiles_r4.sh > gpurun_out/profiles_r4.log 2>&1
tail -12 gpurun_out/profiles_r4.log
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/tests_gpu.log 2>&1; tail -3 gpurun_out/tests_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
