set -x
mkdir -p gpurun_out/r02h
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_boundary.py "tests/test_gpu_benchconfig.py::test_vsmt4_depth32_bench_configuration_two_jobs_in_flight" -m gpu -x -q > gpurun_out/r02h/gputests.txt 2>&1; tail -5 gpurun_out/r02h/gputests.txt
timeout 900 python bench.py > gpurun_out/r02h/bench_default.txt 2>&1; tail -1 gpurun_out/r02h/bench_default.txt | cut -c1-250
timeout 600 python bench.py --cpu-proofs 0 --unfold 5 > gpurun_out/r02h/bench_unfold5.txt 2>&1; tail -1 gpurun_out/r02h/bench_unfold5.txt | cut -c1-250
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r02h/kt -o out -- python bench.py --steps 3 --warmup 1 --cpu-proofs 0 > gpurun_out/r02h/kt.log 2>&1
python tools/rocprof_summary.py stats gpurun_out/r02h/kt > gpurun_out/r02h/kernel_stats.txt 2>&1; head -14 gpurun_out/r02h/kernel_stats.txt
rm -rf gpurun_out/r02h/kt
