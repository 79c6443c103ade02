mkdir -p gpurun_out/r02m
timeout 1200 python -m pytest tests/test_gpu_frontend.py tests/test_gpu_golden.py "tests/test_gpu_benchconfig.py::test_vsmt4_depth32_bench_configuration_two_jobs_in_flight" "tests/test_gpu_benchconfig.py::test_vsmt2_depth32_batch_1024_config_c3" -m gpu -x -q > gpurun_out/r02m/gputests.txt 2>&1; tail -3 gpurun_out/r02m/gputests.txt
for i in 1 2; do timeout 600 python bench.py --cpu-proofs 0 --steps 9 > gpurun_out/r02m/bench_$i.txt 2>&1; tail -1 gpurun_out/r02m/bench_$i.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],1), d['phase_ms_per_step'])"; done
