mkdir -p gpurun_out/r02ab
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_boundary.py "tests/test_gpu_benchconfig.py::test_vsmt4_depth32_bench_configuration_two_jobs_in_flight" -m gpu -x -q > gpurun_out/r02ab/gputests.txt 2>&1; tail -3 gpurun_out/r02ab/gputests.txt
for i in 1 2; do timeout 600 python bench.py --cpu-proofs 0 --steps 9 > gpurun_out/r02ab/bench_$i.txt 2>&1; tail -1 gpurun_out/r02ab/bench_$i.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],1))"; done
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/r02ab/kt -o out -- python bench.py --steps 6 --warmup 1 --cpu-proofs 0 > gpurun_out/r02ab/kt.log 2>&1
python tools/_dump_trace.py gpurun_out/r02ab/kt > gpurun_out/r02ab/trace.csv; rm -rf gpurun_out/r02ab/kt
