mkdir -p gpurun_out/r02s
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_frontend.py tests/test_gpu_golden.py tests/test_gpu_parity.py tests/test_gpu_boundary.py "tests/test_gpu_benchconfig.py::test_vsmt4_depth32_bench_configuration_two_jobs_in_flight" "tests/test_gpu_benchconfig.py::test_vsmt2_depth32_batch_1024_config_c3" -m gpu -x -q > gpurun_out/r02s/gputests.txt 2>&1; tail -3 gpurun_out/r02s/gputests.txt
for i in 1 2; do timeout 600 python bench.py --cpu-proofs 0 --steps 9 > gpurun_out/r02s/bench_$i.txt 2>&1; tail -1 gpurun_out/r02s/bench_$i.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']), round(d['ms_per_step'],1), 'msm ms/step', round(r['avg_launch_ms']*r['launches_per_step'],1))"; done
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/r02s/kt -o out -- python bench.py --steps 3 --warmup 1 --cpu-proofs 0 > gpurun_out/r02s/kt.log 2>&1
python tools/msm_durs.py gpurun_out/r02s/kt; rm -rf gpurun_out/r02s/kt
