mkdir -p gpurun_out/r02o
export TMPDIR=/tmp
for v in a b a b; do
  if [ $v = a ]; then unset BPR1CS_AO_PLAIN; else export BPR1CS_AO_PLAIN=1; fi
  timeout 600 python bench.py --cpu-proofs 0 --steps 9 > gpurun_out/r02o/bench_$v.txt 2>&1; tail -1 gpurun_out/r02o/bench_$v.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v', round(d['value']), round(d['ms_per_step'],1), 'msm ms/step', round(r['avg_launch_ms']*r['launches_per_step'],1))"
done
unset BPR1CS_AO_PLAIN
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/r02o/kt -o out -- python bench.py --steps 3 --warmup 1 --cpu-proofs 0 > gpurun_out/r02o/kt.log 2>&1
python tools/msm_durs.py gpurun_out/r02o/kt
rm -rf gpurun_out/r02o/kt
timeout 1200 python -m pytest tests/test_gpu_frontend.py tests/test_gpu_golden.py "tests/test_gpu_benchconfig.py::test_vsmt4_depth32_bench_configuration_two_jobs_in_flight" -m gpu -x -q > gpurun_out/r02o/gputests.txt 2>&1; tail -3 gpurun_out/r02o/gputests.txt
