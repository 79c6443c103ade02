mkdir -p gpurun_out/r02v
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_boundary.py "tests/test_gpu_benchconfig.py::test_vsmt4_depth32_bench_configuration_two_jobs_in_flight" -m gpu -x -q > gpurun_out/r02v/gputests.txt 2>&1; tail -3 gpurun_out/r02v/gputests.txt
for v in new single new single; do
  if [ $v = single ]; then export BPR1CS_FOLD_SINGLE=1; else unset BPR1CS_FOLD_SINGLE; fi
  timeout 600 python bench.py --cpu-proofs 0 --steps 9 > gpurun_out/r02v/bench_$v.txt 2>&1; echo $v; tail -1 gpurun_out/r02v/bench_$v.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],1))"; done
for v in new single; do
  if [ $v = single ]; then export BPR1CS_FOLD_SINGLE=1; else unset BPR1CS_FOLD_SINGLE; fi
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r02v/kt_$v -o out -- python bench.py --steps 3 --warmup 1 --cpu-proofs 0 > gpurun_out/r02v/kt_$v.log 2>&1
  python tools/rocprof_summary.py stats gpurun_out/r02v/kt_$v 2>&1 | grep -E "vb_|k_msm_fixed2" ; rm -rf gpurun_out/r02v/kt_$v
done
