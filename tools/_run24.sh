mkdir -p gpurun_out/r02w
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/r02w/kt -o out -- python bench.py --steps 6 --warmup 1 --cpu-proofs 0 > gpurun_out/r02w/kt.log 2>&1
python tools/trace_timeline.py gpurun_out/r02w/kt 2>/dev/null > gpurun_out/r02w/timeline.txt; tail -16 gpurun_out/r02w/timeline.txt; rm -rf gpurun_out/r02w/kt
for v in 16 32 64 8 16; do
  export BPR1CS_VB_CHUNKS=$v
  timeout 600 python bench.py --cpu-proofs 0 --steps 9 > gpurun_out/r02w/bench_$v.txt 2>&1; echo VC $v; tail -1 gpurun_out/r02w/bench_$v.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],1))"; done
