mkdir -p gpurun_out/r02p
export TMPDIR=/tmp
for v in a b a; do
  if [ $v = a ]; then unset BPR1CS_AO_PLAIN; else export BPR1CS_AO_PLAIN=1; fi
  timeout 600 python bench.py --cpu-proofs 0 --steps 9 > gpurun_out/r02p/bench_$v.txt 2>&1; tail -1 gpurun_out/r02p/bench_$v.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v', round(d['value']), round(d['ms_per_step'],1), 'msm ms/step', round(r['avg_launch_ms']*r['launches_per_step'],1))"
done
unset BPR1CS_AO_PLAIN
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/r02p/kt -o out -- python bench.py --steps 3 --warmup 1 --cpu-proofs 0 > gpurun_out/r02p/kt.log 2>&1
python tools/msm_durs.py gpurun_out/r02p/kt
rm -rf gpurun_out/r02p/kt
