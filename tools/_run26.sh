mkdir -p gpurun_out/r02y
for v in 1 3 5 2 4 1; do
  timeout 600 python bench.py --cpu-proofs 0 --steps 16 --rng-mode $v > gpurun_out/r02y/bench_rng$v.txt 2>&1; echo rng $v; tail -1 gpurun_out/r02y/bench_rng$v.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],1))"; done
for v in 3; do
  timeout 600 python bench.py --cpu-proofs 0 --steps 16 --pipeline $v > gpurun_out/r02y/bench_pipe$v.txt 2>&1; echo pipe $v; tail -1 gpurun_out/r02y/bench_pipe$v.txt | cut -c1-300; done
