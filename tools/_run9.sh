set -x
mkdir -p gpurun_out/r02i
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r02i/gputests.txt 2>&1; tail -5 gpurun_out/r02i/gputests.txt
timeout 600 python bench.py --cpu-proofs 0 --rng-mode 5 > gpurun_out/r02i/bench_rng5.txt 2>&1; tail -1 gpurun_out/r02i/bench_rng5.txt | cut -c1-250
timeout 600 python bench.py --cpu-proofs 0 > gpurun_out/r02i/bench_rng1.txt 2>&1; tail -1 gpurun_out/r02i/bench_rng1.txt | cut -c1-250
BPR1CS_FOLD_FUNCTOR=1 timeout 600 python bench.py --cpu-proofs 0 --rng-mode 5 > gpurun_out/r02i/bench_rng5_foldfunctor.txt 2>&1; tail -1 gpurun_out/r02i/bench_rng5_foldfunctor.txt | cut -c1-250
timeout 600 python bench.py --cpu-proofs 0 --rng-mode 5 --pipeline 1 > gpurun_out/r02i/bench_rng5_sync.txt 2>&1; tail -1 gpurun_out/r02i/bench_rng5_sync.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['phase_ms_per_step'])"
timeout 900 python bench.py --cpu-proofs 1 --rng-mode 5 --steps 2 > gpurun_out/r02i/bench_cpu.txt 2>&1; tail -1 gpurun_out/r02i/bench_cpu.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps(d['cpu_baseline'])[:700])"
