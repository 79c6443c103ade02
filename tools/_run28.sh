mkdir -p gpurun_out/r02aa
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/r02aa/kt -o out -- python bench.py --steps 6 --warmup 1 --cpu-proofs 0 --rng-mode 5 > gpurun_out/r02aa/kt.log 2>&1
python tools/_dump_trace.py gpurun_out/r02aa/kt > gpurun_out/r02aa/trace5.csv; rm -rf gpurun_out/r02aa/kt
