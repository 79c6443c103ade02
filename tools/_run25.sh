mkdir -p gpurun_out/r02x
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/r02x/kt -o out -- python bench.py --steps 6 --warmup 1 --cpu-proofs 0 > gpurun_out/r02x/kt.log 2>&1
python tools/_dump_trace.py gpurun_out/r02x/kt > gpurun_out/r02x/trace.csv; rm -rf gpurun_out/r02x/kt; wc -l gpurun_out/r02x/trace.csv; tail -1 gpurun_out/r02x/kt.log | cut -c1-200
