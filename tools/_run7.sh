set -x
mkdir -p gpurun_out/r02g
export TMPDIR=/tmp
timeout 900 python bench.py > gpurun_out/r02g/bench_default.txt 2>&1; tail -1 gpurun_out/r02g/bench_default.txt | cut -c1-600
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r02g/kt -o out -- python bench.py --steps 3 --warmup 1 --cpu-proofs 0 > gpurun_out/r02g/kt.log 2>&1
python tools/rocprof_summary.py stats gpurun_out/r02g/kt > gpurun_out/r02g/kernel_stats.txt 2>&1; head -40 gpurun_out/r02g/kernel_stats.txt
python tools/trace_timeline.py gpurun_out/r02g/kt > gpurun_out/r02g/timeline.txt 2>/dev/null; tail -20 gpurun_out/r02g/timeline.txt
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/r02g/pmc_fetch -o out -- python bench.py --steps 3 --warmup 1 --cpu-proofs 0 > gpurun_out/r02g/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/r02g/pmc_write -o out -- python bench.py --steps 3 --warmup 1 --cpu-proofs 0 > gpurun_out/r02g/pmc_write.log 2>&1
python tools/rocprof_summary.py pmc gpurun_out/r02g/pmc_fetch gpurun_out/r02g/pmc_write > gpurun_out/r02g/pmc_hbm.txt 2>&1; head -12 gpurun_out/r02g/pmc_hbm.txt
rm -rf gpurun_out/r02g/kt gpurun_out/r02g/pmc_fetch gpurun_out/r02g/pmc_write
