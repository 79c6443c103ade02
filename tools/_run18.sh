set -x
mkdir -p gpurun_out/r02q
export TMPDIR=/tmp
timeout 900 python bench.py > gpurun_out/r02q/bench_default.txt 2>&1; tail -1 gpurun_out/r02q/bench_default.txt | cut -c1-250
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r02q/kt -o out -- python bench.py --steps 6 --warmup 1 --cpu-proofs 0 > gpurun_out/r02q/kt.log 2>&1
python tools/rocprof_summary.py stats gpurun_out/r02q/kt > gpurun_out/r02q/kernel_stats.txt 2>&1; head -12 gpurun_out/r02q/kernel_stats.txt
python tools/trace_timeline.py gpurun_out/r02q/kt > gpurun_out/r02q/timeline.txt 2>/dev/null
python tools/msm_durs.py gpurun_out/r02q/kt > gpurun_out/r02q/msm_durs.txt; cat gpurun_out/r02q/msm_durs.txt
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/r02q/pmc_fetch -o out -- python bench.py --steps 3 --warmup 1 --cpu-proofs 0 > gpurun_out/r02q/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/r02q/pmc_write -o out -- python bench.py --steps 3 --warmup 1 --cpu-proofs 0 > gpurun_out/r02q/pmc_write.log 2>&1
python tools/rocprof_summary.py pmc gpurun_out/r02q/pmc_fetch gpurun_out/r02q/pmc_write > gpurun_out/r02q/pmc_hbm.txt 2>&1; head -4 gpurun_out/r02q/pmc_hbm.txt
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/r02q/pmc_clk -o out -- python bench.py --steps 3 --warmup 1 --cpu-proofs 0 > gpurun_out/r02q/pmc_clk.log 2>&1
python tools/rocprof_summary.py pmc gpurun_out/r02q/pmc_clk > gpurun_out/r02q/pmc_clk.txt 2>&1; head -8 gpurun_out/r02q/pmc_clk.txt
rm -rf gpurun_out/r02q/kt gpurun_out/r02q/pmc_fetch gpurun_out/r02q/pmc_write gpurun_out/r02q/pmc_clk
timeout 1700 python -m pytest tests -m gpu -x -q > gpurun_out/r02q/gputests.txt 2>&1; tail -3 gpurun_out/r02q/gputests.txt
