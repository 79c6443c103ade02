mkdir -p gpurun_out/r02t
for p in 0 2 3 4 1; do echo "### pattern $p"; timeout 300 tools/msm_ubench 32768 1024 11 2 2097152 3 $p 2>&1 | grep -E "occ3|functor \(no|DIFFER"; done > gpurun_out/r02t/patterns.txt 2>&1
cat gpurun_out/r02t/patterns.txt
