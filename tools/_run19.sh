mkdir -p gpurun_out/r02r
export TMPDIR=/tmp
for v in a b; do
  if [ $v = a ]; then unset BPR1CS_NO_ALEN; else export BPR1CS_NO_ALEN=1; fi
  timeout 600 rocprofv3 --kernel-trace -d gpurun_out/r02r/kt$v -o out -- python bench.py --steps 4 --warmup 1 --cpu-proofs 0 > gpurun_out/r02r/kt$v.log 2>&1
  echo $v; python tools/msm_durs.py gpurun_out/r02r/kt$v
  rm -rf gpurun_out/r02r/kt$v
done
