set -x
mkdir -p gpurun_out/r02k
cd tools
for cfg in "1024 11" "2048 11" "4096 11" "1024 10" "2048 10" "1024 9"; do
  set -- $cfg
  ./msm_ubench 32768 $1 $2 2 2097152 3 > ../gpurun_out/r02k/msm_B$1_W$2.txt 2>&1
  echo "== B=$1 W=$2"; grep -E "table|merged" ../gpurun_out/r02k/msm_B$1_W$2.txt | grep -E "table|occ3" | tail -2
done
cd ..
timeout 600 python bench.py --cpu-proofs 0 --pipeline 1 > gpurun_out/r02k/bench_sync.txt 2>&1; tail -1 gpurun_out/r02k/bench_sync.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['phase_ms_per_step'])"
