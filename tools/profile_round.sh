#!/bin/bash
# Everything profiles/<tag>_* is made from, in one call on the GPU box:  gpurun -- 'bash tools/profile_round.sh r04a [tests]'
# (counters in their own passes with --kernel-trace only; raw rocprofv3 databases are removed, the summaries stay under gpurun_out/<tag>/)
tag=${1:-rXX}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
B="--steps 16 --warmup 4 --cpu-proofs 0 --configs none --latency 0"
python tools/kernel_isa_stats.py k_msm_fixed2 --hash > $out/kernel_hash.txt
# the driver's command (all configurations in the one line), then the same under torchrun with one rank, then one job in flight
t0=$(date +%s); timeout 900 python bench.py --steps 20 --warmup 5 2> $out/bench_default.err | grep -a "^{" | tail -1 > $out/bench_default.json; cut -c1-200 $out/bench_default.json; echo "driver command wall seconds: $(( $(date +%s) - t0 ))" | tee $out/bench_default_wall.txt
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 16 --warmup 4 --cpu-proofs 0 --configs none --latency 0 2> $out/bench_torchrun.err | grep -a "^{" | tail -1 > $out/bench_torchrun_1rank.json
timeout 600 python bench.py --opt jobs_in_flight=1 --steps 8 --warmup 4 --cpu-proofs 0 --configs none --latency 0 2>/dev/null | grep -a "^{" | tail -1 > $out/bench_sync.json
for c in c1 c2 c3 c5 vsmt4_d128 vsmt2_d253; do
  s=16; [ $c = vsmt4_d128 ] && s=6; [ $c = vsmt2_d253 ] && s=8
  timeout 900 python bench.py --config $c --steps $s --warmup 4 2>&1 | grep -a "^{" | tail -1 > $out/bench_$c.json
  timeout 900 rocprofv3 --kernel-trace --stats -d $out/kt_$c -o out -- python bench.py --config $c --steps $s --warmup 4 --cpu-proofs 0 > $out/kt_$c.log 2>&1
  python tools/rocprof_summary.py stats $out/kt_$c > $out/${c}_kernel_stats.txt 2>&1
  rm -rf $out/kt_$c
done
timeout 600 rocprofv3 --kernel-trace --stats -d $out/kt -o out -- python bench.py $B > $out/kt.log 2>&1
python tools/rocprof_summary.py stats $out/kt > $out/kernel_stats.txt 2>&1
python tools/trace_perjob.py $out/kt 30 > $out/pipeline_perjob.txt 2>&1; head -8 $out/pipeline_perjob.txt
python tools/trace_timeline.py $out/kt > $out/pipeline_timeline.txt 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d $out/pmc_$c -o out -- python bench.py --steps 8 --warmup 4 --cpu-proofs 0 --configs none --latency 0 > $out/pmc_$c.log 2>&1
done
python tools/rocprof_summary.py pmc $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE > $out/pmc_hbm_traffic.txt 2>&1
python tools/rocprof_summary.py pmc $out/pmc_GRBM_GUI_ACTIVE > $out/pmc_clock.txt 2>&1
rm -rf $out/kt $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_GRBM_GUI_ACTIVE
# the reference's call shape - ONE proof per prove() - launch by launch (tools/trace_lastcall.py: the last call of the probe)
for c in c4 c1; do
  timeout 600 rocprofv3 --kernel-trace -d $out/kt_lat_$c -o out -- python tools/latency_probe.py --cases $c --batches 1 --reps 3 --no-device-program > $out/kt_lat_$c.log 2>&1
  gap=5; [ $c = c1 ] && gap=2
  python tools/trace_lastcall.py $out/kt_lat_$c $gap 3000 > $out/latency_${c}_one_proof_timeline.txt 2>&1
  rm -rf $out/kt_lat_$c
done
# ... and the verifier half: one proof per verify()
timeout 300 rocprofv3 --kernel-trace -d $out/kt_v -o out -- python tools/verify_probe.py c4 4 > $out/kt_verify_c4.log 2>&1
python tools/trace_lastcall.py $out/kt_v 20 400 erify_finish > $out/verify_c4_one_proof_timeline.txt 2>&1
rm -rf $out/kt_v
[ -x tools/ubench ] && timeout 300 tools/ubench > $out/ubench.txt 2>&1
[ -x tools/ubench_latency ] && timeout 120 tools/ubench_latency > $out/ubench_latency.txt 2>&1
# the plain C caller alone on the device (the library sizes its jobs from the whole device), and the host-side chain's rate on this box
timeout 900 python tools/c_caller_standalone.py 8192 20480 > $out/c_caller_standalone.json 2>$out/c_caller_standalone.err
g++ -O3 -std=c++17 -DBPR1CS_HOST_ONLY -Ibulletproofs-r1cs-gadgets_amd/csrc tools/host_chain_bench.cpp -o /tmp/hcb -pthread 2>/dev/null && /tmp/hcb > $out/host_chain_rate.txt 2>&1
timeout 600 python tools/latency_probe.py --cases c1,c4 --batches 1,8,64 --reps 3 --no-device-program > $out/latency_probe.txt 2>&1
if [ "$2" = tests ]; then timeout 1700 python -m pytest tests -m gpu -x -q > $out/gputests.txt 2>&1; tail -3 $out/gputests.txt; fi
# the driver's smoke call
( echo "# python -c 'import __graft_entry__ as g; g.smoke()' ($tag)"; timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 ) > $out/smoke.txt
