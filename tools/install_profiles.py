import sys, os, json
"""Copy the summaries of `gpurun -- bash tools/profile_round.sh <tag>` from gpurun_out/<tag>/ into profiles/<tag>_* with a header
that says which command produced them (bench.py reads table_format= / proofs_per_launch= from the PMC header)."""
tag = sys.argv[1]; src = "gpurun_out/" + tag; dst = "profiles/" + tag + "_"
commit = os.popen("git rev-parse --short HEAD").read().strip()
khash = open(os.path.join(src, "kernel_hash.txt")).read().strip() if os.path.exists(os.path.join(src, "kernel_hash.txt")) else "unknown"
cfg = json.loads(open(os.path.join(src, "bench_default.json")).read().strip())["config"]
what = "default c4, library defaults: ONE bpr1cs_prove_batch call per timed region, cut by the library into device jobs of %d proofs, %d jobs in flight, W=%d tables (%d windows; the circuit's merged S-box tables one bit narrower), unfold %d, IPA tail on the job's own stream, shared arenas" % (
    cfg["proofs_per_device_job"], cfg["jobs_in_flight"], cfg["table_window_bits"], cfg["table_windows"], cfg["ipa_unfold_rounds"])
hdr = {
 "kernel_stats.txt": "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 16 --warmup 4 --cpu-proofs 0 --configs none   (%s)\n# %s, code of commit %s: prove jobs of VSMT-4 depth-32 proofs + verify (per proof, batched, sharded through RCCL) + the two rate probes; K_build_table / K_merge_points / k_poseidon_team / K_triple_ones_point / K_range_sum_points are one-time setup\n# k_msm_fixed2: 7 launches per job = A_I(rest)+A_I(merged)+A_O(-1 form)+ones | S | 4 x (L_k + R_k) | folded generators (2 sides)\n" % (what, tag, commit),
 "pipeline_perjob.txt": "# rocprofv3 --kernel-trace -- python bench.py --steps 16 --warmup 4 --cpu-proofs 0 --configs none ; tools/trace_perjob.py (%s, same database as %s_kernel_stats.txt; %s)\n" % (tag, tag, what),
 "pipeline_timeline.txt": "# same database ; tools/trace_timeline.py (%s): front kernels of every job; heavy kernels split by whether a front kernel ran at the same time\n" % tag,
 "pmc_hbm_traffic.txt": "# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate runs, --kernel-trace only) -- python bench.py --steps 8 --warmup 4 --cpu-proofs 0 --configs none\n# code of commit " + commit + ", kernel_isa_sha256=" + khash + " (tools/kernel_isa_stats.py k_msm_fixed2 --hash of the profiled library; bench.py quotes this file only for that build)\n# %s: table_format=%d, W=%d, %d windows, proofs_per_launch=%d, 7 launches of the kernel per device job, unfold %d\n# values as reported by the counters in KB (x1024 = bytes); gfx950: FETCH_SIZE under-reports wide coalesced reads 2x, uncalibrated for 128-byte gathers\n" % (tag, cfg["table_format"], cfg["table_window_bits"], cfg["table_windows"], cfg["proofs_per_device_job"], cfg["ipa_unfold_rounds"]),
 "pmc_clock.txt": "# rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -- python bench.py --steps 8 --warmup 4 --cpu-proofs 0 --configs none (%s; kernel_isa_sha256=%s).  GRBM_GUI_ACTIVE is summed over the 8 XCDs: effective clock = value / 8 / duration\n" % (tag, khash),
 "ubench.txt": "# tools/ubench on MI355X (%s).  Cycle figures assume 2.4 GHz; the chip clocks to its power budget (see %s_pmc_clock.txt), so short kernels (first block) and sustained ones (last lines, >= 120 ms) differ.  Last two lines: the Karatsuba kill-test\n" % (tag, tag),
}
for c in ("c1", "c2", "c3", "c5", "vsmt4_d128", "vsmt2_d253"):
    hdr["%s_kernel_stats.txt" % c] = "# rocprofv3 --kernel-trace --stats -- python bench.py --config %s --warmup 4 --cpu-proofs 0 (%s, commit %s; the unprofiled line of the same command: %s_bench_%s.json)\n" % (c, tag, commit, tag, c)
for c in ("c4", "c1"):
    hdr["latency_%s_one_proof_timeline.txt" % c] = "# rocprofv3 --kernel-trace -- python tools/latency_probe.py --cases %s --batches 1 --reps 3 --no-device-program ; tools/trace_lastcall.py (%s, commit %s): every launch of the last bpr1cs_prove_batch_transcripts(batch 1, host wires) call%s, then the totals per kernel\n" % (
        c, tag, commit, " (c1: the window holds the three timed calls, divide the totals by 3)" if c == "c1" else "")
hdr["verify_c4_one_proof_timeline.txt"] = "# rocprofv3 --kernel-trace -- python tools/verify_probe.py c4 4 ; tools/trace_lastcall.py ... erify_finish (%s, commit %s): every launch of the last bpr1cs_verify_batch(batch 1) call of one depth-32 tree proof\n" % (tag, commit)
hdr["ubench_latency.txt"] = "# tools/ubench_latency on MI355X (%s): ONE wavefront alone on the chip - nanoseconds and shader cycles per dependent operation (the single-commitment / single-proof kernels)\n" % tag
hdr["host_chain_rate.txt"] = "# tools/host_chain_bench.cpp on the GPU box's host CPU (%s): the TranscriptRng chain of a depth-32 proof (csrc/host_chain.hpp) on 1 .. 64 threads at once\n" % tag
hdr["latency_probe.txt"] = "# python tools/latency_probe.py --cases c1,c4 --batches 1,8,64 --reps 3 --no-device-program (%s, commit %s): bpr1cs_gadget_prove_on / _verify_on, wall ms and stage ms per call\n" % (tag, commit)
hdr["gputests.txt"] = "# python -m pytest tests -m gpu -x -q (%s, commit %s)\n" % (tag, commit)
hdr["smoke.txt"] = ""   # (the file carries its own header)
names = {"ubench.txt": "ubench_gfx950.txt"}
for f, h in hdr.items():
    p = os.path.join(src, f)
    if not os.path.exists(p): print("missing", p); continue
    body = open(p).read()
    if f == "pmc_clock.txt":
        rows = [l.split(" | ") for l in body.splitlines() if " | " in l and not l.startswith("#")]
        clk = {r[0]: float(r[2]) / 8 / (float(r[3]) * 1e6) for r in rows}
        h += "#   " + "   ".join("%s: %.2f GHz" % (k, clk[k]) for k in ("k_msm_fixed2", "k_probe_mad", "K_ipa_vb_fold2", "K_ipa_vb_win", "K_build_table") if k in clk) + "\n"
    open(dst + names.get(f, f), "w").write(h + body)
for f in ["bench_default.json", "bench_torchrun_1rank.json", "bench_sync.json", "c_caller_standalone.json"] + ["bench_%s.json" % c for c in ("c1", "c2", "c3", "c5", "vsmt4_d128", "vsmt2_d253")]:
    p = os.path.join(src, f)
    if os.path.exists(p):
        lines = [l for l in open(p).read().strip().split("\n") if l.startswith("{")]
        if not lines: print("empty", p); continue
        line = lines[-1]
        json.loads(line)
        open(dst + f, "w").write(line + "\n")
print(sorted(x for x in os.listdir("profiles") if x.startswith(tag)))
