import sys, os, json
tag = sys.argv[1]; src = "gpurun_out/" + tag; dst = "profiles/" + tag + "_"
commit = os.popen("git rev-parse --short HEAD").read().strip()
hdr = {
 "kernel_stats.txt": "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 6 --warmup 1 --cpu-proofs 0   (default: 2 batches in flight, W=11 limb-form tables in 128-byte slots, unfold 4, rng mode 1)\n# round 2 final (%s, code of commit %s): 8 prove jobs x 1024 VSMT-4 depth-32 proofs + verify (per proof, batched, split) + the two rate probes; K_build_table / K_merge_points / k_poseidon_team / K_triple_ones_point / K_range_sum_points are one-time setup\n# k_msm_fixed2: 7 launches per job = A_I(rest)+A_I(merged)+A_O(-1 form)+ones | S | 4 x (L_k + R_k) | folded generators (2 sides)\n" % (tag, commit),
 "pipeline_perjob.txt": "# rocprofv3 --kernel-trace -- python bench.py --steps 6 --warmup 1 --cpu-proofs 0 ; tools/trace_perjob.py (%s, same database as %s_kernel_stats.txt)\n" % (tag, tag),
 "pipeline_timeline.txt": "# same database ; tools/trace_timeline.py (%s): front kernels of every job; heavy kernels split by whether a front kernel ran at the same time\n" % tag,
 "pmc_hbm_traffic.txt": "# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate runs, --kernel-trace only) -- python bench.py --steps 3 --warmup 1 --cpu-proofs 0\n# round 2 final (%s): table_format=1 (limb form, 128-byte slots), W=11, 23 windows, launches_per_step=6 counted by the library's statistics (+1 launch for the folded generators), 1024 proofs, unfold 4\n# values as reported by the counters in KB (x1024 = bytes); gfx950: FETCH_SIZE under-reports wide coalesced reads 2x, uncalibrated for 128-byte gathers\n" % tag,
 "pmc_clock.txt": "# rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -- python bench.py --steps 3 --warmup 1 --cpu-proofs 0 (%s).  GRBM_GUI_ACTIVE is summed over the 8 XCDs: effective clock = value / 8 / duration\n" % tag,
 "ubench.txt": "# tools/ubench on MI355X (%s).  Cycle figures assume 2.4 GHz; the chip clocks to its power budget (see %s_pmc_clock.txt), so short kernels (first block) and sustained ones (last lines, >= 120 ms) differ\n" % (tag, tag),
 "msm_ubench.txt": "# tools/msm_ubench 32768 1024 11 3 2097152 3  (%s: limb-128 tables, one IPA-round-shaped pair of 32768-term MSMs for 1024 proofs)\n" % tag,
}
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
for f in ("bench_default.json", "bench_torchrun_1rank.json", "bench_sync.json"):
    p = os.path.join(src, f)
    if os.path.exists(p):
        line = open(p).read().strip()
        json.loads(line)
        open(dst + f, "w").write(line + "\n")
print(sorted(x for x in os.listdir("profiles") if x.startswith(tag)))
