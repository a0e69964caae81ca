"""Collects a round's rocprofv3 outputs (gpurun_out/*_<tag>*, written by scratch/r03_profile.sh) into
profiles/<round>/ and profiles/pmc_traffic[_c4|_c5].json (what bench.py quotes as roofline.traffic / valu_frac, with provenance).
usage: python scratch/summarize_pmc.py <tag> <round> [c2|c4|c5]"""
import csv, glob, json, os, shutil, sys, collections, datetime
tag, rnd = sys.argv[1], sys.argv[2]
cname = sys.argv[3] if len(sys.argv) > 3 else "c2"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go, pr = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles", rnd)
os.makedirs(pr, exist_ok=True)
bench = json.load(open(os.path.join(go, "bench_%s.json" % tag)))
json.dump(bench, open(os.path.join(pr, "bench_%s.json" % cname), "w"), indent=1)
bprof = json.load(open(os.path.join(go, "bench_prof_%s.json" % tag)))
json.dump(bprof, open(os.path.join(pr, "bench_under_rocprof%s.json" % ("" if cname == "c2" else "_" + cname)), "w"), indent=1)
ks = glob.glob(os.path.join(go, "prof_%s" % tag, "*", "*kernel_stats.csv"))[0]
shutil.copy(ks, os.path.join(pr, "kernel_stats_bench_%s.csv" % cname))
kern_ms, kname, calls, best = 0.0, "", 0, 0.0
for r in csv.DictReader(open(ks)):
    if "biwfa" in r["Name"]:  # a call may launch several flavours (config 5: three): the time is the call's, the name the dominant one's
        tot = float(r["AverageNs"]) * 1e-6 * int(r["Calls"])
        kern_ms += tot
        calls += int(r["Calls"])
        if tot > best:
            best, kname = tot, r["Name"]
ctr = collections.defaultdict(float)
for f in glob.glob(os.path.join(go, "pmc_%s_[0-9]" % tag, "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "biwfa" in r["Kernel_Name"]:
            ctr[r["Counter_Name"]] += float(r["Counter_Value"])
rl = bprof["roofline"]
cells = rl["cell_steps_per_launch"] * max(calls, 1)  # whole call (all launches), like kern_ms and the counters
algo = rl["algorithmic_bytes_per_launch"] * max(calls, 1)
fetch, write = ctr["FETCH_SIZE"] * 1024, ctr["WRITE_SIZE"] * 1024
clock_ghz = rl.get("shader_clock_ghz") or 2.4  # the clock the profiled run itself measured (s_memtime / s_memrealtime)
kcyc = kern_ms * 1e-3 * clock_ghz * 1e9
out = {
    "workload": cname, "pairs": bprof.get("pairs_per_step"), "kernel": kname, "launches": calls, "kernel_ms_under_rocprof": kern_ms,
    "cell_steps_per_launch": cells, "shader_clock_ghz_under_rocprof": clock_ghz,
    "per": "whole call: all of its launches summed (time, cell-steps, counters)",
    "collected": datetime.date.today().isoformat(),
    "command_short": "bench.py%s --steps 1 --warmup 0 --no-cpu-baseline --no-paf" % ("" if cname == "c2" else " --config " + cname),
    "command": "rocprofv3 --kernel-trace --pmc <counters> --output-format csv -- python3 bench.py --steps 1 --warmup 0 "
               "--no-cpu-baseline --no-paf   (separate passes, scratch/r02_profile.sh)",
    "counters": dict(ctr),
    "correction": "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE*1024 is exactly half of the bytes this kernel's "
                  "8 B/lane (and 16 B/lane) row loads read, WRITE_SIZE*1024 is exact -- calibrated on known byte counts in the "
                  "kernel's own access widths and step pattern, profiles/r02/fetch_calibration.json",
    "hbm_bytes_per_launch": 2 * fetch + write,
    "hbm_read_bytes_per_launch": 2 * fetch, "hbm_write_bytes_per_launch": write,
    "algorithmic_bytes_per_launch": algo,
    "traffic_over_algorithmic": (2 * fetch + write) / algo,
    "hbm_TBps_measured": (2 * fetch + write) / (kern_ms * 1e-3) / 1e12,
    "l2_hit_rate": ctr["TCC_HIT_sum"] / max(ctr["TCC_HIT_sum"] + ctr["TCC_MISS_sum"], 1),
    "lds_bank_conflict_fraction": ctr["SQ_LDS_BANK_CONFLICT"] / max(ctr["SQ_LDS_IDX_ACTIVE"], 1),
    "valu_insts_per_cell_step": ctr["SQ_INSTS_VALU"] / cells,
    "salu_insts_per_cell_step": ctr["SQ_INSTS_SALU"] / cells,
    "lds_insts_per_cell_step": ctr["SQ_INSTS_LDS"] / cells,
    "vmem_rd_insts_per_cell_step": ctr["SQ_INSTS_VMEM_RD"] / cells,
    "vmem_wr_insts_per_cell_step": ctr["SQ_INSTS_VMEM_WR"] / cells,
    "wave_wait_fraction": ctr["SQ_WAIT_ANY"] / max(ctr["SQ_WAVE_CYCLES"], 1),
    "wave_active_fraction": ctr["SQ_ACTIVE_INST_ANY"] / max(ctr["SQ_WAVE_CYCLES"], 1),
    "valu_frac": ctr["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024 * kcyc),  # quad-cycles x 4 / (1024 SIMDs x kernel cycles at the measured clock)
    "ta_addr_fifo_full_fraction_of_cu_cycles": ctr["SQ_VMEM_TA_ADDR_FIFO_FULL"] / max(ctr["SQ_BUSY_CU_CYCLES"], 1),
}
json.dump(out, open(os.path.join(pr, "pmc_bench_%s.json" % cname), "w"), indent=1)
json.dump(out, open(os.path.join(root, "profiles", "pmc_traffic%s.json" % ("" if cname == "c2" else "_" + cname)), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k not in ("counters", "command", "correction")}, indent=1))
