"""Collects a round's rocprofv3 outputs (gpurun_out/*_<tag>*, written by scratch/r02_profile.sh) into
profiles/<round>/ and profiles/pmc_traffic.json (what bench.py quotes as roofline.traffic, with provenance).
usage: python scratch/summarize_pmc.py <tag> <round>"""
import csv, glob, json, os, shutil, sys, collections, datetime
tag, rnd = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go, pr = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles", rnd)
os.makedirs(pr, exist_ok=True)
bench = json.load(open(os.path.join(go, "bench_%s.json" % tag)))
json.dump(bench, open(os.path.join(pr, "bench_c2.json"), "w"), indent=1)
bprof = json.load(open(os.path.join(go, "bench_prof_%s.json" % tag)))
json.dump(bprof, open(os.path.join(pr, "bench_under_rocprof.json"), "w"), indent=1)
ks = glob.glob(os.path.join(go, "prof_%s" % tag, "*", "*kernel_stats.csv"))[0]
shutil.copy(ks, os.path.join(pr, "kernel_stats_bench_c2.csv"))
kern_ms = None
for r in csv.DictReader(open(ks)):
    if "biwfa" in r["Name"]:
        kern_ms, kname, calls = float(r["AverageNs"]) * 1e-6, r["Name"], int(r["Calls"])
ctr = collections.defaultdict(float)
for f in glob.glob(os.path.join(go, "pmc_%s_[0-9]" % tag, "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "biwfa" in r["Kernel_Name"]:
            ctr[r["Counter_Name"]] += float(r["Counter_Value"])
rl = bprof["roofline"]
cells = rl["cell_steps_per_launch"]
fetch, write = ctr["FETCH_SIZE"] * 1024, ctr["WRITE_SIZE"] * 1024
kcyc = kern_ms * 1e-3 * 2.4e9
out = {
    "workload": "c2", "pairs": 65280, "kernel": kname, "launches": calls, "kernel_ms_under_rocprof": kern_ms,
    "collected": datetime.date.today().isoformat(),
    "command_short": "bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-paf",
    "command": "rocprofv3 --kernel-trace --pmc <counters> --output-format csv -- python3 bench.py --steps 1 --warmup 0 "
               "--no-cpu-baseline --no-paf   (separate passes, scratch/r02_profile.sh)",
    "counters": dict(ctr),
    "correction": "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE*1024 is exactly half of the bytes this kernel's "
                  "8 B/lane (and 16 B/lane) row loads read, WRITE_SIZE*1024 is exact -- calibrated on known byte counts in the "
                  "kernel's own access widths and step pattern, profiles/r02/fetch_calibration.json",
    "hbm_bytes_per_launch": 2 * fetch + write,
    "hbm_read_bytes_per_launch": 2 * fetch, "hbm_write_bytes_per_launch": write,
    "algorithmic_bytes_per_launch": rl["algorithmic_bytes_per_launch"],
    "traffic_over_algorithmic": (2 * fetch + write) / rl["algorithmic_bytes_per_launch"],
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
    "valu_frac": ctr["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024 * kcyc),  # quad-cycles x 4 / (1024 SIMDs x kernel cycles at 2.4 GHz)
    "ta_addr_fifo_full_fraction_of_cu_cycles": ctr["SQ_VMEM_TA_ADDR_FIFO_FULL"] / max(ctr["SQ_BUSY_CU_CYCLES"], 1),
}
json.dump(out, open(os.path.join(pr, "pmc_bench_c2.json"), "w"), indent=1)
json.dump(out, open(os.path.join(root, "profiles", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k not in ("counters", "command", "correction")}, indent=1))
