"""Summary of scratch/r02_pmc_c45.sh's passes: python scratch/pmc_c45_sum.py <tag> [dir]"""
import csv, glob, collections, json, sys, os
T = sys.argv[1]
O = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
ctr = collections.defaultdict(float); dur = collections.defaultdict(float)
for i in range(1, 5):
    for f in glob.glob("%s/pmc_%s_%d/*/*counter_collection.csv" % (O, T, i)):
        for r in csv.DictReader(open(f)):
            if "biwfa" in r["Kernel_Name"]: ctr[r["Counter_Name"]] += float(r["Counter_Value"])
    for f in glob.glob("%s/pmc_%s_%d/*/*kernel_trace.csv" % (O, T, i)):
        for r in csv.DictReader(open(f)):
            if "biwfa" in r["Kernel_Name"]: dur[i] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
# per dispatch: duration (pass 4), VALU issue fraction, resident waves, HBM rate
disp = collections.OrderedDict()
for f in glob.glob("%s/pmc_%s_4/*/*kernel_trace.csv" % (O, T)):
    for r in csv.DictReader(open(f)):
        if "biwfa" in r["Kernel_Name"]:
            disp[r["Dispatch_Id"]] = {"kernel": r["Kernel_Name"][:90], "wg": r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")), "grid": r.get("Grid_Size_X", r.get("Grid_Size", "")),
                                      "s": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9}
for i in (1, 2, 3, 4):
    order = []
    for f in glob.glob("%s/pmc_%s_%d/*/*counter_collection.csv" % (O, T, i)):
        for r in csv.DictReader(open(f)):
            if "biwfa" in r["Kernel_Name"]:
                if r["Dispatch_Id"] not in order: order.append(r["Dispatch_Id"])
                key = list(disp.keys())[order.index(r["Dispatch_Id"])] if order.index(r["Dispatch_Id"]) < len(disp) else None
                if key: disp[key][r["Counter_Name"]] = disp[key].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k, d in disp.items():
    cyc = d["s"] * 2.4e9
    d["valu_frac"] = d.get("SQ_ACTIVE_INST_VALU", 0) * 4.0 / (1024 * cyc)
    d["busy_cu_frac"] = d.get("SQ_BUSY_CU_CYCLES", 0) / (256 * cyc)
    d["hbm_TBps"] = (2 * d.get("FETCH_SIZE", 0) + d.get("WRITE_SIZE", 0)) * 1024 / d["s"] / 1e12
    d["waves_per_cu"] = 4 * d.get("SQ_WAVE_CYCLES", 0) / cyc / 256
    print({kk: (round(v, 4) if isinstance(v, float) and v < 1e6 else v) for kk, v in d.items() if kk in ("kernel", "wg", "grid", "s", "valu_frac", "busy_cu_frac", "hbm_TBps", "waves_per_cu", "SQ_WAVES")})
logs = [f for f in ("%s/pmc_%s_%d.log" % (O, T, i) for i in (3, 4, 1, 2)) if os.path.exists(f)]
line = [json.loads(l) for l in open(logs[0]) if l.startswith("{")][0]
cells = line["cell_steps"]; secs = dur[3] or dur[4] or dur[1]
hbm = (2 * ctr["FETCH_SIZE"] + ctr["WRITE_SIZE"]) * 1024
out = {"what": "%s %s" % (line["config"], line["select"]), "pairs": line["pairs"], "kernel_s_by_pass": dict(dur), "kernel_ms_plain": line["kernel_ms"], "cell_steps": cells,
       "cell_steps_per_s": cells / secs, "multi_frac": line["multi_frac"],
       "hbm_bytes": hbm, "hbm_TBps": hbm / dur[1] / 1e12 if dur[1] else None, "hbm_bytes_per_cell_step": hbm / cells,
       "l2_hit": ctr["TCC_HIT_sum"] / max(1, ctr["TCC_HIT_sum"] + ctr["TCC_MISS_sum"]),
       "valu_per_cell_step": ctr["SQ_INSTS_VALU"] / cells, "salu_per_cell_step": ctr["SQ_INSTS_SALU"] / cells, "lds_per_cell_step": ctr["SQ_INSTS_LDS"] / cells,
       "vmem_rd_per_cell_step": ctr["SQ_INSTS_VMEM_RD"] / cells, "vmem_wr_per_cell_step": ctr["SQ_INSTS_VMEM_WR"] / cells,
       "wave_wait_fraction": ctr["SQ_WAIT_ANY"] / max(1, ctr["SQ_WAVE_CYCLES"]), "wave_active_fraction": ctr["SQ_ACTIVE_INST_ANY"] / max(1, ctr["SQ_WAVE_CYCLES"]),
       "valu_frac": ctr["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024 * dur[4] * 2.4e9) if dur[4] else None,
       "counters": dict(ctr)}
json.dump(out, open("%s/pmc_%s.json" % (O, T), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "counters"}, indent=1))
