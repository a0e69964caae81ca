#!/usr/bin/env python3
"""bench.py -- all-pairs BiWFA throughput on MI355X (BASELINE.json metric: aligned base-pairs/sec).

A "step" is one pass of the hot path (awv_align_pairs: pair list -> penalties + CIGARs in HBM) over
one batch of synthetic input.

  N = 1   BASELINE.json configs[1] ("c2"): 256 x 10 kbp, 5 % divergence, -p none => 65,280 directed
          pairs, scores 0,5,8,2,24,1.
  N > 1   BASELINE.json configs[2] ("c3"): the 4096 x 10 kbp read set, the SAME on every GPU, and a stated
          prefix of its -p none pair list -- by default the first N x 65,280 pairs, so that per-GPU
          work is what N = 1 does (weak scaling) -- sharded over the ranks with the cost-balanced
          partition the command-line driver uses (allwave_amd/dist.py::shard_pairs; equal-cost pairs:
          rank r aligns pairs r, r + N, ...).  No collective on the data path: RCCL (torch.distributed
          "nccl") carries the timing barrier and the final reduction of counters only.
  value = (bp aligned by all ranks in K steps) / (max over ranks of the timed region).

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself
(python -m torch.distributed.run, before anything in this process touches a GPU) and relays rank 0's
line; under torchrun, --gpus must equal WORLD_SIZE.

Sequences are resident in HBM before the timed region.  `roofline` names the unit that binds the alignment
kernel and the fraction of that unit's peak it runs at, from measured counters: `bound` "valu" (VALU issue:
SQ_ACTIVE_INST_VALU quad-cycles x 4 / (1024 SIMDs x kernel cycles at the shader clock measured in this run))
or "hbm" (measured HBM bytes / kernel time against 8 TB/s HBM3E) -- whichever fraction is larger; the
counters come from the committed rocprofv3 --pmc passes of this command (profiles/pmc_traffic*.json), carried
over per cell-step; the kernel's duration is measured live (HIP events on the engine's stream).  SURVEY 8(d)'s
algorithmic bytes (12 row elements per cell-step for 2-piece = 24 B with 16-bit rows / 48 B with 32-bit rows,
7 elements for 1-piece, + CIGAR bytes) are reported beside it (`algorithmic_GBps`,
`traffic_over_algorithmic`): since round 2 the kernel keeps rows in registers and moves fewer bytes than that.
`cpu_baseline` is the CPU restatement (oracle/, kind "port") on a bounded sample of the same pairs on this
box's host cores, at all cores and at one.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c1|c4|c5] [--pairs P]
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PAIRS_PER_GPU = 65280   # config 2's pair count: the per-GPU share of the default multi-GPU workload
N_SIMDS = 1024          # 256 CUs x 4
CLOCK_HZ_NOMINAL = 2.4e9  # only when the in-kernel clock stamps are unavailable


def usable_cores():
    """Host threads for the CPU baseline: the cgroup CPU quota if one is set, else the affinity
    mask, capped at 16 = one GPU's share of the box's host cores."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 16))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n, argv):
    """Parent of a bare `bench.py --gpus N`: starts the N ranks as children (nothing here has touched
    a GPU), relays rank 0's JSON line and exits with the launcher's code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n,
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    p = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    line = None
    for ln in p.stdout.decode(errors="replace").splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if line:
        print(line)
    raise SystemExit(p.returncode if p.returncode or line else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default=None, choices=["c1", "c2", "c3", "c4", "c5"],
                    help="default: c2 at N = 1, c3 (same read set on every GPU, prefix of its pair list) at N > 1; c4 / c5: BASELINE "
                         "configs[3] / [4] at their real sizes, pair list from the planner's sparsifier (kernel figures only)")
    ap.add_argument("--pairs", type=int, default=0,
                    help="length of the pair-list prefix (whole job, before sharding); default: all of c1/c2, N x 65,280 of c3")
    ap.add_argument("--workgroups", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU-baseline sample duration (all cores)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-paf", action="store_true", help="skip the end-to-end PAF formatting figures")
    ap.add_argument("--launch-check", action="store_true",
                    help="plumbing check without a GPU: ranks rendezvous over gloo, shard the job, reduce counters and print the line's launch fields")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="debug: all ranks share GPU 0 and synchronise over gloo (multi-rank plumbing on a 1-GPU box)")
    args = ap.parse_args()

    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        self_launch(args.gpus, sys.argv[1:])
    if world_env is not None and int(world_env) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%s (launch one rank per GPU, or run `python bench.py --gpus N` "
                         "bare and let it start the ranks)" % (args.gpus, world_env))

    import numpy as np
    import torch  # first: the engine then binds to the HIP runtime torch already loaded

    from allwave_amd import dist as D
    rank, local_rank, world = D.env()
    if args.launch_check:
        # the launcher, the rendezvous, the shard arithmetic and the reductions of the real run, minus the GPU
        from allwave_amd import synth
        dist = D.init(backend="gloo") if world > 1 else None
        cfg = synth.CONFIGS["c1"]
        _, offs, _ = synth.generate(cfg["nseq"], cfg["length"], cfg["d"], cfg["seed"])
        job = synth.all_pairs(cfg["nseq"])
        mine = D.shard_pairs(job, rank, world, lens=(offs[1:] - offs[:-1]).astype(np.int64), scores=cfg["scores"])
        D.barrier(dist, local_rank)
        tmax, (npairs, key) = D.reduce_max_sum(dist, 0.001 * (rank + 1), [len(mine), int((mine[:, 0] * 8 + mine[:, 1]).sum())])
        sizes = D.gather_floats(dist, len(mine))
        if rank == 0:
            print(json.dumps({"metric": "launch-check", "n_gpus": world, "pairs_total": int(npairs), "pairs_expected": len(job),
                              "pair_key_sum": int(key), "pair_key_expected": int((job[:, 0] * 8 + job[:, 1]).sum()),
                              "per_rank_pairs": sizes, "elapsed_max": tmax}))
        if dist is not None:
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if args.rehearse_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # RCCL (backend "nccl") carries only the timing barrier and the final reduction of counters
    dist = D.init(backend="gloo" if args.rehearse_one_gpu else "nccl", device=local_rank) if world > 1 else None

    from allwave_amd import ffi, synth

    cname = args.config or ("c2" if world == 1 else "c3")
    cfg = synth.CONFIGS[cname]
    scores = cfg["scores"]
    sparse = cfg.get("sparsify", "none") != "none"  # c4 / c5: the planner's pair list (iterator.rs:30-92), long / mixed lengths
    if sparse:
        args.no_paf = True  # (the end-to-end legs below plan `-p none`: a million 100-kbp pairs for config 4)
    if cname == "c3" or world == 1 or sparse:
        # one read set, the same on every GPU; the job is a prefix of its pair list, sharded over the ranks
        kw = {"mixed_lengths": cfg["mixed_lengths"]} if "mixed_lengths" in cfg else {}
        data, offs, ids = synth.generate(cfg["nseq"], cfg["length"], cfg["d"], cfg["seed"], **kw)
        if sparse:
            from allwave_amd import host as H
            all_pairs = np.ascontiguousarray(np.asarray(H.plan_pairs(ids, [bytes(data[offs[i]:offs[i + 1]]) for i in range(cfg["nseq"])],
                                                                     cfg["sparsify"]), dtype=np.int32).reshape(-1, 2))
        else:
            all_pairs = synth.all_pairs(cfg["nseq"])
        prefix = args.pairs if args.pairs > 0 else (min(len(all_pairs), world * PAIRS_PER_GPU) if cname == "c3" else len(all_pairs))
        job = all_pairs[:prefix]
        lens = (offs[1:] - offs[:-1]).astype(np.int64)
        pairs = D.shard_pairs(job, rank, world, lens=lens, scores=scores)
        # per-GPU work is fixed as N grows unless --pairs fixes the whole job (then the shards shrink with N)
        scaling = "strong" if (args.pairs > 0 and world > 1) else "weak"
        what = "%s: %d x %s bp synthetic, %.0f%% divergence, -p %s (%d pairs), scores %s; job = first %d pairs of that list, " \
               "same read set on every GPU, cost-balanced shards (equal costs: rank r aligns pairs r, r+N, ...): %d pairs per GPU per step" \
               % (cname, cfg["nseq"], ("%d-%d" % cfg["mixed_lengths"]) if "mixed_lengths" in cfg else str(cfg["length"]), 100 * cfg["d"],
                  cfg.get("sparsify", "none"), len(all_pairs), ",".join(map(str, scores)), len(job), len(pairs))
    else:
        # c1 / c2 on several GPUs: every rank owns an independent config-sized read set (weak scaling)
        data, offs, _ = synth.generate(cfg["nseq"], cfg["length"], cfg["d"], cfg["seed"] + 1000 * rank)
        pairs = synth.all_pairs(cfg["nseq"])
        if args.pairs > 0:
            pairs = pairs[:args.pairs]
        scaling = "weak"
        what = "%s: %d x %d bp synthetic, %.0f%% divergence, -p none, scores %s, %d pairs per GPU per step (own read set per GPU)" \
               % (cname, cfg["nseq"], cfg["length"], 100 * cfg["d"], ",".join(map(str, scores)), len(pairs))
    eng = ffi.Engine(device=local_rank, workgroups=args.workgroups, flags=ffi.AWV_F_KEEP_ON_DEVICE)
    eng.set_sequences((data, offs))  # resident in HBM before the timed region

    def barrier():
        D.barrier(dist, local_rank)

    res = None
    for _ in range(args.warmup):
        res, _ = eng.align_pairs(scores, pairs, want_cigars=False)
    kernel_ms = 0.0
    launches = cells = multi_cells = ext = bp = done = cig_bytes = restarts = clk_cycles = clk_ticks = 0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res, _ = eng.align_pairs(scores, pairs, want_cigars=False)
        st = eng.stats()
        kernel_ms += st.kernel_ms
        launches += st.launches
        cells += st.cell_steps
        multi_cells += st.multi_cell_steps
        restarts += st.restarts
        clk_cycles += st.clock_cycles
        clk_ticks += st.clock_ticks
        ext += st.extend_steps
        bp += st.aligned_bp
        done += st.pairs_completed
        cig_bytes += int(res["cigar_len"].sum())
    barrier()
    elapsed = time.perf_counter() - t0
    if (res["status"] != 0).any():
        raise SystemExit("bench: %d pairs did not complete" % int((res["status"] != 0).sum()))

    elapsed_max, (bp_all, done_all) = D.reduce_max_sum(dist, elapsed, [bp, done])
    per_rank_kernel_ms = D.gather_floats(dist, kernel_ms / max(args.steps, 1))

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    # algorithmic bytes per cell-step: every component row written once and each source row
    # read once -- 2-piece: 5 written + 7 read = 12 offsets, 1-piece: 3 + 4 = 7 -- at the row element
    # size the launch used (2 B when all lengths < 32760, else 4 B; SURVEY 8d quotes the 4-byte figure)
    # (a pair runs on 16-bit rows when its shorter sequence has fewer than 32760 bases -- for a longer partner in the
    # four-/sixteen-wave flavours, which is where such pairs go -- and on 32-bit rows otherwise; mixed sets are quoted at the
    # element size of the majority of their pairs)
    plens = (offs[1:] - offs[:-1]).astype(np.int64)
    n32 = int((np.minimum(plens[pairs[:, 0]], plens[pairs[:, 1]]) >= 32760).sum()) if len(pairs) else 0
    esz = 4 if 2 * n32 > len(pairs) else 2
    bytes_per_cell = (12 if len(scores) == 6 else 7) * esz
    algo_bytes = cells * bytes_per_cell + cig_bytes  # (extend probes read the LDS-staged packed sequences)
    kern_s = kernel_ms * 1e-3
    achieved = algo_bytes / kern_s / 1e9 if kern_s > 0 else 0.0
    avg_launch_ms = kernel_ms / max(launches, 1)
    # HBM bytes and VALU issue cycles come from the committed rocprofv3 --pmc passes of this same command
    # (profiles/pmc_traffic*.json, written by scratch/summarize_pmc.py), NOT from this run: the counters need the
    # profiler.  They are carried over per cell-step (bytes and VALU quad-cycles per cell-step are properties of the
    # kernel on a workload class: c3's reads are c2's) and multiplied by the cell-steps THIS run counted.
    clk = (clk_cycles / clk_ticks * st.clock_tick_khz * 1e3) if clk_ticks else None  # sustained shader clock of the timed launches, measured in-kernel
    clock_hz = clk or CLOCK_HZ_NOMINAL
    cells_per_launch = cells / max(launches, 1)
    traffic = valu_frac = hbm_frac_measured = valu_per_cell = None
    counters_source = "none: no committed profile of this workload class (scratch/r03_profile.sh)"
    tname = {"c2": "pmc_traffic.json", "c3": "pmc_traffic.json", "c1": None, "c4": "pmc_traffic_c4.json", "c5": "pmc_traffic_c5.json"}[cname]
    tpath = os.path.join(ROOT, "profiles", tname) if tname else None
    if tpath and os.path.exists(tpath):
        try:
            t = json.load(open(tpath))
            pcells = float(t["cell_steps_per_launch"])
            traffic = t["hbm_bytes_per_launch"] / pcells * cells_per_launch
            counters_source = "%s (rocprofv3 --pmc passes of `%s`, FETCH_SIZE x2 per profiles/r02/fetch_calibration.json; collected %s, " \
                              "kernel %.0f ms there), carried over per cell-step -- not measured in this run" \
                              % (os.path.relpath(tpath, ROOT), t.get("command_short", "bench.py --steps 1"), t.get("collected", "?"),
                                 t.get("kernel_ms_under_rocprof", 0.0))
            kcyc = avg_launch_ms * 1e-3 * clock_hz
            q = t.get("counters", {}).get("SQ_ACTIVE_INST_VALU")
            if q:
                # quad-cycles x 4 / (SIMDs x kernel cycles); one VALU wave-instruction costs a SIMD ~4 cycles for this kernel's
                # instruction mix (wall-clock microbenchmark, profiles/r03/valu_issue_rates.json; v_fma_f32 row for reference)
                valu_frac = q / pcells * cells_per_launch * 4.0 / (N_SIMDS * kcyc)
            if t.get("counters", {}).get("SQ_INSTS_VALU"):
                valu_per_cell = t["counters"]["SQ_INSTS_VALU"] / pcells
            hbm_frac_measured = traffic / (avg_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS
        except Exception as ex:  # a malformed profile must not take the bench line down
            traffic = valu_frac = hbm_frac_measured = None
            counters_source = "unreadable profile %s: %s" % (tpath, ex)
    # The bound is the unit the counters say is busiest.  With no counters the line still has to be a fraction of
    # something real: the algorithmic bytes against HBM peak, capped at what HBM can be (the cap is flagged).
    algorithmic_GBps = achieved
    if valu_frac is not None and hbm_frac_measured is not None and valu_frac >= hbm_frac_measured:
        peak_rate = N_SIMDS * clock_hz / 4.0 / 1e9  # VALU wave-instructions per ns the chip can issue
        roof = {"bound": "valu", "achieved": valu_frac * peak_rate, "peak": peak_rate, "unit": "G wave-instr/s", "frac": valu_frac}
    elif hbm_frac_measured is not None:
        roof = {"bound": "hbm", "achieved": hbm_frac_measured * HBM_PEAK_GBPS, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": hbm_frac_measured}
    else:
        roof = {"bound": "hbm", "achieved": min(achieved, HBM_PEAK_GBPS), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": min(achieved / HBM_PEAK_GBPS, 1.0), "uncounted": "no PMC profile: algorithmic bytes / time, capped at peak"}
    roof.update({"traffic": traffic, "traffic_over_algorithmic": (traffic / (algo_bytes / max(launches, 1))) if traffic else None,
                 "counters_source": counters_source, "hbm_frac_measured": hbm_frac_measured,
                 "hbm_GBps_measured": (hbm_frac_measured * HBM_PEAK_GBPS) if hbm_frac_measured is not None else None,
                 "valu_frac": valu_frac, "valu_insts_per_cell_step": valu_per_cell,
                 "shader_clock_ghz": (clk / 1e9) if clk else None, "shader_clock_source": "s_memtime / s_memrealtime stamps of every workgroup of the timed launches" if clk else "nominal",
                 "kernel": "biwfa_align_kernel", "avg_launch_ms": avg_launch_ms,
                 "cell_steps_per_launch": cells_per_launch, "bytes_per_cell_step": bytes_per_cell,
                 "algorithmic_bytes_per_launch": algo_bytes / max(launches, 1),
                 "algorithmic_GBps": algorithmic_GBps,  # SURVEY 8(d)'s figure: what a kernel keeping nothing on chip would have to move, per second (not a fraction of anything)
                 "multi_step_cell_fraction": multi_cells / max(cells, 1), "restarted_searches_per_launch": restarts / max(launches, 1)})
    out = {
        "metric": "aligned base-pairs/sec (whole node) + PAF lines/sec, all-pairs 10 kbp",
        "value": bp_all / elapsed_max,
        "unit": "bp/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed_max / max(args.steps, 1),
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": "int32" if esz == 4 else ("int16" if n32 == 0 else "int16 (%d of %d pairs on int32 rows)" % (n32, len(pairs))),
        "data": "synthetic",
        "config": {"workload": what,
                   "parallelism": "pairs sharded over %d GPU(s), no collective on the data path" % world},
        "pairs_per_s_kernel": done_all / elapsed_max,
        "pairs_per_step": len(pairs),
        "per_rank_kernel_ms": per_rank_kernel_ms,
        "kernel_ms_max_over_mean": (max(per_rank_kernel_ms) / (sum(per_rank_kernel_ms) / len(per_rank_kernel_ms))) if per_rank_kernel_ms else None,
        "roofline": roof,
    }

    if world == 1 and not args.no_paf:
        # secondary figures (never `value`): the whole boundary end to end -- sequences on the host ->
        # upload -> [orientation] -> align -> CIGARs over PCIe -> alignment_to_paf text (C++ host mirror)
        # into a counting sink, repeated --steps times.  "forward": no orientation pass (what `value`
        # times); "mash": the reference CLI's default (main.rs:313), sketches on the host threads.
        from allwave_amd import host as H
        nsub = cfg["nseq"]  # the whole workload: 65,280 pairs, 148 MB of PAF text for config 2
        seqs = [bytes(data[offs[i]:offs[i + 1]]) for i in range(nsub)]
        ids = ["s%05d" % i for i in range(nsub)]
        sc = ",".join(map(str, scores))
        thr = usable_cores()
        # (a first identical call creates the host library's engine and its arenas: one-time set-up, not timed)
        H.all_pairs_paf_count(ids, seqs, sc, orientation="forward", device=local_rank, format_threads=thr)
        for key, orient in (("paf_end_to_end", "forward"), ("paf_end_to_end_mash", "mash")):
            runs = [H.all_pairs_paf_count(ids, seqs, sc, orientation=orient, device=local_rank, format_threads=thr)
                    for _ in range(max(args.steps, 1))]
            secs = [r[2] for r in runs]
            nb, nl = runs[-1][0], runs[-1][1]
            tot_bp = sum(len(s) for s in seqs) * (nsub - 1)
            out[key] = {"lines_per_s": nl * len(runs) / sum(secs), "bp_per_s": tot_bp * len(runs) / sum(secs),
                        "pairs": nl, "paf_bytes": nb, "seconds_each": secs, "d2h_ms": runs[-1][3].d2h_ms,
                        "what": "all %d sequences all-pairs, orientation=%s: H2D + %skernel + CIGAR D2H over PCIe + PAF "
                                "formatting on %d host threads into a counting sink, mean of %d runs (engine already created)"
                                % (nsub, orient, "mash sketches + " if orient == "mash" else "", thr, len(runs))}

        # the metric's second half, by name: fully formatted PAF lines per second through the boundary, the reference CLI's
        # default orientation (main.rs:313) -- `pairs_per_s_kernel` above formats no text
        out["paf_lines_per_s"] = out["paf_end_to_end_mash"]["lines_per_s"]
        out["paf_lines_per_s_what"] = "paf_end_to_end_mash.lines_per_s (end to end incl. PCIe and host formatting; not part of `value`)"

    if world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O  # the reported CPU baseline (kind "port"), never the product
        cores = usable_cores()
        probe = pairs[:min(len(pairs), 4 * cores)]
        # baseline mode of the port: exact overlap pre-filter and known-optimum stop on (2.4x faster than the plain WFA2-order
        # search, identical results -- tests/test_oracle.py); the parity check below uses its output
        secs, _, _, _ = O.all_pairs(data, offs, probe, scores, nthreads=cores, fast_overlap=True)
        rate = len(probe) / max(secs, 1e-6)
        nsample = int(min(len(pairs), max(len(probe), rate * args.cpu_seconds)))
        sample = pairs[:nsample]
        secs, ores, ost, _ = O.all_pairs(data, offs, sample, scores, nthreads=cores, fast_overlap=True)
        sbp = int(sum(int(offs[a + 1] - offs[a]) for a, _ in sample))
        out["cpu_baseline"] = {"value": sbp / secs, "unit": "bp/s", "cores": cores, "kind": "port",
                               "sample": "first %d pairs of the same workload, %.1f s, %d threads (one aligner per thread, exact overlap pre-filter and known-optimum stop on)"
                                         % (nsample, secs, cores),
                               "cell_steps": int(ost.cell_steps)}
        # ... and on ONE host thread (T = 1 beside T = all cores), a quarter of the time budget
        n1 = int(max(4, min(nsample, rate / cores * args.cpu_seconds / 4)))
        s1 = pairs[:n1]
        secs1, _, _, _ = O.all_pairs(data, offs, s1, scores, nthreads=1, fast_overlap=True)
        out["cpu_baseline_t1"] = {"value": int(sum(int(offs[a + 1] - offs[a]) for a, _ in s1)) / secs1, "unit": "bp/s", "cores": 1,
                                  "kind": "port", "sample": "first %d pairs of the same workload, %.1f s, 1 thread" % (n1, secs1)}
        g = res[:nsample]
        mism = int(((g["penalty"] != ores["penalty"]) | (g["cigar_len"] != ores["cigar_len"].astype(np.uint32)) |
                    (g["num_matches"] != ores["num_matches"]) | (g["num_mismatches"] != ores["num_mismatches"]) |
                    (g["num_ins"] != ores["num_ins_text"]) | (g["num_del"] != ores["num_del_pattern"])).sum())
        # the roofline's unit cross-checked ("counted by the CPU oracle"): the kernel's own cell-step count on exactly
        # the sampled pairs (one more launch, after the timed region) beside the oracle's on the same pairs
        eng.align_pairs(scores, sample, want_cigars=False)
        gcells = int(eng.stats().cell_steps)
        out["parity"] = {"pairs": nsample, "mismatches": mism,
                         "what": "penalty, cigar_len and M/X/I/D counts vs the CPU oracle on the sampled pairs",
                         "cell_steps_kernel_on_sample": gcells, "cell_steps_oracle_on_sample": int(ost.cell_steps),
                         "cell_steps_kernel_over_oracle": gcells / max(int(ost.cell_steps), 1)}
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
