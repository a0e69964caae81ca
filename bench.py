#!/usr/bin/env python3
"""bench.py -- all-pairs BiWFA throughput on MI355X (BASELINE.json metric: aligned base-pairs/sec).

A "step" is one pass of the hot path (awv_align_pairs: pair list -> penalties + CIGARs in HBM) over
one batch of synthetic input: at N=1 the batch is BASELINE.json configs[1] (256 x 10 kbp, 5 %
divergence, -p none => 65,280 directed pairs, scores 0,5,8,2,24,1).  With N>1 every rank aligns a
config-2-sized shard of its own (independent pairs, no data-path collective; weak scaling); the
value is (bp aligned by all ranks in K steps) / (max over ranks of the timed region).

Sequences are resident in HBM before the timed region.  `roofline` prices the alignment kernel:
algorithmic bytes (12 row elements per cell-step for 2-piece = 24 B with 16-bit rows / 48 B with 32-bit
rows, 7 elements for 1-piece, + CIGAR bytes) over the kernel's HIP-event duration, against 8 TB/s HBM3E.  `cpu_baseline` is the CPU
restatement (oracle/, kind "port") on a bounded sample of the same pairs on this box's host cores.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2] [--pairs P]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c2", choices=["c1", "c2"])
    ap.add_argument("--pairs", type=int, default=0, help="truncate the pair list (debug; reported in config)")
    ap.add_argument("--workgroups", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline sample duration")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-paf", action="store_true", help="skip the end-to-end PAF formatting figure")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="debug: all ranks share GPU 0 and synchronise over gloo (multi-rank plumbing on a 1-GPU box)")
    args = ap.parse_args()

    import numpy as np
    import torch  # first: the engine then binds to the HIP runtime torch already loaded

    from allwave_amd import dist as D
    rank, local_rank, world = D.env()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if args.rehearse_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # RCCL (backend "nccl") carries only the timing barrier and the final reduction of counters
    dist = D.init(backend="gloo" if args.rehearse_one_gpu else "nccl") if world > 1 else None

    from allwave_amd import ffi, synth

    cfg = synth.CONFIGS[args.config]
    scores = cfg["scores"]
    # every rank owns an independent config-sized shard (weak scaling): its own seeded read set
    data, offs, _ = synth.generate(cfg["nseq"], cfg["length"], cfg["d"], cfg["seed"] + 1000 * rank)
    pairs = synth.all_pairs(cfg["nseq"])
    if args.pairs > 0:
        pairs = pairs[:args.pairs]
    eng = ffi.Engine(device=local_rank, workgroups=args.workgroups, flags=ffi.AWV_F_KEEP_ON_DEVICE)
    eng.set_sequences((data, offs))  # resident in HBM before the timed region

    def barrier():
        D.barrier(dist, local_rank)

    res = None
    for _ in range(args.warmup):
        res, _ = eng.align_pairs(scores, pairs, want_cigars=False)
    kernel_ms = 0.0
    launches = cells = ext = bp = done = cig_bytes = 0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res, _ = eng.align_pairs(scores, pairs, want_cigars=False)
        st = eng.stats()
        kernel_ms += st.kernel_ms
        launches += st.launches
        cells += st.cell_steps
        ext += st.extend_steps
        bp += st.aligned_bp
        done += st.pairs_completed
        cig_bytes += int(res["cigar_len"].sum())
    barrier()
    elapsed = time.perf_counter() - t0
    if (res["status"] != 0).any():
        raise SystemExit("bench: %d pairs did not complete" % int((res["status"] != 0).sum()))

    elapsed_max, (bp_all, done_all) = D.reduce_max_sum(dist, elapsed, [bp, done])

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    # algorithmic bytes per cell-step: every component row written once and each source row
    # read once -- 2-piece: 5 written + 7 read = 12 offsets, 1-piece: 3 + 4 = 7 -- at the row element
    # size the launch used (2 B when all lengths < 32760, else 4 B; SURVEY 8d quotes the 4-byte figure)
    esz = 2 if cfg["length"] < 32000 else 4
    bytes_per_cell = (12 if len(scores) == 6 else 7) * esz
    algo_bytes = cells * bytes_per_cell + cig_bytes  # (extend probes read the LDS-staged packed sequences)
    kern_s = kernel_ms * 1e-3
    achieved = algo_bytes / kern_s / 1e9 if kern_s > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            t = json.load(open(tpath))
            if t.get("workload") == args.config and int(t.get("pairs", 0)) == len(pairs):
                traffic = t.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    out = {
        "metric": "aligned base-pairs/sec (whole node) + PAF lines/sec, all-pairs 10 kbp",
        "value": bp_all / elapsed_max,
        "unit": "bp/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed_max / max(args.steps, 1),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int16" if cfg["length"] < 32000 else "int32",
        "data": "synthetic",
        "config": {"workload": "%s: %d x %d bp synthetic, %.0f%% divergence, -p none, scores %s, %d pairs per GPU per step"
                               % (args.config, cfg["nseq"], cfg["length"], 100 * cfg["d"],
                                  ",".join(map(str, scores)), len(pairs)),
                   "parallelism": "pairs sharded over %d GPU(s), no collective on the data path" % world},
        "paf_lines_per_s": done_all / elapsed_max,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                     "kernel": "biwfa_align_kernel", "avg_launch_ms": kernel_ms / max(launches, 1),
                     "cell_steps_per_launch": cells / max(launches, 1), "bytes_per_cell_step": bytes_per_cell,
                     "algorithmic_bytes_per_launch": algo_bytes / max(launches, 1)},
    }

    if world == 1 and not args.no_paf:
        # secondary figure (not `value`): the whole boundary end to end -- sequences on the host ->
        # upload -> align -> CIGARs over PCIe -> alignment_to_paf text (C++ host mirror) into a sink
        from allwave_amd import host as H
        nsub = cfg["nseq"]  # the whole workload: 65,280 pairs, 148 MB of PAF text for config 2
        seqs = [bytes(data[offs[i]:offs[i + 1]]) for i in range(nsub)]
        # (a first identical call creates the host library's engine and its arenas -- as many workgroups
        # and as wide rows as the measured call needs: one-time set-up, not timed)
        H.all_pairs_paf_count(["s%05d" % i for i in range(nsub)], seqs, ",".join(map(str, scores)),
                              orientation="forward", device=local_rank, format_threads=usable_cores())
        nb, nl, secs, hst = H.all_pairs_paf_count(["s%05d" % i for i in range(nsub)], seqs,
                                                  ",".join(map(str, scores)), orientation="forward",
                                                  device=local_rank, format_threads=usable_cores())
        out["paf_end_to_end"] = {"lines_per_s": nl / secs, "bp_per_s": sum(len(s) for s in seqs) * (nsub - 1) / secs,
                                 "pairs": nl, "paf_bytes": nb, "seconds": secs, "d2h_ms": hst.d2h_ms,
                                 "what": "all %d sequences all-pairs: H2D + kernel + CIGAR D2H over PCIe + PAF "
                                         "formatting on %d host threads into a memory sink (engine already created)" % (nsub, usable_cores())}

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
        g = res[:nsample]
        mism = int(((g["penalty"] != ores["penalty"]) | (g["cigar_len"] != ores["cigar_len"].astype(np.uint32)) |
                    (g["num_matches"] != ores["num_matches"]) | (g["num_mismatches"] != ores["num_mismatches"]) |
                    (g["num_ins"] != ores["num_ins_text"]) | (g["num_del"] != ores["num_del_pattern"])).sum())
        out["parity"] = {"pairs": nsample, "mismatches": mism,
                         "what": "penalty, cigar_len and M/X/I/D counts vs the CPU oracle on the sampled pairs"}
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
