"""Worker for tests/test_dist.py: run under torch.distributed.run with the gloo backend (CPU).
Each rank takes its shard of an all-pairs list, computes per-pair penalties with the CPU oracle
(standing in for the GPU engine, which is absent on the CPU box), and the ranks reduce their
counters exactly as bench.py does."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from allwave_amd import dist as D, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    out_path = sys.argv[1]
    rank, local_rank, world = D.env()
    dist = D.init(backend="gloo")
    data, offs, _ = synth.generate(6, 300, 0.05, 5)
    pairs = synth.all_pairs(6)
    mine = D.shard_pairs(pairs, rank, world)
    D.barrier(dist)
    secs, res, st, _ = O.all_pairs(data, offs, mine, (0, 5, 8, 2, 24, 1), nthreads=1)
    D.barrier(dist)
    bp = sum(int(offs[a + 1] - offs[a]) for a, _ in mine)
    tmax, (bp_all, n_all, pen_all) = D.reduce_max_sum(dist, 0.5 + rank, [bp, len(mine), int(res["penalty"].sum())])
    # optional result gather: each rank's "PAF text" (here: one line per pair with its penalty)
    text = "".join("s%05d\ts%05d\t%d\n" % (a, b, p) for (a, b), p in zip(mine, res["penalty"])).encode()
    parts = D.gather_bytes(dist, text)
    if rank == 0:
        lines = sorted(b"".join(parts).decode().splitlines())
        json.dump(dict(world=world, tmax=tmax, bp=bp_all, n=n_all, pen=pen_all, lines=lines), open(out_path, "w"))
    else:
        assert parts is None
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
