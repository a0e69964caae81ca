#!/bin/bash
# All GPUs of one node, one process per GPU, no exchange step: every process plans the
# same pair list and aligns its cost-balanced shard (--shard R/N) on its own device; the PAF shards are
# concatenated at the end (line order is not significant: the reference's own order is nondeterministic
# for more than one thread, src/iterator.rs:222-233).
#   usage: allwave_hip_node.sh <ngpus> <out.paf> -i in.fa [any other allwave_hip option except -o/--device/--shard]
# (-t is per process: give each of the N processes its share of the host's cores, e.g. -t $(( $(nproc) / N )).)
# A shard that fails -- including on a PAF write error -- fails the run; nothing is concatenated then.
set -u
here="$(cd "$(dirname "$0")" && pwd)"
n=${1:?ngpus}; out=${2:?out.paf}; shift 2
pids=()
for ((r = 0; r < n; ++r)); do
  "$here/allwave_hip" "$@" --device "$r" --shard "$r/$n" -o "$out.shard$r" &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait "$p" || rc=$?; done
[ $rc -ne 0 ] && { echo "allwave_hip_node: a shard failed (exit $rc)" >&2; exit $rc; }
: > "$out"
for ((r = 0; r < n; ++r)); do cat "$out.shard$r" >> "$out" && rm -f "$out.shard$r"; done
