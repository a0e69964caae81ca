#!/bin/bash
# One-command pin of the HIP build against the reference: run a built allwave (https://github.com/pangenome/allwave, cargo
# build --release) on the committed FASTA files and diff its PAF against the lines the HIP build produces for them.
#   usage: tests/golden/pin/pin.sh /path/to/allwave            (exit 0 = byte-identical PAF on both read sets)
# The reads are all on the forward strand, so the CLI's default mash orientation reports '+' everywhere; the order of the
# reference's output is unspecified for -t > 1, hence the sort.  (Not run in the HIP build environment: no cargo there.)
set -euo pipefail
BIN=${1:?path to the allwave binary}
D=$(cd "$(dirname "$0")" && pwd)
rc=0
"$BIN" -i "$D/c1.fa" -p none -s 0,1,1,1 -t 4 --no-progress | sort > /tmp/pin_c1.paf
diff <(sort "$D/c1.expected.paf") /tmp/pin_c1.paf > /tmp/pin_c1.diff && echo "c1: identical (56 lines)" || { echo "c1: DIFFERS, see /tmp/pin_c1.diff"; rc=1; }
"$BIN" -i "$D/c2_8x10k.fa" -p none -s 0,5,8,2,24,1 -t 4 --no-progress | sort > /tmp/pin_c2.paf
diff <(sort "$D/c2_8x10k.expected.paf") /tmp/pin_c2.paf > /tmp/pin_c2.diff && echo "c2_8x10k: identical (56 lines)" || { echo "c2_8x10k: DIFFERS, see /tmp/pin_c2.diff"; rc=1; }
exit $rc
