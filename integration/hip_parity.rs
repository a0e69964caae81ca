//! hip_parity.rs -- the one-command pin of the HIP engine's oracle against the REAL lib_wfa2 / WFA2-lib.
//!
//! NOT COMPILED IN THE BUILD ENVIRONMENT of the HIP engine (no cargo / rustc there, and lib_wfa2 cannot be
//! fetched offline).  A maintainer with a working allwave checkout drops this file into allwave's `tests/`
//! directory, copies `tests/golden/oracle_kats.tsv` from the HIP repo next to it (or points AWV_KATS at it) and
//! runs
//!
//!     cargo test --release --test hip_parity -- --nocapture
//!
//! What it does: every vector of the HIP build's golden file (pattern, text, scores -> penalty, CIGAR -- produced
//! by the build's CPU oracle, which the HIP kernels match byte for byte) is aligned by lib_wfa2's
//! AffineWavefronts, constructed and configured exactly as allwave does it (src/alignment.rs:263-289 for the
//! constructor by mode, :226-228 for scope / span / heuristic, :231-236 for align / score / cigar), and the score
//! and the op bytes are compared.  The first run of this file turns "parity unpinned" (HIP repo: DESIGN.md
//! section 3, oracle/biwfa_oracle.h) into a yes or a list of the vectors on which WFA2-lib's tie-breaking differs
//! from the restatement (backtrace priority, breakpoint component / scan order: SURVEY.md A.5 / A.6).
//!
//! The same comparison at the PAF level, without writing Rust: tests/golden/pin/pin.sh in the HIP repo runs the
//! allwave binary on two committed FASTA files and diffs its PAF against the committed expected lines.

use lib_wfa2::affine_wavefront::{
    AffineWavefronts, AlignmentScope, AlignmentSpan, AlignmentStatus, HeuristicStrategy, MemoryMode,
};
use std::fs;

/// allwave's mode detection (src/types.rs:107-116) and aligner construction (src/alignment.rs:263-289)
fn create_wfa_aligner(s: &[i32]) -> AffineWavefronts {
    let (m, x, o, e) = (s[0], s[1], s[2], s[3]);
    if s.len() == 6 {
        AffineWavefronts::with_penalties_affine2p_and_memory_mode(m, x, o, e, s[4], s[5], MemoryMode::Ultralow)
    } else if o == e && o == x {
        // "EditDistance" in allwave is gap-affine (x, x, x)
        AffineWavefronts::with_penalties_and_memory_mode(m, x, x, x, MemoryMode::Ultralow)
    } else {
        AffineWavefronts::with_penalties_and_memory_mode(m, x, o, e, MemoryMode::Ultralow)
    }
}

/// cigar_bytes_to_string of src/alignment.rs:347-376 (M -> '=', I <-> D swapped into standard CIGAR letters)
fn cigar_bytes_to_string(ops: &[u8]) -> String {
    let mut out = String::new();
    let mut i = 0;
    while i < ops.len() {
        let mut j = i;
        while j < ops.len() && ops[j] == ops[i] {
            j += 1;
        }
        let c = match ops[i] {
            b'M' => '=',
            b'X' => 'X',
            b'I' => 'D',
            b'D' => 'I',
            _ => '?',
        };
        out.push_str(&format!("{}{}", j - i, c));
        i = j;
    }
    out
}

#[test]
fn hip_oracle_vectors_match_wfa2() {
    let path = std::env::var("AWV_KATS").unwrap_or_else(|_| "tests/oracle_kats.tsv".to_string());
    let text = fs::read_to_string(&path).unwrap_or_else(|e| panic!("cannot read {path}: {e}"));
    let (mut n, mut bad_score, mut bad_cigar) = (0usize, Vec::new(), Vec::new());
    for line in text.lines().filter(|l| !l.starts_with('#') && !l.trim().is_empty()) {
        let f: Vec<&str> = line.split('\t').collect();
        assert_eq!(f.len(), 6, "malformed vector line: {line}");
        let scores: Vec<i32> = f[1].split(',').map(|v| v.parse().unwrap()).collect();
        let (pattern, txt) = (f[2].as_bytes(), f[3].as_bytes());
        let want_penalty: i32 = f[4].parse().unwrap();
        let mut wf = create_wfa_aligner(&scores);
        wf.set_alignment_scope(AlignmentScope::Alignment);
        wf.set_alignment_span(AlignmentSpan::End2End);
        wf.set_heuristic(&HeuristicStrategy::None);
        let status = wf.align(pattern, txt); // (pattern = query, text = target: src/alignment.rs:231)
        assert!(matches!(status, AlignmentStatus::Completed), "{}: status {:?}", f[0], status);
        n += 1;
        // WFA2 reports gap-affine scores as the negated penalty (HIP ABI: awv_result.score = -penalty)
        if wf.score() != -want_penalty {
            bad_score.push(format!("{}: score {} != {}", f[0], wf.score(), -want_penalty));
        }
        let got = cigar_bytes_to_string(wf.cigar());
        if got != f[5] {
            bad_cigar.push(format!("{} [{}]:\n   wfa2 {}\n   hip  {}", f[0], f[1], got, f[5]));
        }
    }
    println!("{n} vectors: {} score mismatches, {} CIGAR mismatches", bad_score.len(), bad_cigar.len());
    for b in bad_score.iter().chain(bad_cigar.iter()) {
        println!("{b}");
    }
    assert!(bad_score.is_empty(), "penalties differ: the HIP engine's optimum is wrong or the score sign convention differs");
    assert!(bad_cigar.is_empty(), "CIGAR tie-breaking differs from WFA2-lib on the vectors listed above");
}
