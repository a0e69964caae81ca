// main.cpp -- `allwave_hip`: command-line driver with the reference's flags (src/main.rs:30-80) over
// the MI355X engine.  FASTA in (plain, or .gz/BGZF through zlib), PAF out.  SURVEY.md 8f-1.
//   -i/--input  -o/--output  -s/--scores  -x/--preset  -t/--threads  -p/--sparsification
//   --no-progress  --mash-matrix  --wfa-orientation  -k/--keep-prefixes  -e/--exclude-prefixes
// Extensions: --device N (GPU ordinal), --forward-only (skip orientation: all '+').
// -t sets the host threads used for PAF formatting / sketching (alignment itself runs on the GPU).
#include <zlib.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <stdexcept>

#include "allwave.hpp"
#include "planner.hpp"

using namespace allwave;

namespace {

struct Args {
  std::string input, output, scores = "0,5,8,2,24,1", preset, sparsification = "giant:0.99", keep, exclude;
  bool have_output = false, have_scores = false, have_preset = false, no_progress = false, mash_matrix = false;
  bool wfa_orientation = false, forward_only = false, have_keep = false, have_exclude = false;
  int threads = 1, device = 0;
  long shard_rank = 0, shard_world = 1;  // --shard R/N: this process aligns pairs R, R+N, ... (one process per GPU)
};

[[noreturn]] void die(const std::string& m, int code = 2) {
  std::cerr << "error: " << m << "\n";
  std::exit(code);
}

// parse_ani_preset (main.rs:83-124)
std::string parse_ani_preset(const std::string& preset) {
  double ani = 0;
  auto parse = [](const std::string& t, double& v) {
    size_t used = 0;
    try { v = std::stod(t, &used); } catch (...) { return false; }
    return used == t.size() && !t.empty();
  };
  if (preset.find('.') != std::string::npos) {
    double v;
    if (!parse(preset, v) || !(v > 0.0 && v <= 1.0)) die("Invalid ANI value: " + preset + ". Use 0.5-1.0 or 50%-100%");
    ani = v * 100.0;
  } else if (!preset.empty() && preset.back() == '%') {
    double v;
    if (!parse(preset.substr(0, preset.size() - 1), v) || !(v >= 50.0 && v <= 100.0)) die("Invalid ANI percentage: " + preset + ". Use 50%-100%");
    ani = v;
  } else {
    double v;
    if (!parse(preset, v) || !(v >= 50.0 && v <= 100.0)) die("Invalid ANI percentage: " + preset + ". Use 50%-100% or 50-100");
    ani = v;
  }
  if (ani >= 95.0) return "0,7,12,2,36,1";
  if (ani >= 85.0) return "0,5,8,2,24,1";
  if (ani >= 75.0) return "0,4,6,2,18,1";
  if (ani >= 65.0) return "0,3,4,1";
  return "0,1,1,1";
}

// FASTA: id = first word of the header, sequence bytes verbatim (no upper-casing), any line width
std::vector<Sequence> read_fasta(const std::string& path) {
  std::vector<Sequence> seqs;
  gzFile f = gzopen(path.c_str(), "rb");  // transparently reads plain files, gzip and BGZF
  if (!f) die("cannot open " + path, 1);
  gzbuffer(f, 1 << 20);
  std::string line;
  std::vector<char> buf(1 << 16);
  bool have = false;
  auto handle = [&](const std::string& l) {
    if (!l.empty() && l[0] == '>') {
      Sequence s;
      size_t e = l.find_first_of(" \t", 1);
      s.id = l.substr(1, e == std::string::npos ? std::string::npos : e - 1);
      seqs.push_back(std::move(s));
      have = true;
    } else if (have) {
      seqs.back().seq.insert(seqs.back().seq.end(), l.begin(), l.end());
    }
  };
  while (gzgets(f, buf.data(), (int)buf.size())) {
    const size_t n = strlen(buf.data());
    line.append(buf.data(), n);
    if (n && buf[n - 1] == '\n') {
      while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
      handle(line);
      line.clear();
    }
  }
  if (!line.empty()) handle(line);
  gzclose(f);
  return seqs;
}

std::vector<std::string> split_trim(const std::string& s) {
  std::vector<std::string> out;
  std::stringstream ss(s);
  std::string t;
  while (std::getline(ss, t, ',')) {
    size_t b = t.find_first_not_of(" \t"), e = t.find_last_not_of(" \t");
    out.push_back(b == std::string::npos ? "" : t.substr(b, e - b + 1));
  }
  return out;
}

}  // namespace

int main(int argc, char** argv) {
  Args a;
  for (int i = 1; i < argc; ++i) {
    const std::string k = argv[i];
    auto val = [&]() -> std::string {
      if (i + 1 >= argc) die("missing value for " + k);
      return argv[++i];
    };
    if (k == "-i" || k == "--input") a.input = val();
    else if (k == "-o" || k == "--output") { a.output = val(); a.have_output = true; }
    else if (k == "-s" || k == "--scores") { a.scores = val(); a.have_scores = true; }
    else if (k == "-x" || k == "--preset") { a.preset = val(); a.have_preset = true; }
    else if (k == "-t" || k == "--threads") a.threads = std::max(1, atoi(val().c_str()));
    else if (k == "-p" || k == "--sparsification") a.sparsification = val();
    else if (k == "--no-progress") a.no_progress = true;
    else if (k == "--mash-matrix") a.mash_matrix = true;
    else if (k == "--wfa-orientation") a.wfa_orientation = true;
    else if (k == "--forward-only") a.forward_only = true;
    else if (k == "-k" || k == "--keep-prefixes") { a.keep = val(); a.have_keep = true; }
    else if (k == "-e" || k == "--exclude-prefixes") { a.exclude = val(); a.have_exclude = true; }
    else if (k == "--device") a.device = atoi(val().c_str());
    else if (k == "--shard") {
      const std::string v = val();
      char* end = nullptr;
      a.shard_rank = strtol(v.c_str(), &end, 10);
      if (!end || *end != '/') die("--shard expects R/N, e.g. 3/8");
      const char* w = end + 1;
      a.shard_world = strtol(w, &end, 10);
      if (end == w || *end != 0 || a.shard_world < 1 || a.shard_rank < 0 || a.shard_rank >= a.shard_world) die("--shard expects R/N with 0 <= R < N");
    }
    else if (k == "-h" || k == "--help") {
      std::cout << "usage: allwave_hip -i in.fa [-o out.paf] [-s m,x,o,e[,o2,e2] | -x ANI] [-p none|auto|random:f|giant:p|tree:n:f:r[:k]]\n"
                   "                   [-t threads] [--wfa-orientation|--forward-only] [-k prefixes | -e prefixes] [--mash-matrix] [--device N] [--shard R/N]\n";
      return 0;
    } else die("unexpected argument: " + k);
  }
  if (a.input.empty()) die("the following required arguments were not provided: --input <INPUT>");
  if (a.have_scores && a.have_preset) die("the argument '--scores' cannot be used with '--preset'");
  if (a.have_keep && a.have_exclude) die("the argument '--keep-prefixes' cannot be used with '--exclude-prefixes'");

  SparsificationStrategy strategy;
  try { strategy = SparsificationStrategy::parse(a.sparsification); } catch (const std::exception& e) { die(e.what(), 1); }

  std::vector<Sequence> sequences = read_fasta(a.input);
  auto filter = [&](const std::string& list, bool keep) {  // main.rs:236-278
    const auto prefixes = split_trim(list);
    const size_t before = sequences.size();
    std::vector<Sequence> kept;
    for (auto& s : sequences) {
      bool any = false;
      for (const auto& p : prefixes) any = any || s.id.compare(0, p.size(), p) == 0;
      if (any == keep) kept.push_back(std::move(s));
    }
    sequences.swap(kept);
    if (sequences.size() != before)
      std::cerr << (keep ? "Kept sequences with prefixes: " : "Excluded sequences with prefixes: ") << before << " -> "
                << sequences.size() << " (prefixes: " << list << ")\n";
    if (sequences.empty()) die(keep ? "No sequences match the specified keep prefixes" : "All sequences were excluded by the specified prefixes", 1);
  };
  if (a.have_keep) filter(a.keep, true);
  if (a.have_exclude) filter(a.exclude, false);

  planner::set_host_threads(a.threads);  // -t: sketching, mash orientation, PAF formatting
  if (a.mash_matrix) {  // main.rs:281-293
    const size_t k = strategy.kind == SparsificationStrategy::TreeSampling && strategy.kmer_size ? *strategy.kmer_size : 15;
    std::cout << planner::format_distance_matrix(sequences, planner::compute_distance_matrix(sequences, k, 1000));
    return 0;
  }

  std::string scores = a.scores;
  if (a.have_preset) {
    scores = parse_ani_preset(a.preset);
    std::cerr << "Using ANI preset " << a.preset << " -> alignment scores: " << scores << "\n";
  }
  AlignmentParams params;
  try { params = parse_scores(scores); } catch (const std::exception& e) { die(e.what(), 1); }

  try {
    AllPairIterator it = AllPairIterator::with_options(sequences, params, true, !a.wfa_orientation, strategy);
    if (a.forward_only) it.with_orientation(Orientation::ForwardOnly);
    it.with_device(a.device);
    it.with_shard((size_t)a.shard_rank, (size_t)a.shard_world);
    const size_t total = it.pair_count();
    // a short-lived process with little work: taking the ring arena as it comes beats choosing the
    // fastest of four candidates (1-3 s once per engine for up to 6 % of the kernel time)
    if (total < 2000000) set_engine_flags(AWV_F_NO_ARENA_PROBE);
    std::ofstream fout;
    if (a.have_output) {
      fout.open(a.output, std::ios::binary);
      if (!fout) die("cannot create " + a.output, 1);
    }
    std::ostream& out = a.have_output ? (std::ostream&)fout : (std::ostream&)std::cout;
    const auto t0 = std::chrono::steady_clock::now();
    size_t done = 0;
    it.for_each_paf_batch([&](const std::string& chunk) {
      out.write(chunk.data(), (std::streamsize)chunk.size());
      // a short write (disk full, closed pipe) must not pass for a complete PAF: the reference propagates
      // the writer's error (main.rs:355 `writeln!(..)?`, joined at :451-453); throwing here makes the
      // engine stop with AWV_ERR_SINK before any further batch
      if (!out) throw std::runtime_error("write error on the PAF output");
      for (char c : chunk) done += c == '\n';
    }, a.threads);
    out.flush();
    if (!out) die("write error on the PAF output (flush)", 1);
    if (a.have_output) {
      fout.close();
      if (fout.fail()) die("write error on the PAF output (close)", 1);
    }
    if (done != total) die("internal: wrote " + std::to_string(done) + " of " + std::to_string(total) + " PAF lines", 1);
    if (!a.no_progress) {
      const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      char buf[160];
      snprintf(buf, sizeof(buf), "[%.1fs] %zu/%zu (100.0%%) %.1f alignments/sec", secs, done, total, done / std::max(secs, 1e-9));
      std::cerr << buf << "\n";
    }
  } catch (const std::exception& e) {
    die(e.what(), 1);
  }
  return 0;
}
