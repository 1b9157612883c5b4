// hunter_hip::ShardedSolver (include/hunter_hip.hpp): one batch over G contexts — G devices, or the same device named G times — must
// return bit for bit what ONE context of the whole batch returns.
//   sharded_test <params.bin> ranges <total> <world>          prints the split (no GPU)
//   sharded_test <params.bin> run <problem.bin> <G> <sqp_calls>
//       problem.bin as written by tests/test_cpp_adapter.py (see adapter_test.cpp); all G shards on device 0.
//       prints "identical" and exits 0 when trajectories, status words, performance indices and the WBC outputs agree exactly.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "hunter_hip.hpp"

using namespace hunter_hip;

template <class T>
static void readv(std::FILE* f, std::vector<T>& v, size_t n) {
  v.resize(n);
  if (n && std::fread(v.data(), sizeof(T), n, f) != n) { std::fprintf(stderr, "short read\n"); std::exit(2); }
}
template <class T>
static bool same(const std::vector<T>& a, const std::vector<T>& b, const char* what) {
  if (a.size() == b.size() && (a.empty() || std::memcmp(a.data(), b.data(), a.size() * sizeof(T)) == 0)) return true;
  std::printf("DIFFERENT: %s\n", what);
  return false;
}

int main(int argc, char** argv) {
  if (argc < 5) return 64;
  const std::string what = argv[2];
  if (what == "ranges") {
    const int total = std::atoi(argv[3]), world = std::atoi(argv[4]);
    for (int r = 0; r < world; ++r) {
      const std::pair<int, int> p = ShardedSolver::shardRange(total, world, r);
      std::printf("%d %d\n", p.first, p.second);
    }
    return 0;
  }
  if (what != "run" || argc < 6) return 64;
  hb_model model;
  hb_config config;
  loadPackagedParameters(argv[1], model, config);
  std::FILE* f = std::fopen(argv[3], "rb");
  if (!f) return 66;
  std::vector<int32_t> head;
  readv(f, head, 2);
  const size_t B = size_t(head[0]), N = size_t(head[1]);
  ReferenceTables refs;
  readv(f, refs.nNodes, B);
  readv(f, refs.t, B * (N + 1));
  readv(f, refs.mode, B * N);
  readv(f, refs.xRef, B * N * HB_NX);
  readv(f, refs.swingRef, B * N * HB_NC * HB_SWING_REF);
  vector_t x0, rbd, tNow;
  readv(f, x0, B * HB_NX);
  readv(f, rbd, B * HB_NRBD);
  readv(f, tNow, B);
  std::fclose(f);
  const int G = std::atoi(argv[4]), calls = std::atoi(argv[5]);

  // one context of the whole batch
  Context ctx(model, config, int(B), int(N));
  MpcMrtInterface one(ctx);
  one.setReferences(refs);
  one.resetMpcNode(x0);
  for (int c = 0; c < calls; ++c) one.advanceMpc();
  vector_t xs1, us1;
  one.getSolution(xs1, us1);
  const vector_t perf1 = one.getPerformanceIndices();
  ControlOutput o1;
  controllerUpdate(one, tNow, rbd, nullptr, 0.002, o1);

  // the same batch over G shards (all on device 0)
  ShardedSolver sh(model, config, int(B), int(N), std::vector<int>(size_t(G), 0));
  sh.setReferences(refs);
  sh.resetMpcNode(x0);
  std::vector<int32_t> st;
  for (int c = 0; c < calls; ++c) st = sh.advanceMpc();
  vector_t xs2, us2;
  sh.getSolution(xs2, us2);
  const vector_t perf2 = sh.getPerformanceIndices();
  ControlOutput o2;
  sh.controllerUpdate(tNow, rbd, nullptr, 0.002, o2);

  bool ok = same(xs1, xs2, "state trajectories") & same(us1, us2, "input trajectories") & same(perf1, perf2, "performance indices") &
            same(one.mpcStatus(), st, "MPC status words") & same(o1.x, o2.x, "WBC solutions") & same(o1.optimizedState, o2.optimizedState, "optimized states") &
            same(o1.optimizedInput, o2.optimizedInput, "optimized inputs") & same(o1.plannedMode, o2.plannedMode, "planned modes") &
            same(o1.status, o2.status, "WBC status words");
  double amax = 0.0;
  for (double v : o1.x) amax = std::fmax(amax, std::fabs(v));
  if (!(amax > 1.0)) { std::printf("DIFFERENT: the WBC solution is trivial (max |x| = %g)\n", amax); ok = false; }
  std::printf("%s: %d shards of %zu instances x %zu nodes, %d SQP calls\n", ok ? "identical" : "MISMATCH", sh.shards(), B, N, calls);
  return ok ? 0 : 1;
}
