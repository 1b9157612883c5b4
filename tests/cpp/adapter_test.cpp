// Exercises include/hunter_hip.hpp (the C++ host adapter) the way the reference's control loop drives its solver:
//   adapter_test <params.bin> nogpu
//       no device: the constructor must throw hunter_hip::Error(HB_ERR_NO_GPU) — the product has no CPU path.
//   adapter_test <params.bin> run <problem.bin> <result.bin> <sqp_calls> <wbc_type>
//       problem.bin (written by tests/test_cpp_adapter.py): int32 batch, maxNodes; nNodes[B]; t[B][N+1]; mode[B][N];
//       xRef[B][N][22]; swingRef[B][N][4][6]; x0[B][22]; rbd[B][32]; tNow[B]
//       result.bin: x[B][38], optimizedState[B][22], optimizedInput[B][22], plannedMode[B], status[B] (as doubles),
//       direct[B][38] (Wbc::update on the evaluated policy), stateTrajectory, inputTrajectory
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "hunter_hip.hpp"

using namespace hunter_hip;

template <class T>
static void readv(std::FILE* f, std::vector<T>& v, size_t n) {
  v.resize(n);
  if (n && std::fread(v.data(), sizeof(T), n, f) != n) { std::fprintf(stderr, "short read\n"); std::exit(2); }
}
static void writev(std::FILE* f, const vector_t& v) { std::fwrite(v.data(), sizeof(double), v.size(), f); }

int main(int argc, char** argv) {
  if (argc < 3) return 64;
  hb_model model;
  hb_config config;
  loadPackagedParameters(argv[1], model, config);
  const std::string what = argv[2];
  if (what == "nogpu") {
    try {
      Context ctx(model, config, 1, 20);
    } catch (const Error& e) {
      std::printf("threw status %d: %s\n", e.status(), e.what());
      return e.status() == HB_ERR_NO_GPU ? 0 : 3;
    }
    std::printf("a context was created: a GPU is visible\n");
    return 4;
  }
  if (what == "lcm" && argc >= 4) {
    // adapter_test <params.bin> lcm <vectors.bin>: host codec of include/hunter_lcm.h against byte vectors produced by
    // the reference's generated message classes (written by tests/test_cpp_adapter.py from tests/golden/ref_lcm.json):
    // per message int32 type, int32 n_fields, int32 n_bytes, int64 timestamp, fields[n_fields], bytes[n_bytes].  No GPU.
    std::FILE* f = std::fopen(argv[3], "rb");
    if (!f) return 2;
    int32_t head[3];
    int n_ok = 0;
    while (std::fread(head, sizeof(int32_t), 3, f) == 3) {
      int64_t ts;
      if (std::fread(&ts, sizeof ts, 1, f) != 1) return 2;
      vector_t fields;
      std::vector<uint8_t> bytes;
      readv(f, fields, size_t(head[1]));
      readv(f, bytes, size_t(head[2]));
      if (hb_lcm_encoded_size(head[0]) != head[2] || hb_lcm_field_count(head[0]) != head[1]) return 5;
      std::vector<uint8_t> out(bytes.size());
      if (hb_lcm_encode(head[0], 1, &ts, fields.data(), out.data()) != HB_OK || out != bytes) return 6;
      int64_t ts2 = 0;
      vector_t f2(fields.size());
      if (hb_lcm_decode(head[0], 1, bytes.data(), &ts2, f2.data()) != HB_OK || ts2 != ts ||
          std::memcmp(f2.data(), fields.data(), fields.size() * sizeof(double)) != 0)
        return 7;
      bytes[2] ^= 0x10;  // foreign fingerprint
      if (hb_lcm_decode(head[0], 1, bytes.data(), &ts2, f2.data()) != HB_ERR_ARG) return 8;
      ++n_ok;
    }
    std::fclose(f);
    std::printf("ok: %d lcm messages\n", n_ok);
    return n_ok > 0 ? 0 : 9;
  }
  if (what == "gait") {
    // adapter_test <params.bin> gait <t_start> <lower> <upper> [<lower2> <upper2>]: trot template inserted at t_start,
    // then getModeSchedule windows; prints "n ev... | modes..." per window (compared with refgen.GaitSchedule)
    ModeSchedule init;
    init.modeSequence = {3};
    GaitSchedule gs(init, ModeSequenceTemplate{{0.0, 1.0}, {3}}, 0.1);
    gs.insertModeSequenceTemplate(ModeSequenceTemplate{{0.0, 0.3, 0.6}, {2, 1}}, std::atof(argv[3]), std::atof(argv[3]) + 2.0);
    for (int a = 4; a + 1 < argc; a += 2) {
      const ModeSchedule ms = gs.getModeSchedule(std::atof(argv[a]), std::atof(argv[a + 1]));
      std::printf("%zu", ms.eventTimes.size());
      for (double e : ms.eventTimes) std::printf(" %.17g", e);
      std::printf(" |");
      for (int m : ms.modeSequence) std::printf(" %d", m);
      std::printf("\n");
    }
    return 0;
  }
  if (what == "refs" && argc >= 5) {
    // adapter_test <params.bin> refs <problem.bin> <result.bin>: GaitSchedule + ReferenceManager + two SQP solves in C++
    std::FILE* f = std::fopen(argv[3], "rb");
    if (!f) return 66;
    std::vector<int32_t> head;
    readv(f, head, 2);
    const size_t B = size_t(head[0]), N = size_t(head[1]);
    vector_t x0, cmd, t0;
    readv(f, x0, B * HB_NX);
    readv(f, cmd, B * 4);
    readv(f, t0, B);
    std::vector<double> rg;  // hb_refgen_config as doubles (5 + 12 + 10) + joint_ik
    readv(f, rg, 28);
    std::fclose(f);
    hb_refgen_config rcfg;
    rcfg.dt = rg[0]; rcfg.com_height = rg[1]; rcfg.next_position_z = rg[2]; rcfg.swing_height = rg[3]; rcfg.swing_time_scale = rg[4];
    for (int i = 0; i < 12; ++i) rcfg.feet_bias[i / 3][i % 3] = rg[5 + i];
    for (int i = 0; i < 10; ++i) rcfg.default_joints[i] = rg[17 + i];
    rcfg.joint_ik = int32_t(rg[27]);
    rcfg.reserved = 0;
    Context ctx(model, config, int(B), int(N));
    std::vector<GaitSchedule> gaits;
    for (size_t i = 0; i < B; ++i) {
      ModeSchedule init;
      init.modeSequence = {3};
      GaitSchedule gs(init, ModeSequenceTemplate{{0.0, 1.0}, {3}}, 0.1);
      gs.insertModeSequenceTemplate(ModeSequenceTemplate{{0.0, 0.3, 0.6}, {2, 1}}, 0.1, 3.0);   // "trot" of gait.info
      gaits.push_back(gs);
    }
    ReferenceManager refs(ctx, rcfg, gaits);
    MpcMrtInterface mpcMrt(ctx);
    const double horizon = (double(N) - 8.0) * rcfg.dt;
    refs.preSolverRun(t0, horizon, cmd, &x0);
    mpcMrt.resetMpcNode(x0);
    mpcMrt.advanceMpc();
    refs.preSolverRun(t0, horizon, cmd, &x0);   // second call: planner state persists on the device
    mpcMrt.advanceMpc();
    vector_t xs, us;
    mpcMrt.getSolution(xs, us);
    std::FILE* g = std::fopen(argv[4], "wb");
    if (!g) return 73;
    writev(g, xs);
    writev(g, us);
    std::fclose(g);
    std::printf("ok refs\n");
    return 0;
  }
  if (what != "run" || argc < 7) return 64;
  std::FILE* f = std::fopen(argv[3], "rb");
  if (!f) return 66;
  std::vector<int32_t> head;
  readv(f, head, 2);
  const size_t B = size_t(head[0]), N = size_t(head[1]);
  ReferenceTables tables;
  vector_t x0, rbd, tNow;
  readv(f, tables.nNodes, B);
  readv(f, tables.t, B * (N + 1));
  readv(f, tables.mode, B * N);
  readv(f, tables.xRef, B * N * HB_NX);
  readv(f, tables.swingRef, B * N * HB_NC * HB_SWING_REF);
  readv(f, x0, B * HB_NX);
  readv(f, rbd, B * HB_NRBD);
  readv(f, tNow, B);
  std::fclose(f);
  const int calls = std::atoi(argv[5]);
  config.wbc_type = std::atoi(argv[6]);

  Context ctx(model, config, int(B), int(N));
  MpcMrtInterface mpcMrt(ctx);
  mpcMrt.setReferences(tables);
  mpcMrt.resetMpcNode(x0);
  std::vector<SystemObservation> obs(B);
  for (size_t i = 0; i < B; ++i) {
    obs[i].time = tNow[i];
    obs[i].state.assign(x0.begin() + i * HB_NX, x0.begin() + (i + 1) * HB_NX);
  }
  mpcMrt.setCurrentObservation(obs);
  for (int k = 0; k < calls; ++k) mpcMrt.advanceMpc();   // MPC thread body (LeggedController.cpp:396-412)
  ControlOutput out;
  controllerUpdate(mpcMrt, tNow, rbd, nullptr, 0.002, out);  // control thread (LeggedController.cpp:151-185)
  Wbc wbc(ctx);
  const vector_t direct = wbc.update(out.optimizedState, out.optimizedInput, rbd, out.plannedMode, 0.002);
  vector_t xs, us;
  mpcMrt.getSolution(xs, us);

  std::FILE* g = std::fopen(argv[4], "wb");
  if (!g) return 73;
  writev(g, out.x);
  writev(g, out.optimizedState);
  writev(g, out.optimizedInput);
  vector_t tmp(out.plannedMode.begin(), out.plannedMode.end());
  writev(g, tmp);
  tmp.assign(out.status.begin(), out.status.end());
  writev(g, tmp);
  writev(g, direct);
  writev(g, xs);
  writev(g, us);
  std::fclose(g);
  const hb_stats st = ctx.stats();
  std::printf("ok: %lld mpc solves, %lld wbc solves\n", (long long)st.n_mpc_solves, (long long)st.n_wbc_solves);
  // error behaviour of the adapter: wrong sizes are std::invalid_argument, like the reference's loaders
  try {
    mpcMrt.resetMpcNode(vector_t(3, 0.0));
    return 5;
  } catch (const std::invalid_argument&) {
  }
  return 0;
}
