// LCM transport loop-back: the plant side publishes LOWSTATE, the controller side receives it, answers with LOWCMD, the plant side
// receives that (legged_examples/legged_mujoco/src/LeggedMujocoSim.cpp read / write; mujoco/src/lcm_interface/LcmInterface.cpp:14,
// 104-109) — over real UDP sockets on this host (include/hunter_lcm_udp.hpp).
//   lcm_udp_test host <url>                 codec only (no GPU): LOWSTATE out, decoded on arrival; LOWCMD back, decoded on arrival
//   lcm_udp_test device <url> <params.bin>  the controller side is the device path: hb_estimator_update_lcm on the received
//                                           bytes, an MPC + WBC update, hb_joint_command_lcm -> LOWCMD over the socket
#include <cmath>
#include <cstdio>
#include <cstring>
#include <hunter_hip.hpp>
#include <hunter_lcm_udp.hpp>
using namespace hunter_hip;
int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const std::string mode = argv[1], url = argv[2];
  try {
    LcmUdp plantBus(url), ctrlBus(url);   // two endpoints of the same bus
    // plant side: a LOWSTATE message of the default stance
    double state[40] = {0};
    state[0] = 1.0;                       // quaternion w x y z
    state[9] = 9.81;                      // accelerometer z
    const double qj[10] = {0.09, 0.01, 0.40, 0.93, 0.53, -0.09, -0.01, -0.40, 0.93, -0.53};
    for (int j = 0; j < 10; ++j) state[10 + j] = qj[j];
    const int64_t stamp = 123456789;
    uint8_t wire[HB_LCM_LOW_STATE_BYTES];
    if (hb_lcm_encode(HB_LCM_LOW_STATE, 1, &stamp, state, wire) != HB_OK) return 1;
    plantBus.publish("LOWSTATE", wire, HB_LCM_LOW_STATE_BYTES);
    std::string ch;
    std::vector<uint8_t> pl;
    // every endpoint of a bus sees every datagram (its own too, like LCM): pick by channel
    bool got = false;
    for (int k = 0; k < 4 && !got; ++k) got = ctrlBus.receive(ch, pl, 1000) && ch == "LOWSTATE";
    if (!got || pl.size() != HB_LCM_LOW_STATE_BYTES || std::memcmp(pl.data(), wire, pl.size()) != 0) { std::printf("RESULT error lowstate_not_received\n"); return 0; }
    std::vector<uint8_t> cmd(HB_LCM_LOW_CMD_BYTES);
    if (mode == "host") {
      int64_t ts = 0;
      double f[40];
      if (hb_lcm_decode(HB_LCM_LOW_STATE, 1, pl.data(), &ts, f) != HB_OK || ts != stamp || std::memcmp(f, state, sizeof(f)) != 0) { std::printf("RESULT error decode\n"); return 0; }
      double c[60];
      for (int i = 0; i < 60; ++i) c[i] = 0.25 * i - 3.0;
      hb_lcm_encode(HB_LCM_LOW_CMD, 1, &stamp, c, cmd.data());
    } else {
      const Parameters P = loadParametersBlob(argv[3]);
      Context ctx(P.model, P.config, 1, 24, 0);
      LcmBridge bridge(ctx, P.estimator, P.gains, nullptr);
      const std::vector<int32_t> contact(4, 1);
      const vector_t rbd = bridge.read(0.002, pl, contact);   // hb_estimator_update_lcm on the bytes that crossed the socket
      // one MPC + WBC update on the estimated state so that there is a command to send
      MpcMrtInterface mrt(ctx);
      ReferenceManager refs(ctx, P.refgen, std::vector<GaitSchedule>{GaitSchedule(ModeSchedule{P.initialEventTimes, P.initialModes},
                            ModeSequenceTemplate{P.defaultTemplate.switchingTimes, P.defaultTemplate.modes}, P.phaseTransitionStanceTime)});
      SystemObservation obs;
      obs.time = 0.0;
      obs.state = bridge.observationState();
      const vector_t t0{0.0}, cmdVel(4, 0.0);
      refs.preSolverRun(t0, 0.3, cmdVel, &obs.state);
      mrt.resetMpcNode(obs.state);
      mrt.setCurrentObservation(obs);
      mrt.advanceMpc();
      ControlOutput out;
      controllerUpdate(mrt, t0, rbd, nullptr, 0.002, out);
      cmd = bridge.write(0.002, stamp + 2000000);            // hb_joint_command_lcm
    }
    ctrlBus.publish("LOWCMD", cmd.data(), int(cmd.size()));
    got = false;
    for (int k = 0; k < 4 && !got; ++k) got = plantBus.receive(ch, pl, 1000) && ch == "LOWCMD";
    if (!got || pl.size() != HB_LCM_LOW_CMD_BYTES) { std::printf("RESULT error lowcmd_not_received\n"); return 0; }
    int64_t ts = 0;
    double c[60];
    if (hb_lcm_decode(HB_LCM_LOW_CMD, 1, pl.data(), &ts, c) != HB_OK) { std::printf("RESULT error lowcmd_decode\n"); return 0; }
    bool finite = true;
    for (double v : c) finite = finite && std::isfinite(v);
    std::printf("RESULT ok 1 stamp %lld finite %d pos0 %.6f kp0 %.3f ff3 %.4f\n", (long long)ts, finite ? 1 : 0, c[0], c[40], c[33]);
  } catch (const std::exception& e) {
    std::printf("RESULT exception 1\n");
    std::fprintf(stderr, "%s\n", e.what());
  }
  return 0;
}
