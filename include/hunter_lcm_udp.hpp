// LCM transport for the low-level messages, without liblcm: the UDP provider of LCM restricted to short messages ("LC02"
// datagrams, hb_lcm_frame / hb_lcm_unframe).  The reference's MuJoCo bridge and hardware bridge construct `lcm::LCM lcm_`
// with the default provider (mujoco/include/lcm_interface/LcmInterface.h:18, legged_examples/legged_mujoco/include/mujoco_lcm/
// MujocoLcm.h:13), i.e. "udpm://239.255.76.67:7667?ttl=0": multicast group 239.255.76.67, port 7667, datagrams kept on the host.
// low_cmd_t / low_state_t (496 / 336 bytes) are far below the fragmentation limit, so every message is one datagram.
//   LcmUdp bus;                               // default provider: joins the group, sends to it (every endpoint sees every datagram)
//   bus.publish("LOWCMD", bytes, n);  bus.receive(channel, payload, timeout_ms);
// Header-only, POSIX sockets.  Used with hunter_hip::LcmBridge (hunter_hip.hpp): receive LOWSTATE -> read(), write() -> publish LOWCMD.
#pragma once
#include <arpa/inet.h>
#include <netinet/in.h>
#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>

#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "hunter_lcm.h"

namespace hunter_hip {

class LcmUdp {
 public:
  explicit LcmUdp(const std::string& url = "udpm://239.255.76.67:7667?ttl=0") {
    const size_t p0 = url.find("://");
    if (p0 == std::string::npos) throw std::invalid_argument("[hunter_hip] LCM url: " + url);
    if (url.compare(0, p0, "udpm") != 0) throw std::invalid_argument("[hunter_hip] LCM url: only the udpm:// provider is implemented: " + url);
    std::string rest = url.substr(p0 + 3);
    int ttl = 0;
    const size_t q = rest.find('?');
    if (q != std::string::npos) {
      const size_t t = rest.find("ttl=", q);
      if (t != std::string::npos) ttl = std::atoi(rest.c_str() + t + 4);
      rest.erase(q);
    }
    const size_t c = rest.find(':');
    const std::string host = c == std::string::npos ? rest : rest.substr(0, c);
    const int port = c == std::string::npos ? 7667 : std::atoi(rest.c_str() + c + 1);
    if (port < 1 || port > 65535) throw std::invalid_argument("[hunter_hip] LCM url port out of range: " + url);
    if (ttl < 0 || ttl > 255) throw std::invalid_argument("[hunter_hip] LCM url ttl out of range: " + url);
    std::memset(&dest_, 0, sizeof(dest_));
    dest_.sin_family = AF_INET;
    dest_.sin_port = htons(uint16_t(port));
    if (inet_pton(AF_INET, host.c_str(), &dest_.sin_addr) != 1) throw std::invalid_argument("[hunter_hip] LCM url host: " + host);
    rx_ = ::socket(AF_INET, SOCK_DGRAM, 0);
    tx_ = ::socket(AF_INET, SOCK_DGRAM, 0);
    if (rx_ < 0 || tx_ < 0) { closeAll(); throw std::runtime_error("[hunter_hip] LCM: socket() failed"); }   // (no destructor runs for a throwing constructor)
    const int one = 1;
    ::setsockopt(rx_, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
#ifdef SO_REUSEPORT
    ::setsockopt(rx_, SOL_SOCKET, SO_REUSEPORT, &one, sizeof(one));
#endif
    sockaddr_in local{};
    local.sin_family = AF_INET;
    local.sin_port = htons(uint16_t(port));
    local.sin_addr.s_addr = htonl(INADDR_ANY);
    if (::bind(rx_, reinterpret_cast<sockaddr*>(&local), sizeof(local)) != 0) { closeAll(); throw std::runtime_error("[hunter_hip] LCM: bind() failed"); }
    {
      ip_mreq mreq{};
      mreq.imr_multiaddr = dest_.sin_addr;
      mreq.imr_interface.s_addr = htonl(INADDR_ANY);
      if (::setsockopt(rx_, IPPROTO_IP, IP_ADD_MEMBERSHIP, &mreq, sizeof(mreq)) != 0) { closeAll(); throw std::runtime_error("[hunter_hip] LCM: cannot join the multicast group (no multicast route?)"); }
      const unsigned char t = static_cast<unsigned char>(ttl), loop = 1;
      // ttl = 0 keeps the datagrams on this host and LOOP delivers them to the other endpoints of it: without either the bus is
      // silently something else than the reference's, so a failure is an error
      if (::setsockopt(tx_, IPPROTO_IP, IP_MULTICAST_TTL, &t, sizeof(t)) != 0 ||
          ::setsockopt(tx_, IPPROTO_IP, IP_MULTICAST_LOOP, &loop, sizeof(loop)) != 0) {
        closeAll();
        throw std::runtime_error("[hunter_hip] LCM: cannot set the multicast ttl / loop options");
      }
    }
  }
  ~LcmUdp() { closeAll(); }
  LcmUdp(const LcmUdp&) = delete;
  LcmUdp& operator=(const LcmUdp&) = delete;

  void publish(const std::string& channel, const uint8_t* payload, int len) {
    std::vector<uint8_t> frame(size_t(len) + channel.size() + 16);
    const int32_t n = hb_lcm_frame(channel.c_str(), seq_++, payload, len, frame.data(), int32_t(frame.size()));
    if (n < 0) throw std::invalid_argument("[hunter_hip] LCM: message does not fit a short datagram");
    if (::sendto(tx_, frame.data(), size_t(n), 0, reinterpret_cast<const sockaddr*>(&dest_), sizeof(dest_)) != n)
      throw std::runtime_error("[hunter_hip] LCM: sendto() failed");
  }
  // one datagram; false on timeout.  Malformed datagrams are dropped (like liblcm) and the wait continues.
  bool receive(std::string& channel, std::vector<uint8_t>& payload, int timeout_ms) {
    uint8_t buf[65536];
    // ONE deadline for the call: foreign / malformed datagrams on a busy port do not restart the wait
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms < 0 ? 0 : timeout_ms);
    for (;;) {
      int wait_ms = -1;   // (timeout_ms < 0: wait for ever, as poll does)
      if (timeout_ms >= 0) {
        const auto left = std::chrono::duration_cast<std::chrono::milliseconds>(deadline - std::chrono::steady_clock::now()).count();
        wait_ms = left > 0 ? int(left) : 0;
      }
      pollfd pfd{rx_, POLLIN, 0};
      if (::poll(&pfd, 1, wait_ms) <= 0) return false;
      const ssize_t n = ::recv(rx_, buf, sizeof(buf), 0);
      if (n <= 0) return false;
      char ch[64];
      int32_t off = 0;
      const int32_t plen = hb_lcm_unframe(buf, int32_t(n), ch, int32_t(sizeof(ch)), nullptr, &off);
      if (plen < 0) continue;
      channel = ch;
      payload.assign(buf + off, buf + off + plen);
      return true;
    }
  }

 private:
  void closeAll() {
    if (rx_ >= 0) ::close(rx_);
    if (tx_ >= 0) ::close(tx_);
    rx_ = tx_ = -1;
  }
  int rx_ = -1, tx_ = -1;
  sockaddr_in dest_{};
  uint32_t seq_ = 0;
};

}  // namespace hunter_hip
