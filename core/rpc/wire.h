// Wire format between libgpumon clients and the gpumon-hostengine daemon:
// length-prefixed binary frames over a unix or TCP socket.  Plays the role of
// the libdcgm <-> nv-hostengine connection the reference opens with
// dcgmConnect_v2 (bindings/go/dcgm/admin.go:109-134; default TCP port 5555,
// bindings/go/samples/dcgm/README.md:19-21).
//
// frame   := header payload
// header  := u32 magic 'GMRP' | u32 op | u32 id | u32 payload_len   (little endian)
// request : op = Op, id = caller-chosen correlation id
// reply   : op = Op | kReplyBit, same id, payload = i32 status + results
// notify  : op = OP_NOTIFY_*, id = registration / subscription id
//
// ABI structs travel as raw bytes guarded by their version word (sizeof |
// version << 24), so a client and a daemon built from different headers fail
// with GM_ST_VER_MISMATCH instead of misreading each other.
#pragma once

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace gpumon {

const uint32_t kRpcMagic = 0x50524d47;  // "GMRP"
const uint32_t kReplyBit = 0x80000000u;
const uint32_t kMaxFrame = 64u << 20;
const uint32_t kProtocolVersion = 1;
const int kDefaultPort = 5555;

enum Op : uint32_t {
  OP_HELLO = 1,
  OP_DEVICE_COUNT, OP_SUPPORTED_DEVICES, OP_DEVICE_ATTRIBUTES, OP_DEVICE_TOPOLOGY,
  OP_GROUP_CREATE, OP_GROUP_ADD, OP_GROUP_REMOVE, OP_GROUP_DEVICES, OP_GROUP_DESTROY,
  OP_FG_CREATE, OP_FG_DESTROY,
  OP_WATCH_FIELDS, OP_UNWATCH_FIELDS, OP_UPDATE_ALL, OP_LATEST_VALUES, OP_VALUES_SINCE,
  OP_INJECT_VALUE, OP_FAKE_CTL,
  OP_DEVICE_STATUS, OP_DEVICE_PROCESSES, OP_DEVICE_MODE,
  OP_HEALTH_SET, OP_HEALTH_GET, OP_HEALTH_CHECK,
  OP_POLICY_SET, OP_POLICY_GET, OP_POLICY_REGISTER, OP_POLICY_UNREGISTER,
  OP_WATCH_PID_FIELDS, OP_PID_INFO,
  OP_INTROSPECT_TOGGLE, OP_INTROSPECT,
  OP_EVENT_SUBSCRIBE, OP_EVENT_UNSUBSCRIBE,
  OP_XGMI_LINKS, OP_HEALTH_THRESHOLDS_SET, OP_HEALTH_THRESHOLDS_GET, OP_DIAG_SUBMIT, OP_DIAG_RESULTS, OP_PROCESS_SOURCE, OP_METRICS_TABLE_SOURCE,
  OP_NOTIFY_VIOLATION = 1000,
  OP_NOTIFY_EVENT = 1001,
};

struct FrameHeader {
  uint32_t magic, op, id, len;
};

class WireWriter {
 public:
  void u32(uint32_t v) { raw(&v, 4); }
  void i32(int32_t v) { raw(&v, 4); }
  void i64(int64_t v) { raw(&v, 8); }
  void f64(double v) { raw(&v, 8); }
  void str(const std::string& s) {
    u32((uint32_t)s.size());
    raw(s.data(), s.size());
  }
  template <typename T>
  void pod(const T& t) {
    u32((uint32_t)sizeof(T));
    raw(&t, sizeof(T));
  }
  void raw(const void* p, size_t n) { buf_.append((const char*)p, n); }
  const std::string& data() const { return buf_; }

 private:
  std::string buf_;
};

class WireReader {
 public:
  WireReader(const char* p, size_t n) : p_(p), end_(p + n) {}
  bool ok() const { return ok_; }
  uint32_t u32() { uint32_t v = 0; raw(&v, 4); return v; }
  int32_t i32() { int32_t v = 0; raw(&v, 4); return v; }
  int64_t i64() { int64_t v = 0; raw(&v, 8); return v; }
  double f64() { double v = 0; raw(&v, 8); return v; }
  std::string str() {
    uint32_t n = u32();
    if (!ok_ || n > (size_t)(end_ - p_)) { ok_ = false; return std::string(); }
    std::string s(p_, n);
    p_ += n;
    return s;
  }
  // Fails (and leaves *t untouched) when the peer's struct has another size.
  template <typename T>
  bool pod(T* t) {
    uint32_t n = u32();
    if (!ok_ || n != sizeof(T) || n > (size_t)(end_ - p_)) { ok_ = false; return false; }
    memcpy(t, p_, n);
    p_ += n;
    return true;
  }
  void raw(void* dst, size_t n) {
    if (!ok_ || n > (size_t)(end_ - p_)) { ok_ = false; return; }
    memcpy(dst, p_, n);
    p_ += n;
  }

 private:
  const char* p_;
  const char* end_;
  bool ok_ = true;
};

// Blocking socket helpers shared by client and server.
bool rpc_send_frame(int fd, uint32_t op, uint32_t id, const std::string& payload);
// Returns 1 on success, 0 on orderly EOF, -1 on error / malformed frame.
int rpc_recv_frame(int fd, FrameHeader* h, std::string* payload);
// address: unix socket path when is_unix, else "host[:port]" / ":port".
int rpc_listen(const std::string& address, bool is_unix, std::string* error);
int rpc_connect(const std::string& address, bool is_unix, int timeout_ms, std::string* error);

}  // namespace gpumon
