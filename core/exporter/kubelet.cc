// Kubelet PodResources client: one unary gRPC call, PodResourcesLister.List,
// over the kubelet's unix socket — written directly against HTTP/2 framing and
// the protobuf wire format because the call is tiny (an empty request, a
// response made of five nested string-only messages) and the image carries no
// gRPC C++ runtime.  Covers what the reference does with grpc-go
// (exporters/prometheus-dcgm/k8s/pod-gpu-metrics-exporter/src/kubelet_server.go:20-53,
// device_pod.go:26-46; messages: vendor/k8s.io/kubernetes/pkg/kubelet/apis/
// podresources/v1alpha1/api.proto:19-48).  Speaks the v1 service first and
// falls back to v1alpha1, which newer kubelets no longer serve.
#include <cerrno>
#include <cstring>
#include <poll.h>
#include <sys/socket.h>
#include <sys/un.h>
#include <unistd.h>

#include "../common/util.h"
#include "exporter.h"

namespace gpumon {
namespace {

enum { F_DATA = 0, F_HEADERS = 1, F_RST = 3, F_SETTINGS = 4, F_PING = 6, F_GOAWAY = 7, F_WINDOW = 8, F_CONT = 9 };
const uint32_t kWindow = 16u * 1024 * 1024;  // the reference's 16 MiB receive cap

struct Conn {
  int fd = -1;
  int64_t deadline_us = 0;
  std::string err;
  uint64_t consumed = 0;  // DATA bytes received since the connection window was last topped up

  bool wait(short ev) {
    while (true) {
      int64_t left = deadline_us - monotonic_us();
      if (left <= 0) { err = "timed out"; return false; }
      pollfd p = {fd, ev, 0};
      int rc = poll(&p, 1, (int)(left / 1000) + 1);
      if (rc > 0) return true;
      if (rc < 0 && errno != EINTR) { err = strerror(errno); return false; }
    }
  }
  bool send_all(const std::string& b) {
    size_t off = 0;
    while (off < b.size()) {
      if (!wait(POLLOUT)) return false;
      ssize_t n = send(fd, b.data() + off, b.size() - off, MSG_NOSIGNAL);
      if (n < 0) {
        if (errno == EAGAIN || errno == EINTR) continue;
        err = strerror(errno);
        return false;
      }
      off += (size_t)n;
    }
    return true;
  }
  bool recv_n(char* dst, size_t n) {
    size_t off = 0;
    while (off < n) {
      if (!wait(POLLIN)) return false;
      ssize_t r = recv(fd, dst + off, n - off, 0);
      if (r == 0) { err = "connection closed by kubelet"; return false; }
      if (r < 0) {
        if (errno == EAGAIN || errno == EINTR) continue;
        err = strerror(errno);
        return false;
      }
      off += (size_t)r;
    }
    return true;
  }
};

std::string frame(uint8_t type, uint8_t flags, uint32_t stream, const std::string& payload) {
  std::string f;
  uint32_t n = (uint32_t)payload.size();
  f += (char)(n >> 16); f += (char)(n >> 8); f += (char)n;
  f += (char)type; f += (char)flags;
  f += (char)((stream >> 24) & 0x7f); f += (char)(stream >> 16); f += (char)(stream >> 8); f += (char)stream;
  f += payload;
  return f;
}

std::string u32be(uint32_t v) {
  std::string s;
  s += (char)(v >> 24); s += (char)(v >> 16); s += (char)(v >> 8); s += (char)v;
  return s;
}

// HPACK string literal, no Huffman (lengths here are < 127).
void hp_str(std::string* o, const std::string& s) {
  *o += (char)s.size();
  *o += s;
}

std::string request_headers(const std::string& path) {
  std::string h;
  h += (char)0x83;  // :method POST   (static table 3)
  h += (char)0x86;  // :scheme http   (static table 6)
  h += (char)0x04;  // :path, literal without indexing, name = static 4
  hp_str(&h, path);
  h += (char)0x01;  // :authority, name = static 1
  hp_str(&h, "localhost");
  h += (char)0x0f; h += (char)0x10;  // content-type, name = static 31 (15 + 16)
  hp_str(&h, "application/grpc");
  h += (char)0x00;  // new name
  hp_str(&h, "te");
  hp_str(&h, "trailers");
  h += (char)0x00;
  hp_str(&h, "user-agent");
  hp_str(&h, "gpumon-pod-exporter");
  return h;
}

// ---- protobuf wire format ----------------------------------------------------
struct PbReader {
  const uint8_t* p;
  const uint8_t* end;
  bool ok = true;
  PbReader(const uint8_t* b, size_t n) : p(b), end(b + n) {}
  bool done() const { return p >= end || !ok; }
  uint64_t varint() {
    uint64_t v = 0;
    for (int shift = 0; shift < 64 && p < end; shift += 7) {
      uint8_t b = *p++;
      v |= (uint64_t)(b & 0x7f) << shift;
      if (!(b & 0x80)) return v;
    }
    ok = false;
    return 0;
  }
  // Next field: number + wire type; for length-delimited fields sets sub/len.
  bool next(uint32_t* field, uint32_t* wt, const uint8_t** sub, size_t* len) {
    if (done()) return false;
    uint64_t key = varint();
    if (!ok) return false;
    *field = (uint32_t)(key >> 3);
    *wt = (uint32_t)(key & 7);
    switch (*wt) {
      case 0: varint(); break;
      case 1: if (end - p < 8) ok = false; else p += 8; break;
      case 5: if (end - p < 4) ok = false; else p += 4; break;
      case 2: {
        uint64_t n = varint();
        if (!ok || n > (uint64_t)(end - p)) { ok = false; return false; }
        *sub = p;
        *len = (size_t)n;
        p += n;
        break;
      }
      default: ok = false;
    }
    return ok;
  }
};

bool resource_wanted(const std::string& name, const std::string& csv) {
  for (auto& w : split(csv, ','))
    if (trim(w) == name) return true;
  return false;
}

bool decode_list_response(const uint8_t* b, size_t n, const std::string& resources, PodMap* out) {
  PbReader top(b, n);
  uint32_t f, wt;
  const uint8_t* s;
  size_t l;
  while (top.next(&f, &wt, &s, &l)) {
    if (f != 1 || wt != 2) continue;  // repeated PodResources pod_resources = 1
    PbReader pod(s, l);
    std::string pod_name, pod_ns;
    std::vector<std::pair<const uint8_t*, size_t>> containers;
    const uint8_t* s2;
    size_t l2;
    while (pod.next(&f, &wt, &s2, &l2)) {
      if (wt != 2) continue;
      if (f == 1) pod_name.assign((const char*)s2, l2);
      else if (f == 2) pod_ns.assign((const char*)s2, l2);
      else if (f == 3) containers.emplace_back(s2, l2);
    }
    if (!pod.ok) return false;
    for (auto& c : containers) {
      PbReader cont(c.first, c.second);
      std::string cname;
      std::vector<std::pair<const uint8_t*, size_t>> devices;
      while (cont.next(&f, &wt, &s2, &l2)) {
        if (wt != 2) continue;
        if (f == 1) cname.assign((const char*)s2, l2);
        else if (f == 2) devices.emplace_back(s2, l2);
      }
      if (!cont.ok) return false;
      for (auto& d : devices) {
        PbReader dev(d.first, d.second);
        std::string rname;
        std::vector<std::string> ids;
        while (dev.next(&f, &wt, &s2, &l2)) {
          if (wt != 2) continue;
          if (f == 1) rname.assign((const char*)s2, l2);
          else if (f == 2) ids.emplace_back((const char*)s2, l2);
        }
        if (!dev.ok) return false;
        if (!resource_wanted(rname, resources)) continue;
        for (auto& id : ids) (*out)[id] = PodInfo{pod_name, pod_ns, cname};
      }
    }
  }
  return top.ok;
}

// One unary call on `stream`.  Returns 1 = got a message, 0 = stream ended
// without one (e.g. UNIMPLEMENTED), -1 = transport error (c.err set).
int unary_call(Conn& c, uint32_t stream, const std::string& path, std::string* message) {
  std::string out = frame(F_HEADERS, 0x4 /*END_HEADERS*/, stream, request_headers(path));
  out += frame(F_DATA, 0x1 /*END_STREAM*/, stream, std::string("\0\0\0\0\0", 5));  // empty ListPodResourcesRequest
  if (!c.send_all(out)) return -1;
  std::string body, trailer_block;
  uint64_t consumed = 0;  // of this stream's window
  while (true) {
    char hdr[9];
    if (!c.recv_n(hdr, 9)) return -1;
    uint32_t len = ((uint8_t)hdr[0] << 16) | ((uint8_t)hdr[1] << 8) | (uint8_t)hdr[2];
    uint8_t type = (uint8_t)hdr[3], flags = (uint8_t)hdr[4];
    uint32_t sid = (((uint8_t)hdr[5] & 0x7f) << 24) | ((uint8_t)hdr[6] << 16) | ((uint8_t)hdr[7] << 8) | (uint8_t)hdr[8];
    if (len > kWindow) { c.err = "oversized frame"; return -1; }
    std::string payload(len, 0);
    if (len && !c.recv_n(&payload[0], len)) return -1;
    bool end_stream = false;
    switch (type) {
      case F_SETTINGS:
        if (!(flags & 0x1) && !c.send_all(frame(F_SETTINGS, 0x1, 0, ""))) return -1;
        break;
      case F_PING:
        if (!(flags & 0x1) && !c.send_all(frame(F_PING, 0x1, 0, payload))) return -1;
        break;
      case F_GOAWAY:
        c.err = "kubelet sent GOAWAY";
        return -1;
      case F_RST:
        if (sid == stream) { c.err = "stream reset by kubelet"; return 0; }
        break;
      case F_HEADERS:
      case F_CONT:
        if (sid == stream) {
          trailer_block += payload;
          end_stream = type == F_HEADERS && (flags & 0x1);
        }
        break;
      case F_DATA:
        if (sid == stream) {
          size_t off = 0, n = payload.size();
          if (flags & 0x8) {  // PADDED
            if (n < 1) { c.err = "bad padding"; return -1; }
            uint8_t pad = (uint8_t)payload[0];
            off = 1;
            if ((size_t)pad + 1 > n) { c.err = "bad padding"; return -1; }
            n -= pad;
          }
          body.append(payload, off, n - off);
          if (body.size() > kWindow) { c.err = "response larger than 16 MiB"; return -1; }
          consumed += len;
          if (consumed > kWindow / 2) {  // top the stream's flow-control window back up
            if (!c.send_all(frame(F_WINDOW, 0, stream, u32be((uint32_t)consumed)))) return -1;
            consumed = 0;
          }
          end_stream = flags & 0x1;
        }
        // the connection's window is charged for every DATA frame, whichever stream it belongs to, and lives as
        // long as the connection does
        c.consumed += len;
        if (c.consumed > kWindow / 2) {
          if (!c.send_all(frame(F_WINDOW, 0, 0, u32be((uint32_t)c.consumed)))) return -1;
          c.consumed = 0;
        }
        break;
      default: break;
    }
    if (end_stream) break;
  }
  if (body.size() < 5) {
    // Trailers-only reply.  HPACK strings may be Huffman coded; a plain-text
    // grpc-message is surfaced when present.
    size_t p = trailer_block.find("grpc-message");
    c.err = p != std::string::npos ? "rpc failed: " + trailer_block.substr(p + 13, 80) : "rpc returned no message";
    return 0;
  }
  if (body[0] != 0) { c.err = "compressed gRPC message not supported"; return -1; }
  uint32_t mlen = ((uint8_t)body[1] << 24) | ((uint8_t)body[2] << 16) | ((uint8_t)body[3] << 8) | (uint8_t)body[4];
  if ((size_t)mlen + 5 > body.size()) { c.err = "truncated gRPC message"; return -1; }
  message->assign(body, 5, mlen);
  return 1;
}

}  // namespace

// ---- persistent client ---------------------------------------------------------
struct KubeletClient::Impl {
  Conn c;
  uint32_t next_stream = 1;
  int service = 0;  // index into kPaths of the service that answered last
};

static const char* kPaths[] = {"/v1.PodResourcesLister/List", "/v1alpha1.PodResourcesLister/List"};

KubeletClient::KubeletClient(const std::string& socket_path, const std::string& resource_names)
    : socket_path_(socket_path), resource_names_(resource_names), impl_(new Impl) {}

KubeletClient::~KubeletClient() {
  disconnect();
  delete impl_;
}

void KubeletClient::disconnect() {
  if (impl_->c.fd >= 0) close(impl_->c.fd);
  impl_->c.fd = -1;
}

gm_status_t KubeletClient::connect_now(std::string* error) {
  Conn& c = impl_->c;
  sockaddr_un sa;
  memset(&sa, 0, sizeof sa);
  sa.sun_family = AF_UNIX;
  if (socket_path_.size() >= sizeof sa.sun_path) {
    *error = "socket path too long";
    return GM_ST_BADPARAM;
  }
  strcpy(sa.sun_path, socket_path_.c_str());
  c.fd = socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC | SOCK_NONBLOCK, 0);
  if (c.fd < 0) {
    *error = strerror(errno);
    return GM_ST_CONNECTION_NOT_VALID;
  }
  auto fail = [&](gm_status_t st, const std::string& msg) {
    *error = msg;
    disconnect();
    return st;
  };
  if (connect(c.fd, (sockaddr*)&sa, sizeof sa) != 0) {
    if (errno != EINPROGRESS && errno != EAGAIN)
      return fail(GM_ST_CONNECTION_NOT_VALID, strprintf("failure connecting to %s: %s", socket_path_.c_str(), strerror(errno)));
    if (!c.wait(POLLOUT)) return fail(GM_ST_TIMEOUT, "connect: " + c.err);
    int soerr = 0;
    socklen_t sl = sizeof soerr;
    getsockopt(c.fd, SOL_SOCKET, SO_ERROR, &soerr, &sl);
    if (soerr) return fail(GM_ST_CONNECTION_NOT_VALID, strprintf("failure connecting to %s: %s", socket_path_.c_str(), strerror(soerr)));
  }
  std::string hello = "PRI * HTTP/2.0\r\n\r\nSM\r\n\r\n";
  std::string settings;
  settings += std::string("\x00\x02", 2) + u32be(0);        // ENABLE_PUSH = 0
  settings += std::string("\x00\x04", 2) + u32be(kWindow);  // INITIAL_WINDOW_SIZE
  hello += frame(F_SETTINGS, 0, 0, settings);
  hello += frame(F_WINDOW, 0, 0, u32be(kWindow - 65535));
  if (!c.send_all(hello)) return fail(GM_ST_CONNECTION_NOT_VALID, c.err);
  impl_->next_stream = 1;
  c.consumed = 0;
  connects_++;
  return GM_ST_OK;
}

// One List() on the connection as it is.  *transport = the connection is no longer usable.
gm_status_t KubeletClient::list_once(PodMap* out, std::string* error, bool* transport) {
  Conn& c = impl_->c;
  *transport = false;
  // the service that answered last time first; the other one if that one is gone (a kubelet upgrade)
  for (int k = 0; k < 2; k++) {
    const int svc = (impl_->service + k) % 2;
    std::string msg;
    const uint32_t stream = impl_->next_stream;
    impl_->next_stream += 2;
    int rc = unary_call(c, stream, kPaths[svc], &msg);
    if (rc < 0) {
      *transport = true;
      *error = "failure getting pod resources: " + c.err;
      return c.err == "timed out" ? GM_ST_TIMEOUT : GM_ST_CONNECTION_NOT_VALID;
    }
    if (rc == 0) continue;  // not served under this name: try the other service
    impl_->service = svc;
    out->clear();
    if (!decode_list_response((const uint8_t*)msg.data(), msg.size(), resource_names_, out)) {
      *error = "malformed ListPodResourcesResponse";
      return GM_ST_GENERIC_ERROR;
    }
    return GM_ST_OK;
  }
  *error = "failure getting pod resources: " + c.err;
  return GM_ST_NOT_SUPPORTED;
}

gm_status_t KubeletClient::list(int timeout_ms, PodMap* out, std::string* error) {
  std::string err;
  Conn& c = impl_->c;
  c.deadline_us = monotonic_us() + (int64_t)timeout_ms * 1000;
  c.err.clear();
  // A connection kept from the last call may have been closed under us (kubelet restart, GOAWAY): one fresh
  // connection is tried before the call counts as failed.  A timeout is not retried: the time is spent.
  for (int attempt = 0; attempt < 2; attempt++) {
    const bool reused = c.fd >= 0 && impl_->next_stream < (1u << 30);
    if (!reused) {
      disconnect();
      gm_status_t st = connect_now(&err);
      if (st != GM_ST_OK) {
        if (error) *error = err;
        return st;
      }
    }
    bool transport = false;
    gm_status_t st = list_once(out, &err, &transport);
    if (!transport) {
      if (error && st != GM_ST_OK) *error = err;
      return st;
    }
    disconnect();
    if (!reused || st == GM_ST_TIMEOUT) {
      if (error) *error = err;
      return st;
    }
  }
  if (error) *error = err;
  return GM_ST_CONNECTION_NOT_VALID;
}

gm_status_t kubelet_list_pod_resources(const std::string& socket_path, const std::string& resource_names,
                                       int timeout_ms, PodMap* out, std::string* error) {
  KubeletClient k(socket_path, resource_names);
  std::string err;
  gm_status_t st = k.list(timeout_ms, out, &err);
  if (error) *error = err;
  return st;
}

}  // namespace gpumon
