#include "wire.h"

#include <arpa/inet.h>
#include <cerrno>
#include <fcntl.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/un.h>
#include <unistd.h>

#include "../common/util.h"

namespace gpumon {

static bool send_all(int fd, const char* p, size_t n) {
  while (n) {
    ssize_t w = send(fd, p, n, MSG_NOSIGNAL);
    if (w < 0) {
      if (errno == EINTR) continue;
      return false;
    }
    p += w;
    n -= (size_t)w;
  }
  return true;
}

static int recv_all(int fd, char* p, size_t n) {
  size_t got = 0;
  while (got < n) {
    ssize_t r = recv(fd, p + got, n - got, 0);
    if (r == 0) return got == 0 ? 0 : -1;
    if (r < 0) {
      if (errno == EINTR) continue;
      return -1;
    }
    got += (size_t)r;
  }
  return 1;
}

bool rpc_send_frame(int fd, uint32_t op, uint32_t id, const std::string& payload) {
  FrameHeader h = {kRpcMagic, op, id, (uint32_t)payload.size()};
  std::string buf((const char*)&h, sizeof h);
  buf += payload;
  return send_all(fd, buf.data(), buf.size());
}

int rpc_recv_frame(int fd, FrameHeader* h, std::string* payload) {
  int rc = recv_all(fd, (char*)h, sizeof *h);
  if (rc <= 0) return rc;
  if (h->magic != kRpcMagic || h->len > kMaxFrame) return -1;
  payload->resize(h->len);
  if (h->len && recv_all(fd, &(*payload)[0], h->len) != 1) return -1;
  return 1;
}

static void split_host_port(const std::string& address, std::string* host, int* port) {
  *host = address;
  *port = kDefaultPort;
  size_t c = address.rfind(':');
  if (c != std::string::npos) {
    *host = address.substr(0, c);
    *port = atoi(address.substr(c + 1).c_str());
  }
  if (host->empty()) *host = "0.0.0.0";
  if (*host == "localhost") *host = "127.0.0.1";
}

int rpc_listen(const std::string& address, bool is_unix, std::string* error) {
  int fd;
  if (is_unix) {
    sockaddr_un sa;
    memset(&sa, 0, sizeof sa);
    sa.sun_family = AF_UNIX;
    if (address.size() >= sizeof sa.sun_path) {
      *error = "socket path too long";
      return -1;
    }
    strcpy(sa.sun_path, address.c_str());
    unlink(address.c_str());
    fd = socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0);
    if (fd < 0 || bind(fd, (sockaddr*)&sa, sizeof sa) != 0 || listen(fd, 64) != 0) {
      *error = strprintf("cannot listen on %s: %s", address.c_str(), strerror(errno));
      if (fd >= 0) close(fd);
      return -1;
    }
    chmod(address.c_str(), 0660);
    return fd;
  }
  std::string host;
  int port;
  split_host_port(address, &host, &port);
  fd = socket(AF_INET, SOCK_STREAM | SOCK_CLOEXEC, 0);
  int one = 1;
  if (fd >= 0) setsockopt(fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
  sockaddr_in sa;
  memset(&sa, 0, sizeof sa);
  sa.sin_family = AF_INET;
  sa.sin_port = htons((uint16_t)port);
  if (inet_pton(AF_INET, host.c_str(), &sa.sin_addr) != 1) sa.sin_addr.s_addr = htonl(INADDR_ANY);
  if (fd < 0 || bind(fd, (sockaddr*)&sa, sizeof sa) != 0 || listen(fd, 64) != 0) {
    *error = strprintf("cannot listen on %s: %s", address.c_str(), strerror(errno));
    if (fd >= 0) close(fd);
    return -1;
  }
  return fd;
}

int rpc_connect(const std::string& address, bool is_unix, int timeout_ms, std::string* error) {
  int64_t deadline = monotonic_us() + (int64_t)timeout_ms * 1000;
  while (true) {
    int fd = -1;
    int rc = -1;
    if (is_unix) {
      sockaddr_un sa;
      memset(&sa, 0, sizeof sa);
      sa.sun_family = AF_UNIX;
      if (address.size() >= sizeof sa.sun_path) {
        *error = "socket path too long";
        return -1;
      }
      strcpy(sa.sun_path, address.c_str());
      fd = socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0);
      if (fd >= 0) rc = connect(fd, (sockaddr*)&sa, sizeof sa);
    } else {
      std::string host;
      int port;
      split_host_port(address, &host, &port);
      if (host == "0.0.0.0") host = "127.0.0.1";
      addrinfo hints, *res = nullptr;
      memset(&hints, 0, sizeof hints);
      hints.ai_family = AF_INET;
      hints.ai_socktype = SOCK_STREAM;
      if (getaddrinfo(host.c_str(), strprintf("%d", port).c_str(), &hints, &res) == 0 && res) {
        fd = socket(AF_INET, SOCK_STREAM | SOCK_CLOEXEC, 0);
        if (fd >= 0) rc = connect(fd, res->ai_addr, res->ai_addrlen);
        freeaddrinfo(res);
        int one = 1;
        if (fd >= 0) setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
      }
    }
    if (fd >= 0 && rc == 0) return fd;
    int e = errno;
    if (fd >= 0) close(fd);
    // A freshly spawned daemon may not be listening yet.
    if (monotonic_us() >= deadline) {
      *error = strprintf("cannot connect to %s: %s", address.c_str(), strerror(e));
      return -1;
    }
    usleep(20000);
  }
}

}  // namespace gpumon
