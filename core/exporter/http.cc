#include "http.h"

#include <arpa/inet.h>
#include <cerrno>
#include <cstring>
#include <fcntl.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>

#include "../common/util.h"

namespace gpumon {

const char* http_status_text(int code) {
  switch (code) {
    case 200: return "OK";
    case 400: return "Bad Request";
    case 404: return "Not Found";
    case 405: return "Method Not Allowed";
    case 408: return "Request Timeout";
    case 413: return "Payload Too Large";
    case 500: return "Internal Server Error";
    case 503: return "Service Unavailable";
  }
  return "Unknown";
}

HttpServer::HttpServer() {}
HttpServer::~HttpServer() { stop(); }

static void set_nonblock(int fd) {
  int fl = fcntl(fd, F_GETFL, 0);
  fcntl(fd, F_SETFL, fl | O_NONBLOCK);
  fcntl(fd, F_SETFD, FD_CLOEXEC);
}

bool HttpServer::start(const std::string& address, HttpHandler handler, int timeout_ms) {
  // "host:port", ":port", "port"; an IPv6 host in brackets: "[::1]:9400", "[::]:9400" (both families where the
  // host allows dual-stack sockets).
  std::string host = "0.0.0.0", port = address;
  bool v6 = false;
  if (!address.empty() && address[0] == '[') {
    const size_t close_br = address.find(']');
    if (close_br == std::string::npos) {
      GM_LOGE("http: malformed listen address %s", address.c_str());
      return false;
    }
    host = address.substr(1, close_br - 1);
    port = close_br + 1 < address.size() && address[close_br + 1] == ':' ? address.substr(close_br + 2) : "0";
    v6 = true;
  } else {
    size_t c = address.rfind(':');
    if (c != std::string::npos) {
      if (c > 0) host = address.substr(0, c);
      port = address.substr(c + 1);
    }
  }
  if (host == "localhost") host = "127.0.0.1";
  int fd = socket(v6 ? AF_INET6 : AF_INET, SOCK_STREAM, 0);
  if (fd < 0) {
    GM_LOGE("http: cannot open a socket for %s: %s", address.c_str(), strerror(errno));
    return false;
  }
  int one = 1;
  setsockopt(fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
  sockaddr_storage ss;
  memset(&ss, 0, sizeof ss);
  socklen_t sl;
  if (v6) {
    sockaddr_in6* sa = (sockaddr_in6*)&ss;
    sa->sin6_family = AF_INET6;
    sa->sin6_port = htons((uint16_t)atoi(port.c_str()));
    if (inet_pton(AF_INET6, host.c_str(), &sa->sin6_addr) != 1) sa->sin6_addr = in6addr_any;
    int off = 0;  // the wildcard address takes IPv4 clients too
    setsockopt(fd, IPPROTO_IPV6, IPV6_V6ONLY, &off, sizeof off);
    sl = sizeof *sa;
  } else {
    sockaddr_in* sa = (sockaddr_in*)&ss;
    sa->sin_family = AF_INET;
    sa->sin_port = htons((uint16_t)atoi(port.c_str()));
    if (inet_pton(AF_INET, host.c_str(), &sa->sin_addr) != 1) sa->sin_addr.s_addr = htonl(INADDR_ANY);
    sl = sizeof *sa;
  }
  if (bind(fd, (sockaddr*)&ss, sl) != 0 || listen(fd, 128) != 0) {
    GM_LOGE("http: cannot listen on %s: %s", address.c_str(), strerror(errno));
    close(fd);
    return false;
  }
  sl = sizeof ss;
  getsockname(fd, (sockaddr*)&ss, &sl);
  port_ = ntohs(v6 ? ((sockaddr_in6*)&ss)->sin6_port : ((sockaddr_in*)&ss)->sin_port);
  set_nonblock(fd);
  int p[2];
  if (pipe(p) != 0) {
    close(fd);
    return false;
  }
  set_nonblock(p[0]);
  set_nonblock(p[1]);
  listen_fd_ = fd;
  wake_rd_ = p[0];
  wake_wr_ = p[1];
  handler_ = std::move(handler);
  timeout_ms_ = timeout_ms;
  running_ = true;
  th_ = std::thread([this] { loop(); });
  return true;
}

void HttpServer::stop() {
  if (!running_.exchange(false)) return;
  char b = 1;
  if (write(wake_wr_, &b, 1) < 0) {}
  if (th_.joinable()) th_.join();
  close(listen_fd_);
  close(wake_rd_);
  close(wake_wr_);
  listen_fd_ = wake_rd_ = wake_wr_ = -1;
}

// Parses complete requests out of c.in, appends responses to c.out.
// Returns false when the connection must be dropped.
bool HttpServer::handle_input(Conn& c) {
  while (true) {
    size_t end = c.in.find("\r\n\r\n");
    if (end == std::string::npos) return c.in.size() <= 16384;
    std::string head = c.in.substr(0, end);
    c.in.erase(0, end + 4);
    HttpRequest req;
    HttpResponse resp;
    std::vector<std::string> lines = split(head, '\n');
    std::vector<std::string> rl = split(trim(lines[0]), ' ');
    bool keep = true;
    if (rl.size() < 3) {
      resp.status = 400;
      resp.body = "malformed request line\n";
      keep = false;
    } else {
      req.method = rl[0];
      std::string target = rl[1];
      size_t q = target.find('?');
      req.path = q == std::string::npos ? target : target.substr(0, q);
      req.query = q == std::string::npos ? "" : target.substr(q + 1);
      if (rl[2] == "HTTP/1.0") keep = false;
      for (size_t i = 1; i < lines.size(); i++) {
        size_t col = lines[i].find(':');
        if (col == std::string::npos) continue;
        std::string k = trim(lines[i].substr(0, col)), v = trim(lines[i].substr(col + 1));
        for (auto& ch : k) ch = (char)tolower((unsigned char)ch);
        req.headers[k] = v;
      }
      auto ch = req.headers.find("connection");
      if (ch != req.headers.end()) {
        std::string v = ch->second;
        for (auto& x : v) x = (char)tolower((unsigned char)x);
        if (v == "close") keep = false;
        if (v == "keep-alive") keep = true;
      }
      if (req.headers.count("content-length") && atoi(req.headers["content-length"].c_str()) > 0) {
        resp.status = 413;
        resp.body = "request bodies are not accepted\n";
        keep = false;
      } else if (req.method != "GET" && req.method != "HEAD") {
        resp.status = 405;
        resp.body = "method not allowed\n";
      } else {
        handler_(req, &resp);
      }
    }
    requests_++;
    std::string hdr = strprintf("HTTP/1.1 %d %s\r\nContent-Type: %s\r\nContent-Length: %zu\r\nConnection: %s\r\n\r\n",
                                resp.status, http_status_text(resp.status), resp.content_type.c_str(),
                                resp.body.size(), keep ? "keep-alive" : "close");
    c.out += hdr;
    if (req.method != "HEAD") c.out += resp.body;
    if (!keep) {
      c.close_after = true;
      return true;
    }
  }
}

void HttpServer::loop() {
  std::vector<Conn> conns;
  std::vector<pollfd> pfds;
  while (running_) {
    pfds.clear();
    pfds.push_back({listen_fd_, POLLIN, 0});
    pfds.push_back({wake_rd_, POLLIN, 0});
    for (auto& c : conns) pfds.push_back({c.fd, (short)(c.out_off < c.out.size() ? POLLOUT : POLLIN), 0});
    int rc = poll(pfds.data(), pfds.size(), 1000);
    if (rc < 0 && errno != EINTR) break;
    int64_t now = monotonic_us();
    if (pfds[0].revents & POLLIN) {
      while (true) {
        int fd = accept(listen_fd_, nullptr, nullptr);
        if (fd < 0) break;
        set_nonblock(fd);
        int one = 1;
        setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
        Conn c;
        c.fd = fd;
        c.deadline_us = now + (int64_t)timeout_ms_ * 1000;
        if (conns.size() >= 256) {
          close(fd);
          continue;
        }
        conns.push_back(std::move(c));
      }
    }
    size_t npolled = pfds.size() - 2;
    for (size_t i = 0; i < conns.size();) {
      Conn& c = conns[i];
      bool drop = false;
      short rev = i < npolled ? pfds[i + 2].revents : 0;
      if (rev & (POLLERR | POLLHUP | POLLNVAL)) drop = !(rev & POLLIN);
      if (!drop && (rev & POLLIN)) {
        char buf[8192];
        ssize_t n = read(c.fd, buf, sizeof buf);
        if (n == 0) drop = true;
        else if (n < 0) drop = errno != EAGAIN && errno != EINTR;
        else {
          c.in.append(buf, (size_t)n);
          c.deadline_us = now + (int64_t)timeout_ms_ * 1000;
          if (!handle_input(c)) drop = true;
        }
      }
      if (!drop && c.out_off < c.out.size()) {
        ssize_t n = send(c.fd, c.out.data() + c.out_off, c.out.size() - c.out_off, MSG_NOSIGNAL);
        if (n > 0) {
          c.out_off += (size_t)n;
          c.deadline_us = now + (int64_t)timeout_ms_ * 1000;
        } else if (n < 0 && errno != EAGAIN && errno != EINTR) {
          drop = true;
        }
        if (c.out_off == c.out.size()) {
          c.out.clear();
          c.out_off = 0;
          if (c.close_after) drop = true;
        }
      }
      if (!drop && now > c.deadline_us) drop = true;
      if (drop) {
        close(c.fd);
        conns.erase(conns.begin() + (long)i);
        if (i < npolled) {
          pfds.erase(pfds.begin() + (long)(i + 2));
          npolled--;
        }
      } else {
        i++;
      }
    }
  }
  for (auto& c : conns) close(c.fd);
}

}  // namespace gpumon
