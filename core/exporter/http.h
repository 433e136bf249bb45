// Minimal dependency-free HTTP/1.1 server: one poll()-driven thread, keep-alive,
// GET/HEAD only, per-connection read/write timeouts.  Serves /metrics,
// /gpu/metrics, /healthz and the REST routes; replaces the Go net/http servers
// of the reference (bindings/go/samples/dcgm/restApi/server.go:19-88,
// exporters/prometheus-dcgm/k8s/pod-gpu-metrics-exporter/src/http.go:11-52).
#pragma once

#include <atomic>
#include <functional>
#include <map>
#include <string>
#include <thread>
#include <vector>

namespace gpumon {

struct HttpRequest {
  std::string method, path, query;
  std::map<std::string, std::string> headers;  // lower-cased names
};

struct HttpResponse {
  int status = 200;
  std::string content_type = "text/plain; charset=utf-8";
  std::string body;
};

using HttpHandler = std::function<void(const HttpRequest&, HttpResponse*)>;

class HttpServer {
 public:
  HttpServer();
  ~HttpServer();
  // address: "host:port", ":port" or "port"; port 0 picks a free port.
  bool start(const std::string& address, HttpHandler handler, int timeout_ms = 10000);
  void stop();
  int port() const { return port_; }
  uint64_t requests() const { return requests_.load(); }

 private:
  struct Conn {
    int fd;
    std::string in, out;
    size_t out_off = 0;
    int64_t deadline_us;
    bool close_after = false;
  };
  void loop();
  bool handle_input(Conn& c);

  int listen_fd_ = -1, wake_rd_ = -1, wake_wr_ = -1;
  int port_ = 0;
  int timeout_ms_ = 10000;
  HttpHandler handler_;
  std::atomic<bool> running_{false};
  std::atomic<uint64_t> requests_{0};
  std::thread th_;
};

const char* http_status_text(int code);

}  // namespace gpumon
