// REST routes.  Same URL scheme, status codes and text-or-JSON selection as the
// reference's sample server (bindings/go/samples/dcgm/restApi/server.go:40-71,
// handlers/utils.go:95-156): ids that do not parse as uint32 -> 400, unknown id
// or uuid -> 404, unsupported GPU -> 500, a trailing "/json" selects JSON.
// Every route is also served under the vendor-neutral /gpumon prefix.
#include "../common/util.h"
#include "exporter.h"
#include "views.h"

namespace gpumon {
namespace {

bool parse_u32(const std::string& s, uint32_t* out) {
  if (s.empty() || s.size() > 10) return false;
  uint64_t v = 0;
  for (char c : s) {
    if (c < '0' || c > '9') return false;
    v = v * 10 + (uint64_t)(c - '0');
  }
  if (v > 0xffffffffull) return false;
  *out = (uint32_t)v;
  return true;
}

void fail(HttpResponse* r, int code, const std::string& msg) {
  r->status = code;
  r->body = msg + "\n";
}

// Resolves {id|uuid} to a GPU index; writes the error response itself.
bool resolve_gpu(Engine* e, const std::string& kind, const std::string& key, HttpResponse* resp, uint32_t* gpu) {
  uint32_t n = 0;
  if (e->device_count(&n) != GM_ST_OK) {
    fail(resp, 500, "engine unavailable");
    return false;
  }
  if (kind == "id") {
    if (!parse_u32(key, gpu)) {
      fail(resp, 400, "strconv.ParseUint: parsing \"" + key + "\": invalid syntax");
      return false;
    }
    if (*gpu >= n) {
      fail(resp, 404, "404 page not found");
      return false;
    }
  } else {
    bool found = false;
    for (uint32_t i = 0; i < n && !found; i++) {
      gm_device_attributes_t a;
      a.version = GM_DEVICE_ATTRIBUTES_VERSION;
      if (e->device_attributes(i, &a) == GM_ST_OK && key == a.uuid) {
        *gpu = i;
        found = true;
      }
    }
    if (!found) {
      fail(resp, 404, "404 page not found");
      return false;
    }
  }
  std::vector<uint32_t> sup;
  e->supported_devices(&sup);
  bool ok = false;
  for (uint32_t s : sup) ok = ok || s == *gpu;
  if (!ok) {
    fail(resp, 500, strprintf("Error adding GPU %u to group: This GPU is not supported by the engine", *gpu));
    return false;
  }
  return true;
}

}  // namespace

bool handle_rest(Engine* e, const HttpRequest& req, HttpResponse* resp) {
  std::vector<std::string> seg;
  for (auto& s : split(req.path, '/'))
    if (!s.empty()) seg.push_back(s);
  if (seg.empty() || (seg[0] != "dcgm" && seg[0] != "gpumon")) return false;
  bool json = seg.size() > 1 && seg.back() == "json";
  if (json) seg.pop_back();
  std::string body;
  gm_status_t st = GM_ST_OK;
  uint32_t gpu = 0;

  if (seg.size() == 5 && seg[1] == "device" && (seg[2] == "info" || seg[2] == "status") &&
      (seg[3] == "id" || seg[3] == "uuid")) {
    if (!resolve_gpu(e, seg[3], seg[4], resp, &gpu)) return true;
    st = seg[2] == "info" ? view_device_info(e, gpu, json, &body) : view_device_status(e, gpu, json, &body);
  } else if (seg.size() == 4 && seg[1] == "health" && (seg[2] == "id" || seg[2] == "uuid")) {
    if (!resolve_gpu(e, seg[2], seg[3], resp, &gpu)) return true;
    st = view_health(e, gpu, json, &body);
  } else if (seg.size() == 5 && seg[1] == "process" && seg[2] == "info" && seg[3] == "pid") {
    uint32_t pid;
    if (!parse_u32(seg[4], &pid)) {
      fail(resp, 400, "strconv.ParseUint: parsing \"" + seg[4] + "\": invalid syntax");
      return true;
    }
    st = view_process_info(e, pid, json, &body);
    if (st == GM_ST_NO_DATA) {
      fail(resp, 404, strprintf("no GPU activity recorded for pid %u", pid));
      return true;
    }
  } else if (seg.size() == 4 && (seg[1] == "links" || seg[1] == "diag") && (seg[2] == "id" || seg[2] == "uuid")) {
    if (!resolve_gpu(e, seg[2], seg[3], resp, &gpu)) return true;
    st = seg[1] == "links" ? view_links(e, gpu, json, &body) : view_diag_results(e, gpu, json, &body);
  } else if (seg.size() == 2 && seg[1] == "status") {
    st = view_hostengine(e, json, &body);
  } else if (seg.size() == 2 && seg[1] == "topology") {
    st = view_topology_matrix(e, &body);
  } else {
    return false;
  }
  if (st != GM_ST_OK) {
    fail(resp, 500, status_string(st));
    return true;
  }
  resp->status = 200;
  resp->content_type = json ? "application/json" : "text/plain; charset=utf-8";
  resp->body = body;
  return true;
}

}  // namespace gpumon
