// Pod attribution: device-id -> {pod, namespace, container} maps and the text
// rewrite that appends pod labels to exposition samples.  Semantics follow the
// reference's rewriter (exporters/prometheus-dcgm/k8s/pod-gpu-metrics-exporter/
// src/device_pod.go:26-113): a sample is attributed when its uuid (2nd label)
// or "<prefix><gpu index>" (1st label) is a device id a pod holds.  Unlike the
// reference, labels are parsed (quotes and escapes honoured) instead of split
// on "," / "}" characters, and a sample is never labelled twice.
#include <algorithm>

#include "../common/json.h"
#include "../common/util.h"
#include "exporter.h"

namespace gpumon {

PodMap parse_pod_map_tsv(const std::string& tsv) {
  PodMap m;
  for (auto& line : split(tsv, '\n')) {
    if (line.empty() || line[0] == '#') continue;
    auto f = split(line, '\t');
    if (f.size() < 4 || f[0].empty()) continue;
    m[f[0]] = PodInfo{f[1], f[2], f[3]};
  }
  return m;
}

std::string pod_map_to_tsv(const PodMap& m) {
  std::string out;
  for (auto& kv : m) out += kv.first + "\t" + kv.second.pod + "\t" + kv.second.ns + "\t" + kv.second.container + "\n";
  return out;
}

static bool resource_wanted(const std::string& name, const std::string& wanted_csv) {
  for (auto& w : split(wanted_csv, ','))
    if (trim(w) == name) return true;
  return false;
}

// kubelet_internal_checkpoint: {"Data":{"PodDeviceEntries":[{"PodUID","ContainerName",
// "ResourceName","DeviceIDs": [ids] | {"<numa>":[ids]}}]}}.  It knows pod UIDs,
// not names, so the UID stands in for the pod name and the namespace is empty.
bool parse_device_plugin_checkpoint(const std::string& json, const std::string& resource_names, PodMap* out) {
  Json root;
  if (!json_parse(json, &root)) return false;
  const Json* data = root.get("Data");
  const Json* entries = data ? data->get("PodDeviceEntries") : nullptr;
  if (!entries || entries->type != Json::ARR) return false;
  for (auto& e : entries->arr) {
    const Json* res = e.get("ResourceName");
    const Json* uid = e.get("PodUID");
    const Json* cont = e.get("ContainerName");
    const Json* ids = e.get("DeviceIDs");
    if (!res || !uid || !ids || res->type != Json::STR || !resource_wanted(res->str, resource_names)) continue;
    PodInfo info{uid->str, "", cont && cont->type == Json::STR ? cont->str : ""};
    auto take = [&](const Json& arr) {
      for (auto& id : arr.arr)
        if (id.type == Json::STR) (*out)[id.str] = info;
    };
    if (ids->type == Json::ARR) take(*ids);
    else if (ids->type == Json::OBJ)
      for (auto& kv : ids->obj)
        if (kv.second.type == Json::ARR) take(kv.second);
  }
  return true;
}

namespace {

// Parses `{name="value",...}` starting at s[pos] == '{'.  Returns the offset of
// the closing brace or npos; fills the label values in order.
size_t parse_labels(const std::string& s, size_t pos, std::vector<std::pair<std::string, std::string>>* labels) {
  size_t p = pos + 1;
  while (p < s.size()) {
    while (p < s.size() && (s[p] == ' ' || s[p] == ',')) p++;
    if (p < s.size() && s[p] == '}') return p;
    size_t eq = s.find('=', p);
    if (eq == std::string::npos || eq + 1 >= s.size() || s[eq + 1] != '"') return std::string::npos;
    std::string name = s.substr(p, eq - p), value;
    p = eq + 2;
    while (p < s.size() && s[p] != '"') {
      if (s[p] == '\\' && p + 1 < s.size()) {
        char e = s[p + 1];
        value += e == 'n' ? '\n' : e;
        p += 2;
      } else {
        value += s[p++];
      }
    }
    if (p >= s.size()) return std::string::npos;
    p++;  // closing quote
    labels->emplace_back(name, value);
  }
  return std::string::npos;
}

std::string escape_label(const std::string& v) {
  std::string o;
  for (char c : v) {
    if (c == '\\') o += "\\\\";
    else if (c == '"') o += "\\\"";
    else if (c == '\n') o += "\\n";
    else o += c;
  }
  return o;
}

}  // namespace

std::string relabel_text(const std::string& text, const PodMap& m, const std::string& index_prefix) {
  std::string out;
  out.reserve(text.size() + text.size() / 2);
  size_t start = 0;
  std::vector<std::pair<std::string, std::string>> labels;
  while (start < text.size()) {
    size_t nl = text.find('\n', start);
    size_t end = nl == std::string::npos ? text.size() : nl;
    std::string line = text.substr(start, end - start);
    start = nl == std::string::npos ? text.size() : nl + 1;
    size_t brace = line.find('{');
    if (!line.empty() && line[0] != '#' && brace != std::string::npos) {
      labels.clear();
      size_t close = parse_labels(line, brace, &labels);
      if (close != std::string::npos && labels.size() >= 2) {
        bool already = false;
        for (auto& l : labels) already = already || l.first == "pod_name";
        const PodInfo* p = nullptr;
        auto it = m.find(labels[1].second);  // uuid is the second label by contract
        if (it != m.end()) p = &it->second;
        if (!p && (it = m.find(index_prefix + labels[0].second)) != m.end()) p = &it->second;
        if (p && !already) {
          std::string add = ",pod_name=\"" + escape_label(p->pod) + "\",pod_namespace=\"" + escape_label(p->ns) +
                            "\",container_name=\"" + escape_label(p->container) + "\"";
          line.insert(close, add);
        }
      }
    }
    out += line;
    if (nl != std::string::npos) out += '\n';
  }
  return out;
}

}  // namespace gpumon
