// Text and JSON renderings of engine objects, shared by the REST routes and the
// CLI tools.  Layouts follow the reference's text/template blocks with AMD
// vocabulary (bindings/go/samples/dcgm/restApi/handlers/utils.go:19-92,
// bindings/go/samples/dcgm/deviceInfo/main.go:13-33, topology/main.go:11-71).
#pragma once

#include <string>

#include "../engine/engine.h"

namespace gpumon {

gm_status_t view_device_info(Engine* e, uint32_t gpu, bool json, std::string* out);
gm_status_t view_device_status(Engine* e, uint32_t gpu, bool json, std::string* out);
gm_status_t view_health(Engine* e, uint32_t gpu, bool json, std::string* out);
gm_status_t view_process_info(Engine* e, uint32_t pid, bool json, std::string* out);
gm_status_t view_hostengine(Engine* e, bool json, std::string* out);
gm_status_t view_topology_matrix(Engine* e, std::string* out);
// Per-link table of one GPU: state, far end, rate, traffic (the reference's NVLink walk, bindings/go/nvml/bindings.go:268-318).
gm_status_t view_links(Engine* e, uint32_t gpu, bool json, std::string* out);
// Latest verdict of every active diagnostic that has reported for the GPU.
gm_status_t view_diag_results(Engine* e, uint32_t gpu, bool json, std::string* out);
std::string view_violation(const gm_violation_t& v);
std::string format_time_us(int64_t ts_us);  // "Running" for 0

}  // namespace gpumon
