// Field registry: the single table that ties an AMD-native field id to its
// type, its backend data source and, for exported fields, the Prometheus
// family (name / type / HELP) it renders as.
//
// The 36 + 5 families of the reference's exporter keep their exact names, types
// and HELP text (exporters/prometheus-dcgm/dcgm-exporter/dcgm-exporter:121-184);
// AMD-only families are added under gpumon_* names.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "gpumon.h"

namespace gpumon {

// Backend data sources.  One backend call fills every field of a source, so the
// sampler asks for sources, not fields: a tick that needs twenty fields of
// SRC_METRICS costs one amdsmi_get_gpu_metrics_info call.
enum Source : uint32_t {
  SRC_STATIC = 1u << 0,    // identity / limits, read once
  SRC_METRICS = 1u << 1,   // the bulk gpu-metrics table
  SRC_VRAM = 1u << 2,      // VRAM / visible VRAM / GTT usage
  SRC_ECC = 1u << 3,       // RAS error counts (total + per block)
  SRC_BAD_PAGES = 1u << 4, // retired-page table
  SRC_ENERGY = 1u << 5,    // energy accumulator
  SRC_PERF = 1u << 6,      // DPM performance level, power cap
  SRC_VIOLATION = 1u << 7, // throttle residency / violation status
  SRC_EVENTS = 1u << 8,    // engine-derived from the driver event stream
  SRC_PROF = 1u << 9,      // profiling-class utilisation (-p)
  SRC_PROCS = 1u << 10,    // process list (accounting only, never a field)
};

struct FieldDef {
  uint16_t id;
  uint8_t type;        // gm_field_type_t
  uint32_t source;     // exactly one Source bit
  const char* tag;
  const char* unit;
  uint8_t prom_type;   // 0 none, 1 gauge, 2 counter
  uint8_t prom_class;  // 0 contract, 1 profiling, 2 AMD extras
  const char* prom_name;
  const char* prom_help;
  // Extras that are one member of a labelled family ("link", "stack", "xcd").
  const char* extra_label;  // label name or nullptr
  int extra_index;          // label value
};

// Dense slot index of a field id, or -1.  Slots index every per-GPU value array.
int field_slot(uint16_t id);
const FieldDef* field_def(uint16_t id);
const FieldDef& field_at(int slot);
int field_count();
const FieldDef* field_by_tag(const std::string& tag);

// Ordered field ids of the exporter's contract set (order of the reference's
// awk program), the profiling set and the AMD extras.
const std::vector<uint16_t>& contract_fields();
const std::vector<uint16_t>& profiling_fields();
const std::vector<uint16_t>& extra_fields();

}  // namespace gpumon
