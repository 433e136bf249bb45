// ReplicateBackend: every real GPU shown N times.
//
// A measuring aid, not a product feature: on a host with one GPU it answers "what does monitoring an 8-GPU node
// cost with the REAL device library?" — every replica's sample is a full round through libamd_smi.so to the real
// device, so the sampler, the cache and the exporter carry N label sets at the true per-call price.  Replica k of
// GPU g has index g + k * real_count, a UUID with the suffix "-r<k>" and PCI domain k.
//   GPUMON_BACKEND=amdsmi GPUMON_BACKEND_SPEC=replicate=8[,lib=/path/libamd_smi.so]
#include "../common/util.h"
#include "backend.h"

namespace gpumon {
namespace {

class ReplicateBackend : public Backend {
 public:
  ReplicateBackend(std::unique_ptr<Backend> inner, uint32_t times) : inner_(std::move(inner)), times_(times ? times : 1) {}
  const char* name() const override { return inner_->name(); }
  gm_status_t init() override {
    gm_status_t st = inner_->init();
    if (st == GM_ST_OK) real_ = inner_->device_count();
    return st;
  }
  void shutdown() override { inner_->shutdown(); }
  uint32_t device_count() override { return std::min<uint32_t>(real_ * times_, GM_MAX_GPUS); }
  gm_status_t static_info(uint32_t gpu, GpuStatic* out) override {
    if (!real_ || gpu >= device_count()) return GM_ST_BADPARAM;
    gm_status_t st = inner_->static_info(gpu % real_, out);
    if (st != GM_ST_OK) return st;
    const uint32_t k = gpu / real_;
    out->index = gpu;
    if (k) {
      out->uuid += strprintf("-r%u", k);
      if (out->pci_bdf.size() > 4) out->pci_bdf = strprintf("%04x", k) + out->pci_bdf.substr(4);
    }
    return GM_ST_OK;
  }
  gm_status_t sample(uint32_t gpu, uint32_t sources, Sample* out) override {
    if (!real_ || gpu >= device_count()) return GM_ST_BADPARAM;
    return inner_->sample(gpu % real_, sources, out);
  }
  gm_status_t processes(uint32_t gpu, std::vector<ProcSample>* out) override {
    if (!real_ || gpu >= device_count()) return GM_ST_BADPARAM;
    return inner_->processes(gpu % real_, out);
  }
  const char* process_source(uint32_t gpu) override { return real_ ? inner_->process_source(gpu % real_) : "none"; }
  std::string metrics_table_source(uint32_t gpu) override { return real_ ? inner_->metrics_table_source(gpu % real_) : "n/a"; }
  gm_status_t peer_link(uint32_t gpu, uint32_t peer, PeerLink* out) override {
    if (!real_ || gpu >= device_count() || peer >= device_count() || gpu == peer) return GM_ST_BADPARAM;
    if (gpu % real_ == peer % real_) return GM_ST_NOT_SUPPORTED;  // a GPU and its own replica
    return inner_->peer_link(gpu % real_, peer % real_, out);
  }
  gm_status_t xgmi_links(uint32_t gpu, std::vector<XgmiLink>* out) override {
    if (!real_ || gpu >= device_count()) return GM_ST_BADPARAM;
    return inner_->xgmi_links(gpu % real_, out);
  }
  gm_status_t poll_events(int timeout_ms, std::vector<DeviceEvent>* out) override { return inner_->poll_events(timeout_ms, out); }
  bool delivers_ras_events() override { return inner_->delivers_ras_events(); }
  gm_status_t control(const std::string& command) override { return inner_->control(command); }
  void refresh_handles() override { inner_->refresh_handles(); }
  gm_status_t runtime_version(uint32_t* major, uint32_t* minor, std::string* text) override {
    return inner_->runtime_version(major, minor, text);
  }

 private:
  std::unique_ptr<Backend> inner_;
  uint32_t times_, real_ = 0;
};

}  // namespace

std::unique_ptr<Backend> make_replicate_backend(std::unique_ptr<Backend> inner, uint32_t times) {
  return std::unique_ptr<Backend>(new ReplicateBackend(std::move(inner), times));
}

}  // namespace gpumon
