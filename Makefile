# Build of the MI355X-native GPU monitoring stack.
#   make            -> build/libgpumon.so, build/bin/{gpumon,gpumon-exporter,gpumon-hostengine,gpumon-pod-exporter}
#   make diag       -> build/libgpumon_diag.so (HIP kernels, --offload-arch=gfx950) + build/bin/gpumon-diag
#   make prof       -> build/libgpumon_prof.so (device-wide CU / MFMA counters, optional) + build/bin/gpumon-prof-probe
#   make tsan|asan  -> sanitizer builds of the core + C++ tests (CPU, fake backend)
#   make test       -> C++ unit tests
ROCM      ?= /opt/rocm
CXX       ?= g++
HIPCC     ?= $(ROCM)/bin/hipcc
ARCH      ?= gfx950
BUILD     ?= build
SAN       ?=
OPT       ?= -O2
CXXFLAGS  += -std=c++17 $(OPT) -g -fPIC -Wall -Wextra -Wno-unused-parameter -Wno-missing-field-initializers \
             -fvisibility=hidden -pthread -Iinclude -I$(ROCM)/include $(SAN)
LDFLAGS   += -pthread -ldl $(SAN)

CORE_SRC := core/common/util.cc core/common/json.cc core/fields.cc \
            core/backend/fake_backend.cc core/backend/amdsmi_backend.cc core/backend/replicate_backend.cc core/backend/procscan.cc core/backend/metrics_table.cc \
            core/engine/engine.cc core/engine/health.cc core/engine/policy.cc core/engine/accounting.cc \
            core/rpc/wire.cc core/rpc/server.cc core/rpc/client.cc \
            core/exporter/http.cc core/exporter/exporter.cc core/exporter/podmap.cc core/exporter/kubelet.cc core/exporter/diag_results.cc core/prof/prof_child.cc \
            core/exporter/views.cc core/exporter/rest.cc core/exporter/pod_exporter.cc core/capi.cc core/smi.cc
CORE_OBJ := $(CORE_SRC:%.cc=$(BUILD)/obj/%.o)
BINS     := gpumon gpumon-exporter gpumon-hostengine gpumon-pod-exporter
DIAG_HIP := diag/hip/hbm_stream.hip diag/hip/memtest.hip diag/hip/mfma_gemm.hip diag/hip/diag_api.hip \
            diag/rccl/rccl_sweep.hip diag/hip/link_test.hip diag/hip/pcie_test.hip
DIAG_OBJ := $(DIAG_HIP:%.hip=$(BUILD)/obj/%.o)
HIPFLAGS += --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Iinclude -Idiag -Wno-unused-result -Wno-inline-asm

.PHONY: all core bins diag prof test tsan asan clean
all: core bins
core: $(BUILD)/libgpumon.so
bins: $(BINS:%=$(BUILD)/bin/%)

$(BUILD)/obj/%.o: %.cc
	@mkdir -p $(dir $@)
	$(CXX) $(CXXFLAGS) -MMD -MP -c $< -o $@

$(BUILD)/libgpumon.so: $(CORE_OBJ)
	$(CXX) -shared -o $@ $^ $(LDFLAGS)

$(BUILD)/bin/%: cmd/%.cc cmd/cli_common.h include/gpumon.h $(BUILD)/libgpumon.so
	@mkdir -p $(dir $@)
	$(CXX) $(CXXFLAGS) -fvisibility=default -o $@ $< -L$(BUILD) -lgpumon -Wl,-rpath,'$$ORIGIN/..' $(LDFLAGS)

# ---- active diagnostics: hand-written CDNA4 kernels --------------------------
$(BUILD)/obj/%.o: %.hip
	@mkdir -p $(dir $@)
	$(HIPCC) $(HIPFLAGS) -MMD -MP -c $< -o $@

diag: $(BUILD)/libgpumon_diag.so $(BUILD)/bin/gpumon-diag
$(BUILD)/libgpumon_diag.so: $(DIAG_OBJ)
	$(HIPCC) --offload-arch=$(ARCH) -shared -o $@ $^ -ldl

$(BUILD)/bin/gpumon-diag: cmd/gpumon-diag.cc cmd/cli_common.h cmd/diag_expect.h diag/gpumon_diag.h $(BUILD)/libgpumon_diag.so $(BUILD)/libgpumon.so
	@mkdir -p $(dir $@)
	$(CXX) $(CXXFLAGS) -fvisibility=default -o $@ $< -L$(BUILD) -lgpumon_diag -lgpumon \
	    -Wl,-rpath,'$$ORIGIN/..' -Wl,-rpath,$(ROCM)/lib -Wl,-rpath-link,$(ROCM)/lib $(LDFLAGS)

# ---- optional profiling-counter plugin (rocprofiler-sdk device counting + HSA) -----
# Loaded by libgpumon.so with dlopen when GPUMON_PROF_COUNTERS=1; nothing links against it.
prof: $(BUILD)/libgpumon_prof.so $(BUILD)/bin/gpumon-prof-probe
$(BUILD)/libgpumon_prof.so: core/prof/prof_counters.cc core/prof/prof_counters.h
	@mkdir -p $(BUILD)
	$(CXX) $(CXXFLAGS) -D__HIP_PLATFORM_AMD__ -Wno-deprecated-declarations -shared -o $@ $< -L$(ROCM)/lib -lrocprofiler-sdk -lhsa-runtime64 -Wl,-rpath,$(ROCM)/lib $(LDFLAGS)
$(BUILD)/bin/gpumon-prof-probe: cmd/gpumon-prof-probe.cc core/prof/prof_counters.h
	@mkdir -p $(dir $@)
	$(CXX) $(CXXFLAGS) -o $@ $< $(LDFLAGS)

# Timing-only ablation builds of the GEMM pipeline (fragment reads / DMA / MFMAs
# removed one at a time); used by scripts/gpu_gemm_ablate.py, never by the product.
diag-ablate: $(BUILD)/libgpumon_diag_ablate.so
$(BUILD)/libgpumon_diag_ablate.so: $(DIAG_HIP) diag/gpumon_diag.h diag/hip/diag_common.h
	@mkdir -p $(BUILD)
	$(HIPCC) $(HIPFLAGS) -DGMD_GEMM_ABLATE -shared -o $@ $(DIAG_HIP) -ldl

# ---- tests -------------------------------------------------------------------
TESTS := $(patsubst tests/cpp/%.cc,$(BUILD)/tests/%,$(wildcard tests/cpp/*.cc))
$(BUILD)/tests/%: tests/cpp/%.cc $(BUILD)/libgpumon.so
	@mkdir -p $(dir $@)
	$(CXX) $(CXXFLAGS) -fvisibility=default -I. -o $@ $< $(CORE_OBJ) $(LDFLAGS)
test: $(TESTS)
	@for t in $(TESTS); do echo "== $$t"; $$t || exit 1; done

tsan:
	$(MAKE) BUILD=build-tsan SAN="-fsanitize=thread" OPT=-O1 test
asan:
	$(MAKE) BUILD=build-asan SAN="-fsanitize=address,undefined -fno-omit-frame-pointer" OPT=-O1 test

clean:
	rm -rf build build-tsan build-asan

-include $(CORE_OBJ:.o=.d) $(DIAG_OBJ:.o=.d)
