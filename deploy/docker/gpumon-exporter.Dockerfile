# Node exporter image: gpumon-exporter with the engine embedded (-e).
# Build context: the repository root.
#   docker build -f deploy/docker/gpumon-exporter.Dockerfile -t gpumon/exporter .
#   docker run --device /dev/kfd --device /dev/dri -v /run/prometheus:/run/prometheus gpumon/exporter
# libamd_smi.so comes from the ROCm base image and is loaded at run time; the
# build itself needs neither a GPU nor the driver.
FROM rocm/dev-ubuntu-22.04:7.2 AS build
RUN apt-get update && apt-get install -y --no-install-recommends make g++ && rm -rf /var/lib/apt/lists/*
WORKDIR /src
COPY Makefile ./
COPY include include
COPY core core
COPY cmd cmd
RUN make -j"$(nproc)" all prof

FROM rocm/dev-ubuntu-22.04:7.2
COPY --from=build /src/build/libgpumon.so /src/build/libgpumon_prof.so /usr/local/lib/
# gpumon-prof-probe is what --prof-counters=child starts (found next to the library: ../bin)
COPY --from=build /src/build/bin/gpumon /src/build/bin/gpumon-exporter /src/build/bin/gpumon-hostengine /src/build/bin/gpumon-prof-probe /usr/local/bin/
RUN ldconfig
VOLUME /run/prometheus
EXPOSE 9400
# Same default behaviour as the reference image: collect every second into
# /run/prometheus/dcgm.prom with the engine in-process; also serve :9400/metrics.
# Add "-p" for the profiling metrics and "--prof-counters" for MFMA pipe activity from hardware counters
# (libgpumon_prof.so, next to libgpumon.so, is only loaded then).
ENTRYPOINT ["gpumon-exporter", "-e", "--listen", ":9400"]
