# Pod attribution sidecar: watches /run/prometheus/dcgm.prom, adds pod labels from
# the kubelet PodResources API, serves :9400/gpu/metrics.  Needs no GPU access:
# the build stage uses the ROCm image only for the amdsmi.h header that
# libgpumon's device backend compiles against; the device library itself is
# loaded lazily and never touched by this binary, so the runtime stage is plain.
#   docker build -f deploy/docker/pod-exporter.Dockerfile -t gpumon/pod-exporter .
FROM rocm/dev-ubuntu-22.04:7.2 AS build
RUN apt-get update && apt-get install -y --no-install-recommends make g++ && rm -rf /var/lib/apt/lists/*
WORKDIR /src
COPY Makefile ./
COPY include include
COPY core core
COPY cmd cmd
RUN make -j"$(nproc)" all

FROM ubuntu:22.04
COPY --from=build /src/build/libgpumon.so /usr/local/lib/
COPY --from=build /src/build/bin/gpumon-pod-exporter /usr/local/bin/
RUN ldconfig
EXPOSE 9400
ENTRYPOINT ["gpumon-pod-exporter", "-logtostderr", "-v", "1"]
