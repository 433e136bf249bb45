# Active diagnostics image: gpumon-diag with the gfx950 kernels (HBM stream, memtest, MFMA / GEMM stress, xGMI link
# test, RCCL sweeps).  hipcc cross-compiles the kernels in the build stage; no GPU is needed to build.
# Build context: the repository root.
#   docker build -f deploy/docker/gpumon-diag.Dockerfile -t gpumon/diag .
#   docker run --device /dev/kfd --device /dev/dri --group-add video --group-add render \
#              -v /var/lib/gpumon/diag:/var/lib/gpumon/diag gpumon/diag
# librccl.so and libamd_smi.so come from the ROCm base image and are loaded at run time.
FROM rocm/dev-ubuntu-22.04:7.2-complete AS build
RUN apt-get update && apt-get install -y --no-install-recommends make g++ && rm -rf /var/lib/apt/lists/*
WORKDIR /src
COPY Makefile ./
COPY include include
COPY core core
COPY cmd cmd
COPY diag diag
RUN make -j"$(nproc)" all diag

FROM rocm/dev-ubuntu-22.04:7.2-complete
COPY --from=build /src/build/libgpumon.so /src/build/libgpumon_diag.so /usr/local/lib/
COPY --from=build /src/build/bin/gpumon-diag /src/build/bin/gpumon /usr/local/bin/
COPY deploy/docker/gpumon-diag-all.sh /usr/local/bin/gpumon-diag-all
RUN ldconfig && mkdir -p /var/lib/gpumon/diag
VOLUME /var/lib/gpumon/diag
# Level 2 (HBM stream, memtest over most of the HBM, MFMA peak and a short GEMM stress) on every GPU the container
# sees, one after the other, then the xGMI link test and an RCCL sweep when there is more than one
# (GPUMON_DIAG_LEVEL / GPUMON_DIAG_DIR / GPUMON_DIAG_SUBMIT, see the script).  The reports are what
# `gpumon-exporter --diag-results /var/lib/gpumon/diag` turns into gpumon_diag_* metrics and into health incidents;
# the exit code is 0 only when every test passed.
ENTRYPOINT ["gpumon-diag-all"]
