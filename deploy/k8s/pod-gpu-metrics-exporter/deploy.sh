#!/bin/bash
# Builds and pushes both images; REGISTRY defaults to the local daemon only.
set -euo pipefail
cd "$(dirname "$0")/../../.."
REGISTRY=${REGISTRY:-}
TAG=${TAG:-latest}
docker build -f deploy/docker/gpumon-exporter.Dockerfile -t "${REGISTRY}gpumon/exporter:${TAG}" .
docker build -f deploy/docker/pod-exporter.Dockerfile -t "${REGISTRY}gpumon/pod-exporter:${TAG}" .
if [ -n "$REGISTRY" ]; then
  docker push "${REGISTRY}gpumon/exporter:${TAG}"
  docker push "${REGISTRY}gpumon/pod-exporter:${TAG}"
fi
