module github.com/k8s-gpu-monitor-amd/gpumon/bindings/go

go 1.21
