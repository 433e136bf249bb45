package hostengine

// Names of the reference's dcgm package for the types that carry an AMD-neutral name here,
// so code written against it keeps compiling: DcgmStatus (bindings/go/dcgm/
// hostengine_status.go:13-16) and XIDErrorInfo (process_info.go:46-49).
type (
	DcgmStatus   = Status
	XIDErrorInfo = CriticalEvents
)

// The reference's policy condition names (bindings/go/dcgm/policy.go:23-31) for the conditions that carry an AMD
// name here: double-bit ECC = uncorrectable ECC, retired pages = bad pages, NVLink = xGMI, XID = GPU event.
const (
	DbePolicy     = EccUePolicy
	MaxRtPgPolicy = MaxBadPgPolicy
	NvlinkPolicy  = XgmiPolicy
	XidPolicy     = GpuEventPolicy
)
