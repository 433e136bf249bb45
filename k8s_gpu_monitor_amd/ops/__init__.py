"""Active diagnostics: ctypes front ends of the HIP kernels in diag/hip."""
