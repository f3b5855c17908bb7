"""Small host-side helpers of the boundary (no kernels here)."""
