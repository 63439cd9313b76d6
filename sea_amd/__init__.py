"""sea_amd: the SEA temporal-rollout hot path on MI355X — HIP kernels behind a C ABI (csrc/, include/sea_hip.h) and the host-side mirror of the reference modules."""
