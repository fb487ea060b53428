"""HIP sources of libpantheon_hip.so and the in-tree build script (`python -m pantheonrl_amd.csrc.build`)."""
