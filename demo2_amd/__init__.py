"""demo2_amd: MI355X-native (gfx950) implementation of DeMo's PIFE -> HDM -> ATMoE hot path.

Host side in Python mirroring the reference's `make_model` / `forward` surface; compute in hand-written HIP
kernels behind the C ABI of include/demo_hip.h (demo2_amd/lib/libdemo_hip.so)."""
__all__ = ["ops", "modeling", "config"]
