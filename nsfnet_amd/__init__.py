"""nsfnet_amd - MI355X-native PINN training engine for the lid-driven-cavity NSFnet /
ev-NSFnet workload.  The per-step hot path is a hand-written HIP (gfx950) pipeline
behind the C ABI of include/nsfnet_pinn.h; this package is the Python host side that
mirrors the reference's pinn_solver / net class surface."""

__version__ = "0.1.0"
