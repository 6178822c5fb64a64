"""islam_amd: MI355X (gfx950) implementation of iSLAM's bilevel hot path behind the reference's own
Python call surface (TartanVO / run_pvgo / IMUModule).  Compute lives in islam_amd/lib/libislam_hip.so
(hand-written HIP, C ABI in include/islam_hip.h); there is no CPU fallback."""
__version__ = '0.1.0'
