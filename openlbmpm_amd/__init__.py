"""openlbmpm_amd -- MI355X-native (gfx950) replacement for the GPU collision-streaming
path of PorousMediaSimulation/openLBMPM.  Compute lives in liblbmpm_hip.so (hand-written
HIP, C ABI in include/lbmpm.h); this package is the thin host side."""
__version__ = "0.1.0"
