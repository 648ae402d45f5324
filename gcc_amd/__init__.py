"""gcc_amd: MI355X-native GAN-compression (GCC) training hot path.

HIP/CDNA4 kernels behind a C ABI (include/gcc_hip.h, gcc_amd/csrc) + the host-side mirror of the
reference's model surface (gcc_amd/models, options, train).  See DESIGN.md."""
__version__ = '0.1.0'
