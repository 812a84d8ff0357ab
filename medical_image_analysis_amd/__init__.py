"""medical_image_analysis_amd -- MI355X (gfx950) native hot path of MambaXray-VL.

The package is a thin host-side mirror of the reference's operator / module surface
(`selective_scan_fn`, `causal_conv1d_fn`, `mamba_inner_fn[_no_out_proj]`, `Mamba`, `ARM`,
`VisionMamba` ...) over a C-ABI shared library of hand-written HIP kernels
(csrc/*.hip -> libmxvl.so, declared in include/mxvl.h).  There is NO CPU or PyTorch fallback:
every operator raises if libmxvl.so is missing or the tensors are not on a HIP device.
"""
__version__ = "0.1.0"
