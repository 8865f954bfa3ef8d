"""multimodal_amd - MI355X-native hot path of the NeoX/MAGMA ParallelTransformerLayer.

`multimodal_amd.mpu`, `multimodal_amd.model` and `multimodal_amd.fused_kernels` mirror the reference's
`megatron.mpu`, `megatron.model` and `megatron.fused_kernels` for the layer path; everything computes in
libmm_hip.so (include/mm_hip.h).  There is no CPU or eager fallback.
"""
__version__ = "0.1.0"
