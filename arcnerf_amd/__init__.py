"""arcnerf_amd — MI355X (gfx950) native volumetric-rendering hot path behind ArcNerf's plugin API.

The compute lives in arcnerf_amd/lib/libarcnerf_hip.so (hand-written HIP, C ABI in include/arcnerf_hip.h); this
package is the host-side mirror of the reference's encoder / model / chunk_processing interface.
"""
__version__ = '0.1.0'
