"""MI355X-native (gfx950) cross-modal encode / fuse / score hot path of globc/mrAudio.

Everything numeric runs in ``libmra_hip.so`` (hand-written HIP kernels behind the C ABI of
``include/mra.h``); this package is the Python host side with the reference's interface:
``mraudio_amd.models.xinstructblip.XInstructBLIP`` (reference ``models/xinstructblip.py``),
``mraudio_amd.qformer.QFormer`` (the LAVIS Q-Former seam, ``.bert(...)``), the scorer, the
processors' call signatures and the clip-axis sharding over RCCL.
"""
from ._lib import LIB_PATH, MraError  # noqa: F401

__version__ = "0.1"
