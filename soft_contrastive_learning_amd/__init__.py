"""MI355X-native hot path of soft-contrastive visual localisation.

``model.nets`` / ``model.losses`` / ``pointnetvlad_cls`` mirror the reference's Python
interface for the path; everything under them runs in ``libscl_hip.so`` (hand-written
gfx950 kernels behind the C-ABI of ``include/scl_hip.h``).  Importing the package does
not load the library; the first op does, and raises if it was never built.
"""
__version__ = "0.1.0"
