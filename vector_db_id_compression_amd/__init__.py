"""MI355X-native per-inverted-list ID codecs (ROC / bits-back ANS, Elias-Fano, packed bits).

Hot path of facebookresearch/vector_db_id_compression rebuilt as hand-written HIP for gfx950 behind a
C-ABI (include/vidc.h); this package is the thin Python host side mirroring the reference's
`custom_invlists` / `altid` SWIG modules.
"""
from . import _lib  # noqa: F401
from ._lib import VIDC_PREC_EXACT, VIDC_PREC_REFERENCE, VidcError  # noqa: F401

__all__ = ["VidcError", "VIDC_PREC_REFERENCE", "VIDC_PREC_EXACT"]
