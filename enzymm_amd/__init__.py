"""enzymm_amd -- MI355X-native geometric template matcher behind enzymm's PyJess boundary.

``enzymm_amd.pyjess`` mirrors the part of the ``pyjess`` API that enzymm uses; the search
runs in ``libenzymm_amd.so`` (hand-written HIP for gfx950, C ABI in ``include/enzymm_amd.h``).
"""
__version__ = "0.1.0"
