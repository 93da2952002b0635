"""latok_amd -- MI355X-native (gfx950, HIP) implementation of latok's character-feature-matrix + split-mask path.

Layout mirrors the reference package so it can stand in for it:

    latok_amd.latok                    <- reference C extension ``latok.latok`` (latok/core/src/latok/latok.c:373-378)
    latok_amd.core.offsets             <- reference latok/core/offsets.py
    latok_amd.core.latok_utils         <- reference latok/core/latok_utils.py
    latok_amd.core.default_tokenizer   <- reference latok/core/default_tokenizer.py
    latok_amd.batch                    <- additive: whole-batch entry points over the fused kernel

All compute goes through ``liblatok_hip.so`` (C ABI in include/latok_hip.h).  There is no CPU fallback: importing is
harmless, but every compute call raises ``RuntimeError`` when the library or a HIP device is missing.
"""
__version__ = "0.1.0"
