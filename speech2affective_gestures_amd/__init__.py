"""speech2affective_gestures_amd -- MI355X-native hot path of Speech2AffectiveGestures (the GAN training step).

Import surface mirrors the reference:  ``net.multimodal_context_net_v2``, ``net.tcn``, ``net.utils.tgcn``,
``net.utils.graph`` and ``processor_v2`` live under this package with the same class names, constructor
and forward signatures and ``state_dict`` keys.  All arithmetic runs in ``libs2ag_hip.so`` (hand-written
gfx950 kernels, C ABI in include/s2ag_hip.h); there is no CPU fallback.
"""
__version__ = '0.1.0'
