"""behavenet_amd: MI355X-native implementation of BehaveNet's conv-autoencoder hot path.

Python is glue (config, module containers, training loop); all arithmetic on the path runs in
the hand-written HIP kernels of ``libbehavenet_hip.so`` (see ``include/behavenet_hip.h``).
"""

__version__ = '0.1.0'
