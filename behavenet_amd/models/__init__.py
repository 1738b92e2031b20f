from behavenet_amd.models.aes import AE, ConditionalAE, AEMSP  # noqa: F401
from behavenet_amd.models.vaes import VAE, ConditionalVAE, BetaTCVAE, PSVAE, MSPSVAE  # noqa: F401
from behavenet_amd.models.decoders import ConvDecoder  # noqa: F401
