from .vae import PosteriorMatchingVAE  # noqa: F401
from .networks import get_network, ConvEncoder, ConvDecoder, ResidualMLP  # noqa: F401
from .distributions import get_distribution  # noqa: F401
