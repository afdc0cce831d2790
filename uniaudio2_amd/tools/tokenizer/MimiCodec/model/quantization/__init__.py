from .vq import ResidualVectorQuantizer, SplitResidualVectorQuantizer  # noqa: F401
