from .HyperParameters import HyperParameters
