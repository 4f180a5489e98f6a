from .Distribution import Distribution
from .GaussianPrior import GaussianPrior
from .Sampled import Sampled
from . import tfd
