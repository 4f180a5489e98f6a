from .Distribution import Distribution
from .GaussianPrior import GaussianPrior
from .Sampled import Sampled
from .MultivariateNormalDiagPlusLowRank import MultivariateNormalDiagPlusLowRank
from . import tfd
