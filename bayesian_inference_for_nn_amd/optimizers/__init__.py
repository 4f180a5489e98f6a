from .Optimizer import Optimizer
from .BBB import BBB
from .HMC import HMC
from .SGLD import SGLD
from .SGD import SGD
from .SVGD import SVGD
from .SWAG import SWAG
