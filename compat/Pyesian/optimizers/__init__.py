from bayesian_inference_for_nn_amd.optimizers import *  # noqa: F401,F403
from bayesian_inference_for_nn_amd.optimizers import BBB, HMC, SGD, SGLD, SVGD, SWAG, Optimizer  # noqa: F401
