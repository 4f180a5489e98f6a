from bayesian_inference_for_nn_amd.distributions import *  # noqa: F401,F403
from bayesian_inference_for_nn_amd.distributions import Distribution, GaussianPrior, Sampled, tfd  # noqa: F401
