from bayesian_inference_for_nn_amd.nn import *  # noqa: F401,F403
from bayesian_inference_for_nn_amd.nn import BayesianModel  # noqa: F401
