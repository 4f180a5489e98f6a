from bayesian_inference_for_nn_amd.datasets import *  # noqa: F401,F403
from bayesian_inference_for_nn_amd.datasets import ArrayDataset, Dataset  # noqa: F401
