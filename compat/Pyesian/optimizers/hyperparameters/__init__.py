from bayesian_inference_for_nn_amd.optimizers.hyperparameters import HyperParameters  # noqa: F401
