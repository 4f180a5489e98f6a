"""The reference's package path, served by bayesian_inference_for_nn_amd (see compat/README.md)."""
