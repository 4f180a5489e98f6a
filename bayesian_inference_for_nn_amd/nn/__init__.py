from .BayesianModel import BayesianModel
from .model import DenseNet, model_from_json, sequential_json
