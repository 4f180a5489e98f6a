"""A driver in the style of the reference's HMC_classification.py / simple_regression_example.py,
written against `import tensorflow as tf` and `from Pyesian...` to exercise compat/ end to end:
    PYTHONPATH=.:compat python examples/hmc_classification_compat.py
(The reference's own scripts run the same way with their path in place of this file.)"""
import numpy as np
import tensorflow as tf
from Pyesian.datasets import Dataset
from Pyesian.distributions import GaussianPrior
from Pyesian.nn import BayesianModel
from Pyesian.optimizers import BBB, HMC, SGD
from Pyesian.optimizers.hyperparameters import HyperParameters
from Pyesian.visualisations import Metrics

from bayesian_inference_for_nn_amd import synth

np.random.seed(42)

# --- HMC on moons, as HMC_classification.py:21-76
x, y = synth.moons(2000, noise=0.2)
dataset = Dataset(tf.data.Dataset.from_tensor_slices((x, y)), tf.keras.losses.SparseCategoricalCrossentropy, "Classification")
model = tf.keras.models.Sequential([
    tf.keras.layers.Dense(50, activation='relu', input_shape=(2,)),
    tf.keras.layers.Dense(2, activation=tf.keras.activations.softmax)
])
optimizer = HMC()
optimizer.compile(HyperParameters(epsilon=0.002, m=0.5, L=20), model.to_json(), dataset, verbose=False, prior=GaussianPrior(0.0, 1.0))
optimizer.train(60)
bayesian_model: BayesianModel = optimizer.result()
x_test, y_true = next(iter(dataset.test_data.batch(dataset.test_size)))
_, preds = bayesian_model.predict(x_test, nb_samples=100)
preds = preds.numpy() if hasattr(preds, "numpy") else preds
pred_labels = tf.argmax(preds, axis=1).numpy()
acc = float((pred_labels == y_true.numpy()).mean())
print(f"HMC moons: accepted {optimizer._accepted_runs}/{optimizer._total_runs}, test accuracy {100 * acc:.1f} %")
assert acc > 0.8

# --- SGD regression, as simple_regression_example.py:11-38
x = tf.random.uniform(shape=(600, 1), minval=1, maxval=20, dtype=tf.float32)
y = 2 * x + 2
dataset = Dataset(tf.data.Dataset.from_tensor_slices((x, y)), tf.keras.losses.MeanSquaredError, "Regression")
model = tf.keras.models.Sequential()
model.add(tf.keras.layers.Dense(1, activation='linear', input_shape=(1,)))
optimizer = SGD()
optimizer.compile(HyperParameters(lr=1e-3, frequency=1), model.to_json(), dataset, verbose=False, starting_model=model)
optimizer.train(3000)
bm = optimizer.result()
Metrics(bm, dataset).summary()
w, b = bm._model.layers[0].trainable_variables
print(f"SGD linreg: w = {w[0, 0]:.3f}, b = {b[0]:.3f}")
assert abs(w[0, 0] - 2.0) < 0.15

# --- BBB on moons, as simple_classification_example.py:9-34 (result() used as a model)
x, y = synth.moons(2000)
dataset = Dataset(tf.data.Dataset.from_tensor_slices((x, y)), tf.keras.losses.SparseCategoricalCrossentropy, "Classification")
model = tf.keras.Sequential()
model.add(tf.keras.layers.Dense(50, activation='relu', input_shape=(2,)))
model.add(tf.keras.layers.Dense(2, activation=tf.keras.activations.softmax))
optimizer = BBB()
optimizer.compile(HyperParameters(lr=0.5, alpha=0.0, batch_size=1000), model.to_json(), dataset, verbose=False, prior=GaussianPrior(0.0, -1.0))
optimizer.train(600)
bayesian_model = optimizer.result()
m = Metrics(bayesian_model, dataset).summary()
assert m["accuracy"] > 0.8
print("compat example ok")
