"""Minimal stand-ins for the reference's evaluation helpers (out of the hot path's scope): enough
for the driver scripts' `Metrics(model, dataset).summary()` and `Plotter(...)` calls to run."""

import numpy as np


class Metrics:
    def __init__(self, model, dataset):
        self._model = model[0] if isinstance(model, tuple) else model
        self._dataset = dataset

    def summary(self, nb_samples: int = 100):
        x, y = self._dataset.test_data.as_numpy()
        _, mean = self._model.predict(x, nb_samples)
        mean = np.asarray(mean)
        if self._dataset.likelihood_model == "Classification":
            acc = float((mean.argmax(axis=1) == np.asarray(y).reshape(-1)).mean())
            print(f"Accuracy: {100 * acc:.2f} %")
            return {"accuracy": acc}
        mse = float(((mean - np.asarray(y).reshape(mean.shape)) ** 2).mean())
        print(f"MSE: {mse:.6f}")
        return {"mse": mse}


class Plotter:
    def __init__(self, model, dataset):
        self._model = model[0] if isinstance(model, tuple) else model
        self._dataset = dataset

    def _plot(self, name):
        try:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
        except Exception:
            print(f"[Plotter] matplotlib unavailable; skipped {name}")
            return
        x, y = self._dataset.test_data.as_numpy()
        _, mean = self._model.predict(x, 20)
        fig = plt.figure(figsize=(5, 4))
        x2 = np.asarray(x).reshape(len(x), -1)
        if x2.shape[1] >= 2:
            plt.scatter(x2[:, 0], x2[:, 1], c=np.asarray(mean).argmax(axis=1) if np.asarray(mean).shape[1] > 1 else np.asarray(mean)[:, 0], s=6)
        else:
            plt.scatter(x2[:, 0], np.asarray(mean)[:, 0], s=6)
        fig.savefig(name + ".png", dpi=80)
        plt.close(fig)

    def plot_decision_boundaries(self, n_samples=100, n_boundaries=10, **kw):
        self._plot("decision_boundaries")

    def plot_uncertainty_area(self, uncertainty_threshold=0.9, **kw):
        self._plot("uncertainty_area")

    def regression_uncertainty(self, *a, **kw):
        self._plot("regression_uncertainty")

    def __getattr__(self, name):
        if name.startswith("plot") or name.startswith("compare") or name.startswith("learning"):
            return lambda *a, **k: self._plot(name)
        raise AttributeError(name)
