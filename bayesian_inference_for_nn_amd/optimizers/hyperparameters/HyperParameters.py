"""Hyper-parameter bag (mirrors Pyesian/optimizers/hyperparameters/HyperParameters.py:6-62:
attribute access, AttributeError for unknown keys, default batch_size = 64, the
"key value key value" text format whose values are all floats)."""

import copy


class HyperParameters:
    def __init__(self, **kwargs):
        self._params = copy.deepcopy(kwargs)
        if "batch_size" not in kwargs:
            self._params["batch_size"] = 64
        self.connectors = "._-"

    def __getattr__(self, item):
        params = self.__dict__.get("_params", {})
        if item in params:
            return params[item]
        raise AttributeError("'HyperParameters' object has no attribute " + str(item))

    def from_file(self, fn):
        with open(fn, "r") as f:
            return self.parse(f.read())

    def parse(self, text: str):
        keys, values = [], []
        k, v, s = "", "", 0
        for c in text:
            if s == 0:
                if c.isalnum() or c in self.connectors:
                    k += c
                elif k:
                    keys.append(k)
                    k = ""
                    s = 1
            else:
                if c.isdigit() or c in "-.":
                    v += c
                elif v:
                    values.append(float(v))
                    v = ""
                    s = 0
        if k:
            keys.append(k)
            for _ in range(len(keys) - len(values)):
                values.append(0.0)
        elif v:
            values.append(float(v))
        for i in range(len(keys)):
            self._params[keys[i]] = values[i]
        return self
