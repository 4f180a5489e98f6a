"""Hyper-parameter bag with the reference's contract
(Pyesian/optimizers/hyperparameters/HyperParameters.py:6-62): keyword construction, attribute
access, ``AttributeError`` for unknown names (the optimizers probe optional keys with
``hasattr``), ``batch_size`` defaulting to 64, and the whitespace text format
``"name value name value"`` in which every value becomes a float."""

from copy import deepcopy

_KEY_EXTRA = "._-"       # characters allowed in a name besides letters and digits
_NUM_CHARS = "-."        # characters allowed in a number besides digits


def _is_key_char(ch: str) -> bool:
    return ch.isalnum() or ch in _KEY_EXTRA


def _is_num_char(ch: str) -> bool:
    return ch.isdigit() or ch in _NUM_CHARS


class HyperParameters:
    def __init__(self, **kwargs):
        self._params = deepcopy(kwargs)
        self._params.setdefault("batch_size", 64)
        self.connectors = _KEY_EXTRA

    def __getattr__(self, name):
        try:
            return self.__dict__["_params"][name]
        except KeyError:
            raise AttributeError("'HyperParameters' object has no attribute " + str(name)) from None

    def from_file(self, path):
        with open(path, "r") as handle:
            return self.parse(handle.read())

    def parse(self, text: str):
        """Alternates between a name (letters, digits, '.', '_', '-') and a number (digits, '-', '.');
        anything else separates tokens, and non-numeric characters while a number is expected are
        skipped.  Names left without a number at the end of the text get 0.0."""
        names, numbers = [], []
        token, want_number = "", False
        for ch in text:
            belongs = _is_num_char(ch) if want_number else _is_key_char(ch)
            if belongs:
                token += ch
                continue
            if token:
                if want_number:
                    numbers.append(float(token))
                else:
                    names.append(token)
                token, want_number = "", not want_number
        if token:
            if want_number:
                numbers.append(float(token))
            else:
                names.append(token)
                numbers.extend([0.0] * (len(names) - len(numbers)))
        for position, name in enumerate(names):
            self._params[name] = numbers[position]      # a name whose number never came raises IndexError, as the reference
        return self
