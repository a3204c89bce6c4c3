"""BasicBrain: same surface as ReinLife/Models/utils.py:1-14 (method / input_dim / output_dim)."""


class BasicBrain:
    def __init__(self, input_dim, output_dim, method):
        self._method = method
        self.input_dim = input_dim
        self.output_dim = output_dim

    @property
    def method(self):
        return self._method

    @method.setter
    def method(self, method):
        self._method = method
