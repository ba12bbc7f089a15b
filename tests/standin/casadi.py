"""Stand-in for the handful of casadi names the SSM evaluator touches (casadi is not installable here).

TEST INFRASTRUCTURE, written for this repo: put ``tests/standin`` on sys.path and ``import casadi`` resolves
to this module.  It mimics what matters for the contract between ``CasadiSSMEvaluator`` and a
StateSpaceModel: ``Callback.construct`` queries the declared arities / sparsities like casadi does,
``DM`` is column-major under ``reshape`` (casadi's convention) and converts with ``np.array``."""
import numpy as _np


class Sparsity(object):
    def __init__(self, rows, cols):
        self.rows, self.cols = int(rows), int(cols)

    @staticmethod
    def dense(rows, cols=1):
        return Sparsity(rows, cols)

    @property
    def shape(self):
        return (self.rows, self.cols)

    def __eq__(self, other):
        return tuple(other) == self.shape if not isinstance(other, Sparsity) else other.shape == self.shape

    def __iter__(self):
        return iter(self.shape)


class DM(object):
    """dense matrix: always 2-D, column-major reshape"""

    def __init__(self, x):
        a = _np.array(x, dtype=_np.float64)
        if a.ndim == 0:
            a = a.reshape(1, 1)
        elif a.ndim == 1:
            a = a.reshape(-1, 1)
        self._a = a

    @property
    def shape(self):
        return self._a.shape

    @property
    def T(self):
        return DM(self._a.T)

    def reshape(self, shape):
        return DM(_np.reshape(self._a, shape, order="F"))

    def __array__(self, dtype=None, copy=None):
        return self._a.astype(dtype) if dtype is not None else self._a

    def full(self):
        return self._a.copy()


def vertcat(*args):
    return DM(_np.vstack([_np.array(a, dtype=_np.float64).reshape(_np.array(a).shape[0], -1) for a in args]))


def reshape(a, shape):
    return _np.reshape(a, shape, order="F")


class Callback(object):
    def __init__(self):
        self.constructed = None

    def construct(self, name, opts=None):
        # casadi asks for the signature at construction time: a callback with inconsistent declarations fails here
        self.signature_in = [self.get_sparsity_in(i).shape for i in range(self.get_n_in())]
        self.signature_out = [self.get_sparsity_out(i).shape for i in range(self.get_n_out())]
        self.constructed = name

    def __call__(self, *args):
        """numeric call: checks the argument shapes against the declared sparsities, then eval()"""
        assert len(args) == len(self.signature_in), "arity"
        dm = [DM(a) for a in args]
        for a, s in zip(dm, self.signature_in):
            assert a.shape == s, (a.shape, s)
        out = self.eval(dm)
        assert len(out) == len(self.signature_out)
        out = [DM(o) for o in out]
        for o, s in zip(out, self.signature_out):
            assert o.shape == s, (o.shape, s)
        return out
