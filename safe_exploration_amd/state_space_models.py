"""State-space-model surface of the hot path.

Mirrors the interface of /root/reference/safe_exploration/state_space_models.py:14-211
(``StateSpaceModel``): attribute names, method names, argument meaning and the
NotImplementedError behaviour of the abstract methods, and the CasADi callback wrapper
``CasadiSSMEvaluator`` (state_space_models.py:214-566) through which the MPC's IPOPT loop evaluates a
state-space model once per iteration (:278-303 forward, :384-417 Jacobian, :534-562 reverse).  The
wrapper is built lazily on whatever ``casadi`` module is importable (``casadi.Callback`` is its base
class); this package never needs casadi for anything else.
"""
import copy

import numpy as np

_EVALUATOR_CACHE = {}


def _dense(x):
    """casadi DM / numpy -> 2-D float64 numpy array."""
    a = np.array(x, dtype=np.float64)
    return a.reshape(-1, 1) if a.ndim < 2 else a


def _casadi_version(cas):
    """(major, minor) of the casadi module; (3, 5) when it does not say (the API generation of the reference:
    ``jacobian_old``, one stacked Jacobian output per callback)."""
    try:
        parts = str(getattr(cas, "__version__", "3.5")).split(".")
        return int(parts[0]), int("".join(ch for ch in parts[1] if ch.isdigit()) or 0)
    except Exception:
        return 3, 5


def _evaluator_class(cas):
    """``CasadiSSMEvaluator`` on top of the given casadi module (cached per module)."""
    cached = _EVALUATOR_CACHE.get(id(cas))
    if cached is not None:
        return cached

    def dense_in(ssm, linearize_mu, with_outputs, with_seeds):
        """sparsity list of [state, action | outputs | output seeds] (state_space_models.py:262-278,369-381,501-527)"""
        n, m = ssm.num_states, ssm.num_actions
        outs = [cas.Sparsity.dense(n, 1), cas.Sparsity.dense(n, 1)]
        if linearize_mu:
            outs.append(cas.Sparsity.dense(n, n + m))
        sp = [cas.Sparsity.dense(n, 1), cas.Sparsity.dense(m, 1)]
        if with_outputs:
            sp += outs
        if with_seeds:
            sp += outs
        return sp

    class _Derivative(cas.Callback):
        """Shared shell of the two nested callbacks of the reference (JacFun :328-417, BackFun :455-562): no
        derivatives of their own, dense inputs, evaluation delegated to a bound method of the evaluator."""

        def __init__(self, name, ssm, sp_in, sp_out, fn, opts):
            cas.Callback.__init__(self)
            self.ssm, self._sp_in, self._sp_out, self._fn = ssm, sp_in, sp_out, fn
            self.construct(name, opts)

        def get_n_in(self):
            return len(self._sp_in)

        def get_n_out(self):
            return len(self._sp_out)

        def get_sparsity_in(self, i):
            return self._sp_in[i]

        def get_sparsity_out(self, i):
            return self._sp_out[i]

        def has_reverse(self, nadj):
            return False

        def has_forward(self, nfwd):
            return False

        def has_jacobian(self):
            return False

        def eval(self, arg):
            return self._fn(arg)

    class CasadiSSMEvaluator(cas.Callback):
        """casadi.Callback evaluating a StateSpaceModel: inputs (state n x 1, action m x 1), outputs
        (mean n x 1, variance n x 1[, jac_mean n x (n+m)]) -- state_space_models.py:214-303.

        Derivatives: ``get_jacobian`` returns a callback producing the stacked (2n [+ n(n+m)]) x (n+m) matrix
        [jac_mean; jac_variance[; d jac_mean/dz]] (:305-419); ``get_reverse`` one producing the adjoints of
        (state, action) for the output seeds (:435-566).  One model evaluation per call of either."""

        def __init__(self, ssm, linearize_mu=True, has_jacobian=True, has_reverse=False, opts={}, jac_mu_order="F"):
            cas.Callback.__init__(self)
            self.v_has_jacobian = has_jacobian
            self.v_has_reverse = has_reverse
            self.v_has_forward = False
            if not (has_jacobian or has_reverse):
                raise ValueError("Need to specify either has_jacobian or has_reverse")
            self.ssm = ssm
            self.linearize_mu = linearize_mu
            # Row order of the d jac_mean/dz block of the stacked Jacobian.  CasADi numbers the entries of a dense
            # matrix output column by column (its ``vec``), and the full Jacobian it asks a callback for has one row
            # per such entry: row j*n + i <-> jac_mean[i, j] -- "F", the default here.  The reference states the same
            # intent ("The reshaping rule has to follow the casadi rule", utils.py:357-380) but its helper flattens
            # row-major (row i*D + j <-> jac_mean[i, j], "C"); its only test of the block sums over all entries
            # (test_state_space_models.py:263-286) and cannot tell the two apart.  "C" reproduces the reference's
            # rows exactly.  The reverse callback is unaffected (its seed arrives as an n x D matrix).
            # tests/test_casadi_real.py settles it against IPOPT's derivative checker wherever casadi is installed.
            # (constructor argument jac_mu_order="C" for runs that must match the reference's rows byte for byte)
            if jac_mu_order not in ("F", "C"):
                raise ValueError("jac_mu_order must be 'F' (CasADi's vec rule) or 'C' (the reference helper's rows)")
            self.jac_mu_order = jac_mu_order
            self.construct("CasadiModelEvaluator", opts)

        def get_n_in(self):
            return 2

        def get_n_out(self):
            return 3 if self.linearize_mu else 2

        def get_sparsity_in(self, i):
            return dense_in(self.ssm, self.linearize_mu, False, False)[i]

        def get_sparsity_out(self, i):
            return dense_in(self.ssm, self.linearize_mu, True, False)[2 + i]

        def eval(self, arg):
            state, action = _dense(arg[0]), _dense(arg[1])
            if self.linearize_mu:
                mu, sigma, jac_mu = self.ssm.linearize_predict(state.T, action.T, False, False)
                return [mu, sigma, jac_mu]
            mu, sigma = self.ssm.predict(state.T, action.T)
            return [np.reshape(mu, (-1, 1)), np.reshape(sigma, (-1, 1))]

        # -- Jacobian callback ------------------------------------------------------------------------
        def _eval_jacobian(self, arg):
            state, action = _dense(arg[0]), _dense(arg[1])
            n, D = self.ssm.num_states, self.ssm.num_states + self.ssm.num_actions
            if self.linearize_mu:
                _, _, jac_mu, jac_sigma, hess_mu = self.ssm.linearize_predict(state.T, action.T, True, False)
                # (n, D, D) -> (n D, D): "C": row i*D + j holds d jac_mu[i, j] / dz (utils.py:357-380), "F": row j*n + i
                if self.jac_mu_order == "F":
                    hess_mu = np.transpose(np.reshape(hess_mu, (n, D, D)), (1, 0, 2))
                return [np.vstack((jac_mu, jac_sigma, np.reshape(hess_mu, (n * D, D))))]
            _, _, jac_mu, jac_sigma = self.ssm.predict(state.T, action.T, True, False)
            return [np.vstack((np.reshape(jac_mu, (n, D)), np.reshape(jac_sigma, (n, D))))]

        def _eval_jacobian_blocks(self, arg):
            # CasADi >= 3.6: one block per (output, input) pair, output-major: d out_o / d in_i of shape
            # numel(out_o) x numel(in_i), rows in CasADi's vec (column-major) order of out_o
            (stacked,) = self._eval_jacobian_order(arg, "F")
            n, m = self.ssm.num_states, self.ssm.num_actions
            bounds = [0, n, 2 * n] + ([2 * n + n * (n + m)] if self.linearize_mu else [])
            blocks = []
            for o in range(len(bounds) - 1):
                rows = stacked[bounds[o]:bounds[o + 1]]
                blocks += [np.ascontiguousarray(rows[:, :n]), np.ascontiguousarray(rows[:, n:])]
            return blocks

        def _eval_jacobian_order(self, arg, order):
            keep, self.jac_mu_order = self.jac_mu_order, order
            try:
                return self._eval_jacobian(arg)
            finally:
                self.jac_mu_order = keep

        def get_jacobian(self, name, inames, onames, opts):
            n, m = self.ssm.num_states, self.ssm.num_actions
            D = n + m
            rows = 2 * n + (n * D if self.linearize_mu else 0)
            if _casadi_version(cas) >= (3, 6):
                # (the 3.6 API change: Function.jacobian() delivers the blocks jac_<out>_<in> instead of one matrix)
                sizes = [n, n] + ([n * D] if self.linearize_mu else [])
                sp_out = [cas.Sparsity.dense(r, c) for r in sizes for c in (n, m)]
                self.jac_callback = _Derivative(name, self.ssm, dense_in(self.ssm, self.linearize_mu, True, False),
                                                sp_out, self._eval_jacobian_blocks, opts)
                return self.jac_callback
            self.jac_callback = _Derivative(name, self.ssm, dense_in(self.ssm, self.linearize_mu, True, False),
                                            [cas.Sparsity.dense(rows, D)], self._eval_jacobian, opts)
            return self.jac_callback

        def has_reverse(self, nadj):
            return self.v_has_reverse and nadj == 1

        def has_forward(self, nfwd):
            return self.v_has_forward

        def has_jacobian(self):
            return self.v_has_jacobian

        # -- reverse-mode callback --------------------------------------------------------------------
        def _eval_reverse(self, arg):
            # arg = [state, action | outputs | seeds]; the seeds follow the n_out outputs (:556-562)
            n_out = self.get_n_out()
            seeds = [_dense(a) for a in arg[2 + n_out:2 + 2 * n_out]]
            # the adjoint needs the linearisation at THIS point: re-evaluate instead of trusting a cache that an
            # interleaved forward call may have overwritten
            state, action = _dense(arg[0]), _dense(arg[1])
            if self.linearize_mu:
                self.ssm.linearize_predict(state.T, action.T, True, False)
                # jac_mean seed flattened row-major: entry i*D + j multiplies d jac_mean[i, j] / dz -- what
                # get_linearize_reverse expects (the seed arrives as an n x D matrix, so this has nothing to do with the
                # row order of the third block of the stacked Jacobian, jac_mu_order above)
                seed = np.concatenate((seeds[0].reshape(-1), seeds[1].reshape(-1), seeds[2].reshape(-1)))
                adj_state, adj_action = self.ssm.get_linearize_reverse(seed)
            else:
                self.ssm.predict(state.T, action.T, True, False)
                seed = np.concatenate((seeds[0].reshape(-1), seeds[1].reshape(-1)))
                adj_state, adj_action = self.ssm.get_reverse(seed)
            return [cas.DM(np.reshape(adj_state, (-1, 1))), cas.DM(np.reshape(adj_action, (-1, 1)))]

        def get_reverse(self, nadj, name, inames, onames, opts):
            if not self.v_has_reverse:
                raise ValueError("Calling reverse even though it is not provided! This should not happen.")
            n, m = self.ssm.num_states, self.ssm.num_actions
            self.reverse_callback = _Derivative(name, self.ssm, dense_in(self.ssm, self.linearize_mu, True, True),
                                                [cas.Sparsity.dense(n, 1), cas.Sparsity.dense(m, 1)],
                                                self._eval_reverse, opts)
            return self.reverse_callback

    _EVALUATOR_CACHE[id(cas)] = CasadiSSMEvaluator
    return CasadiSSMEvaluator


def __getattr__(name):
    # ``from safe_exploration_amd.state_space_models import CasadiSSMEvaluator`` where casadi is installed
    if name == "CasadiSSMEvaluator":
        import casadi
        return _evaluator_class(casadi)
    raise AttributeError(name)


class StateSpaceModel(object):
    """x_{t+1} = f(x_t, u_t) with uncertainty information; x in (1 x n), u in (1 x m).

    Attributes (state_space_models.py:41-48): num_states, num_actions, has_jacobian, has_reverse.
    """

    def __init__(self, num_states, num_actions, has_jacobian=True, has_reverse=False):
        self.num_states = num_states
        self.num_actions = num_actions
        self._forward_cache = None
        self._linearize_forward_cache = None
        self.has_jacobian = has_jacobian
        self.has_reverse = has_reverse

    def __call__(self, states, actions):
        # state_space_models.py:50-72
        return self.predict(states, actions, True, False)

    def predict(self, states, actions, jacobians=False, full_cov=False):
        raise NotImplementedError("Need to implement this in a subclass!")

    def linearize_predict(self, states, actions, jacobians=False, full_cov=False):
        raise NotImplementedError("Need to implement this in a subclass when using the predefined "
                                  "get_forward_model_casadi() method")

    def get_forward_model_casadi(self, linearize_mu=True):
        """state_space_models.py:140-166: a ``CasadiSSMEvaluator`` around ``copy.deepcopy(self)``, with this
        model's ``has_jacobian`` / ``has_reverse`` flags."""
        try:
            import casadi
        except ImportError as exc:
            raise ImportError("get_forward_model_casadi needs casadi (the CasADi MPC is the caller of "
                              "this surface)") from exc
        return _evaluator_class(casadi)(copy.deepcopy(self), linearize_mu, self.has_jacobian, self.has_reverse)

    def get_reverse(self, seed):
        raise NotImplementedError("Need to implement this in a sublass when providing reverse AD "
                                  "for forward model")

    def get_linearize_reverse(self, seed):
        raise NotImplementedError("Need to implement this in a sublass when providing reverse AD "
                                  "for linearized forward")

    def update_model(self, train_x, train_y, opt_hyp=False, replace_old=False):
        raise NotImplementedError("Need to implement this in subclass")
