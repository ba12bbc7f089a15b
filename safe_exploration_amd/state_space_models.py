"""State-space-model surface of the hot path.

Mirrors the interface of /root/reference/safe_exploration/state_space_models.py:14-211
(``StateSpaceModel``): attribute names, method names, argument meaning and the
NotImplementedError behaviour of the abstract methods.  The CasADi callback wrapper of the
reference (``CasadiSSMEvaluator``, state_space_models.py:214-566) is a *consumer* of this surface
and only exists where casadi is importable; it is out of scope of the MI355X hot path.
"""

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
        """state_space_models.py:140-166 wraps ``copy.deepcopy(self)`` in a ``casadi.Callback``.

        The callback class is CasADi glue of the *caller* (it only uses ``linearize_predict`` / ``predict`` /
        ``get_reverse`` / ``get_linearize_reverse`` and ``copy.deepcopy`` of this object, all provided here), so
        when the reference package is importable its own ``CasadiSSMEvaluator`` is used unchanged on top of this
        model; nothing CasADi-specific is re-implemented (casadi is not installed where this library is built and
        tested)."""
        try:
            import casadi  # noqa: F401
        except ImportError as exc:
            raise ImportError("get_forward_model_casadi needs casadi (the CasADi MPC is the caller of "
                              "this surface, not part of the MI355X hot path)") from exc
        try:
            from safe_exploration.state_space_models import CasadiSSMEvaluator
        except ImportError as exc:
            raise NotImplementedError("no CasADi callback class available: install the reference package "
                                      "(safe_exploration.state_space_models.CasadiSSMEvaluator works on this "
                                      "model unchanged)") from exc
        import copy
        return CasadiSSMEvaluator(copy.deepcopy(self), linearize_mu)

    def get_reverse(self, seed):
        raise NotImplementedError("Need to implement this in a sublass when providing reverse AD "
                                  "for forward model")

    def get_linearize_reverse(self, seed):
        raise NotImplementedError("Need to implement this in a sublass when providing reverse AD "
                                  "for linearized forward")

    def update_model(self, train_x, train_y, opt_hyp=False, replace_old=False):
        raise NotImplementedError("Need to implement this in subclass")
