"""Mirror of fbs.sdes (fbs/sdes/__init__.py:1-3) for the sampler hot path."""
from .linear import (make_ou_sde, make_linear_sde, make_gaussian_bw_sb, LinearSDE, StationaryConstLinearSDE, StationaryLinLinearSDE,
                     StationaryExpLinearSDE)
from .simulators import reverse_simulator, euler_maruyama, discrete_time_simulator, doob_bridge_simulator
