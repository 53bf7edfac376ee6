"""fbs/samplers/common.py:5-9."""
from typing import Any, NamedTuple


class MCMCState(NamedTuple):
    acceptance_prob: Any
    is_accepted: Any
    prop_log_ell: Any
    log_ell: Any
