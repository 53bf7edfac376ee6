"""Mirror of fbs.samplers (fbs/samplers/__init__.py:1-3)."""
from .smc import bootstrap_filter, pmcmc_kernel, twisted_smc
from .resampling import multinomial, systematic, stratified, killing
from .gibbs import gibbs_init, gibbs_kernel
