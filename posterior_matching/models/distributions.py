"""reference import path posterior_matching.models.distributions -> the MI355X-native implementation."""
from posterior_matching_amd.models.distributions import *  # noqa: F401,F403
from posterior_matching_amd.models import distributions as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
