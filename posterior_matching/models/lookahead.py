"""reference import path posterior_matching.models.lookahead -> the MI355X-native implementation."""
from posterior_matching_amd.models.lookahead import *  # noqa: F401,F403
from posterior_matching_amd.models import lookahead as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
