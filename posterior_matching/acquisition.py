"""reference import path posterior_matching.acquisition -> the MI355X-native implementation."""
from posterior_matching_amd.acquisition import *  # noqa: F401,F403
from posterior_matching_amd import acquisition as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
