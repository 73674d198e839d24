"""reference import path posterior_matching.clustering -> the MI355X-native implementation."""
from posterior_matching_amd.clustering import *  # noqa: F401,F403
from posterior_matching_amd import clustering as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
