"""reference import path posterior_matching.models.vade -> the MI355X-native implementation."""
from posterior_matching_amd.models.vade import *  # noqa: F401,F403
from posterior_matching_amd.models import vade as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
