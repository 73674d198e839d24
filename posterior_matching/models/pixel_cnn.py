"""reference import path posterior_matching.models.pixel_cnn -> the MI355X-native implementation."""
from posterior_matching_amd.models.pixel_cnn import *  # noqa: F401,F403
from posterior_matching_amd.models import pixel_cnn as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
