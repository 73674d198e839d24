"""reference import path posterior_matching.masking"""
from posterior_matching_amd.masking import *  # noqa: F401,F403
from posterior_matching_amd.masking import get_mask_generator  # noqa: F401
