"""MI355X-native Posterior-Matching VAE training path (hand-written HIP behind a C ABI).

Drop-in for the hot path of lupalab/posterior-matching: `posterior_matching_amd.models` mirrors
`posterior_matching.models` (PosteriorMatchingVAE, get_network, get_distribution) and
`train_pm_vae.py --config configs/<name>.py` is the same entry point.
"""
__version__ = "0.1.0"
