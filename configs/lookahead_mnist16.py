"""Lookahead posteriors for a PM-VAE on 16 x 16 MNIST: the configuration of the reference's configs/lookahead_mnist16.py (values
are data), expressed with this repo's ConfigDict."""
from posterior_matching_amd.config_dict import ConfigDict


def get_config():
    config = ConfigDict()

    config.data = ConfigDict()
    config.data.dataset = "mnist16"
    config.data.train_split = "train"
    config.data.validation_split = "test"
    config.data.train_batch_size = 32
    config.data.val_batch_size = 32
    config.data.mask_generator = "UniformMaskGenerator"
    config.data.mask_generator_kwargs = ConfigDict()
    config.data.mask_generator_kwargs.bounds = (0.0, 0.20)

    # run directory of `train_pm_vae.py` holding the trained VAE (model_config.json + train_state.pkl)
    config.pm_vae_dir = "runs/pm-vae-mnist16-20220302-160842"

    config.model = ConfigDict()
    config.model.lookahead_subsample = 16
    config.model.model_samples = 64

    config.steps = 40000
    config.validation_freq = 5000

    config.lr_schedule = ConfigDict()
    config.lr_schedule.init_value = 0.001
    config.lr_schedule.decay_rate = 0.9
    config.lr_schedule.transition_steps = 5000

    return config
