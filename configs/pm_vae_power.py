"""PM-VAE on UCI POWER: the values of the reference's configs/pm_vae_power.py as one nested literal."""
from posterior_matching_amd.config_dict import ConfigDict


def get_config():
    return ConfigDict(
    {'data': {'dataset': 'power',
              'train_split': 'train',
              'validation_split': 'val',
              'train_batch_size': 512,
              'val_batch_size': 512,
              'training_noise': 0.001,
              'mask_generator': 'BernoulliMaskGenerator'},
     'model': {'latent_dim': 16,
               'encoder_net': 'ResidualMLP',
               'decoder_net': 'ResidualMLP',
               'decoder_dist': 'IdentityGaussian',
               'posterior_dist': 'TriLGaussian',
               'decoder_dist_config': {'event_size': 6},
               'masked_posterior_dist': 'AutoregressiveGMM',
               'masked_posterior_config': {'hidden_units': 256, 'residual_blocks': 3},
               'encoder_net_config': {'residual_blocks': 2, 'hidden_units': 256, 'layer_norm': False},
               'decoder_net_config': {'residual_blocks': 2, 'hidden_units': 256, 'layer_norm': False},
               'matching_ll_stop_gradients': True},
     'beta': {'schedule': 'cyclic', 'low_value': 0.0, 'high_value': 1.0, 'period': 50000, 'delay': 1000},
     'steps': 200000,
     'validation_freq': 1000,
     'save_final_state': True,
     'weight_decay': 1e-05,
     'lr_schedule': {'init_value': 0.001, 'decay_rate': 0.9, 'transition_steps': 5000}}
    )
