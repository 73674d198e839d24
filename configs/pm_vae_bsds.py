"""PM-VAE on BSDS300 patches: the values of the reference's configs/pm_vae_bsds.py as one nested literal (the one reference
config with the monotonic beta schedule; `masked_posterior_*` are NOT read by from_config - see configs/pm_vae_gas.py)."""
from posterior_matching_amd.config_dict import ConfigDict


def get_config():
    return ConfigDict(
    {'data': {'dataset': 'bsds',
              'train_split': 'train',
              'validation_split': 'val',
              'train_batch_size': 512,
              'val_batch_size': 512,
              'training_noise': 0.001,
              'mask_generator': 'BernoulliMaskGenerator'},
     'model': {'latent_dim': 64,
               'encoder_net': 'ResidualMLP',
               'decoder_net': 'ResidualMLP',
               'decoder_dist': 'IdentityGaussian',
               'posterior_dist': 'TriLGaussian',
               'decoder_dist_config': {'event_size': 63},
               'masked_posterior_dist': 'AutoregressiveGMM',
               'masked_posterior_config': {'hidden_units': 256, 'residual_blocks': 3},
               'encoder_net_config': {'residual_blocks': 5, 'hidden_units': 256, 'layer_norm': True},
               'decoder_net_config': {'residual_blocks': 5, 'hidden_units': 256, 'layer_norm': True},
               'matching_ll_stop_gradients': True},
     'beta': {'schedule': 'monotonic',
              'low_value': 0.0,
              'high_value': 1.0,
              'transition_steps': 200000,
              'transition_begin': 30000},
     'steps': 200000,
     'validation_freq': 1000,
     'save_final_state': True,
     'weight_decay': 1e-05,
     'lr_schedule': {'init_value': 0.001, 'decay_rate': 0.9, 'transition_steps': 5000}}
    )
