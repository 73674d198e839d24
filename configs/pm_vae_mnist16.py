"""PM-VAE on 16 x 16 MNIST (the model the lookahead posteriors are trained for): the values of the reference's
configs/pm_vae_mnist16.py as one nested literal."""
from posterior_matching_amd.config_dict import ConfigDict


def get_config():
    return ConfigDict(
    {'data': {'dataset': 'mnist16',
              'train_split': 'train',
              'validation_split': 'test',
              'train_batch_size': 128,
              'val_batch_size': 128,
              'mask_generator': 'UniformMaskGenerator',
              'mask_generator_kwargs': {'bounds': (0.0, 0.2)}},
     'model': {'latent_dim': 10,
               'encoder_net': 'ConvEncoder',
               'decoder_net': 'ConvDecoder',
               'posterior_dist': 'TriLGaussian',
               'decoder_dist': 'Bernoulli',
               'encoder_net_config': {'conv_layers': [(32, 3, 1), (32, 3, 2), (64, 3, 2), (64, 1, 1)]},
               'decoder_net_config': {'conv_layers': [(64, 8, 1), (64, 5, 2), (32, 5, 1), (32, 5, 1), (1, 3, 1)]}},
     'steps': 200000,
     'validation_freq': 10000,
     'lr_schedule': {'init_value': 0.001, 'decay_rate': 0.9, 'transition_steps': 5000}}
    )
