"""PM-VaDE on MNIST (a partial encoder + AutoregressiveGMM matched to a frozen VaDE): the values of the reference's
configs/pm_vade_mnist.py as one nested literal.  `vade_dir`: run directory of train_vade.py (model_config.json + train_state.pkl)."""
from posterior_matching_amd.config_dict import ConfigDict


def get_config():
    return ConfigDict(
    {'data': {'dataset': 'mnist',
              'train_split': 'train',
              'validation_split': 'test',
              'train_batch_size': 128,
              'val_batch_size': 128},
     'vade_dir': 'runs/vade-mnist-20220305-121540',
     'model': {'encoder_net': 'ConvEncoder',
               'decoder_net': 'ConvDecoder',
               'decoder_dist': 'Bernoulli',
               'latent_dim': 10,
               'num_components': 10,
               'partial_posterior_dist': 'AutoregressiveGMM',
               'partial_posterior_dist_config': {'num_components': 10, 'residual_blocks': 2, 'hidden_units': 256},
               'encoder_net_config': {'conv_layers': [(32, 5, 1), (32, 5, 2), (64, 5, 1), (64, 5, 2), (128, 7, 1)]},
               'decoder_net_config': {'conv_layers': [(64, 7, 1),
                                                      (64, 5, 2),
                                                      (32, 5, 1),
                                                      (32, 5, 2),
                                                      (32, 5, 1),
                                                      (1, 5, 1)]}},
     'steps': 160000,
     'validation_freq': 5000,
     'lr_schedule': {'init_value': 0.001, 'decay_rate': 0.9, 'staircase': False, 'transition_steps': 4687}}
    )
