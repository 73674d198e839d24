"""VaDE on MNIST (GMM-prior VAE): the values of the reference's configs/vade_mnist.py as one nested literal."""
from posterior_matching_amd.config_dict import ConfigDict


def get_config():
    return ConfigDict(
    {'data': {'dataset': 'mnist',
              'train_split': 'train',
              'validation_split': 'test',
              'train_batch_size': 128,
              'val_batch_size': 128},
     'model': {'encoder_net': 'ConvEncoder',
               'decoder_net': 'ConvDecoder',
               'decoder_dist': 'Bernoulli',
               'latent_dim': 10,
               'num_components': 10,
               'encoder_net_config': {'conv_layers': [(32, 5, 1), (32, 5, 2), (64, 5, 1), (64, 5, 2), (128, 7, 1)]},
               'decoder_net_config': {'conv_layers': [(64, 7, 1),
                                                      (64, 5, 2),
                                                      (32, 5, 1),
                                                      (32, 5, 2),
                                                      (32, 5, 1),
                                                      (1, 5, 1)]}},
     'pretrain_steps': 70312,
     'steps': 140625,
     'validation_freq': 1000,
     'cluster_pred_num_samples': 50,
     'pretrain_lr': 0.002,
     'lr_schedule': {'init_value': 0.002, 'decay_rate': 0.9, 'staircase': False, 'transition_steps': 4687},
     'adam': {'eps': 0.0001}}
    )
