"""PM-VQVAE (stage 2) on MNIST: the configuration of the reference's configs/pm_vqvae_mnist.py (values
are data), expressed with this repo's ConfigDict."""
from posterior_matching_amd.config_dict import ConfigDict


def get_config():
    config = ConfigDict()

    config.data = ConfigDict()
    config.data.dataset = "mnist"
    config.data.train_split = "train"
    config.data.validation_split = "test"
    config.data.train_batch_size = 32
    config.data.val_batch_size = 32
    config.data.mask_generator = "MNISTMaskGenerator"

    # run directory of `train_vqvae.py` holding the trained VQVAE (model_config.json + train_state.pkl)
    config.vqvae_dir = "runs/vqvae-mnist-20220227-111235"

    config.pixel_cnn = ConfigDict()
    config.pixel_cnn.image_shape = (7, 7)
    config.pixel_cnn.num_resnet = 8
    config.pixel_cnn.num_hierarchies = 1
    config.pixel_cnn.num_filters = 128
    config.pixel_cnn.dropout = 0.5

    config.conditional_dim = 512

    config.steps = 120000
    config.validation_freq = 1000

    config.lr_schedule = ConfigDict()
    config.lr_schedule.init_value = 3e-4
    config.lr_schedule.decay_rate = 0.999995
    config.lr_schedule.transition_steps = 1

    return config
