"""The two PM-VAE configurations the hot path is checked on, as plain dicts.

Values restate configs/pm_vae_mnist.py:4-50 and configs/pm_vae_gas.py:4-59 of the reference
(data, not code); the oracle takes plain dicts so that it does not depend on the product's
ConfigDict.  tests/test_configs.py checks that the product's configs/*.py agree with these.
"""


def pm_vae_mnist():
    return {
        "data": {"dataset": "mnist", "train_split": "train", "validation_split": "test",
                 "train_batch_size": 256, "val_batch_size": 256, "mask_generator": "MNISTMaskGenerator"},
        "model": {
            "latent_dim": 32, "encoder_net": "ConvEncoder", "decoder_net": "ConvDecoder",
            "posterior_dist": "TriLGaussian", "partial_posterior_dist": "AutoregressiveGMM",
            "decoder_dist": "Bernoulli",
            "encoder_net_config": {"conv_layers": [(32, 5, 1), (32, 5, 2), (64, 5, 1), (64, 5, 2), (128, 7, 1)]},
            "decoder_net_config": {"conv_layers": [(64, 7, 1), (64, 5, 2), (32, 5, 1), (32, 5, 2), (32, 5, 1), (1, 5, 1)]},
        },
        "steps": 80000, "validation_freq": 1000,
        "lr_schedule": {"init_value": 0.001, "decay_rate": 0.9, "transition_steps": 5000},
    }


def pm_vae_gas():
    return {
        "data": {"dataset": "gas", "train_split": "train", "validation_split": "val",
                 "train_batch_size": 512, "val_batch_size": 512, "training_noise": 0.001,
                 "mask_generator": "BernoulliMaskGenerator"},
        "model": {
            "latent_dim": 16, "encoder_net": "ResidualMLP", "decoder_net": "ResidualMLP",
            "decoder_dist": "IdentityGaussian", "posterior_dist": "TriLGaussian",
            "decoder_dist_config": {"event_size": 8},
            "masked_posterior_dist": "AutoregressiveGMM",
            "masked_posterior_config": {"hidden_units": 256, "residual_blocks": 3},
            "encoder_net_config": {"residual_blocks": 2, "hidden_units": 256, "layer_norm": False},
            "decoder_net_config": {"residual_blocks": 2, "hidden_units": 256, "layer_norm": False},
            "matching_ll_stop_gradients": True,
        },
        "beta": {"schedule": "cyclic", "low_value": 0.0, "high_value": 1.0, "period": 50000, "delay": 1000},
        "steps": 200000, "validation_freq": 1000, "save_final_state": True, "weight_decay": 0.00001,
        "lr_schedule": {"init_value": 0.001, "decay_rate": 0.9, "transition_steps": 5000},
    }


def vqvae_mnist():
    """configs/vqvae_mnist.py:4-30 of the reference (stage-1 VQ-VAE; BASELINE quotes it at batch 256)."""
    return {
        "data": {"dataset": "mnist", "train_split": "train", "validation_split": "test",
                 "train_batch_size": 32, "val_batch_size": 32},
        "model": {"embedding_dim": 64, "num_embeddings": 256, "hidden_units": 32, "residual_hidden_units": 32,
                  "residual_blocks": 2, "decay": 0.99, "use_ema": True, "commitment_cost": 0.25,
                  "output_channels": 1},
        "steps": 60000, "validation_freq": 1000, "learning_rate": 3e-4,
    }


def pm_vqvae_mnist():
    """configs/pm_vqvae_mnist.py:4-40 of the reference (stage 2: PixelCNN partial posterior over the 7x7 codes)."""
    return {
        "data": {"dataset": "mnist", "train_split": "train", "validation_split": "test",
                 "train_batch_size": 32, "val_batch_size": 32, "mask_generator": "MNISTMaskGenerator"},
        "vqvae_dir": "runs/vqvae-mnist-20220227-111235",
        "pixel_cnn": {"image_shape": (7, 7), "num_resnet": 8, "num_hierarchies": 1, "num_filters": 128, "dropout": 0.5},
        "conditional_dim": 512,
        "steps": 120000, "validation_freq": 1000,
        "lr_schedule": {"init_value": 3e-4, "decay_rate": 0.999995, "transition_steps": 1},
    }


def pm_vdvae_mnist():
    """configs/pm_vdvae_mnist.py:4-35 of the reference (per-device batch 16; BASELINE quotes global 64 on 8 GPUs)."""
    return {
        "data": {"dataset": "mnist", "train_split": "train", "validation_split": "test",
                 "train_batch_size": 16, "val_batch_size": 16, "mask_generator": "MNISTMaskGenerator"},
        "model": {"image_shape": (28, 28, 1), "encoder_blocks": "28x6,28d2,14x4,14d2,7x2,7d2,3x2,3d2,1x2",
                  "decoder_blocks": "1x2,3m1,3x2,7m3,7x2,14m7,14x4,28m14,28x6", "latent_dim": 16, "width": 192,
                  "bottleneck_multiple": 0.25, "no_bias_above": 64, "num_mixtures": 10, "custom_width_string": None},
        "ema_rate": 0.999, "gradient_clip": 200.0, "lr": 0.00015,
        "steps": 500000, "validation_freq": 5000,
    }


def vqvae_celeb_a():
    """configs/vqvae_celeb_a.py:4-30 of the reference (stage-1 VQ-VAE on 64 x 64 RGB; 16 x 16 code grid, K = 512)."""
    return {
        "data": {"dataset": "celeb_a", "train_split": "train", "validation_split": "validation",
                 "train_batch_size": 64, "val_batch_size": 64},
        "model": {"embedding_dim": 64, "num_embeddings": 512, "hidden_units": 128, "residual_hidden_units": 32,
                  "residual_blocks": 2, "decay": 0.99, "use_ema": True, "commitment_cost": 0.25,
                  "output_channels": 3},
        "steps": 100000, "validation_freq": 1000, "learning_rate": 3e-4,
    }


def pm_vqvae_celeb_a():
    """configs/pm_vqvae_celeb_a.py:4-40 of the reference (stage 2: 12-resnet PixelCNN over the 16 x 16 codes; per-device
    batch 32; BASELINE quotes global 128 on 8 GPUs = 16 per GPU)."""
    return {
        "data": {"dataset": "celeb_a", "train_split": "train", "validation_split": "validation",
                 "train_batch_size": 32, "val_batch_size": 32, "mask_generator": "CelebAMaskGenerator"},
        "vqvae_dir": "runs/vqvae-celeb_a-20220227-111869",
        "pixel_cnn": {"image_shape": (16, 16), "num_resnet": 12, "num_hierarchies": 1, "num_filters": 128, "dropout": 0.5},
        "conditional_dim": 512,
        "steps": 150000, "validation_freq": 2000,
        "lr_schedule": {"init_value": 3e-4, "decay_rate": 0.999995, "transition_steps": 1},
    }


def pm_vae_miniboone():
    """configs/pm_vae_miniboone.py:4-61 of the reference: the ResidualMLP(layer_norm=True, dropout=0.5) family."""
    return {
        "data": {"dataset": "miniboone", "train_split": "train", "validation_split": "val",
                 "train_batch_size": 1024, "val_batch_size": 1024, "training_noise": 0.001,
                 "mask_generator": "BernoulliMaskGenerator"},
        "model": {
            "latent_dim": 32, "encoder_net": "ResidualMLP", "decoder_net": "ResidualMLP",
            "decoder_dist": "IdentityGaussian", "posterior_dist": "TriLGaussian",
            "decoder_dist_config": {"event_size": 43},
            "masked_posterior_dist": "AutoregressiveGMM",
            "masked_posterior_config": {"hidden_units": 256, "residual_blocks": 3},
            "encoder_net_config": {"residual_blocks": 5, "hidden_units": 256, "layer_norm": True, "dropout": 0.5},
            "decoder_net_config": {"residual_blocks": 2, "hidden_units": 256, "layer_norm": True, "dropout": 0.5},
            "matching_ll_stop_gradients": True,
        },
        "beta": {"schedule": "cyclic", "low_value": 0.0, "high_value": 1.0, "period": 5000, "delay": 2000},
        "steps": 22000, "validation_freq": 1000, "save_final_state": True, "weight_decay": 0.00001,
        "lr_schedule": {"init_value": 0.001, "decay_rate": 0.9, "transition_steps": 1000},
    }


def vade_mnist():
    """configs/vade_mnist.py:4-55 of the reference (VaDE: conv encoder / decoder, 10 latent dimensions, 10 components)."""
    return {
        "data": {"dataset": "mnist", "train_split": "train", "validation_split": "test",
                 "train_batch_size": 128, "val_batch_size": 128},
        "model": {"encoder_net": "ConvEncoder", "decoder_net": "ConvDecoder", "decoder_dist": "Bernoulli",
                  "latent_dim": 10, "num_components": 10,
                  "encoder_net_config": {"conv_layers": [(32, 5, 1), (32, 5, 2), (64, 5, 1), (64, 5, 2), (128, 7, 1)]},
                  "decoder_net_config": {"conv_layers": [(64, 7, 1), (64, 5, 2), (32, 5, 1), (32, 5, 2), (32, 5, 1), (1, 5, 1)]}},
        "pretrain_steps": int(60000 / 128 * 150), "steps": int(60000 / 128 * 300), "validation_freq": 1000,
        "cluster_pred_num_samples": 50, "pretrain_lr": 0.002,
        "lr_schedule": {"init_value": 0.002, "decay_rate": 0.9, "staircase": False, "transition_steps": int(60000 / 128 * 10)},
        "adam": {"eps": 1e-4},
    }


def pm_vade_mnist():
    """configs/pm_vade_mnist.py:4-62 of the reference (partial encoder + AutoregressiveGMM on a frozen VaDE)."""
    m = dict(vade_mnist()["model"], partial_posterior_dist="AutoregressiveGMM",
             partial_posterior_dist_config={"num_components": 10, "residual_blocks": 2, "hidden_units": 256})
    return {
        "data": {"dataset": "mnist", "train_split": "train", "validation_split": "test",
                 "train_batch_size": 128, "val_batch_size": 128},
        "vade_dir": "runs/vade-mnist-20220305-121540", "model": m, "steps": 160000, "validation_freq": 5000,
        "lr_schedule": {"init_value": 0.001, "decay_rate": 0.9, "staircase": False, "transition_steps": int(60000 / 128 * 10)},
    }


def pm_vae_mnist16():
    """configs/pm_vae_mnist16.py:4-53 of the reference (the PM-VAE the lookahead posteriors are trained for; 16 x 16 MNIST)."""
    return {
        "data": {"dataset": "mnist16", "train_split": "train", "validation_split": "test", "train_batch_size": 128,
                 "val_batch_size": 128, "mask_generator": "UniformMaskGenerator", "mask_generator_kwargs": {"bounds": (0.0, 0.2)}},
        "model": {"latent_dim": 10, "encoder_net": "ConvEncoder", "decoder_net": "ConvDecoder", "posterior_dist": "TriLGaussian",
                  "decoder_dist": "Bernoulli",
                  "encoder_net_config": {"conv_layers": [(32, 3, 1), (32, 3, 2), (64, 3, 2), (64, 1, 1)]},
                  "decoder_net_config": {"conv_layers": [(64, 8, 1), (64, 5, 2), (32, 5, 1), (32, 5, 1), (1, 3, 1)]}},
        "steps": 200000, "validation_freq": 10000,
        "lr_schedule": {"init_value": 0.001, "decay_rate": 0.9, "transition_steps": 5000},
    }


def lookahead_mnist16():
    """configs/lookahead_mnist16.py:4-36 of the reference."""
    return {
        "data": {"dataset": "mnist16", "train_split": "train", "validation_split": "test", "train_batch_size": 32,
                 "val_batch_size": 32, "mask_generator": "UniformMaskGenerator", "mask_generator_kwargs": {"bounds": (0.0, 0.20)}},
        "pm_vae_dir": "runs/pm-vae-mnist16-20220302-160842",
        "model": {"lookahead_subsample": 16, "model_samples": 64},
        "steps": 40000, "validation_freq": 5000,
        "lr_schedule": {"init_value": 0.001, "decay_rate": 0.9, "transition_steps": 5000},
    }


def _uci(dataset, event_size):
    c = pm_vae_gas()
    c["data"]["dataset"] = dataset
    c["model"]["decoder_dist_config"]["event_size"] = event_size
    return c


def pm_vae_power():
    """configs/pm_vae_power.py of the reference: the gas configuration with 6 features"""
    return _uci("power", 6)


def pm_vae_hepmass():
    """configs/pm_vae_hepmass.py of the reference: the gas configuration with 21 features"""
    return _uci("hepmass", 21)


def pm_vae_bsds():
    """configs/pm_vae_bsds.py of the reference: 63 features, latent 64, 5-block LayerNorm MLPs, the monotonic beta schedule"""
    c = _uci("bsds", 63)
    c["model"]["latent_dim"] = 64
    for net in ("encoder_net_config", "decoder_net_config"):
        c["model"][net].update(residual_blocks=5, layer_norm=True)
    c["beta"] = {"schedule": "monotonic", "low_value": 0.0, "high_value": 1.0, "transition_steps": 200000, "transition_begin": 30000}
    return c
