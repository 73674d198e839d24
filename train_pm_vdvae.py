#!/usr/bin/env python3
"""Train a Posterior-Matching VDVAE on the MI355X-native path.

Same entry point as the reference's train_pm_vdvae.py (the only data-parallel script of the reference):

    python train_pm_vdvae.py --config configs/pm_vdvae_mnist.py
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train_pm_vdvae.py --config ...

The configured batch size is per device (one process per GPU); gradients are summed over RCCL.  Data
are synthetic raw-pixel (0..255) batches of the dataset's shape, or a .npy given with --data.
"""
import argparse
import json
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from posterior_matching_amd import optim  # noqa: E402
from posterior_matching_amd.config_dict import apply_overrides, load_config_file  # noqa: E402
from posterior_matching_amd.data import make_dataset  # noqa: E402
from posterior_matching_amd.models.vdvae import PosteriorMatchingVDVAE  # noqa: E402
from posterior_matching_amd.parallel import env_world  # noqa: E402
from posterior_matching_amd.trainer import CheckpointCallback, LearningRateLoggerCallback, Trainer, VDVAELoss  # noqa: E402
from posterior_matching_amd.utils import Callback, TensorBoardCallback, configure_environment, make_run_dir  # noqa: E402

configure_environment()


class ReconstructionCallback(Callback):
    """reference train_pm_vdvae.py:33-95: reconstructions and 8 imputations of 8 validation images, logged as
    uint8 strips (the unconditional samples of the reference need forward_prior, which is not built here)."""

    def __init__(self, model, dataset, num_examples: int = 8):
        self._model, self._batches, self._n, self._i = model, dataset.batches, num_examples, 0

    def on_validation_end(self, train_state, step, logs):
        import numpy as np
        import torch

        batch = self._batches[self._i % len(self._batches)]
        self._i += 1
        x, b = batch["image"][:self._n].contiguous(), batch["mask"][:self._n].contiguous()
        m = self._model
        eps = [torch.randn(s, device=x.device) for s in m.eps_shapes(x.shape[0])]
        m(x, b, eps)
        rec = m.reconstruction()
        imp = m.impute(x, b, num_samples=8, seed=random.randint(0, int(2e9)))
        torch.cuda.synchronize()
        x_o = torch.where(b == 1, x, torch.full_like(x, 127.5))
        tiles = imp.permute(0, 2, 1, 3, 4).reshape(x.shape[0], x.shape[1], 8 * x.shape[2], x.shape[3])
        logs["reconstructions"] = torch.cat([x, rec], dim=2).cpu().numpy().astype(np.uint8)
        logs["imputations"] = torch.cat([x, x_o, tiles], dim=2).cpu().numpy().astype(np.uint8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("--data", default=None, help="optional .npy with raw-pixel training examples")
    ap.add_argument("--device_masks", action="store_true",
                    help="draw a fresh mask for every training batch on the GPU (pm_image_mask_mixture & co.) "
                         "instead of cycling host-generated masks")
    args, rest = ap.parse_known_args()
    config = load_config_file(args.config)
    apply_overrides(config, [r[len("--config."):] for r in rest if r.startswith("--config.")])
    if "seed" not in config:
        config.seed = random.randint(0, int(2e9))
    config.lock()

    rank, local_rank, world = env_world()
    import numpy as np
    import torch

    device = torch.device("cuda", local_rank)
    arrays = np.load(args.data) if args.data else None
    # load_datasets(config.data, normalize_images=False) (train_pm_vdvae.py:107): raw 0..255 pixel values
    train_dataset = make_dataset(config.data, config.data.train_batch_size, 64, config.seed + rank, device,
                                     training=True, arrays=arrays, normalize_images=False,
                                     device_masks=args.device_masks)
    val_dataset = make_dataset(config.data, config.data.val_batch_size, 8, config.seed + 10007 + rank, device,
                                   training=False, arrays=arrays, normalize_images=False)

    model = PosteriorMatchingVDVAE(**config.model, device=device, seed=config.seed)
    loss_fn = VDVAELoss(config, model)                      # loss_fn of the reference's train_pm_vdvae.py:109-120

    warm_up_steps = config.get("warm_up", 0)
    if warm_up_steps > 0:
        schedule = optim.linear_schedule(0, config.lr, warm_up_steps)
    else:
        schedule = optim.constant_schedule(config.lr)
    optimizer = optim.chain(
        optim.clip_by_global_norm(config.gradient_clip),
        optim.scale_by_adam(**config.get("adam", {})),
        optim.add_decayed_weights(config.get("weight_decay", 0.0), mask="ndim != 1"),
        optim.scale_by_schedule(schedule),
        optim.scale(-1),
    )

    trainer = Trainer(loss_fn, optimizer, seed=config.seed, num_devices=world, skip_nonfinite_updates=True,
                      ema_rate=config.get("ema_rate", 0.999), use_ema_for_eval=True)

    run_dir = make_run_dir(prefix=f"pm-vdvae-{config.data.dataset}")
    if rank == 0:
        print("Using run directory:", run_dir)
        with open(os.path.join(run_dir, "model_config.json"), "w") as fp:
            json.dump(config.model.to_dict(), fp)

    callbacks = [
        CheckpointCallback(os.path.join(run_dir, "train_state.pkl")),
        ReconstructionCallback(model, val_dataset),
        LearningRateLoggerCallback(schedule),
        TensorBoardCallback(os.path.join(run_dir, "tb")),
    ]
    trainer.fit(train_dataset, config.steps, val_dataset=val_dataset, validation_freq=config.validation_freq,
                callbacks=callbacks)


if __name__ == "__main__":
    main()
