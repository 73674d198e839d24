"""Lookahead posteriors for active feature acquisition (reference posterior_matching/models/lookahead.py:14-227): same class
names and constructor arguments; the frozen PM-VAE is evaluated through the engine's layers, the model-specific steps are the
kernels of csrc/pm_lookahead.hip.

Rows of the sampled tensors are (example b, model sample z, subsampled feature s) - the reference stacks the model samples in
front; the averages it takes do not depend on the order.  The PM-VAE's parameters live on `pm_vae.store`, the lookahead encoder's
(the only trainable modules of train_lookahead_posterior.py:61-62) on `self.store`."""
from __future__ import annotations

from typing import Any, Dict, Mapping, Optional, Tuple

import numpy as np
import torch

from .. import ops
from .core import Feat, Module, ParamStore
from .distributions import AutoregressiveGMM, TriLGaussian
from .networks import get_network
from .vae import PosteriorMatchingVAE
from ..ops import LayerGeom


class LookaheadBlock(Module):
    """reference lookahead.py:14-42: Flatten -> hk.Linear(2 * event_size * num_features) -> one diagonal Gaussian per feature:
    params [B, F, 2k] = (loc | raw), scale = softplus(raw) + 1e-5 (applied inside pm_lookahead_ll_* / pm_lookahead_info_gains)."""

    def __init__(self, event_size: int, num_features: int, w_init=None, b_init=None, name: Optional[str] = None):
        super().__init__(name)
        if w_init is not None or b_init is not None:
            raise NotImplementedError("custom initialisers")
        self._event_size, self._num_features = int(event_size), int(num_features)
        self._num_params = 2 * self._event_size

    def build(self, store: ParamStore, prefix: str, in_shape) -> Tuple[int, ...]:
        self.attach(store, prefix)
        fin = int(np.prod(in_shape))
        fout = self._num_params * self._num_features
        self._fin = fin
        self.g_lin = LayerGeom.dense(fin, fout)
        store.add(f"{prefix}/linear/w", (fin, fout), fan_in=fin)
        store.add(f"{prefix}/linear/b", (fout,))
        self._ws = (store.request_split(f"{prefix}/linear/w", self.g_lin, "fwd"),
                    store.request_split(f"{prefix}/linear/w", self.g_lin, "dgrad"))
        return (self._num_features, self._num_params)

    def __call__(self, feat: Feat) -> torch.Tensor:
        B = feat.t.shape[0]
        self._feat = feat
        out = self.buf("params", (B, self._num_features, self._num_params))
        ops.layer_forward(self.g_lin, feat.t.view(B, self._fin), self.P("linear/w"), self.P("linear/b"), out.view(B, -1),
                          in_act=feat.in_act, wsplit=self.store.split_view(self._ws[0]))
        return out

    def backward(self, dparams: torch.Tensor) -> torch.Tensor:
        """accumulates the Linear's gradients; returns d / d(features) (shaped like the features)"""
        feat = self._feat
        B = feat.t.shape[0]
        flat, dflat = feat.t.view(B, self._fin), dparams.view(B, -1)
        self.wgrad(self.g_lin, flat, dflat, self.G("linear/w"), self.G("linear/b"), in_act=feat.in_act)
        dfeat = self.buf("dfeat", (B, self._fin))
        ops.layer_dgrad(self.g_lin, dflat, self.P("linear/w"), dfeat, aux=flat, aux_act=feat.grad_act,
                        wsplit=self.store.split_view(self._ws[1]))
        return dfeat.view(feat.t.shape)


class LookaheadPosterior:
    """reference lookahead.py:45-227."""

    def __init__(self, pm_vae: PosteriorMatchingVAE, lookahead_encoder_net, num_features: int, lookahead_subsample: int = 16,
                 model_samples: int = 64, name: Optional[str] = None, device: Optional[str] = None, seed: int = 1):
        self.name = name or "lookahead_posterior"
        self.pm_vae = pm_vae
        self.lookahead_encoder_net = lookahead_encoder_net
        self.lookahead_block = LookaheadBlock(pm_vae.latent_dim, num_features, name="lookahead_block")
        self._num_features, self._lookahead_subsample, self._model_samples = int(num_features), int(lookahead_subsample), int(model_samples)
        if self._lookahead_subsample > self._num_features:
            raise ValueError("lookahead_subsample exceeds num_features (jax.random.choice(..., replace=False) raises too)")
        self.store: Optional[ParamStore] = None
        self._device, self._seed = device, seed
        self._step_dev = None

    @classmethod
    def from_config(cls, config: Mapping[str, Any], pm_vae_config: Mapping[str, Any], name: Optional[str] = None,
                    device: Optional[str] = None, seed: int = 1) -> "LookaheadPosterior":
        """reference lookahead.py:84-126"""
        pm_vae = PosteriorMatchingVAE.from_config(pm_vae_config, device=device, seed=seed)
        net = get_network(config.get("lookahead_encoder_net", pm_vae_config["encoder_net"]),
                          config.get("lookahead_encoder_net_config", pm_vae_config.get("encoder_net_config")),
                          name="lookahead_encoder_net")
        return cls(pm_vae, net, config["num_features"], config.get("lookahead_subsample", 16), config.get("model_samples", 64),
                   name=name, device=device, seed=seed)

    @property
    def latent_dim(self) -> int:
        return self.pm_vae.latent_dim

    def init(self, x_shape, device=None, seed: Optional[int] = None) -> None:
        vae = self.pm_vae
        if vae.store is None:
            vae.init(x_shape, device or self._device, seed)
        x_shape = tuple(int(s) for s in x_shape)
        if len(x_shape) != 3:
            raise NotImplementedError("LookaheadPosterior: image data ([H, W, C] with a [H, W, 1] mask) only - the one reference "
                                      "config is configs/lookahead_mnist16.py")
        if self._num_features != x_shape[0] * x_shape[1]:
            raise ValueError(f"num_features = {self._num_features} but the mask has {x_shape[0] * x_shape[1]} entries")
        self._x_shape = x_shape
        xb_shape = x_shape[:-1] + (x_shape[-1] + 1,)
        store = ParamStore()
        self.ws = vae.ws
        self.lookahead_encoder_net.ws = self.lookahead_block.ws = self.ws
        f = self.lookahead_encoder_net.build(store, "lookahead_encoder_net", xb_shape)
        self.lookahead_block.build(store, "lookahead_block", f)
        store.allocate(vae.store.device, (self._seed if seed is None else seed) + 2)
        self.store = store

    # ------------------------------------------------------------------------------------------------------------------------
    def _counter(self) -> torch.Tensor:
        if self._step_dev is None:
            self._step_dev = torch.zeros(1, dtype=torch.int32, device=self.store.device)
        return self._step_dev

    def _noise(self, B: int, noise, seed: int) -> Dict[str, torch.Tensor]:
        """device Philox draws for whatever `noise` does not hold (the reference draws from hk.next_rng_key())"""
        Z, S, k, dev = self._model_samples, self._lookahead_subsample, self.latent_dim, self.store.device
        noise = dict(noise or {})
        pp = self.pm_vae.partial_posterior_dist
        want = {"eps": (B, Z, k), "eps_look": (B, Z, S, k)}
        if isinstance(pp, AutoregressiveGMM):
            want["gumbel"] = (B, Z, k, pp._num_components)
            want["gumbel_look"] = (B, Z, S, k, pp._num_components)
        step, drew = self._counter(), False
        for i, (name, shape) in enumerate(want.items()):
            if name not in noise:
                t = self.ws.get(f"lookahead/noise_{name}", shape)
                (ops.gumbel_fill if name.startswith("gumbel") else ops.normal_fill)(t, seed, step, stream_id=31 + i)
                noise[name], drew = t, True
        if drew:
            ops.counter_increment(step)
        return noise

    def draw_indices(self, rng: np.random.Generator) -> torch.Tensor:
        """jax.random.choice(key, num_features, (lookahead_subsample,), replace=False) (lookahead.py:151-156) with the host's
        generator: int32 [S] on the device"""
        inds = rng.choice(self._num_features, size=self._lookahead_subsample, replace=False).astype(np.int32)
        return torch.from_numpy(inds).to(self.store.device)

    def model_one_step_z(self, x: torch.Tensor, b: torch.Tensor, noise, inds: torch.Tensor) -> torch.Tensor:
        """lookahead.py:128-185 (all of it under stop_gradient): z ~ q(z | x_o) -> decoder mean -> the S lookahead masks ->
        z' ~ q(z | x_o, x_i); returns [B, Z, S, k]"""
        vae = self.pm_vae
        B, Z, S, k = x.shape[0], self._model_samples, self._lookahead_subsample, self.latent_dim
        flat = lambda t, lead: t.reshape((-1,) + tuple(t.shape[lead:])).contiguous()   # noqa: E731  (merge `lead` batch axes)
        nz = {"eps": flat(noise["eps"], 2)}
        if "gumbel" in noise:
            nz["gumbel"] = flat(noise["gumbel"], 2)
        z, _ = vae._partial_posterior_samples(x, b, nz, Z)                               # rows b*Z + z
        mean = vae.decoder_dist.mean(vae.decoder_net(Feat(z), is_training=False))       # decoder(z).mean()
        imp = self.ws.get("lookahead/samples", (B, Z) + tuple(x.shape[1:]))
        imp.view(-1).copy_(mean.reshape(-1))
        ops.impute_blend(x, b, imp, lo=1.0, hi=0.0)                                     # where(b == 1, x_o, sample): no clipping
        C_ = x.shape[-1]
        xl = self.ws.get("lookahead/x_look_b", (B * Z * S,) + tuple(x.shape[1:-1]) + (C_ + 1,))
        ops.lookahead_inputs(imp, b, inds, xl)                                          # [x_look | b_look], already masked
        pfeat = vae.partial_encoder_net(Feat(xl), is_training=False)
        pp = vae.partial_posterior_dist
        e2 = flat(noise["eps_look"], 3)
        if isinstance(pp, AutoregressiveGMM):
            z2, _ = pp.sample_n(pfeat, (flat(noise["gumbel_look"], 3), e2), 1, "look")
        else:
            z2, _ = pp.sample_n(pfeat, e2, 1, "look")
        return z2.view(B, Z, S, k)

    def __call__(self, x: torch.Tensor, b: torch.Tensor, is_training: bool = False, noise=None, inds: Optional[torch.Tensor] = None,
                 seed: int = 0) -> torch.Tensor:
        """lookahead.py:128-203 -> lookahead_lls [B].  noise / inds: explicit draws (parity mode), see oracle/lookahead_oracle.py"""
        if self.store is None:
            self.init(x.shape[1:], x.device)
        B = x.shape[0]
        noise = self._noise(B, noise, seed)
        if inds is None:
            inds = self.draw_indices(np.random.default_rng(seed))
        zs = self.model_one_step_z(x, b, noise, inds)
        xob = self.ws.get("lookahead/x_o_b", tuple(x.shape[:-1]) + (x.shape[-1] + b.shape[-1],))
        ops.mask_concat(x, b, xob)
        params = self.lookahead_block(self.lookahead_encoder_net(Feat(xob), is_training=is_training))
        ll = self.ws.get("lookahead/lls", (B,))
        ops.lookahead_ll_fwd(params, inds, zs, b, ll)
        self._saved = (params, inds, zs, b)
        return ll

    def backward(self, g: torch.Tensor) -> None:
        """accumulates d (sum_b g[b] lookahead_lls[b]) / d (lookahead encoder parameters) into self.store's gradient buffer"""
        params, inds, zs, b = self._saved
        dparams = self.ws.get("lookahead/dparams", tuple(params.shape))
        ops.lookahead_ll_bwd(params, inds, zs, b, g, dparams)
        dfeat = self.lookahead_block.backward(dparams)
        self.lookahead_encoder_net.backward(dfeat, need_input_grad=False)
        self.ws.join_aux()

    def expected_info_gains(self, x: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        """lookahead.py:205-227 for ONE instance (x, b without a batch axis): entropy of q(z | x) minus the entropy of every
        lookahead posterior, -inf where b == 1; [num_features]"""
        if self.store is None:
            self.init(x.shape, x.device)
        vae = self.pm_vae
        x1, b1 = x.unsqueeze(0).contiguous(), b.unsqueeze(0).contiguous()
        post = vae.posterior_dist
        cur = self.ws.get("lookahead/cur_ent", (1,))
        ops.gaussian_entropy(post._linear_fwd(vae.encoder_net(Feat(x1), is_training=False)), cur, self.latent_dim,
                             isinstance(post, TriLGaussian))
        xob = self.ws.get("lookahead/x_o_b", tuple(x1.shape[:-1]) + (x1.shape[-1] + b1.shape[-1],))
        ops.mask_concat(x1, b1, xob)
        params = self.lookahead_block(self.lookahead_encoder_net(Feat(xob), is_training=False))
        gains = self.ws.get("lookahead/gains", (self._num_features,))
        ops.lookahead_info_gains(params[0], cur, b1.reshape(-1), gains)
        return gains

    # ------------------------------------------------------------------------------------------------------------------------
    def params_dict(self) -> Dict[str, torch.Tensor]:
        out = dict(self.pm_vae.params_dict())
        out.update(self.store.to_dict("p"))
        return out

    def load_params(self, values, require_trainable: bool = False) -> None:
        """The frozen PM-VAE must be covered completely (KeyError otherwise: a checkpoint with another prefix would leave it at
        random values and every number downstream would still look plausible); the lookahead networks may be absent (training
        starts them from scratch) unless `require_trainable` (evaluation).  Unused checkpoint keys are reported."""
        import warnings

        used = self.pm_vae.store.load_matching(values, True, "frozen PM-VAE parameters")
        if self.store is not None:
            used |= self.store.load_matching(values, require_trainable, "lookahead parameters")
        extra = [k for k in values if k not in used]
        if extra:
            warnings.warn(f"LookaheadPosterior.load_params: {len(extra)} checkpoint entries were not used: {extra[:6]}")
