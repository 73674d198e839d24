"""Host-side mirrors of posterior_matching/models/vade.py of the reference: VADE and PosteriorMatchingVADE.

Same constructor / from_config contract and method names (`elbo`, `predict_cluster`, `posterior_matching_ll`,
`partial_predict_cluster`; reference vade.py:18-265); every arithmetic operation runs in libpmhip.so.  The encoder /
decoder networks, the DiagonalGaussian head and the AutoregressiveGMM partial posterior are the PM-VAE's; what is new is the
mixture prior (csrc/pm_vade.hip).  VADE.elbo's five terms collapse to  log p(x | z) + log p(z) - log q(z | x)  with the mixture
marginal p(z) = sum_c pi_c N(z; mu_c, diag exp(log_scale_c)^2)  (oracle/vade_oracle.py states the ELBO term by term, as the
reference writes it, and checks the identity).  JAX's autodiff is replaced by explicit backward passes; the random draws behind
hk.next_rng_key() are explicit `eps` tensors or device Philox draws.
"""
from __future__ import annotations

from typing import Any, Dict, Mapping, Optional

import torch

from .. import ops
from .core import Feat, Module, ParamStore, Workspace
from .distributions import AutoregressiveGMM, DiagonalGaussian, get_distribution
from .networks import get_network


class VADE(Module):
    """The Variational Deep Embedding model (reference vade.py:18-150)."""

    def __init__(self, num_components: int, latent_dim: int, encoder_net, decoder_net, decoder_dist,
                 device: Optional[str] = None, seed: int = 1):
        super().__init__("vade")
        if num_components > 64:
            raise NotImplementedError("the mixture kernels hold one component per lane: num_components <= 64")
        self.num_components, self.latent_dim = int(num_components), int(latent_dim)
        self.encoder_net, self.decoder_net, self.decoder_dist = encoder_net, decoder_net, decoder_dist
        self.posterior_dist = DiagonalGaussian(latent_dim, name="posterior_dist")      # hk.Sequential([encoder_net, DiagonalGaussian])
        self._device, self._seed = device, seed
        self.store: Optional[ParamStore] = None

    @classmethod
    def from_config(cls, config: Mapping[str, Any], device: Optional[str] = None, seed: int = 1) -> "VADE":
        """reference vade.py:67-94"""
        encoder_net = get_network(config["encoder_net"], config.get("encoder_net_config"), name="encoder_net")
        decoder_net = get_network(config["decoder_net"], config.get("decoder_net_config"), name="decoder_net")
        decoder_dist = get_distribution(config["decoder_dist"], config.get("decoder_dist_config"), name="decoder_dist")
        return cls(config["num_components"], config["latent_dim"], encoder_net, decoder_net, decoder_dist, device=device,
                   seed=seed)

    # -- parameters -----------------------------------------------------------------------------------------------------------
    def _build(self, store: ParamStore, ws: Workspace, x_shape) -> None:
        C, k = self.num_components, self.latent_dim
        store.add("vade/logits", (C,), fan_in=0)          # hk.initializers.Constant(0)   (vade.py:40-44)
        store.add("vade/mu", (C, k), fan_in=-2)           # hk.initializers.RandomNormal()   (:45-49)
        store.add("vade/log_scale", (C, k), fan_in=-2)    # (:50-54)
        for m in (self.encoder_net, self.posterior_dist, self.decoder_net, self.decoder_dist):
            m.ws = ws
        f = self.encoder_net.build(store, "encoder_net", x_shape)
        self.posterior_dist.build(store, "posterior_dist", f)
        f = self.decoder_net.build(store, "decoder_net", (k,))
        self.decoder_dist.build(store, "decoder_dist", f)

    def init(self, x_shape, device=None, seed: Optional[int] = None) -> None:
        if not torch.cuda.is_available():
            raise RuntimeError("posterior_matching_amd needs an MI355X: there is no CPU fallback path")
        from .. import _lib

        _lib.load()
        device = torch.device(device or self._device or "cuda:0")
        store, ws = ParamStore(), Workspace(device)
        x_shape = tuple(int(s) for s in x_shape)
        self._build(store, ws, x_shape)
        store.allocate(device, self._seed if seed is None else seed)
        self.store, self.ws, self._x_shape = store, ws, x_shape
        self.attach(store, "")

    @property
    def num_params(self) -> int:
        return self.store.num_params

    def _mix(self, kind: str = "p"):
        d = self.store.p if kind == "p" else self.store.g
        return d["vade/mu"], d["vade/log_scale"], d["vade/logits"]

    def _step_counter(self) -> torch.Tensor:
        if getattr(self, "_eval_step", None) is None:
            self._eval_step = torch.zeros(1, dtype=torch.int32, device=self.store.device)
        return self._eval_step

    def _draw(self, shape, seed: int, stream_id: int) -> torch.Tensor:
        step = self._step_counter()
        eps = torch.empty(shape, device=self.store.device)
        ops.normal_fill(eps, seed, step, stream_id=stream_id)
        ops.counter_increment(step)
        return eps

    # -- vade.py:117-150 ----------------------------------------------------------------------------------------------------
    def elbo(self, x: torch.Tensor, eps: Optional[torch.Tensor] = None, is_training: bool = False, seed: int = 0) -> torch.Tensor:
        """The VaDE evidence lower bound of x, [B].  eps [B, latent_dim]: the N(0,1) draw behind posterior.sample (device
        Philox draw when None)."""
        if self.store is None:
            self.init(x.shape[1:], x.device)
        B = x.shape[0]
        if eps is None:
            eps = self._draw((B, self.latent_dim), seed, 21)
        feat = self.encoder_net(Feat(x), is_training=is_training)
        z, lq = self.posterior_dist.sample_and_log_prob(feat, eps)
        dec = self.decoder_net(Feat(z), is_training=is_training)
        rec = self.decoder_dist.log_prob_sum(dec, x)
        lpz = self.ws.get("vade/log_p_z", (B,))
        mu, ls, logits = self._mix()
        ops.vade_prior_fwd(z, mu, ls, logits, lpz)
        out, neg = self.ws.get("vade/elbo", (B,)), self.ws.get("vade/neg_lq", (B,))
        ops.scale_shift(lq, -1.0, 0.0, neg)
        ops.scale_shift(rec, 1.0, 0.0, out)
        ops.axpy1(lpz, out)
        ops.axpy1(neg, out)
        self._z, self._x = z, x
        return out

    def backward_elbo(self, g: torch.Tensor) -> None:
        """accumulates d (sum_b g[b] elbo[b]) / d params into the flat gradient buffer (zero it first)"""
        B = g.shape[0]
        dpre = self.decoder_dist.backward(g)
        dz = self.decoder_net.backward(dpre, need_input_grad=True)
        dz_p = self.ws.get("vade/dz_prior", tuple(self._z.shape))
        gmu, gls, glogits = self._mix("g")
        mu, ls, logits = self._mix()
        ops.vade_prior_bwd(self._z, mu, ls, logits, g, dz_p, gmu, gls, glogits)
        ops.axpy1(dz_p, dz)
        g_lq = self.ws.get("vade/g_lq", (B,))
        ops.scale_shift(g, -1.0, 0.0, g_lq)                      # log q enters the ELBO with a minus sign
        denc = self.posterior_dist.backward_sample_log_prob(dz, g_lq)
        self.encoder_net.backward(denc, need_input_grad=False)

    # -- train_vade.py:45-49: the pre-training autoencoder ----------------------------------------------------------------------
    def reconstruction_ll_at_mean(self, x: torch.Tensor, is_training: bool = False) -> torch.Tensor:
        """decoder(encoder(x).mean()).log_prob(x), [B]"""
        if self.store is None:
            self.init(x.shape[1:], x.device)
        feat = self.encoder_net(Feat(x), is_training=is_training)
        loc = self.posterior_dist.mean(feat)
        dec = self.decoder_net(Feat(loc), is_training=is_training)
        return self.decoder_dist.log_prob_sum(dec, x)

    def backward_reconstruction_at_mean(self, g: torch.Tensor) -> None:
        dpre = self.decoder_dist.backward(g)
        dloc = self.decoder_net.backward(dpre, need_input_grad=True)
        denc = self.posterior_dist.backward_mean(dloc)
        self.encoder_net.backward(denc, need_input_grad=False)

    def encode_mean(self, x: torch.Tensor) -> torch.Tensor:
        """encoder(x).mean() [B, latent_dim] (train_vade.py:63-65: the latents the initial GMM is fitted on)"""
        if self.store is None:
            self.init(x.shape[1:], x.device)
        return self.posterior_dist.mean(self.encoder_net(Feat(x), is_training=False))

    # -- vade.py:96-115 -------------------------------------------------------------------------------------------------------
    def predict_cluster(self, x: torch.Tensor, num_samples: int = 10, eps: Optional[torch.Tensor] = None,
                        seed: int = 0) -> torch.Tensor:
        """q(c | x) [B, num_components]: mean over `num_samples` posterior samples of softmax_c(log p(z | c) + log pi_c).
        eps [B, num_samples, latent_dim] (device Philox draw when None)."""
        if self.store is None:
            self.init(x.shape[1:], x.device)
        B, S, k = x.shape[0], int(num_samples), self.latent_dim
        if eps is None:
            eps = self._draw((B, S, k), seed, 22)
        feat = self.encoder_net(Feat(x), is_training=False)
        z, _ = self.posterior_dist.sample_n(feat, eps.reshape(B * S, k).contiguous(), S, "cluster")
        probs = self.ws.get("vade/cluster_probs", (B, self.num_components))
        mu, ls, logits = self._mix()
        ops.vade_cluster_probs(z, mu, ls, logits, probs, S)
        return probs

    def zero_grad(self) -> None:
        self.store.zero_grad()

    def params_dict(self) -> Dict[str, torch.Tensor]:
        return self.store.to_dict("p")

    def grads_dict(self) -> Dict[str, torch.Tensor]:
        return self.store.to_dict("g")

    def load_params(self, values, require_all: bool = False) -> None:
        """require_all: every parameter of the VaDE must be in `values` (KeyError otherwise)"""
        import warnings

        used = self.store.load_matching(values, require_all, "VaDE parameters")
        extra = [k for k in values if k not in used]
        if extra:
            warnings.warn(f"VADE.load_params: {len(extra)} checkpoint entries were not used: {extra[:6]}")


class PosteriorMatchingVADE(VADE):
    """A VADE with an additional Posterior Matching (partial) encoder (reference vade.py:153-265).  The VaDE's own parameters
    live on `self.store`; the partial encoder and its distribution on `self.partial_store` - the only trainable modules of
    train_pm_vade.py (its trainable_predicate: "partial_" in module_name), so the optimizer only ever sees that store."""

    def __init__(self, num_components: int, latent_dim: int, encoder_net, partial_encoder_net, partial_posterior_dist,
                 decoder_net, decoder_dist, device: Optional[str] = None, seed: int = 1):
        super().__init__(num_components, latent_dim, encoder_net, decoder_net, decoder_dist, device=device, seed=seed)
        self.partial_encoder_net, self.partial_posterior_dist = partial_encoder_net, partial_posterior_dist
        self.partial_store: Optional[ParamStore] = None

    @classmethod
    def from_config(cls, config: Mapping[str, Any], device: Optional[str] = None, seed: int = 1) -> "PosteriorMatchingVADE":
        """reference vade.py:183-223"""
        encoder_net = get_network(config["encoder_net"], config.get("encoder_net_config"), name="encoder_net")
        partial_encoder_net = get_network(config.get("partial_encoder_net", config["encoder_net"]),
                                          config.get("partial_encoder_net_config", config.get("encoder_net_config")),
                                          name="partial_encoder_net")
        pp_cfg = dict(config.get("partial_posterior_dist_config") or {})
        pp_cfg["event_size"] = config["latent_dim"]
        partial_posterior_dist = get_distribution(config.get("partial_posterior_dist", "TriLGaussian"), pp_cfg,
                                                  name="partial_posterior_dist")
        decoder_net = get_network(config["decoder_net"], config.get("decoder_net_config"), name="decoder_net")
        decoder_dist = get_distribution(config["decoder_dist"], config.get("decoder_dist_config"), name="decoder_dist")
        return cls(config["num_components"], config["latent_dim"], encoder_net, partial_encoder_net, partial_posterior_dist,
                   decoder_net, decoder_dist, device=device, seed=seed)

    def init(self, x_shape, device=None, seed: Optional[int] = None) -> None:
        super().init(x_shape, device, seed)
        x_shape = self._x_shape
        # [x*b | b] (vade.py:236, 257): image masks are [B,H,W,1], feature masks have the features' shape
        xb_shape = x_shape[:-1] + ((x_shape[-1] + 1,) if len(x_shape) == 3 else (2 * x_shape[-1],))
        store = ParamStore()
        self.partial_encoder_net.ws = self.partial_posterior_dist.ws = self.ws
        f = self.partial_encoder_net.build(store, "partial_encoder_net", xb_shape)
        self.partial_posterior_dist.build(store, "partial_posterior_dist", f)
        store.allocate(self.store.device, (self._seed if seed is None else seed) + 1)
        self.partial_store = store

    def _partial_feat(self, x: torch.Tensor, b: torch.Tensor, is_training: bool) -> Feat:
        xob = self.ws.get("vade/x_o_b", tuple(x.shape[:-1]) + (x.shape[-1] + b.shape[-1],))
        ops.mask_concat(x, b, xob)
        return self.partial_encoder_net(Feat(xob), is_training=is_training)

    # -- vade.py:247-265 ----------------------------------------------------------------------------------------------------
    def posterior_matching_ll(self, x: torch.Tensor, b: torch.Tensor, eps: Optional[torch.Tensor] = None,
                              is_training: bool = False, seed: int = 0) -> torch.Tensor:
        """log q(z | x_o) [B] of a sample z ~ q(z | x) of the full encoder (no gradient flows into z: jax.lax.stop_gradient)"""
        if self.store is None:
            self.init(x.shape[1:], x.device)
        B = x.shape[0]
        if eps is None:
            eps = self._draw((B, self.latent_dim), seed, 23)
        feat = self.encoder_net(Feat(x), is_training=False)           # frozen module (train_pm_vade.py:59-60)
        z, _ = self.posterior_dist.sample_and_kl(feat, eps)
        pfeat = self._partial_feat(x, b, is_training)
        return self.partial_posterior_dist.log_prob(pfeat, z)

    def backward_posterior_matching_ll(self, g: torch.Tensor) -> None:
        """accumulates d (sum_b g[b] ll[b]) / d (partial encoder parameters) into partial_store's gradient buffer"""
        dpenc = self.partial_posterior_dist.backward_log_prob(g, None)
        self.partial_encoder_net.backward(dpenc, need_input_grad=False)

    # -- vade.py:225-245 ----------------------------------------------------------------------------------------------------
    def partial_predict_cluster(self, x: torch.Tensor, b: torch.Tensor, num_samples: int = 10, noise=None,
                                seed: int = 0) -> torch.Tensor:
        """q(c | x_o) [B, num_components] from `num_samples` samples of the partial posterior.  noise: {"eps" [B,S,k],
        "gumbel" [B,S,k,nc] (AutoregressiveGMM only)}; device Philox draws when absent."""
        if self.store is None:
            self.init(x.shape[1:], x.device)
        B, S, k = x.shape[0], int(num_samples), self.latent_dim
        noise = dict(noise or {})
        pp = self.partial_posterior_dist
        if "eps" not in noise:
            noise["eps"] = self._draw((B, S, k), seed, 24)
        flat = lambda t: t.reshape((B * S,) + tuple(t.shape[2:])).contiguous()   # noqa: E731
        pfeat = self._partial_feat(x, b, False)
        if isinstance(pp, AutoregressiveGMM):
            if "gumbel" not in noise:
                gm = torch.empty((B, S, k, pp._num_components), device=x.device)
                step = self._step_counter()
                ops.gumbel_fill(gm, seed, step, stream_id=25)
                ops.counter_increment(step)
                noise["gumbel"] = gm
            z, _ = pp.sample_n(pfeat, (flat(noise["gumbel"]), flat(noise["eps"])), S, "pcluster")
        else:
            z, _ = pp.sample_n(pfeat, flat(noise["eps"]), S, "pcluster")
        probs = self.ws.get("vade/partial_cluster_probs", (B, self.num_components))
        mu, ls, logits = self._mix()
        ops.vade_cluster_probs(z, mu, ls, logits, probs, S)
        return probs

    def partial_params_dict(self) -> Dict[str, torch.Tensor]:
        return self.partial_store.to_dict("p")

    def load_params(self, values, require_trainable: bool = False) -> None:
        """The frozen VaDE must be covered completely (KeyError otherwise); the partial encoder may be absent unless
        `require_trainable` (evaluation of a trained PM-VaDE).  Unused checkpoint keys are reported."""
        import warnings

        used = self.store.load_matching(values, True, "frozen VaDE parameters")
        if self.partial_store is not None:
            used |= self.partial_store.load_matching(values, require_trainable, "partial-encoder parameters")
        extra = [k for k in values if k not in used]
        if extra:
            warnings.warn(f"PosteriorMatchingVADE.load_params: {len(extra)} checkpoint entries were not used: {extra[:6]}")
