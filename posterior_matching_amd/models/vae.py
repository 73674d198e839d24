"""Host-side mirror of posterior_matching/models/vae.py of the reference: PosteriorMatchingVAE.

Same constructor / from_config / __call__ contract (reference vae.py:35-144); the per-step
arithmetic runs in libpmhip.so.  JAX's functional autodiff is replaced by an explicit
`backward(g_rec, g_kl, g_mll)` over the buffers the forward left in HBM.
"""
from __future__ import annotations

import os
from typing import Any, Dict, Mapping, Optional

import torch

from .. import ops
from ..ops import ACT_NONE
from .core import Feat, Module, ParamStore, Workspace
from .distributions import AutoregressiveGMM, Bernoulli, DiagonalGaussian, TriLGaussian, get_distribution
from .networks import get_network


class PosteriorMatchingVAE(Module):
    """A VAE with an extra (partial) encoder trained by Posterior Matching (reference vae.py:16-59)."""

    def __init__(self, latent_dim: int, encoder_net, decoder_net, partial_encoder_net, posterior_dist, decoder_dist,
                 partial_posterior_dist, matching_ll_stop_gradients: bool = False, name: Optional[str] = None,
                 device: Optional[str] = None, seed: int = 1):
        super().__init__(name)
        self.latent_dim = latent_dim
        self.encoder_net, self.posterior_dist = encoder_net, posterior_dist
        self.decoder_net, self.decoder_dist = decoder_net, decoder_dist
        self.partial_encoder_net, self.partial_posterior_dist = partial_encoder_net, partial_posterior_dist
        self._matching_ll_stop_gradients = matching_ll_stop_gradients
        self._device = device
        self._seed = seed
        self.concurrent = True   # run the ELBO and the posterior-matching chains on two HIP streams
        self.lend_wgrad = "enc"  # which weight gradients of the ELBO chain run on the side stream (see backward)
        # encoder layers whose weight gradient stays on the main stream although "enc" is lent: the first layer's (the LAST of
        # the backward pass, a thin lane kernel) - with it lent the side queue ends ~45 us behind the main one
        # (profiles/r04_stamp_timeline_pm_vae.txt); same-box pairs 1.376 / 1.360 vs 1.381 / 1.389 ms
        # (after the decoder's five bias-gradient launches became one - pm_colsum_part_multi - the main queue had room for one
        # more: layers 0 AND 4 kept: 1.310 / 1.315 / 1.318 / 1.320 vs 1.327 / 1.346 / 1.331 / 1.332 ms, four same-box pairs,
        # profiles/r04_ab_enc_keep.txt; "4" alone, "0,1,4", "0,3,4", "0,2", "0,1": no better than "0")
        self.enc_keep_wgrad = "0,4"
        self.store: Optional[ParamStore] = None
        if not isinstance(posterior_dist, (TriLGaussian, DiagonalGaussian)):
            raise NotImplementedError("posterior_dist must be TriLGaussian or DiagonalGaussian")
        if not isinstance(partial_posterior_dist, (AutoregressiveGMM, TriLGaussian, DiagonalGaussian)):
            raise NotImplementedError("partial_posterior_dist must be AutoregressiveGMM, TriLGaussian or DiagonalGaussian")

    @classmethod
    def from_config(cls, config: Mapping[str, Any], name: Optional[str] = None, device: Optional[str] = None,
                    seed: int = 1) -> "PosteriorMatchingVAE":
        """reference vae.py:61-118, including its key quirk: only `partial_posterior_dist[_config]`
        is read, so configs that set `masked_posterior_dist` (configs/pm_vae_gas.py:24-27) get
        partial posterior = posterior_dist.  The reference's in-place mutation of the shared
        `posterior_dist_config` dict is harmless here and not reproduced."""
        encoder_net = get_network(config["encoder_net"], config.get("encoder_net_config"), name="encoder_net")
        decoder_net = get_network(config["decoder_net"], config.get("decoder_net_config"), name="decoder_net")
        partial_encoder_net = get_network(
            config.get("partial_encoder_net", config["encoder_net"]),
            config.get("partial_encoder_net_config", config.get("encoder_net_config")),
            name="partial_encoder_net")
        posterior_dist_config = dict(config.get("posterior_dist_config", {}) or {})
        posterior_dist_config["event_size"] = config["latent_dim"]
        partial_posterior_dist_config = dict(config.get("partial_posterior_dist_config", posterior_dist_config) or {})
        partial_posterior_dist_config["event_size"] = config["latent_dim"]
        posterior_dist = get_distribution(config["posterior_dist"], posterior_dist_config, name="posterior_dist")
        decoder_dist = get_distribution(config["decoder_dist"], config.get("decoder_dist_config"), name="decoder_dist")
        partial_posterior_dist = get_distribution(config.get("partial_posterior_dist", config["posterior_dist"]),
                                                  partial_posterior_dist_config, name="partial_posterior_dist")
        return cls(config["latent_dim"], encoder_net, decoder_net, partial_encoder_net, posterior_dist, decoder_dist,
                   partial_posterior_dist, config.get("matching_ll_stop_gradients", False), name=name, device=device,
                   seed=seed)

    # ------------------------------------------------------------------------------------------
    def init(self, x_shape, device=None, seed: Optional[int] = None) -> None:
        """Creates the parameters for per-example inputs of shape `x_shape` ([H,W,C] or [D]);
        the counterpart of haiku's lazy init at the first call (bax: PRNGKey(seed))."""
        if not torch.cuda.is_available():
            raise RuntimeError("posterior_matching_amd needs an MI355X: there is no CPU fallback path")
        from .. import _lib

        _lib.load()
        device = torch.device(device or self._device or "cuda:0")
        store, ws = ParamStore(), Workspace(device)
        x_shape = tuple(int(s) for s in x_shape)
        # [x*b | b] (vae.py:132-133): image masks are [B,H,W,1], feature masks have the features' shape (masking.py:344-348)
        xb_shape = x_shape[:-1] + ((x_shape[-1] + 1,) if len(x_shape) == 3 else (2 * x_shape[-1],))
        mods = [self.encoder_net, self.posterior_dist, self.decoder_net, self.decoder_dist, self.partial_encoder_net,
                self.partial_posterior_dist]
        for m in mods:
            m.ws = ws
        f = self.encoder_net.build(store, "encoder_net", x_shape)
        self.posterior_dist.build(store, "posterior_dist", f)
        f = self.decoder_net.build(store, "decoder_net", (self.latent_dim,))
        self.decoder_dist.build(store, "decoder_dist", f)
        f = self.partial_encoder_net.build(store, "partial_encoder_net", xb_shape)
        self.partial_posterior_dist.build(store, "partial_posterior_dist", f)
        store.allocate(device, self._seed if seed is None else seed)
        self.store, self.ws, self._x_shape = store, ws, x_shape
        self.attach(store, "")

    @property
    def num_params(self) -> int:
        return self.store.num_params

    def __call__(self, x: torch.Tensor, b: torch.Tensor, is_training: bool = False,
                 eps: Optional[torch.Tensor] = None, early_g_mll: Optional[torch.Tensor] = None,
                 early_g_rec: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """reference vae.py:120-144.  `eps` is the N(0,1) draw behind posterior.sample (required:
        the caller owns the RNG, see trainer.py).  Returns per-example `reconstruction_ll`, `kl`,
        `matching_ll` (device tensors owned by the model, overwritten by the next call).

        early_g_mll [B] (train steps, two-stream mode): d loss / d matching_ll, already on the device when the call
        starts (it is -matching_coef / B whatever the forward computes).  The posterior-matching branch then runs its
        BACKWARD pass on the side stream straight behind its forward pass, beside the decoder, instead of waiting for the
        loss; backward() skips that branch.  The gradient buffer must be zero (or hold gradients to accumulate onto) before
        the call.
        early_g_rec [B] (train steps): d loss / d reconstruction_ll, likewise known up front - a Bernoulli decoder head then
        writes d loss / d logits from the launch that sums the log-likelihood (backward() finds it there)."""
        if self.store is None:
            self.init(x.shape[1:], x.device)
        if eps is None:
            raise ValueError("eps (the reparameterisation noise, [B, latent_dim]) must be given")
        self._xin, self._b = x, b
        # The masked (partial) encoder does not depend on the encoder -> sample -> decoder chain:
        # run it on a second HIP stream so that the two chains fill the chip together (most layers
        # launch only 1-2 workgroups per CU).  Under graph capture this becomes a forked branch.
        main, side = torch.cuda.current_stream(x.device), self._side_stream(x.device)
        ops.wait_stream(side, main)
        with torch.cuda.stream(side):
            xob = self.ws.get("x_o_b", tuple(x.shape[:-1]) + (x.shape[-1] + b.shape[-1],))
            ops.mask_concat(x, b, xob)
            pfeat = self.partial_encoder_net(Feat(xob), is_training=is_training)
        feat = self.encoder_net(Feat(x), is_training=is_training)
        z, kl = self.posterior_dist.sample_and_kl(feat, eps)
        # the posterior-matching log-prob only needs z and the masked encoder's features: it runs on the
        # side stream beside the decoder
        ops.wait_stream(side, main)
        self._z = z
        self._pm_backward_done = False
        mll_ready = None
        with torch.cuda.stream(side):
            mll = self.partial_posterior_dist.log_prob(pfeat, z)
            if early_g_mll is not None and is_training and self.concurrent:
                if getattr(self, "_mll_ready", None) is None:
                    self._mll_ready = torch.cuda.Event()
                mll_ready = self._mll_ready
                ops.record_event(mll_ready, side)      # the loss needs matching_ll, not the backward pass queued behind it
                self._pm_backward(early_g_mll, side)
                self._pm_backward_done = True
        dec = self.decoder_net(Feat(z), is_training=is_training)
        if early_g_rec is not None and is_training and isinstance(self.decoder_dist, Bernoulli):
            rec = self.decoder_dist.log_prob_sum(dec, x, early_g=early_g_rec)
        else:
            rec = self.decoder_dist.log_prob_sum(dec, x)
        if mll_ready is not None:
            ops.wait_event(main, mll_ready)
        else:
            ops.wait_stream(main, side)
        self._z = z
        return {"reconstruction_ll": rec, "kl": kl, "matching_ll": mll}

    # ------------------------------------------------------------------------------------------
    # evaluation paths.  S samples per example are laid out sample-minor on the row axis.
    def _eval_noise(self, B: int, S: int, noise, seed, need_posterior: bool):
        """noise: {"eps" [B,S,k], "gumbel" [B,S,k,nc] (AutoregressiveGMM only), "eps_posterior" [B,S,k]} - drawn on the
        device from Philox streams keyed by `seed` when not given (the reference draws from hk.next_rng_key())."""
        k, dev = self.latent_dim, self.store.device
        noise = dict(noise or {})
        seed = self._seed if seed is None else seed
        step = getattr(self, "_eval_step", None)
        if step is None:
            step = self._eval_step = torch.zeros(1, dtype=torch.int32, device=dev)
        drew = False
        if "eps" not in noise:
            noise["eps"] = torch.empty((B, S, k), device=dev)
            ops.normal_fill(noise["eps"], seed, step, stream_id=11)
            drew = True
        if isinstance(self.partial_posterior_dist, AutoregressiveGMM) and "gumbel" not in noise:
            noise["gumbel"] = torch.empty((B, S, k, self.partial_posterior_dist._num_components), device=dev)
            ops.gumbel_fill(noise["gumbel"], seed, step, stream_id=12)
            drew = True
        if need_posterior and "eps_posterior" not in noise:
            noise["eps_posterior"] = torch.empty((B, S, k), device=dev)
            ops.normal_fill(noise["eps_posterior"], seed, step, stream_id=13)
            drew = True
        if drew:
            ops.counter_increment(step)
        return {n: t.reshape((B * S,) + tuple(t.shape[2:])).contiguous() for n, t in noise.items()}

    def _partial_posterior_samples(self, x_o: torch.Tensor, b: torch.Tensor, nz, S: int):
        xob = self.ws.get("x_o_b", tuple(x_o.shape[:-1]) + (x_o.shape[-1] + b.shape[-1],))
        ops.mask_concat(x_o, b, xob)                      # x_o * b, b  (vae.py:158-159: x_o *= b)
        pfeat = self.partial_encoder_net(Feat(xob), is_training=False)
        pp = self.partial_posterior_dist
        if isinstance(pp, AutoregressiveGMM):
            return pp.sample_n(pfeat, (nz["gumbel"], nz["eps"]), S, "pp")
        return pp.sample_n(pfeat, nz["eps"], S, "pp")

    def impute(self, x_o: torch.Tensor, b: torch.Tensor, num_samples: int = 100, noise=None,
               seed: Optional[int] = None) -> torch.Tensor:
        """reference vae.py:146-169: imputations [num_samples, *x_o.shape]: observed values where b == 1, the decoder
        mean of a partial-posterior sample elsewhere."""
        if self.store is None:
            self.init(x_o.shape[1:], x_o.device)
        B, S = x_o.shape[0], int(num_samples)
        nz = self._eval_noise(B, S, noise, seed, need_posterior=False)
        z, _ = self._partial_posterior_samples(x_o, b, nz, S)
        dec = self.decoder_net(Feat(z), is_training=False)
        mean = self.decoder_dist.mean(dec)
        imp = self.ws.get("imputations", (B, S) + tuple(x_o.shape[1:]))
        imp.view(-1).copy_(mean.reshape(-1))
        ops.impute_blend(x_o, b, imp, lo=1.0, hi=0.0)       # lo > hi: no clipping
        return imp.transpose(0, 1)

    def is_log_prob(self, x: torch.Tensor, b: torch.Tensor, num_samples: int = 100, noise=None,
                    seed: Optional[int] = None):
        """reference vae.py:171-226: importance-sampled (log p(x), log p(x_u | x_o)), each [B]."""
        if self.store is None:
            self.init(x.shape[1:], x.device)
        B, S = x.shape[0], int(num_samples)
        nz = self._eval_noise(B, S, noise, seed, need_posterior=True)
        post, pp, dd = self.posterior_dist, self.partial_posterior_dist, self.decoder_dist
        # q(z | x): samples, log q, log p(z), log p(x | z)
        feat = self.encoder_net(Feat(x), is_training=False)
        z, rep = post.sample_n(feat, nz["eps_posterior"], S, "post")
        log_q = post.log_prob_n(rep, z, "post")
        log_pz = self.ws.get("is_log_pz", (B * S,))
        ops.std_normal_logprob(z, log_pz)
        dec = self.decoder_net(Feat(z), is_training=False)
        log_px = dd.log_prob_rep(dec, x, None, S, "x")
        log_p_x = self.ws.get("is_log_p_x", (B,))
        ops.logmeanexp3(log_px, log_pz, log_q, log_p_x, S)
        # q(z | x_o): the same with the observed dimensions only
        z_o, rep_o = self._partial_posterior_samples(x, b, nz, S)
        log_q_o = pp.log_prob_n(rep_o, z_o, "pp")
        log_pz_o = self.ws.get("is_log_pz_o", (B * S,))
        ops.std_normal_logprob(z_o, log_pz_o)
        dec_o = self.decoder_net(Feat(z_o), is_training=False)
        log_pxo = dd.log_prob_rep(dec_o, x, b, S, "xo")
        log_p_xo = self.ws.get("is_log_p_xo", (B,))
        ops.logmeanexp3(log_pxo, log_pz_o, log_q_o, log_p_xo, S)
        neg = self.ws.get("is_neg_log_p_xo", (B, 1))
        ops.scale_shift(log_p_xo.view(B, 1), -1.0, 0.0, neg)
        out = self.ws.get("is_log_p_xu_xo", (B, 1))
        ops.add_cols(log_p_x.view(B, 1), neg, 0, out)       # log p(x_u | x_o) = log p(x) - log p(x_o)
        return log_p_x, out.view(B)

    def expected_info_gains(self, x: torch.Tensor, b: torch.Tensor, num_samples: int = 100, noise=None,
                            seed: Optional[int] = None) -> torch.Tensor:
        """reference vae.py:228-290: for ONE instance (x, b without a batch axis) the expected drop of the partial
        posterior's entropy when feature i becomes observed, -inf where b == 1; shape [num_features].
        noise: {"eps" [1, S, k]} replaces the device Philox draw (parity mode).  Gaussian partial posteriors only - the
        reference's AutoregressiveGMM is a tfd.Autoregressive, whose entropy() is not implemented, so it raises there too."""
        pp = self.partial_posterior_dist
        if isinstance(pp, AutoregressiveGMM):
            raise NotImplementedError("expected_info_gains needs partial_posterior.entropy(): not defined for AutoregressiveGMM")
        if self.store is None:
            self.init(x.shape, x.device)
        S, F = int(num_samples), b.numel()
        x1, b1 = x.unsqueeze(0).contiguous(), b.unsqueeze(0).contiguous()
        nz = self._eval_noise(1, S, noise, seed, need_posterior=False)
        z, _ = self._partial_posterior_samples(x1, b1, nz, S)            # S samples of q(z | x_o)
        mean = self.decoder_dist.mean(self.decoder_net(Feat(z), is_training=False))     # [S, ...]: decoder(z).mean()
        mean = mean.reshape((S,) + tuple(x.shape))
        cand = self.ws.get("ig_cand", (F + 1,) + tuple(x.shape[:-1]) + (x.shape[-1] + b.shape[-1],))
        ents = self.ws.get("ig_ents", (S, F + 1))
        tril = isinstance(pp, TriLGaussian)
        for s in range(S):                                               # hk.scan over the samples (vae.py:270-277)
            ops.info_gain_inputs(x1, b1, mean[s], cand)       # where(b == 1, x_o, x_u): x_o = x * b equals x where b == 1
            pfeat = self.partial_encoder_net(Feat(cand), is_training=False)
            ops.gaussian_entropy(pp._linear_fwd(pfeat), ents[s], self.latent_dim, tril)
        gains = self.ws.get("ig_gains", (F,))
        ops.info_gain_finish(ents, b1.reshape(-1), gains)
        return gains

    def _side_stream(self, device) -> "torch.cuda.Stream":
        if not self.concurrent:
            return torch.cuda.current_stream(device)
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=device, priority=int(os.environ.get("PM_SIDE_PRIO", "0")))
        return self._side

    def _pm_backward(self, g_mll: torch.Tensor, side) -> "torch.cuda.Event":
        """backward pass of the posterior-matching branch (AR-GMM / TriL head + partial encoder) on the current (side) stream;
        returns the event behind which d loss / dz of this branch is complete"""
        want_dz = not self._matching_ll_stop_gradients                      # vae.py:136-137
        dz_pm = self.ws.get("dz_matching", self._z.shape) if want_dz else None
        dpenc = self.partial_posterior_dist.backward_log_prob(g_mll, dz_pm)
        if getattr(self, "_dz_ready", None) is None:
            self._dz_ready = torch.cuda.Event()
        ops.record_event(self._dz_ready, side)
        self.partial_encoder_net.backward(dpenc, need_input_grad=False)
        self.ws.join_aux()
        self.store.grads_ready(["partial_encoder_net", "partial_posterior_dist"])     # data-parallel: bucket is complete
        return self._dz_ready

    def backward(self, g_rec: torch.Tensor, g_kl: torch.Tensor, g_mll: torch.Tensor) -> None:
        """Accumulates d loss / d params into the flat gradient buffer given the per-example
        gradients of the loss w.r.t. the three outputs (zero the buffer first: `zero_grad`)."""
        dev = g_rec.device
        main, side = torch.cuda.current_stream(dev), self._side_stream(dev)
        want_dz = not self._matching_ll_stop_gradients                      # vae.py:136-137
        dz_pm = self.ws.get("dz_matching", self._z.shape) if want_dz else None
        ops.wait_stream(side, main)
        if getattr(self, "_pm_backward_done", False):     # ran behind the branch's forward pass (early_g_mll)
            self._pm_backward_done = False
            dz_ready = self._dz_ready
        else:
            with torch.cuda.stream(side):     # posterior-matching branch: AR-GMM / TriL head + partial encoder
                dz_ready = self._pm_backward(g_mll, side)
        # ELBO branch on the main stream.  It is the longer chain; its weight gradients only feed the optimizer, so
        # (lend_wgrad) they are queued on the side stream behind the posterior-matching branch instead of sitting
        # between the data-gradient kernels of the critical path.
        # lend_wgrad: which of this chain's weight gradients are queued on the side stream (behind the posterior-matching
        # branch, which ends long before this chain does: profiles/r03_*_timeline.txt) instead of sitting between the data-
        # gradient kernels of the critical path.  "enc": the encoder's; "decN": those of the decoder's first N layers (the
        # LAST N of its backward pass); "dec" / "all": every decoder layer (/ and the encoder); "": none.
        lend = os.environ.get("PM_LEND_WGRAD", self.lend_wgrad) if self.concurrent else ""
        parts = [t for t in str(lend).replace("all", "dec,enc").split(",") if t]
        dec_below = 0
        for t in parts:
            if t.startswith("dec"):
                dec_below = int(t[3:]) if t[3:] else 1 << 30
        dpre = self.decoder_dist.backward(g_rec)
        from .networks import ConvDecoder

        if dec_below and isinstance(self.decoder_net, ConvDecoder):
            dz = self.decoder_net.backward(dpre, need_input_grad=True, lend=(side, dec_below))
        else:
            dz = self.decoder_net.backward(dpre, need_input_grad=True)
        if not dec_below:               # the decoder's weight gradients ran on this stream: its bucket is complete (data-parallel)
            self.ws.join_aux()
            self.store.grads_ready(["decoder_net", "decoder_dist"])
        if want_dz:
            ops.wait_event(main, dz_ready)
        if want_dz and isinstance(self.posterior_dist, TriLGaussian):
            denc = self.posterior_dist.backward_sample_kl(dz, g_kl, dz2=dz_pm)       # dz + dz_pm inside the kernel
        else:
            if want_dz:
                ops.axpy1(dz_pm, dz)
            denc = self.posterior_dist.backward_sample_kl(dz, g_kl)
        if "enc" in parts:      # the encoder's weight gradients, issued when the posterior-matching chain has long finished
            self.ws.wgrad_stream = side
            # ... except the layers named here (PM_ENC_KEEP, layer indices): tools/stamp_timeline.py shows both queues busy to
            # the end of the step, the side queue ending ~45 us after the main one with all five lent
            keep = os.environ.get("PM_ENC_KEEP", self.enc_keep_wgrad)
            self.encoder_net.own_stream_wgrad = tuple(int(t) for t in str(keep).split(",") if t.strip() != "")
        self.encoder_net.backward(denc, need_input_grad=False)
        self.ws.wgrad_stream = None
        self.ws.join_aux()
        ops.wait_stream(main, side)

    def zero_grad(self) -> None:
        self.store.zero_grad()

    # checkpoint-style access (reference: TrainState.params pytree)
    def params_dict(self) -> Dict[str, torch.Tensor]:
        return self.store.to_dict("p")

    def grads_dict(self) -> Dict[str, torch.Tensor]:
        return self.store.to_dict("g")

    def load_params(self, values) -> None:
        self.store.load_dict(values)
