"""Host-side mirrors of posterior_matching/models/distributions.py of the reference.

The reference's heads return TFP distribution objects and the model calls .sample /
.kl_divergence / .log_prob on them (vae.py:124-138).  Here each head exposes exactly those
uses as fused HIP calls plus their explicit backward.
"""
from __future__ import annotations

from typing import Any, Mapping, Optional

import torch

from .. import ops
from ..ops import ACT_NONE, ACT_RELU, LayerGeom
from .core import Feat, Module, ParamStore
from .networks import ResidualMLP


class Bernoulli(Module):
    """reference distributions.py:20-25: tfd.Bernoulli(logits = decoder output)."""

    def __init__(self, name: Optional[str] = None):
        super().__init__(name)

    def build(self, store, prefix, feat_shape):
        self.attach(store, prefix)

    def log_prob_sum(self, feat: Feat, x: torch.Tensor, early_g: Optional[torch.Tensor] = None) -> torch.Tensor:
        """early_g [B]: d loss / d ll, already on the device (a train step knows it before the forward pass: -1 / B) - the
        gradient w.r.t. the decoder's last pre-activation then leaves the same launch and backward() only hands it out."""
        if feat.in_act != ACT_NONE:
            raise NotImplementedError("Bernoulli head expects materialised logits (conv decoder)")
        self._feat, self._x = feat, x
        ll = self.buf("ll", (x.shape[0],))
        self._dpre_done = False
        if early_g is not None:
            dpre = self.buf("dpre", feat.t.shape)
            ops.bernoulli_ll_fwd_bwd(feat.t, x, early_g, ll, dpre, feat.grad_act)
            self._dpre_done = True
        else:
            ops.bernoulli_ll_fwd(feat.t, x, ll)
        return ll

    def backward(self, g: torch.Tensor) -> torch.Tensor:
        """returns the gradient w.r.t. the decoder network's last pre-activation."""
        dpre = self.buf("dpre", self._feat.t.shape)
        if getattr(self, "_dpre_done", False):          # written by log_prob_sum(early_g=...) of this forward pass
            self._dpre_done = False
            return dpre
        ops.bernoulli_ll_bwd(self._feat.t, self._x, g, dpre, self._feat.grad_act)
        return dpre

    # -- evaluation paths (vae.py:146-226): rows of `feat` are (example, sample) pairs, sample-minor
    def mean(self, feat: Feat) -> torch.Tensor:
        out = self.buf("mean", feat.t.shape)
        ops.sigmoid(feat.t, out)
        return out

    def log_prob_rep(self, feat: Feat, x: torch.Tensor, w: Optional[torch.Tensor], S: int, tag: str) -> torch.Tensor:
        ll = self.buf(f"ll_rep_{tag}", (x.shape[0] * S,))
        ops.bernoulli_ll_rep_fwd(feat.t, x, w, ll, S)
        return ll


class _LinearHead(Module):
    """Flatten + hk.Linear(out) on a network's features (distributions.py:42-48,73-79,102-108)."""

    def _build_linear(self, store: ParamStore, prefix: str, feat_shape, out: int):
        self.attach(store, prefix)
        fin = 1
        for s in feat_shape:
            fin *= int(s)
        self.geom = LayerGeom.dense(fin, out)
        store.add(f"{prefix}/linear/w", (fin, out), fan_in=fin)
        store.add(f"{prefix}/linear/b", (out,))
        self._ws = (store.request_split(f"{prefix}/linear/w", self.geom, "fwd"),
                    store.request_split(f"{prefix}/linear/w", self.geom, "dgrad"))

    def _linear_fwd(self, feat: Feat) -> torch.Tensor:
        self._feat = feat
        B = feat.t.shape[0]
        out = self.buf("params", (B, self.geom.CO))
        ops.layer_forward(self.geom, feat.t, self.P("linear/w"), self.P("linear/b"), out, in_act=feat.in_act, B=B,
                          wsplit=self.store.split_view(self._ws[0]))
        return out

    def _linear_bwd(self, dparams: torch.Tensor) -> torch.Tensor:
        """returns the gradient w.r.t. the feeding network's last pre-activation."""
        f = self._feat
        B = f.t.shape[0]
        self.wgrad(self.geom, f.t, dparams, self.G("linear/w"), self.G("linear/b"), in_act=f.in_act, B=B)
        dfeat = self.buf("dfeat", (B, self.geom.CI))
        ops.layer_dgrad(self.geom, dparams, self.P("linear/w"), dfeat, aux=f.t, aux_act=f.grad_act, B=B,
                        wsplit=self.store.split_view(self._ws[1]))
        return dfeat.view(f.t.shape)


class IdentityGaussian(_LinearHead):
    """reference distributions.py:28-55: Linear -> loc, one scalar log_scale parameter."""

    def __init__(self, event_size: int, w_init=None, b_init=None, name: Optional[str] = None):
        super().__init__(name)
        if w_init is not None or b_init is not None:
            raise NotImplementedError("custom initialisers")
        self._event_size = event_size

    def build(self, store, prefix, feat_shape):
        self._build_linear(store, prefix, feat_shape, self._event_size)
        store.add(f"{prefix}/log_scale", ())

    def log_prob_sum(self, feat: Feat, x: torch.Tensor) -> torch.Tensor:
        loc = self._linear_fwd(feat)
        self._x, self._loc = x, loc
        ll = self.buf("ll", (x.shape[0],))
        ops.normal_ll_fwd(loc, x, self.P("log_scale"), ll)
        return ll

    def backward(self, g: torch.Tensor) -> torch.Tensor:
        dloc = self.buf("dloc", self._loc.shape)
        ops.normal_ll_bwd(self._loc, self._x, self.P("log_scale"), g, dloc, self.G("log_scale"))
        return self._linear_bwd(dloc)

    def mean(self, feat: Feat) -> torch.Tensor:
        return self._linear_fwd(feat)

    def log_prob_rep(self, feat: Feat, x: torch.Tensor, w: Optional[torch.Tensor], S: int, tag: str) -> torch.Tensor:
        loc = self._linear_fwd(feat)
        ll = self.buf(f"ll_rep_{tag}", (x.shape[0] * S,))
        ops.normal_ll_rep_fwd(loc, x, self.P("log_scale"), w, ll, S)
        return ll


class TriLGaussian(_LinearHead):
    """reference distributions.py:87-113: Linear(k + k(k+1)/2) -> loc, FillScaleTriL."""

    def __init__(self, event_size: int, w_init=None, b_init=None, name: Optional[str] = None):
        super().__init__(name)
        if w_init is not None or b_init is not None:
            raise NotImplementedError("custom initialisers")
        self._event_size = event_size
        self._num_params = event_size + event_size * (event_size + 1) // 2

    def build(self, store, prefix, feat_shape):
        self._build_linear(store, prefix, feat_shape, self._num_params)

    # -- posterior use: sample + KL (vae.py:123-124,130)
    def sample_and_kl(self, feat: Feat, eps: torch.Tensor):
        prm = self._linear_fwd(feat)
        B, k = eps.shape
        self._prm, self._eps = prm, eps
        z, kl = self.buf("z", (B, k)), self.buf("kl", (B,))
        ops.tril_sample_kl_fwd(prm, eps, z, kl)
        return z, kl

    def backward_sample_kl(self, dz: torch.Tensor, g_kl: torch.Tensor, dz2: Optional[torch.Tensor] = None) -> torch.Tensor:
        """dz2: a second gradient w.r.t. z (the posterior-matching branch's), summed inside the kernel"""
        dprm = self.buf("dparams", self._prm.shape)
        ops.tril_sample_kl_bwd(self._prm, self._eps, dz, g_kl, dprm, dz2=dz2)
        return self._linear_bwd(dprm)

    # -- partial-posterior use: log_prob(z) (vae.py:134-138)
    def log_prob(self, feat: Feat, z: torch.Tensor) -> torch.Tensor:
        prm = self._linear_fwd(feat)
        self._prm, self._z = prm, z
        lp = self.buf("lp", (z.shape[0],))
        ops.tril_logprob_fwd(prm, z, lp)
        return lp

    def backward_log_prob(self, g: torch.Tensor, dz: Optional[torch.Tensor]):
        """dz (when given) is WRITTEN with d log_prob / d z; returns d/d(pre-activation) of the feeding network."""
        dprm = self.buf("dparams", self._prm.shape)
        ops.tril_logprob_bwd(self._prm, self._z, g, dprm, dz)
        return self._linear_bwd(dprm)

    # -- evaluation paths (vae.py:146-226): S samples per example, rows b*S + s
    def sample_n(self, feat: Feat, eps: torch.Tensor, S: int, tag: str):
        """eps [B*S, k] -> (z [B*S, k], repeated head parameters for log_prob_n)"""
        prm = self._linear_fwd(feat)
        R, k = eps.shape
        rep = self.buf(f"params_rep_{tag}", (R, prm.shape[1]))
        ops.repeat_rows(prm, rep, S)
        z, kl = self.buf(f"z_n_{tag}", (R, k)), self.buf(f"kl_n_{tag}", (R,))
        ops.tril_sample_kl_fwd(rep, eps, z, kl)
        return z, rep

    def log_prob_n(self, rep: torch.Tensor, z: torch.Tensor, tag: str) -> torch.Tensor:
        lp = self.buf(f"lp_n_{tag}", (z.shape[0],))
        ops.tril_logprob_fwd(rep, z, lp)
        return lp


class DiagonalGaussian(_LinearHead):
    """reference distributions.py:58-84: Linear(2k) -> loc, scale = softplus(raw) + 1e-5, MultivariateNormalDiag.
    Usable as posterior_dist (sample + KL) and as partial_posterior_dist (log_prob), like TriLGaussian."""

    def __init__(self, event_size: int, w_init=None, b_init=None, name: Optional[str] = None):
        super().__init__(name)
        if w_init is not None or b_init is not None:
            raise NotImplementedError("custom initialisers")
        self._event_size = event_size
        self._num_params = 2 * event_size

    def build(self, store, prefix, feat_shape):
        self._build_linear(store, prefix, feat_shape, self._num_params)

    def sample_and_kl(self, feat: Feat, eps: torch.Tensor):
        prm = self._linear_fwd(feat)
        B, k = eps.shape
        self._prm, self._eps = prm, eps
        z, kl = self.buf("z", (B, k)), self.buf("kl", (B,))
        ops.diag_gaussian_sample_kl_fwd(prm, eps, z, kl)
        return z, kl

    def backward_sample_kl(self, dz: torch.Tensor, g_kl: torch.Tensor) -> torch.Tensor:
        dprm = self.buf("dparams", self._prm.shape)
        ops.diag_gaussian_sample_kl_bwd(self._prm, self._eps, dz, g_kl, dprm)
        return self._linear_bwd(dprm)

    def log_prob(self, feat: Feat, z: torch.Tensor) -> torch.Tensor:
        prm = self._linear_fwd(feat)
        self._prm, self._z = prm, z
        lp = self.buf("lp", (z.shape[0],))
        ops.diag_gaussian_logprob_fwd(prm, z, lp)
        return lp

    def backward_log_prob(self, g: torch.Tensor, dz: Optional[torch.Tensor]):
        dprm = self.buf("dparams", self._prm.shape)
        ops.diag_gaussian_logprob_bwd(self._prm, self._z, g, dprm, dz)
        return self._linear_bwd(dprm)

    # -- VaDE's encoder (vade.py:60-62, 128-140): a sample AND the density of that sample under the same parameters
    def sample_and_log_prob(self, feat: Feat, eps: torch.Tensor):
        """z = loc + scale * eps and log q(z | x) [B] (posterior.sample(...), posterior.log_prob(z))"""
        z, _ = self.sample_and_kl(feat, eps)
        self._z = z
        lq = self.buf("lq", (z.shape[0],))
        ops.diag_gaussian_logprob_fwd(self._prm, z, lq)
        return z, lq

    def backward_sample_log_prob(self, dz: torch.Tensor, g_lq: torch.Tensor) -> torch.Tensor:
        """dz: gradient w.r.t. the sample from its other consumers; g_lq [B]: d loss / d log q.  The density's gradient has
        an explicit part (parameters) and a part through z = loc + scale * eps; both go through the existing kernels."""
        dprm = self.buf("dparams", self._prm.shape)
        dz_q = self.buf("dz_lq", self._z.shape)
        ops.diag_gaussian_logprob_bwd(self._prm, self._z, g_lq, dprm, dz_q)          # explicit part, d log q / dz
        ops.axpy1(dz, dz_q)                                                          # total d loss / dz
        dprm_z = self.buf("dparams_z", self._prm.shape)
        zero = self.buf("zero_b", (dz.shape[0],))
        ops.fill_zero(zero)
        ops.diag_gaussian_sample_kl_bwd(self._prm, self._eps, dz_q, zero, dprm_z)    # through the reparameterisation
        ops.axpy1(dprm_z, dprm)
        return self._linear_bwd(dprm)

    def mean(self, feat: Feat) -> torch.Tensor:
        """posterior.mean() = loc [B, k] (train_vade.py:47, 65)"""
        prm = self._linear_fwd(feat)
        self._prm = prm
        k = self._event_size
        loc, zero = self.buf("loc", (prm.shape[0], k)), self.buf("zero_loc", (prm.shape[0], k))
        ops.fill_zero(zero)
        ops.add_cols(zero, prm, 0, loc)                                              # loc = prm[:, :k]
        return loc

    def backward_mean(self, dloc: torch.Tensor) -> torch.Tensor:
        dprm = self.buf("dparams", self._prm.shape)
        ops.fill_zero(dprm)
        ops.copy_cols(dloc, dprm, 0)
        return self._linear_bwd(dprm)

    # -- evaluation paths (vae.py:146-226): S samples per example, rows b*S + s
    def sample_n(self, feat: Feat, eps: torch.Tensor, S: int, tag: str):
        """eps [B*S, k] -> (z [B*S, k], repeated head parameters for log_prob_n)"""
        prm = self._linear_fwd(feat)
        R, k = eps.shape
        rep = self.buf(f"params_rep_{tag}", (R, prm.shape[1]))
        ops.repeat_rows(prm, rep, S)
        z, kl = self.buf(f"z_n_{tag}", (R, k)), self.buf(f"kl_n_{tag}", (R,))
        ops.diag_gaussian_sample_kl_fwd(rep, eps, z, kl)
        return z, rep

    def log_prob_n(self, rep: torch.Tensor, z: torch.Tensor, tag: str) -> torch.Tensor:
        lp = self.buf(f"lp_n_{tag}", (z.shape[0],))
        ops.diag_gaussian_logprob_fwd(rep, z, lp)
        return lp


class AutoregressiveGMM(Module):
    """reference distributions.py:192-223 + the scan of :152-166 + OneDimensionalGMM :116-134.

    The k scan steps are teacher-forced, so they are stacked on the batch axis (row i*B+b = step i
    of example b) and run as ONE batched MLP; the head only computes the 3*num_components columns
    that step i consumes (`out.log_prob(value)[:, i]`, distributions.py:161) as a grouped GEMM.
    """

    def __init__(self, event_size: int, num_components: int = 10, residual_blocks: int = 2,
                 hidden_units: int = 256, name: Optional[str] = None):
        super().__init__(name)
        self._event_dim = event_size
        self._num_components = num_components
        self._residual_blocks = residual_blocks
        self._hidden_units = hidden_units
        self.mlp = ResidualMLP(residual_blocks, hidden_units)

    def build(self, store, prefix, feat_shape):
        self.attach(store, prefix)
        k, nc, hu = self._event_dim, self._num_components, self._hidden_units
        self._ctx_dim = 1
        for s in feat_shape:
            self._ctx_dim *= int(s)
        self.mlp.ws = self.ws
        self.mlp.build(store, f"{prefix}/mlp", (2 * k + self._ctx_dim,))
        store.add(f"{prefix}/gmm/linear/w", (hu, 3 * nc * k), fan_in=hu)
        store.add(f"{prefix}/gmm/linear/b", (3 * nc * k,))
        self.g_head = LayerGeom.dense(hu, 3 * nc)   # one group = one scan step's column slice
        gk = dict(groups=k, w_gs=3 * nc, w_ld=3 * nc * k)
        self._ws_head = (store.request_split(f"{prefix}/gmm/linear/w", self.g_head, "fwd", **gk),
                         store.request_split(f"{prefix}/gmm/linear/w", self.g_head, "dgrad", **gk))

    def _group_kw(self, B):
        k, nc, hu = self._event_dim, self._num_components, self._hidden_units
        return dict(B=B, groups=k, in_gs=B * hu, w_gs=3 * nc, out_gs=B * 3 * nc, bias_gs=3 * nc, w_ld=3 * nc * k)

    def log_prob(self, feat: Feat, z: torch.Tensor) -> torch.Tensor:
        if feat.in_act != ACT_NONE:
            raise NotImplementedError("AutoregressiveGMM expects a materialised context")
        B, k = z.shape
        nc = self._num_components
        self._feat, self._z = feat, z
        inp = self.buf("inp", (k * B, 2 * k + self._ctx_dim))
        ops.argmm_build_input(z, feat.t, inp)
        self.mlp.ws = self.ws
        hfeat = self.mlp(Feat(inp))
        self._hfeat = hfeat
        head = self.buf("head", (k * B, 3 * nc))
        ops.layer_forward(self.g_head, hfeat.t, self.P("gmm/linear/w"), self.P("gmm/linear/b"), head,
                          in_act=hfeat.in_act, wsplit=self.store.split_view(self._ws_head[0]), **self._group_kw(B))
        self._head = head
        mll = self.buf("mll", (B,))
        ops.gmm_logprob_fwd(head, z, mll, nc)
        return mll

    def backward_log_prob(self, g: torch.Tensor, dz: Optional[torch.Tensor]):
        """dz (when given) is WRITTEN with d log_prob / d z (direct + through the scan inputs);
        returns d/d(pre-activation) of the partial encoder."""
        B, k = self._z.shape
        nc, hu = self._num_components, self._hidden_units
        dhead = self.buf("dhead", self._head.shape)
        ops.gmm_logprob_bwd(self._head, self._z, g, dhead, dz, nc, accumulate_dz=False)
        hf = self._hfeat
        self.wgrad(self.g_head, hf.t, dhead, self.G("gmm/linear/w"), self.G("gmm/linear/b"), in_act=hf.in_act,
                        **self._group_kw(B))
        dh = self.mlp.out_grad_buffer(k * B)          # the MLP's grouped weight-gradient launch reads it in place
        ops.layer_dgrad(self.g_head, dhead, self.P("gmm/linear/w"), dh, aux=hf.t, aux_act=hf.grad_act,
                        wsplit=self.store.split_view(self._ws_head[1]), **self._group_kw(B))
        dinp = self.mlp.backward(dh, need_input_grad=True)
        dctx = self.buf("dctx", (B, self._ctx_dim))
        ops.argmm_input_bwd(dinp, dz, dctx, B, k, self._ctx_dim, accumulate_dz=True, ctx=self._feat.t,
                            ctx_act=self._feat.grad_act)
        return dctx.view(self._feat.t.shape)


def _argmm_sample_n(self, feat: Feat, noise, S: int, tag: str):
    """_AutoregressiveDistribution._sample_n (reference distributions.py:168-190): latent dimension i is drawn from
    the network evaluated on z[:, :i].  Every step re-runs the stacked teacher-forced pass and reads step i's rows
    (evaluation path: k network passes per sample, as in the reference's fori_loop).
    noise = (gumbel [B*S, k, nc], eps [B*S, k]) -> (z [B*S, k], repeated context Feat for log_prob_n)"""
    gumbel, eps = noise
    R, k = eps.shape
    nc = self._num_components
    ctx = feat.t.reshape(feat.t.shape[0], -1)
    if feat.in_act != ACT_NONE:
        raise NotImplementedError("AutoregressiveGMM expects a materialised context")
    rep = self.buf(f"ctx_rep_{tag}", (R, ctx.shape[1]))
    ops.repeat_rows(ctx, rep, S)
    z = self.buf(f"z_n_{tag}", (R, k))
    ops.fill_zero(z)
    rfeat = Feat(rep, feat.in_act, feat.grad_act)
    for i in range(k):
        self.log_prob(rfeat, z)                       # leaves the stacked head [k*R, 3nc] in self._head
        ops.gmm_sample_step(self._head, gumbel, eps, z, nc, i)
    return z, rfeat


def _argmm_log_prob_n(self, rfeat: Feat, z: torch.Tensor, tag: str) -> torch.Tensor:
    lp = self.buf(f"lp_n_{tag}", (z.shape[0],))
    lp.copy_(self.log_prob(rfeat, z))
    return lp


AutoregressiveGMM.sample_n = _argmm_sample_n
AutoregressiveGMM.log_prob_n = _argmm_log_prob_n

_DISTRIBUTIONS = {
    "Bernoulli": Bernoulli,
    "IdentityGaussian": IdentityGaussian,
    "DiagonalGaussian": DiagonalGaussian,
    "TriLGaussian": TriLGaussian,
    "AutoregressiveGMM": AutoregressiveGMM,
}


def get_distribution(distribution_type: str, distribution_config: Optional[Mapping[str, Any]] = None,
                     name: Optional[str] = None):
    """reference distributions.py:235-241."""
    distribution_config = dict(distribution_config or {})
    return _DISTRIBUTIONS[distribution_type](**distribution_config, name=name)
