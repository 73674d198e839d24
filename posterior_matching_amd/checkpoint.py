"""Importer for parameters trained by the reference (the JAX-free half of SURVEY.md 8(f)-4).

The reference pickles a bax TrainState whose `.params` / `.state` are haiku trees of jax arrays
(train_pm_vae.py:91,104-106; train_pm_vqvae.py:154-155); reading that pickle needs jax.  On a machine with jax the
one-off export is

    import pickle, numpy as np, jax
    ts = pickle.load(open("train_state.pkl", "rb"))
    flat = {f"{mod}/{leaf}": np.asarray(v) for mod, d in ts.params.items() for leaf, v in d.items()}
    np.savez("params.npz", **flat)            # likewise ts.state -> state.npz, ts.ema_params -> ema.npz

and everything after that is here: `haiku_to_native` renames the haiku module paths to this package's parameter names,
`load_npz` fills a model's ParamStore.  Layouts are identical (NHWC activations, HWIO conv weights, [kh,kw,Cout,Cin]
transposed-conv weights, [in,out] dense) so no array is transposed.

haiku names auto-created modules `conv2_d`, `conv2_d_1`, ... / `linear`, `linear_1`, ... in creation order inside their
parent's scope; the reference passes explicit names only where listed below.  Rules (reference file:line of the module):
  PM-VAE     <net>/conv2_d[_i]            -> <net>/conv_i            networks.py:30-36   (net = encoder_net | partial_encoder_net)
             <net>/conv2_d_transpose[_i]  -> <net>/conv_t_i          networks.py:62-68   (decoder_net)
             <net>/linear[_j]             -> <net>/linear_0 | block_k/linear_{0,1}       networks.py:116-129 (ResidualMLP)
             <dist>/linear                -> <dist>/linear           distributions.py:44,75,104
             <dist>/log_scale             -> <dist>/log_scale        distributions.py:46-48
             <dist>/residual_mlp/linear[_j], <dist>/one_dimensional_gmm/linear -> <dist>/mlp/..., <dist>/gmm/linear
                                                                     distributions.py:212-218 (AutoregressiveGMM)
  VQ-VAE     vqvae/[~/]conv_residual_encoder/enc_1 ...  -> encoder/enc_1 ...   vqvae.py:184-210 (explicit names kept)
             .../conv_residual_stack/res3x3_i           -> encoder|decoder/res3x3_i   vqvae.py:148-162
             vqvae/pre_vq_conv, conv_residual_decoder/dec_j, .../log_scale     vqvae.py:54,232-262
  stage 2    a leading "vqvae/" (train_pm_vqvae.py:123) is stripped for the frozen tree
  VDVAE      "x_bias_{res}]" (stray bracket, vdvae.py:797) -> "x_bias_{res}"
Anything the rules do not cover is matched by natural order and shape inside its top-level scope (`match_by_order`), and
every import ends with a strict check: all native names filled, all shapes equal.
"""
from __future__ import annotations

import re
from collections import OrderedDict
from typing import Dict, Mapping, Optional, Sequence, Tuple

import numpy as np

_SUFFIX = re.compile(r"^(.*?)(?:_(\d+))?$")


def flatten_haiku(tree: Mapping) -> "OrderedDict[str, np.ndarray]":
    """{module: {leaf: array}} (haiku) or an already flat {path: array} -> flat, insertion order kept"""
    flat: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for k, v in tree.items():
        if isinstance(v, Mapping):
            for leaf, arr in v.items():
                flat[f"{k}/{leaf}"] = np.asarray(arr)
        else:
            flat[k] = np.asarray(v)
    return flat


def _index(name: str, base: str) -> Optional[int]:
    """'conv2_d' -> 0, 'conv2_d_3' -> 3 for base 'conv2_d'; None when `name` is not that base"""
    if name == base:
        return 0
    if name.startswith(base + "_") and name[len(base) + 1:].isdigit():
        return int(name[len(base) + 1:])
    return None


def _natural_key(path: str):
    return [(_SUFFIX.match(p).group(1), int(_SUFFIX.match(p).group(2) or 0)) for p in path.split("/")]


def rename(path: str) -> str:
    """one haiku parameter path -> this package's name (see the table in the module docstring); paths no rule covers are
    returned with only the structural clean-ups applied"""
    path = path.replace("/~/", "/")
    if path.startswith("vqvae/"):
        path = path[len("vqvae/"):]
    path = re.sub(r"^posterior_matching_(vae|vdvae)/", "", path)
    path = re.sub(r"x_bias_(\d+)\]", r"x_bias_\1", path)
    parts = path.split("/")
    out = []
    for i, p in enumerate(parts):
        if p in ("conv_residual_stack",):
            continue
        if p == "conv_residual_encoder":
            p = "encoder"
        elif p == "conv_residual_decoder":
            p = "decoder"
        elif p == "residual_mlp":
            p = "mlp"
        elif p == "one_dimensional_gmm":
            p = "gmm"
        elif _index(p, "conv2_d_transpose") is not None:
            p = f"conv_t_{_index(p, 'conv2_d_transpose')}"
        elif _index(p, "conv2_d") is not None:
            p = f"conv_{_index(p, 'conv2_d')}"
        out.append(p)
    # ResidualMLP linears: linear, linear_1, ... -> linear_0, block_k/linear_{0,1}  (only inside a network / mlp scope)
    if len(out) >= 2 and _index(out[-2], "linear") is not None and out[-1] in ("w", "b"):
        scope = out[-3] if len(out) >= 3 else ""
        j = _index(out[-2], "linear")
        is_mlp_scope = scope in ("encoder_net", "decoder_net", "partial_encoder_net", "mlp")
        if is_mlp_scope:
            out[-2:-1] = ["linear_0"] if j == 0 else [f"block_{(j - 1) // 2}", f"linear_{(j - 1) % 2}"]
    return "/".join(out)


def match_by_order(src: Mapping[str, np.ndarray], native_shapes: Mapping[str, Tuple[int, ...]],
                   done: Mapping[str, np.ndarray], strict: bool = True) -> Dict[str, np.ndarray]:
    """Whatever the rules left unmatched: inside each top-level scope, pair the remaining source leaves (natural order of
    their haiku paths) with the remaining native names (creation order), leaf kind by leaf kind (w with w, b with b, ...),
    when - and only when - counts and shape sequences agree."""
    out: Dict[str, np.ndarray] = {}
    ambiguous = []
    todo_native = [n for n in native_shapes if n not in done]
    todo_src = sorted(src, key=_natural_key)
    scope_of = lambda path: path.split("/")[0]   # noqa: E731
    leaf_of = lambda path: path.rsplit("/", 1)[-1]   # noqa: E731
    for scope in OrderedDict.fromkeys(scope_of(n) for n in todo_native):
        nat = [n for n in todo_native if scope_of(n) == scope]
        cand = [s for s in todo_src if scope_of(rename(s)) == scope]
        for leaf in OrderedDict.fromkeys(leaf_of(n) for n in nat):
            nn = [n for n in nat if leaf_of(n) == leaf]
            cc = [s for s in cand if leaf_of(s) == leaf]
            if len(nn) == len(cc) and all(tuple(src[s].shape) == tuple(native_shapes[n]) for n, s in zip(nn, cc)):
                # equal shapes prove nothing when several leaves share one shape (stacks of 256x256 linears, the VDVAE's
                # repeated Blocks): a creation order that differs from haiku's natural name order would permute weights
                # silently.  Pair by order only where every shape in the group is unique; otherwise a rename rule is needed.
                shapes = [tuple(native_shapes[n]) for n in nn]
                if len(set(shapes)) != len(shapes):
                    if strict:
                        continue
                    ambiguous.extend(zip(nn, cc))
                out.update(zip(nn, (src[s] for s in cc)))
    if ambiguous:
        import warnings

        warnings.warn("checkpoint import paired same-shaped leaves by order only: "
                      + ", ".join(f"{c} -> {n}" for n, c in ambiguous[:8]) + (" ..." if len(ambiguous) > 8 else ""))
    return out


def haiku_to_native(params: Mapping, native_shapes: Mapping[str, Sequence[int]]) -> "OrderedDict[str, np.ndarray]":
    """haiku tree (nested or flat) -> {native name: array} for exactly the names in `native_shapes`; raises KeyError /
    ValueError when a name cannot be filled or a shape differs."""
    flat = flatten_haiku(params)
    native_shapes = OrderedDict((k, tuple(int(s) for s in v)) for k, v in native_shapes.items())
    out: Dict[str, np.ndarray] = {}
    used = set()
    for src_name, arr in flat.items():
        name = rename(src_name)
        if name in native_shapes and name not in out and tuple(arr.shape) == native_shapes[name]:
            out[name] = arr
            used.add(src_name)
    rest = OrderedDict((k, v) for k, v in flat.items() if k not in used)
    if len(out) < len(native_shapes) and rest:
        out.update(match_by_order(rest, native_shapes, out))
    missing = [n for n in native_shapes if n not in out]
    if missing:
        raise KeyError(f"{len(missing)} parameters not found in the checkpoint, e.g. {missing[:4]}; unmatched source "
                       f"leaves e.g. {[k for k in rest][:4]}")
    for n, shp in native_shapes.items():
        if tuple(out[n].shape) != shp:
            raise ValueError(f"{n}: checkpoint shape {tuple(out[n].shape)} != model shape {shp}")
    return OrderedDict((n, np.asarray(out[n], dtype=np.float32)) for n in native_shapes)


def load_npz(model, path: str) -> None:
    """fills `model` (anything with .store: PosteriorMatchingVAE, VQVAE, PosteriorMatchingVDVAE, or a stage-2 store holder)
    from an exported params.npz; the model must have been init()-ed so that its parameter names and shapes exist"""
    store = model.store
    shapes = OrderedDict((n, s) for n, (s, _) in store.specs.items())
    with np.load(path) as z:
        store.load_dict(haiku_to_native({k: z[k] for k in z.files}, shapes))


def vq_state_to_native(state: Mapping) -> Dict[str, np.ndarray]:
    """haiku state of hk.nets.VectorQuantizerEMA (module vector_quantizer_ema under vqvae/) -> VQVAE.load_state() keys:
    embeddings, ema_cluster_size/{hidden,average}, ema_dw/{hidden,average}, counter (both EMAs count together)."""
    flat = flatten_haiku(state)
    out: Dict[str, np.ndarray] = {}
    for k, v in flat.items():
        k = k.replace("/~/", "/")
        if k.endswith("/embeddings"):
            out["embeddings"] = np.asarray(v, np.float32)
        for ema in ("ema_cluster_size", "ema_dw"):
            for leaf in ("hidden", "average"):
                if k.endswith(f"{ema}/{leaf}"):
                    out[f"{ema}/{leaf}"] = np.asarray(v, np.float32)
            if k.endswith(f"{ema}/counter"):
                out["counter"] = np.asarray(v).reshape(1).astype(np.int32)
    need = {"embeddings", "ema_cluster_size/hidden", "ema_cluster_size/average", "ema_dw/hidden", "ema_dw/average", "counter"}
    if need - set(out):
        raise KeyError(f"VQ state leaves missing: {sorted(need - set(out))}")
    return out
