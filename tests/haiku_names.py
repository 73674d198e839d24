"""Native parameter name -> the haiku module path the reference's checkpoints use (the inverse of
posterior_matching_amd.checkpoint.rename), so that tests can write reference-shaped trees.  Naming rules: haiku numbers
auto-created modules in creation order inside their parent scope (`conv2_d`, `conv2_d_1`, ..., `linear`, `linear_1`, ...);
the reference passes explicit names elsewhere (networks.py:30-36,62-68,116-129; distributions.py:212-218;
vqvae.py:54,148-262; train_pm_vqvae.py:123; vdvae.py:797)."""
import re


def vae_names(n: str) -> str:
    m = re.match(r"(.*)/conv_t_(\d+)/(w|b)$", n)
    if m:
        return f"{m.group(1)}/conv2_d_transpose{'' if m.group(2) == '0' else '_' + m.group(2)}/{m.group(3)}"
    m = re.match(r"(.*)/conv_(\d+)/(w|b)$", n)
    if m:
        return f"{m.group(1)}/conv2_d{'' if m.group(2) == '0' else '_' + m.group(2)}/{m.group(3)}"
    n = n.replace("/mlp/", "/residual_mlp/").replace("/gmm/", "/one_dimensional_gmm/")
    m = re.match(r"(.*)/block_(\d+)/linear_(\d)/(w|b)$", n)
    if m:
        return f"{m.group(1)}/linear_{1 + 2 * int(m.group(2)) + int(m.group(3))}/{m.group(4)}"
    return re.sub(r"/linear_0/(w|b)$", r"/linear/\1", n)


def vq_names(n: str) -> str:
    n = re.sub(r"^(encoder|decoder)/(res\dx\d_\d+)/", lambda m: f"conv_residual_{m.group(1)}/conv_residual_stack/{m.group(2)}/", n)
    n = re.sub(r"^(encoder|decoder)/", lambda m: f"conv_residual_{m.group(1)}/", n)
    return "vqvae/~/" + n if n.count("/") else "vqvae/" + n


def vdvae_names(n: str) -> str:
    n = "posterior_matching_vdvae/" + n
    return re.sub(r"x_bias_(\d+)$", r"x_bias_\1]", n) if "x_bias" in n else n


def to_tree(native: dict, to_haiku) -> dict:
    """{native name: array} -> haiku's {module path: {leaf: array}}"""
    tree = {}
    for n, v in native.items():
        mod, leaf = to_haiku(n).rsplit("/", 1)
        tree.setdefault(mod, {})[leaf] = v
    return tree


def vq_state_tree(state: dict) -> dict:
    """oracle VQ state ("vq/embeddings", "vq/ema_*/{hidden,average,counter}") -> the haiku state tree of
    hk.nets.VectorQuantizerEMA under vqvae/"""
    base = "vqvae/~/vector_quantizer_ema"
    tree = {base: {"embeddings": state["vq/embeddings"]}}
    for ema in ("ema_cluster_size", "ema_dw"):
        tree[f"{base}/~/{ema}"] = {leaf: state[f"vq/{ema}/{leaf}"] for leaf in ("hidden", "average", "counter")}
    return tree
