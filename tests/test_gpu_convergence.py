"""Long-horizon agreement of the arithmetics: does the default bf16x3 path TRAIN to the same ELBO /
matching-LL as the strict-f32 path, and do both start on the float32 oracle's trajectory?

The reference trains configs/pm_vae_mnist.py for 100 k+ steps (train_pm_vae.py:58-83); the parity tests compare <= 6 optimizer
steps.  Here: 300 optimizer steps at B = 64 on a fixed structured synthetic set (512 images of blurred strokes, rectangle /
Bernoulli masks), identical seeds, data order and reparameterisation noise for every run -

  * the HIP bf16x3 run (the benchmarked arithmetic, launch-plan replay on two streams),
  * the HIP strict-f32 run (every GEMM on the f32 MFMA: the reference's own arithmetic),
  * four more strict-f32 runs: two that differ ONLY in the noise seed, two whose initial parameters are multiplied by
    (1 + 1e-6 N(0, 1)) - a change at float32 rounding level, which is what swapping one arithmetic for another amounts to,
  * the float32 (and float64) torch-CPU oracle for the first 3 steps (an oracle step pair costs ~10 s of host time;
    tests/test_gpu_parity.py::test_bf16x3_training_trajectory_within_1e3 follows the oracle for 4 - 6 steps on other data).

Asserted: (1) steps 0-2: ELBO / KL / matching-LL of both HIP runs within 1e-3 relative of the oracle's float32 trajectory
(x3 the float32-vs-float64 oracle drift where that is larger - the yardstick of test_train_steps_match_oracle);
(2) the mean of the LAST 20 validation ELBO and matching-LL values (a held-out batch with fixed noise, evaluated every 5 steps
over steps 200-299) of the bf16x3 run lies within max(1e-3 relative, 4 sigma) of the five strict-f32 runs' mean, sigma =
their standard deviation, floored at the measured chaos level (1 % of the ELBO, 2 nats of matching-LL).  MEASURED (tools/convergence_probe.py, MI355X): the strict-f32 family ends at validation ELBO -216.4 ...
-225.7 and matching-LL -19.6 ... -26.6 - a 1e-6 perturbation of the initial parameters alone moves the step-300 validation
ELBO by 2 - 5 nats (1 - 2 %), as much as another noise seed does: Adam's first updates are sign-like, entries whose gradient
is rounding noise flip, and at step 300 the ELBO is still climbing ~1 nat per step, so a trajectory that is a few steps
"ahead" reads several nats higher.  A 1e-3 agreement at step 300 is therefore not a property of ANY two float32 runs; what
"trains to the same place" can mean, and what is asserted, is membership in the float32 family's spread (bf16x3: -223.9 /
-20.7, inside both).  The active bound is printed."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

B, STEPS, ORACLE_STEPS, NDATA = 64, 300, 3, 512


def _strokes(n, seed):
    """n images 28x28x1 in [0, 1]: 2-4 random line segments each, blurred with a 3x3 binomial kernel twice"""
    rng = np.random.default_rng(seed)
    img = np.zeros((n, 32, 32), np.float64)
    for i in range(n):
        for _ in range(int(rng.integers(2, 5))):
            p0, p1 = rng.uniform(4, 28, size=2), rng.uniform(4, 28, size=2)
            for t in np.linspace(0.0, 1.0, 40):
                y, x = p0 * (1 - t) + p1 * t
                img[i, int(round(y)), int(round(x))] = 1.0
    k = np.array([0.25, 0.5, 0.25])
    for _ in range(2):
        img = sum(k[j] * np.roll(img, j - 1, axis=1) for j in range(3))
        img = sum(k[j] * np.roll(img, j - 1, axis=2) for j in range(3))
    img = img[:, 2:30, 2:30]
    img /= img.reshape(n, -1).max(axis=1).reshape(n, 1, 1) + 1e-12
    return img[..., None]


def _masks(n, seed):
    rng = np.random.default_rng(seed)
    b = (rng.uniform(size=(n, 28, 28, 1)) < 0.5).astype(np.float64)
    for i in range(0, n, 2):                       # every other example: a rectangle of unobserved pixels instead
        b[i] = 1.0
        y0, x0 = rng.integers(0, 14, size=2)
        b[i, y0:y0 + 14, x0:x0 + 14] = 0.0
    return b


def _run(cfg, xs, data, masks, bf16x3, noise_seed, record_steps=0, perturb=0.0, pseed=0):
    from posterior_matching_amd import optim
    from posterior_matching_amd.engine import PMVAETrainStep
    from tests.test_gpu_parity import _product_model

    dev = torch.device("cuda:0")
    m = _product_model(cfg, xs, seed=11, bf16x3=bf16x3)
    if perturb:            # a rounding-level change of the starting point: p * (1 + perturb * N(0, 1))
        g = torch.Generator().manual_seed(pseed)
        m.load_params({n: t.cpu() * (1.0 + perturb * torch.randn(t.shape, generator=g)) for n, t in m.params_dict().items()})
    opt = optim.chain(optim.scale_by_adam(), optim.add_decayed_weights(cfg.get("weight_decay", 0.0)),
                      optim.scale_by_schedule(optim.exponential_decay(**cfg["lr_schedule"])), optim.scale(-1.0))
    ts = PMVAETrainStep(m, cfg, opt, B, xs, external_eps=True)
    k = cfg["model"]["latent_dim"]
    order = np.random.default_rng(5).permutation(NDATA)          # the same data order for every run
    gen = torch.Generator().manual_seed(noise_seed)
    xv, bv = data[NDATA:NDATA + B].to(dev), masks[NDATA:NDATA + B].to(dev)      # held-out validation batch, fixed noise
    ev = torch.randn((B, k), generator=torch.Generator().manual_seed(999)).to(dev)
    start = {n: t.cpu().clone() for n, t in m.params_dict().items()}
    data_d, masks_d = data.to(dev), masks.to(dev)               # everything a run reads is resident before its first step
    eps_all = torch.randn((STEPS, B, k), generator=gen)
    eps_d = eps_all.to(dev)
    traj, val = [], []
    for step in range(STEPS):
        idx = torch.as_tensor(order[(step * B) % NDATA:(step * B) % NDATA + B].copy())
        if step < record_steps:
            traj.append((idx, eps_all[step].clone()))
        idx_d = idx.to(dev)
        ts.set_batch(data_d[idx_d], masks_d[idx_d], eps_d[step])
        ts.step()
        if step < ORACLE_STEPS:
            met = ts.read_metrics()
            traj_m = (met["reconstruction_ll"] - met["kl"], met["kl"], met["matching_ll"])
            if step < record_steps:
                traj[-1] = traj[-1] + (traj_m,)
            else:
                traj.append(traj_m)
        if step >= 200 and step % 5 == 4:
            out = ts.evaluate(xv, bv, ev)
            val.append((out["reconstruction_ll"] - out["kl"], out["matching_ll"]))
    assert len(val) == 20
    return start, traj, np.array(val).mean(axis=0)


def test_bf16x3_trains_to_the_strict_f32_result_and_both_start_on_the_oracle_trajectory():
    from oracle import pm_vae_oracle as O
    from tests.ref_configs import pm_vae_mnist

    cfg, xs = pm_vae_mnist(), (28, 28, 1)
    data = torch.tensor(_strokes(NDATA + B, 3), dtype=torch.float32)
    masks = torch.tensor(_masks(NDATA + B, 4), dtype=torch.float32)

    start, traj16, val16 = _run(cfg, xs, data, masks, True, noise_seed=21, record_steps=ORACLE_STEPS)
    _, traj32, val32 = _run(cfg, xs, data, masks, False, noise_seed=21)
    family = [val32]
    family.append(_run(cfg, xs, data, masks, False, noise_seed=22)[2])
    family.append(_run(cfg, xs, data, masks, False, noise_seed=23)[2])
    family.append(_run(cfg, xs, data, masks, False, noise_seed=21, perturb=1e-6, pseed=1)[2])
    family.append(_run(cfg, xs, data, masks, False, noise_seed=21, perturb=1e-6, pseed=2)[2])
    family = np.array(family)

    # (1) the first steps against the oracle (float32, and float64 as the yardstick of float32 drift)
    p32 = {n: t.float() for n, t in start.items()}
    p64 = {n: t.double() for n, t in start.items()}
    st32 = [{n: torch.zeros_like(t) for n, t in p.items()} for p in (p32, p32)]
    st64 = [{n: torch.zeros_like(t) for n, t in p.items()} for p in (p64, p64)]
    for step, (idx, eps, got16) in enumerate(traj16):
        x, b = data[idx], masks[idx]
        _, a32, _ = O.train_step(p32, st32[0], st32[1], cfg, x, b, eps, step)
        _, a64, _ = O.train_step(p64, st64[0], st64[1], cfg, x.double(), b.double(), eps.double(), step)
        got32 = traj32[step]
        for j, key in enumerate(("elbo", "kl", "matching_ll")):
            w32 = float(a32["reconstruction_ll"] - a32["kl"]) if key == "elbo" else float(a32[key])
            w64 = float(a64["reconstruction_ll"] - a64["kl"]) if key == "elbo" else float(a64[key])
            tol = max(1e-3 * abs(w32), 3.0 * abs(w32 - w64))
            assert abs(got16[j] - w32) <= tol, ("bf16x3", step, key, got16[j], w32, w64)
            assert abs(got32[j] - w32) <= tol, ("strict f32", step, key, got32[j], w32, w64)

    # (2) where the runs END: the bf16x3 run inside the strict-f32 family's spread of last-20 validation means
    report = {}
    for j, key in enumerate(("val_elbo", "val_matching_ll")):
        mean, std = float(family[:, j].mean()), float(family[:, j].std(ddof=1))
        # five samples of a chaotic quantity estimate its spread poorly (the strict-f32 runs are not even run-to-run
        # identical: their split-K data gradients use atomics), so sigma is floored at the MEASURED sensitivity: 1 % of the
        # value (ELBO) / 2 nats (matching-LL) is what a 1e-6 perturbation of the start does (tools/convergence_probe.py)
        floor = max(0.01 * abs(mean), 2.0 if key == "val_matching_ll" else 0.0)
        sigma = max(std, floor)
        bound = max(1e-3 * abs(mean), 4.0 * sigma)
        report[key] = {"bf16x3": float(val16[j]), "strict_f32_family": family[:, j].tolist(), "family_mean": mean,
                       "family_std": std, "sigma_used": sigma, "bound_4_sigma": bound,
                       "active_bound": "1e-3 relative" if bound == 1e-3 * abs(mean) else
                       ("4 x family std" if sigma == std else "4 x measured chaos floor")}
        assert abs(val16[j] - mean) <= bound, (key, report[key])
    print("convergence parity:", report)
    # and training happened at all: the validation ELBO of the trained model is far above the first step's
    assert val32[0] > traj32[0][0] + 100.0, ("the run did not train", val32, traj32[0])
