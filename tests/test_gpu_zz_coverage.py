"""Kernel-dispatch coverage (collected last: the file name sorts behind every other test module).

Every dispatch rule in libpmhip.so that is keyed on a batch size or a shape (plan_image: B >= 128; rows_sum_v4:
B * ceil(N / 256) >= 128; the 64-column workgroups from 512 workgroups on; ...) is a place where the benchmarked step can
run a kernel that no parity test reaches.  This test runs ONE eager optimizer step of every workload bench.py / tools
time, at the benchmarked batch, records the kernel variants it launches ("<kernel as rocprofv3 names it>[<variant>]",
posterior_matching_amd.ops.coverage_begin) and requires each of them to have been launched by a test of the parity
modules AND followed, in that test, by a comparison against the oracle (tests/conftest.py: confirm_compared, called by the
modules' rel_err helpers; the exact-output modules count every launch) - recorded while those tests ran in this session."""
import pytest
import torch

from tests import conftest

pytestmark = [pytest.mark.gpu, pytest.mark.no_parity_coverage]

WORKLOADS = [("pm_vae_mnist", 256), ("pm_vae_gas", 128), ("vqvae_mnist", 256), ("pm_vqvae_mnist", 256),
             ("pm_vdvae_mnist", 8), ("pm_vdvae_mnist", 16), ("pm_vqvae_celeb_a", 16)]


@pytest.mark.parametrize("name,batch", WORKLOADS)
def test_every_benchmarked_kernel_has_a_parity_test(name, batch):
    from tools.workloads import build, kernels_of_one_step

    tested = conftest.PARITY_KERNELS
    if len(tested) < 40:
        pytest.skip("the parity modules did not run in this session (run the whole -m gpu suite)")
    w = build(name, batch)
    launched = kernels_of_one_step(w)
    del w
    torch.cuda.empty_cache()
    assert len(launched) >= 8, launched
    missing = sorted(k for k in launched if k not in tested)
    only_launched = {k: conftest.UNCOMPARED_KERNELS[k] for k in missing if k in conftest.UNCOMPARED_KERNELS}
    assert not missing, (f"{name} at batch {batch} launches kernels without a parity test that COMPARES a result computed "
                         f"after their launch: {missing}; launched but never compared in: {only_launched}")


def test_coverage_recorder_sees_names_and_variants():
    """the recorder itself: a masked sub-kernel at B = 128 is reported as image_conv_bf16_kernel<..>[masked], the same
    geometry at B = 3 as direct_gemm_bf16_kernel<..>[masked]; entry points with one kernel report their own name"""
    from posterior_matching_amd import ops
    from posterior_matching_amd.ops import LayerGeom

    d = torch.device("cuda:0")
    geom = LayerGeom.masked_conv(7, 7, 32, 32, 3, 3, 2, 2)
    seen = {}
    for B in (3, 128):
        x, w = torch.randn((B, 7, 7, 32), device=d), torch.randn(geom.weight_shape, device=d)
        y = torch.empty((B, 7, 7, 32), device=d)
        ops.coverage_begin()
        ops.layer_forward(geom, x, w, None, y)
        ops.fill_zero(y)
        seen[B] = ops.coverage_end()
    torch.cuda.synchronize()
    assert all(any(k.endswith("[masked]") for k in s) for s in seen.values()), seen
    assert any(k.startswith("pm_") for k in seen[3]), seen
