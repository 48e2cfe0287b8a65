"""-m gpu: the HIP kernels through the C ABI against the oracle / the reference's golden vectors."""
import pytest

import parity_cases as PC

pytestmark = pytest.mark.gpu


def test_gpu_sdf_mlp_forward(gpu):
    PC.check_sdf_mlp_forward(gpu)


def test_gpu_marching_tets_golden(gpu):
    PC.check_mtets_golden(gpu)


def test_gpu_sdf_mlp_backward(gpu):
    PC.check_sdf_mlp_backward(gpu)


def test_gpu_sdf_mlp_deform(gpu):
    PC.check_sdf_mlp_deform(gpu)


def test_gpu_lbs_golden(gpu):
    PC.check_lbs_golden(gpu)
