"""Kernel-logic checks on the host emulation of the .hip sources (toy sizes; see tests/emul/hip_emul.h).

These do NOT replace the -m gpu parity tests: they exist so index maps / MFMA fragment layouts /
compaction order are debugged before a GPU round trip.
"""
import parity_cases as PC


def test_emul_sdf_mlp_forward(emul):
    PC.check_sdf_mlp_forward(emul, n=200)


def test_emul_marching_tets_golden(emul):
    PC.check_mtets_golden(emul)


def test_emul_sdf_mlp_backward(emul):
    PC.check_sdf_mlp_backward(emul, n=96)


def test_emul_lbs_golden(emul):
    PC.check_lbs_golden(emul)
