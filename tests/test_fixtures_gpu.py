"""-m gpu: the reference-generated op / block / step fixtures (G1, G2, G5) straight through the HIP entry points — the
chain is fixture <-> kernel, with neither the oracle nor the numpy emulator in between."""
import pytest
import torch

from tests import fixture_cases as FC

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("dt", [0, 1])
def test_g1_ops_through_hip(golden, dt):
    FC.run_g1(golden("g1_ops.pt"), DEV, dt=dt)


def test_g2_blocks_through_the_engine_block_routines(golden):
    FC.run_g2(golden("g2_blocks.pt"), DEV, tol=1e-4, gtol=5e-4)


def test_g5_steps_through_hip(golden):
    FC.run_g5(golden("g5_steps.pt"), golden("g3_model.pt"), DEV)
