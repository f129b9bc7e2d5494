"""Shape / option sweeps of the drop-in surface and of the K-row driver (tools/surface_sweep.py,
tools/driver_sweep.py): every case must run and leave finite parameters.  They found two sizing bugs the
pinned configurations did not reach (critic-only workspace carve at B = 1 with wide nets; D-input sums of
large batches on the whole-network path).  Parity itself is the other GPU tests' job."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('twin', [False, True])
def test_driver_sweep(twin):
    from tools import driver_sweep
    assert driver_sweep.main(twin, quick=True) == 0


@pytest.mark.parametrize('twin', [False, True])
def test_surface_sweep(twin):
    from tools import surface_sweep
    assert surface_sweep.main(twin) == 0
