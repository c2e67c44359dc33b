"""Runs last (file name): the conv kernel's producer/consumer protocol uses BOUNDED waits on LDS counters, so a protocol bug
would not hang the GPU -- it would be counted.  After everything the suite has launched the count must still be zero."""
import pytest
import torch

from wakeword_jupyterlab_amd import _native as nat

pytestmark = pytest.mark.gpu


def test_no_in_kernel_wait_ever_timed_out():
    torch.cuda.synchronize()
    assert nat.lib.ww_sync_timeouts() == 0
