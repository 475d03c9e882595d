"""device.upload / device.download: every per-call host<->device transfer of the drop-in path goes through one page-locked
staging buffer (pageable copies stalled one cold call in three for 70-100 ms)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_upload_download_roundtrip_and_reuse():
    from vican_amd import device
    dev = torch.device("cuda", torch.cuda.current_device())
    rng = np.random.default_rng(0)
    for rep in range(3):                                       # the staging buffer is reused (and grows with the request)
        n = 1000 * (rep + 1) ** 3
        a = rng.standard_normal((n, 9))                        # float64 -> float32 on the way
        b = rng.integers(0, 2 ** 31 - 1, n).astype(np.int64)   # int64 -> int32
        c = rng.standard_normal((n, 3))[:, ::-1]               # not contiguous
        e = np.zeros((0, 3))                                   # empty
        ta, tb, tc, te = device.upload(dev, [(a, torch.float32), (b, torch.int32), (c, torch.float64), (e, torch.float64)])
        assert ta.dtype == torch.float32 and tb.dtype == torch.int32 and tc.dtype == torch.float64 and te.shape == (0, 3)
        assert all(t.data_ptr() % 256 == 0 for t in (ta, tb, tc))
        assert torch.equal(ta.cpu(), torch.from_numpy(a.astype(np.float32)))
        assert torch.equal(tb.cpu(), torch.from_numpy(b.astype(np.int32)))
        assert torch.equal(tc.cpu(), torch.from_numpy(np.ascontiguousarray(c)))
        # a second upload must not disturb tensors of the first one (their own device allocation; the staging buffer is
        # only reused once its copy has completed)
        device.upload(dev, [(np.ones(n), torch.float64)])
        assert torch.equal(tb.cpu(), torch.from_numpy(b.astype(np.int32)))
        ha, hb, hc = device.download([ta, tb.view(-1, 1).expand(-1, 2), tc.t()])      # views / transposes come back contiguous
        assert ha.dtype == np.float32 and np.array_equal(ha, a.astype(np.float32))
        assert np.array_equal(hb, np.repeat(b.astype(np.int32)[:, None], 2, 1))
        assert np.array_equal(hc, np.ascontiguousarray(c).T) and hc.flags["C_CONTIGUOUS"]
        ha[:] = 0                                              # results are copies, not views of the staging buffer
        assert np.array_equal(device.download([ta])[0], a.astype(np.float32))
