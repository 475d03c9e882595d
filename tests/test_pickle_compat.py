"""Reference edge caches (`torch.save` of a dict holding `vican.geometry.SE3` objects, main.ipynb:68,108)
load through this repository's `vican` shim and expose the same numbers and dtypes."""
import os

import numpy as np
import torch

from util import GOLDEN_DIR


def test_reference_pickle_loads_with_shim():
    d = torch.load(os.path.join(GOLDEN_DIR, "ref_edges_pickle.pt"), weights_only=False)
    exp = np.load(os.path.join(GOLDEN_DIR, "ref_edges_pickle_expect.npz"))
    import vican.geometry
    from vican_amd.geometry import SE3
    assert ["|".join(k) for k in d] == list(exp["keys"])
    for i, v in enumerate(d.values()):
        pose = v["pose"]
        assert type(pose) is SE3 and type(pose) is vican.geometry.SE3
        assert np.array_equal(np.asarray(pose.R(), dtype=np.float64), exp["R"][i])
        assert np.array_equal(np.asarray(pose.t(), dtype=np.float64), exp["t"][i])
        assert str(np.asarray(pose.R()).dtype) == str(exp["R_dtype"][i])
        # algebra follows the reference's float32 round trip
        inv = pose.inv()
        assert inv.R().dtype == np.float32 and np.allclose(inv.R(), np.asarray(pose.R()).T, atol=1e-6)
        assert set(v) >= {"pose", "corners", "reprojected_err", "im_filename"}
