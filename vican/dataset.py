from vican_amd.dataset import (Camera, Dataset, DojoDataset, load_edges, read_cameras,  # noqa: F401
                               read_object_poses, save_edges)
