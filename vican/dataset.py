from vican_amd.dataset import Camera, Dataset, load_edges, read_cameras, read_object_poses, save_edges  # noqa: F401
