"""Import-path shim for ``from vican.cam import estimate_pose_mp`` (main.ipynb cell 1).

The vision front-end (ArUco detection + PnP with OpenCV, reference vican/cam.py:83-260) is OUTSIDE the hot path this
build replaces: the name imports cleanly so the notebook's import cell runs unchanged, and calling
it explains what to do instead (load the precomputed ``cam_marker_edges.pt`` edge cache, which the notebook offers as
the alternative in the same cell, or run the reference's detector and feed its edge dict to ``bipartite_se3sync``)."""
from vican_amd.dataset import Camera  # noqa: F401


def estimate_pose_mp(im_filenames=None, cams=None, aruco=None, marker_size=None, corner_refine=None, brightness=None,
                     contrast=None, flags=None, marker_ids=None) -> dict:
    """Same signature as the reference (vican/cam.py:190-198); not provided by the MI355X build."""
    raise NotImplementedError(
        "vican.cam.estimate_pose_mp (OpenCV ArUco detection + PnP) is not part of the MI355X build of the pose-graph "
        "solver: load the precomputed edges instead - torch.load(os.path.join(DATASET_PATH, 'cam_marker_edges.pt')) as "
        "main.ipynb suggests, or vican.dataset.load_edges(path) - or produce the edge dict with the reference's "
        "detector; bipartite_se3sync / object_bipartite_se3sync accept it unchanged.")
