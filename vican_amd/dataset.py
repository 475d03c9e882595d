"""Data formats either side of the solver: the ``cameras.json`` /
``object_pose_<n>.json`` ground-truth files written by the reference's renderer and the cached
``cam_marker_edges.pt`` edge dictionaries of the notebook.

Mirrors the reader side of the reference's ``Dataset`` (dataset.py:14-101): same attribute names
(``cams[id].extrinsics`` ..., ``object[t]``, ``im_data``), so notebook cells 1 and 9 run unchanged
through the ``vican.dataset`` shim.  Pose estimation from images (``vican.cam.estimate_pose_mp``,
OpenCV) is not on the path and not provided.
"""
from __future__ import annotations

import json
import os

import numpy as np

from .geometry import SE3

__all__ = ["Camera", "Dataset", "DojoDataset", "read_cameras", "read_object_poses", "load_edges", "save_edges"]


class Camera:
    """Perspective camera record (reference cam.py:14-44): intrinsics 3x3, distortion vector,
    extrinsics = camera pose in the world frame."""

    def __init__(self, id, intrinsics, distortion, extrinsics, resolution_x, resolution_y):
        self.id = id
        self.intrinsics = np.asarray(intrinsics).squeeze()
        self.distortion = np.asarray(distortion).squeeze()
        self.extrinsics = extrinsics
        self.resolution_x = resolution_x
        self.resolution_y = resolution_y

    def __repr__(self):
        return "Camera %sx%s id=%s\nIntrinsics:\n%s\nDistortion:\n%s\nExtrinsics:\n%s" % (
            self.resolution_y, self.resolution_x, self.id, self.intrinsics, self.distortion, self.extrinsics)


def read_cameras(path: str) -> dict:
    """``cameras.json`` -> ``{camera id: Camera}`` (dataset.py:39-60): per camera fx, fy, cx, cy,
    distortion, R (3x3), t (3), resolution_x, resolution_y."""
    with open(path) as f:
        data = json.load(f)
    cams = {}
    for k, v in data.items():
        K = np.array([[v["fx"], 0.0, v["cx"]], [0.0, v["fy"], v["cy"]], [0.0, 0.0, 1.0]])
        cams[k] = Camera(id=k, intrinsics=K, distortion=np.array(v["distortion"]),
                         extrinsics=SE3(R=np.array(v["R"]), t=np.array(v["t"])),
                         resolution_x=v["resolution_x"], resolution_y=v["resolution_y"])
    return cams


def read_object_poses(root: str) -> dict:
    """All ``object*`` JSON files of a dataset folder -> ``{timestamp: SE3}`` (dataset.py:63-76)."""
    out = {}
    for fn in os.listdir(root):
        if fn.split("_")[0] != "object":
            continue
        with open(os.path.join(root, fn)) as f:
            for t, pose in json.load(f).items():
                out[t] = SE3(R=np.array(pose["R"]), t=np.array(pose["t"]))
    return out


class Dataset:
    """Folder of renders: ``root/cameras.json`` (required), ``root/object_pose_<n>.json`` (optional),
    images ``root/<timestamp>/<camera_id>.jpg`` (listed, never decoded here)."""

    def __init__(self, root: str):
        self.root = root
        self.cam_path = os.path.join(root, "cameras.json")
        assert os.path.isfile(self.cam_path)
        self.read_cameras()
        self.read_im_data()
        self.read_object()

    def read_cameras(self):
        self.cams = read_cameras(self.cam_path)

    def read_object(self):
        self.object = read_object_poses(self.root)

    def read_im_data(self):
        self.im_data = {"filename": [], "timestamp": [], "cam": [], "cam_id": []}
        for t in os.listdir(self.root):
            d = os.path.join(self.root, t)
            if not (t.isnumeric() and os.path.isdir(d)):
                continue
            for fn in os.listdir(d):
                if fn.endswith(".jpg"):
                    cam_id = fn.split(".")[0]
                    self.im_data["cam_id"].append(cam_id)
                    self.im_data["filename"].append(os.path.join(d, fn))
                    self.im_data["timestamp"].append(t)
                    self.im_data["cam"].append(self.cams[cam_id])


class DojoDataset:
    """Real-capture layout (reference dataset.py:103-181): ``cameras_intrinsics.json`` (per camera
    ``intrinsics`` 3x3 + ``distortion``), ``cameras_transformations_to_origin_ground_truth.json`` (4x4 per
    camera), ``aruco_cube_transformations.json`` (``{"to": {marker: 4x4}}`` -> ``object_constraints``
    holding the INVERSE of each, ready to be passed as ``constraints=``), images under
    ``aruco_images_samples/<timestamp>/<camera_id>.jpg``."""

    def __init__(self, root: str):
        self.root = root
        self.read_cameras()
        self.read_im_data()
        self.read_object_constraints()

    def read_cameras(self):
        with open(os.path.join(self.root, "cameras_intrinsics.json")) as f:
            intr = json.load(f)
        with open(os.path.join(self.root, "cameras_transformations_to_origin_ground_truth.json")) as f:
            extr = json.load(f)
        self.cams = {}
        for c in extr.keys():
            self.cams[c] = Camera(id=c, intrinsics=np.array(intr[c]["intrinsics"]), distortion=np.array(intr[c]["distortion"]),
                                  extrinsics=SE3(pose=np.array(extr[c])), resolution_x=None, resolution_y=None)

    def read_object_constraints(self):
        with open(os.path.join(self.root, "aruco_cube_transformations.json")) as f:
            data = json.load(f)
        self.object_constraints = {m: SE3(pose=np.array(v)).inv() for m, v in data["to"].items()}

    def read_im_data(self):
        path = os.path.join(self.root, "aruco_images_samples")
        self.im_data = {"filename": [], "timestamp": [], "cam": [], "cam_id": []}
        if not os.path.isdir(path):
            raise FileNotFoundError(path)
        for t in os.listdir(path):
            d = os.path.join(path, t)
            if not (t.isnumeric() and os.path.isdir(d)):
                continue
            for fn in os.listdir(d):
                if fn.endswith(".jpg"):
                    cam_id = fn.split(".")[0]
                    self.im_data["cam_id"].append(cam_id)
                    self.im_data["filename"].append(os.path.join(d, fn))
                    self.im_data["timestamp"].append(t)
                    self.im_data["cam"].append(self.cams[cam_id])


def load_edges(path: str) -> dict:
    """Cached edge dictionary of the notebook (``torch.load`` of a pickled
    ``{(cam, "t_m"): {"pose": SE3, ...}}``, main.ipynb:68-71,108-111).  The pickle names
    ``vican.geometry.SE3``; the ``vican`` shim package resolves it to this build's class."""
    import torch
    import vican.geometry  # noqa: F401  (makes the pickled class importable)
    return torch.load(path, weights_only=False)


def save_edges(edges: dict, path: str) -> None:
    import torch
    torch.save(edges, path)
