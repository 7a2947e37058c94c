"""Synthetic ActorsHQ-shaped scene + the training branch of the reference data loader, HBM resident.

The dataset itself is not obtainable (per-user access YAML + network, README.md:53,86), so BASELINE.json's
configs are restated on a procedural scene with the dataset's geometry: an animated
"capsule person" of 10 ellipsoids, 160 cameras on five rings looking at the origin (normalised focal
1.773863, 4x scale, centre-cropped square images), per-frame uint8 {0,255} occupancy grids [z][y][x] with
voxel centres i/(G-1)-0.5 dilated by two voxels, frames 15..64, procedurally shaded RGBA images with the
silhouette as mask. Scene normalisation follows actorshq/dataset/data_loader.py:182-215 exactly
(scene_offset = -aabb.mean, scene_scale = 1/max extent, inverse_krs = inv(P)[:3,:3]^T).

`SyntheticDataLoader.__next__` is the training branch of DataLoader.__next__ (data_loader.py:539-575,
631-660): torch.randint over the pool, the sampler call, InputBatch assembly. The image pool, per-slot camera
tables and ALL frames' occupancy grids live in HBM (288 GB makes the reference's
CPU pool + 8-grid texture ring unnecessary)."""
from __future__ import annotations

import math
import threading
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import ray_sampler_native
from .input_batch import InputBatch
from .occupancy_grid_native import OccupanyGrid


# ---------------------------------------------------------------------------------------------- geometry
def _person_ellipsoids(frame: int, num_frames: int = 50) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """centers (10,3), radii (10,3), albedo (10,3) in metres; y is 'down' (RDF), the person is 1.8 m tall."""
    ph = 2.0 * math.pi * (frame % num_frames) / num_frames
    s, c = math.sin(ph), math.cos(ph)
    centers = np.array([
        [0.00, -0.05 + 0.01 * s, 0.00],                 # torso
        [0.00 + 0.02 * s, -0.72, 0.02 * c],             # head
        [-0.27, -0.25, 0.10 * s],                       # upper arm L
        [-0.33 - 0.03 * c, 0.02, 0.22 * s],             # lower arm L
        [0.27, -0.25, -0.10 * s],                       # upper arm R
        [0.33 + 0.03 * c, 0.02, -0.22 * s],             # lower arm R
        [-0.11, 0.38, 0.08 * s],                        # upper leg L
        [-0.12, 0.73, 0.16 * s + 0.03],                 # lower leg L
        [0.11, 0.38, -0.08 * s],                        # upper leg R
        [0.12, 0.73, -0.16 * s + 0.03],                 # lower leg R
    ], dtype=np.float64)
    radii = np.array([
        [0.20, 0.33, 0.13], [0.11, 0.14, 0.12],
        [0.06, 0.17, 0.06], [0.05, 0.16, 0.05], [0.06, 0.17, 0.06], [0.05, 0.16, 0.05],
        [0.09, 0.22, 0.09], [0.07, 0.20, 0.07], [0.09, 0.22, 0.09], [0.07, 0.20, 0.07],
    ], dtype=np.float64)
    albedo = np.array([
        [0.75, 0.25, 0.20], [0.85, 0.65, 0.55],
        [0.20, 0.45, 0.75], [0.85, 0.65, 0.55], [0.20, 0.45, 0.75], [0.85, 0.65, 0.55],
        [0.25, 0.25, 0.35], [0.30, 0.30, 0.40], [0.25, 0.25, 0.35], [0.30, 0.30, 0.40],
    ], dtype=np.float64)
    return centers, radii, albedo


@dataclass
class Camera:
    """Right-down-forward pinhole camera, cam -> world extrinsics (actorshq/dataset/camera_data.py:17-102)."""
    width: int
    height: int
    rotation_cam2world: np.ndarray  # (3,3)
    translation: np.ndarray         # (3,)
    fx: float
    fy: float
    cx: float
    cy: float

    def projection_matrix_world2pixel(self) -> np.ndarray:
        k = np.array([[self.fx, 0, self.cx], [0, self.fy, self.cy], [0, 0, 1.0]])
        c2w = np.eye(4)
        c2w[:3, :3] = self.rotation_cam2world
        c2w[:3, 3] = self.translation
        p = np.eye(4)
        p[:3] = k @ np.linalg.inv(c2w)[:3]
        return p


def make_cameras(num_cameras: int, width: int, height: int, radius: float = 4.1) -> List[Camera]:
    cams = []
    rings = 5
    per_ring = int(math.ceil(num_cameras / rings))
    fx = 1.773863 * max(width, height) * (1028.0 / 1028.0)
    for i in range(num_cameras):
        ring, k = divmod(i, per_ring)
        ang = 2.0 * math.pi * (k + 0.5 * (ring % 2)) / per_ring
        y = (-0.9 + 1.8 * (ring + 0.5) / rings) * 1.2
        pos = np.array([radius * math.cos(ang), y, radius * math.sin(ang)])
        fwd = -pos / np.linalg.norm(pos)
        down = np.array([0.0, 1.0, 0.0])
        right = np.cross(down, fwd)
        right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        rot = np.stack([right, down, fwd], axis=1)  # columns = camera axes in world space
        cams.append(Camera(width, height, rot, pos, fx, fx, width * 0.5, height * 0.5))
    return cams


class SyntheticScene:
    """Geometry, cameras, normalisation, occupancy grids and images of the synthetic capture."""

    def __init__(self, frame_numbers: Sequence[int], num_cameras: int = 160, width: int = 752, height: int = 752,
                 grid_resolution: int = 256, device: str = "cuda"):
        self.frame_numbers = list(frame_numbers)
        self.device = torch.device(device)
        self.grid_resolution = grid_resolution
        self.width, self.height = width, height
        self.cameras_world = make_cameras(num_cameras, width, height)
        aabbs = []
        for f in self.frame_numbers:
            c, r, _ = _person_ellipsoids(f)
            aabbs.append(np.stack([(c - r).min(0), (c + r).max(0)]))
        aabbs = np.stack(aabbs)
        pad = 0.03  # the dataset's boxes are not tight either
        self.aabb_world = np.stack([aabbs[:, 0].min(0) - pad, aabbs[:, 1].max(0) + pad])
        # data_loader.py:182-184
        self.scene_offset = -self.aabb_world.mean(0)
        self.scene_scale = 1.0 / np.max(self.aabb_world[1] - self.aabb_world[0])
        self.cameras = []
        for cam in self.cameras_world:  # volumetric_dataset.py:get_scaled_cameras
            self.cameras.append(Camera(cam.width, cam.height, cam.rotation_cam2world,
                                       (cam.translation + self.scene_offset) * self.scene_scale, cam.fx, cam.fy, cam.cx,
                                       cam.cy))
        # data_loader.py:194-215
        self.all_inverse_krs = torch.from_numpy(np.stack(
            [np.linalg.inv(c.projection_matrix_world2pixel()) for c in self.cameras], 0))[..., :3, :3] \
            .transpose(-1, -2).float().contiguous().to(self.device)
        self.all_camera_origins = torch.from_numpy(np.stack([c.translation for c in self.cameras], 0)).float() \
            .contiguous().to(self.device)
        self.aabb = torch.from_numpy((self.aabb_world + self.scene_offset) * self.scene_scale).float().contiguous() \
            .to(self.device)

    # ellipsoids of frame f in the NORMALISED scene space
    def _ellipsoids_normalised(self, frame: int):
        c, r, a = _person_ellipsoids(frame)
        c = (c + self.scene_offset) * self.scene_scale
        r = r * self.scene_scale
        t = lambda x: torch.from_numpy(x).float().to(self.device)
        return t(c), t(r), t(a)

    @torch.no_grad()
    def occupancy_grid(self, frame: int) -> torch.Tensor:
        """(G,G,G) uint8 {0,255}, [z][y][x], voxel centre i/(G-1)-0.5
        (actorshq/toolbox/native/occupancy_grid_generation.cu:32-37,80), dilated by 2 voxels."""
        G = self.grid_resolution
        c, r, _ = self._ellipsoids_normalised(frame)
        lin = torch.arange(G, device=self.device, dtype=torch.float32) / (G - 1) - 0.5
        occ = torch.zeros(G, G, G, dtype=torch.bool, device=self.device)
        z = lin.view(G, 1, 1); y = lin.view(1, G, 1); x = lin.view(1, 1, G)
        for k in range(c.shape[0]):
            q = ((x - c[k, 0]) / r[k, 0]) ** 2 + ((y - c[k, 1]) / r[k, 1]) ** 2 + ((z - c[k, 2]) / r[k, 2]) ** 2
            occ |= q <= 1.0
        occ = torch.nn.functional.max_pool3d(occ.float()[None, None], kernel_size=5, stride=1, padding=2)[0, 0] > 0
        return (occ.to(torch.uint8) * 255).contiguous()

    @torch.no_grad()
    def render_rgba(self, camera_number: int, frame: int) -> torch.Tensor:
        """(H*W, 4) uint8 ground-truth image: Lambert-shaded ellipsoids, alpha = silhouette."""
        return self.render_rgba_cameras([camera_number], frame)[0]

    @torch.no_grad()
    def render_rgba_cameras(self, camera_numbers: Sequence[int], frame: int) -> torch.Tensor:
        """(C, H*W, 4) uint8 images of one frame for several cameras at once (same arithmetic per pixel as a single
        image: the capture store below renders 160 cameras x 50 frames with ~50 x 10 batched calls instead of 8000)."""
        cams = torch.as_tensor(list(camera_numbers), dtype=torch.long, device=self.device)
        W, H = self.width, self.height
        m = self.all_inverse_krs[cams].transpose(-1, -2)  # undo the column-major transpose: rows = matrix rows
        ys, xs = torch.meshgrid(torch.arange(H, device=self.device, dtype=torch.float32) + 0.5,
                                torch.arange(W, device=self.device, dtype=torch.float32) + 0.5, indexing="ij")
        pix = torch.stack([xs.reshape(-1), ys.reshape(-1), torch.ones(H * W, device=self.device)], 1)
        d = pix.unsqueeze(0) @ m.transpose(-1, -2)                      # (C, HW, 3)
        d = d / d.norm(dim=2, keepdim=True)
        o = self.all_camera_origins[cams].unsqueeze(1)                  # (C, 1, 3)
        c, r, alb = self._ellipsoids_normalised(frame)
        C = cams.numel()
        best_t = torch.full((C, H * W), float("inf"), device=self.device)
        color = torch.zeros(C, H * W, 3, device=self.device)
        light = torch.tensor([0.4, -0.7, -0.6], device=self.device)
        light = light / light.norm()
        for k in range(c.shape[0]):
            oc = (o - c[k]) / r[k]                                      # (C, 1, 3)
            dk = d / r[k]
            A = (dk * dk).sum(2)
            B = 2.0 * (dk * oc).sum(2)
            Cq = (oc * oc).sum(2) - 1.0                                 # (C, 1)
            disc = B * B - 4.0 * A * Cq
            t = (-B - torch.sqrt(disc.clamp(min=0))) / (2.0 * A)
            hit = (disc > 0) & (t > 0) & (t < best_t)
            p = o + t.unsqueeze(2) * d
            n = (p - c[k]) / (r[k] * r[k])
            n = n / n.norm(dim=2, keepdim=True).clamp(min=1e-8)
            stripes = 0.85 + 0.15 * torch.sin(60.0 * p[..., 1:2] + 25.0 * p[..., 0:1])
            shade = (0.35 + 0.65 * (n @ (-light)).clamp(min=0)).unsqueeze(2) * stripes
            col = alb[k].view(1, 1, 3) * shade
            color = torch.where(hit.unsqueeze(2), col, color)
            best_t = torch.where(hit, t, best_t)
        mask = torch.isfinite(best_t).float().unsqueeze(2)
        rgba = torch.cat([color * mask, mask], 2)
        return (rgba * 255.0).to(torch.uint8).contiguous()  # data_loader.py:441


class ResidentCapture:
    """Every (camera, frame) image of the capture in HBM: (num_cameras, num_frames, P, 4) uint8. At 4x scale the
    reference's whole training set (160 cameras x 50 frames x 752^2 px) is 18 GB -- 6 % of one MI355X's 288 GB -- so
    the loader's "replacer" never decodes or renders anything during training: refilling a pool slot is one 2.3 MB
    device-to-device copy. The reference keeps a 200-image CPU pool and a thread that decodes JPEGs into it
    (data_loader.py:258-309,396-511) because 24 GB GPUs cannot hold the set."""

    def __init__(self, scene: SyntheticScene, camera_numbers: Sequence[int], cams_per_call: int = 16):
        self.camera_numbers = list(camera_numbers)
        self.frame_numbers = list(scene.frame_numbers)
        self.cam_slot = {c: i for i, c in enumerate(self.camera_numbers)}
        self.frame_slot = {f: i for i, f in enumerate(self.frame_numbers)}
        P = scene.width * scene.height
        self.images = torch.empty(len(self.camera_numbers), len(self.frame_numbers), P, 4, dtype=torch.uint8,
                                  device=scene.device)
        for f in self.frame_numbers:
            for a in range(0, len(self.camera_numbers), cams_per_call):
                cams = self.camera_numbers[a:a + cams_per_call]
                self.images[a:a + len(cams), self.frame_slot[f]].copy_(scene.render_rgba_cameras(cams, f))

    @staticmethod
    def fits(scene: SyntheticScene, num_cameras: int, budget_bytes: int) -> bool:
        return num_cameras * len(scene.frame_numbers) * scene.width * scene.height * 4 <= budget_bytes

    def image(self, camera_number: int, frame: int) -> torch.Tensor:
        return self.images[self.cam_slot[camera_number], self.frame_slot[frame]]


class HostCapture(ResidentCapture):
    """The same store in PINNED HOST memory, for captures that do not fit HBM (1x scale: 36 MB per image, 290 GB for the 160 x 50
    images of configs[2]; 1 000 frames at 4x: 361 GB) -- where the reference keeps its images (data_loader.py:258-309: a CPU pool
    filled by a decoding thread, uploaded per batch). Images are shaded once at set-up and copied to the host; the replacer
    thread's refill of a pool slot stays ONE launch of hrf_pool_replace on the replacer's stream: the kernel reads the image
    straight out of the pinned buffer over the host link (pinned allocations are mapped into the device's address space; 16-byte
    coalesced reads, 8 x 2.3 MB per training step at 4x) and writes the HBM pool, under the training kernels of the other
    streams -- no staging copy, no extra synchronisation, the lock / event discipline of the resident store unchanged.
    `camera_numbers` may be a subset of the rig (what fits the host budget): the loader then trains on those cameras."""

    def __init__(self, scene: SyntheticScene, camera_numbers: Sequence[int], cams_per_call: int = 8):
        self.camera_numbers = list(camera_numbers)
        self.frame_numbers = list(scene.frame_numbers)
        self.cam_slot = {c: i for i, c in enumerate(self.camera_numbers)}
        self.frame_slot = {f: i for i, f in enumerate(self.frame_numbers)}
        P = scene.width * scene.height
        self.images = torch.empty(len(self.camera_numbers), len(self.frame_numbers), P, 4, dtype=torch.uint8, pin_memory=True)
        for f in self.frame_numbers:
            for a in range(0, len(self.camera_numbers), cams_per_call):
                cams = self.camera_numbers[a:a + cams_per_call]
                self.images[a:a + len(cams), self.frame_slot[f]].copy_(scene.render_rgba_cameras(cams, f), non_blocking=True)
        if scene.device.type == "cuda":
            torch.cuda.synchronize()

    @staticmethod
    def cameras_that_fit(scene: SyntheticScene, budget_bytes: int) -> int:
        per_camera = len(scene.frame_numbers) * scene.width * scene.height * 4
        return int(budget_bytes // per_camera)


# ---------------------------------------------------------------------------------------------- loader
class SyntheticDataLoader:
    """Training-mode loader with the reference's interface: iterator yielding InputBatch, mutable
    `batch_size`, `pause_replacing()` / `continue_replacing()` (data_loader.py:54-69,519-531)."""

    def __init__(self, scene: SyntheticScene, batch_size: int = 8192, camera_numbers: Optional[Sequence[int]] = None,
                 max_buffer_size: int = 200, max_num_frames_per_batch: int = 8, seed: int = 123,
                 output_samples: bool = True, occupancy: bool = True, camera_seed: Optional[int] = None,
                 capture: Optional[ResidentCapture] = None, frame_synchronous: bool = False):
        """seed: order in which FRAMES enter the pool; camera_seed (default: seed): order of the cameras of a frame.
        Data-parallel ranks pass the same `seed`, their own `camera_seed` and frame_synchronous=True: the pools then
        hold the same frames on every rank at every replacement count, so each rank knows without communication which
        temporal segments can receive gradients anywhere (TrainEngine._exchange_ranges), while the rays still differ.
        (The flag is a promise by the caller: ranks must also replace in lockstep, which the step-paced replacer does.)
        capture: ResidentCapture to refill pool slots from (one device copy); without it images are rendered on demand."""
        self.scene = scene
        self.device = scene.device
        self.batch_size = batch_size
        self.camera_numbers = list(camera_numbers) if camera_numbers is not None else list(range(len(scene.cameras)))
        self.frame_numbers = list(scene.frame_numbers)
        self.max_num_frames_per_batch = min(max_num_frames_per_batch, len(self.frame_numbers))
        self.rng = np.random.RandomState(seed)
        self.camera_rng = np.random.RandomState(seed if camera_seed is None else camera_seed)
        self.frame_synchronous = bool(frame_synchronous)
        self.capture = capture
        # replacer thread state (data_loader.py:323-335,353-354): the lock serialises pool writes against sampler
        # launches; the events order the device work of the two sides, which run on different streams
        self.data_lock = threading.Lock()
        self.replacer_event = threading.Event()
        self._tick = threading.Condition()
        self._queue: List[Tuple[list, list]] = []   # (pairs, slots) batches the replacer thread still has to load
        self._step = 0
        self._present = {0: set()}                  # step -> frames that were in the pool at some time during that step
        self._replacer_thread: Optional[threading.Thread] = None
        self._replacer_stop = False
        self._replacer_stream: Optional[torch.cuda.Stream] = None
        self.pool_read_done: Optional[torch.cuda.Event] = None    # after the last sampler launch that reads the pool
        self.pool_write_done: Optional[torch.cuda.Event] = None   # after the last pool slot write
        self.replacements = 0
        self.resolution = (max(scene.width, scene.height), min(scene.width, scene.height))
        self.num_pixels_per_camera = scene.width * scene.height
        num_pairs = len(self.camera_numbers) * len(self.frame_numbers)
        self.buffer_size = min(max_buffer_size, num_pairs)
        if self.max_num_frames_per_batch > 1:  # data_loader.py:250-253
            self.buffer_size = min(self.buffer_size, len(self.camera_numbers) * (self.max_num_frames_per_batch - 1))
        om = "samples" if output_samples else "rays"
        sp = "occupancy" if occupancy else "aabb"
        self.ray_sampler_func = getattr(ray_sampler_native, f"get_{om}_{sp}_minmax")  # data_loader.py:173-175
        B, P, dev = self.buffer_size, self.num_pixels_per_camera, self.device
        self.pixel_colors = torch.empty(B, P, 4, dtype=torch.uint8, device=dev)       # HBM resident pool
        self.light_mask = torch.zeros(B, P, 1, dtype=torch.bool, device=dev)
        self.frame_numbers_cuda = torch.full((B,), -1, dtype=torch.int32, device=dev)
        self.camera_numbers_cuda = torch.full((B,), -1, dtype=torch.int32, device=dev)
        self.landscape_mode_cuda = torch.empty(B, dtype=torch.bool, device=dev)
        self.inverse_krs_cuda = torch.empty(B, 3, 3, dtype=torch.float32, device=dev)
        self.camera_origins_cuda = torch.empty(B, 3, dtype=torch.float32, device=dev)
        self.grid_texture_objects_cuda = torch.zeros(B, dtype=torch.int64, device=dev)
        self.aabb = scene.aabb
        self.occupancy_grid_resolution = scene.grid_resolution if occupancy else 0
        self.occupancy = occupancy
        self.frame_to_grid_texture = {}
        if occupancy:
            self.grid_ring = OccupanyGrid(scene.grid_resolution, len(self.frame_numbers))
            for f in self.frame_numbers:  # every frame's grid stays resident
                self.frame_to_grid_texture[f] = self.grid_ring.add_grid(scene.occupancy_grid(f))
        self._slot_frames = [-1] * B            # frame loaded into each slot (as enqueued on the device)
        self._slot_frames_logical = [-1] * B    # ... once every scheduled replacement has been applied
        self._busy = False
        self._replacer_error: Optional[BaseException] = None
        self._spec_host = self._spec_dev = None
        # tables hrf_pool_replace indexes: landscape flag per camera number, grid handle per capture frame index
        self._all_landscape = torch.tensor([1 if (c.width >= c.height) else 0 for c in scene.cameras], dtype=torch.uint8,
                                           device=dev)
        self._grid_by_frame = None
        if occupancy and capture is not None:
            self._grid_by_frame = torch.tensor([self.frame_to_grid_texture[f] for f in capture.frame_numbers],
                                               dtype=torch.int64, device=dev)
        self.camera_frame_pairs = self._camera_frame_pair_generator()
        self.pair_load_index = 0
        pairs, slots = self._schedule(B)
        self._load_many(pairs, slots)
        self.iternum = 0
        self._replacing = False

    def _camera_frame_pair_generator(self):
        """data_loader.py:356-394."""
        if self.max_num_frames_per_batch > 1:
            n_per_frame = int(np.ceil(self.buffer_size / (self.max_num_frames_per_batch - 1)))
        else:
            n_per_frame = len(self.camera_numbers)
        n_per_frame = min(n_per_frame, len(self.camera_numbers))
        state = {f: {"next": 0, "cams": list(self.camera_numbers)} for f in self.frame_numbers}
        frames = list(self.frame_numbers)
        while True:
            self.rng.shuffle(frames)
            for f in frames:
                info = state[f]
                for _ in range(n_per_frame):
                    if info["next"] == 0:
                        self.camera_rng.shuffle(info["cams"])
                    yield info["cams"][info["next"]], f
                    info["next"] = (info["next"] + 1) % len(info["cams"])

    def _load(self, pair: Tuple[int, int], slot: int) -> None:
        """_load_and_copy_camera_frame_data (data_loader.py:424-511): image from the resident capture (one device copy)
        or rendered on demand, per-slot camera tables, grid handle."""
        cam_no, frame = pair
        cam = self.scene.cameras[cam_no]
        src = (self.capture.image(cam_no, frame) if self.capture is not None and cam_no in self.capture.cam_slot
               else self.scene.render_rgba(cam_no, frame))
        self.pixel_colors[slot].copy_(src, non_blocking=True)
        self._slot_frames[slot] = frame
        self.frame_numbers_cuda[slot] = frame
        self.camera_numbers_cuda[slot] = cam_no
        self.landscape_mode_cuda[slot] = cam.width > cam.height if cam.width != cam.height else True
        self.inverse_krs_cuda[slot].copy_(self.scene.all_inverse_krs[cam_no])
        self.camera_origins_cuda[slot].copy_(self.scene.all_camera_origins[cam_no])
        if self.occupancy:
            self.grid_texture_objects_cuda[slot] = self.frame_to_grid_texture[frame]

    def _load_many(self, pairs, slots) -> None:
        """Several slots at once. With a resident capture: ONE launch (hrf_pool_replace) fed by one small pinned
        host-to-device copy; otherwise slot by slot."""
        if self.capture is None:
            for pair, slot in zip(pairs, slots):
                self._load(pair, slot)
            return
        from .. import _lib
        from .._lib import check, ptr, stream_ptr
        cap = self.capture
        # a capture may hold a subset of the rig (HostCapture under a memory budget): pairs it does not hold go slot by slot through
        # _load (shaded on demand), the rest in the one launch below
        missing = [(p, sl) for p, sl in zip(pairs, slots) if p[0] not in cap.cam_slot or p[1] not in cap.frame_slot]
        if missing:
            for pair, slot in missing:
                self._load(pair, slot)
            kept = [(p, sl) for p, sl in zip(pairs, slots) if p[0] in cap.cam_slot and p[1] in cap.frame_slot]
            if not kept:
                return
            pairs, slots = [p for p, _ in kept], [sl for _, sl in kept]
        k = len(pairs)
        if self._spec_host is None or self._spec_host.shape[1] < k:
            self._spec_host = torch.empty(8, max(k, 16), 5, dtype=torch.int32).pin_memory()
            self._spec_dev = torch.empty(8, max(k, 16), 5, dtype=torch.int32, device=self.device)
            self._spec_turn = 0
        turn = self._spec_turn = (self._spec_turn + 1) % self._spec_host.shape[0]
        host, dev = self._spec_host[turn], self._spec_dev[turn]
        for i, ((cam_no, frame), slot) in enumerate(zip(pairs, slots)):
            host[i, 0], host[i, 1], host[i, 2], host[i, 3], host[i, 4] = slot, cap.cam_slot[cam_no], cap.frame_slot[frame], cam_no, frame
            self._slot_frames[slot] = frame
        dev[:k].copy_(host[:k], non_blocking=True)
        occ = self.occupancy
        check(_lib.lib().hrf_pool_replace(ptr(dev), k, ptr(cap.images), self.num_pixels_per_camera, len(cap.frame_numbers),
                                          ptr(self.pixel_colors), ptr(self.scene.all_inverse_krs), ptr(self.scene.all_camera_origins),
                                          ptr(self._all_landscape), ptr(self._grid_by_frame) if occ else None,
                                          ptr(self.frame_numbers_cuda), ptr(self.camera_numbers_cuda),
                                          ptr(self.landscape_mode_cuda.view(torch.uint8)), ptr(self.inverse_krs_cuda),
                                          ptr(self.camera_origins_cuda), ptr(self.grid_texture_objects_cuda) if occ else None,
                                          stream_ptr()))

    def frames_in_pool(self):
        """Frame numbers the pool holds once every scheduled replacement has been applied (host bookkeeping only)."""
        return set(f for f in self._slot_frames_logical if f >= 0)

    def frames_superset(self):
        """Frames that were in the pool at some time during this step or the two before it: a superset of the frames of
        any ray that can still be in flight (the collector draws the rays of step t+1 while step t runs), and a function
        of the replacement schedule only -- identical on every rank of a frame-synchronous data-parallel run, whatever
        the timing of the replacer threads."""
        out = set()
        for k in (self._step, self._step - 1, self._step - 2):
            out |= self._present.get(k, set())
        return out

    def _schedule(self, k: int):
        """Next k (pair, slot) replacements of the schedule + host bookkeeping of the frames."""
        pairs = [next(self.camera_frame_pairs) for _ in range(k)]
        slots = [(self.pair_load_index + i) % self.buffer_size for i in range(k)]
        self.pair_load_index += k
        for (_, f), s in zip(pairs, slots):
            self._slot_frames_logical[s] = f
            self._present[self._step].add(f)
        return pairs, slots

    def replace_next(self) -> None:
        """One iteration of the replacer thread's loop (data_loader.py:396-422), run synchronously on the caller's stream."""
        with self.data_lock:
            cur = torch.cuda.current_stream() if self.pixel_colors.is_cuda else None
            if cur is not None and self.pool_read_done is not None:
                cur.wait_event(self.pool_read_done)
            pairs, slots = self._schedule(1)
            self._load(pairs[0], slots[0])
            self.replacements += 1
            if cur is not None:
                ev = torch.cuda.Event()
                ev.record(cur)
                self.pool_write_done = ev

    # ------------------------------------------------------------------ background replacer (data_loader.py:353-354,396-422)
    def start_replacer(self, replacements_per_tick: int = 1) -> None:
        """Daemon thread that refills pool slots while training runs, like the reference's. The reference's thread is
        paced by JPEG decoding; this one is paced by `tick()` (the training loop calls it once per step) so that a run is
        reproducible and data-parallel ranks replace in lockstep: every tick refills `replacements_per_tick` slots on a
        dedicated stream. Ordering against the sampler, which reads the pool on other streams:
          sampler side (`pool_reader()`):  wait(pool_write_done) -> launches -> record(pool_read_done)
          replacer side:                   wait(pool_read_done)  -> copies   -> record(pool_write_done)
        both under data_lock (held only while enqueuing, as at data_loader.py:417-421,548)."""
        if self._replacer_thread is not None:
            return
        self.replacements_per_tick = int(replacements_per_tick)
        self._replacer_stream = torch.cuda.Stream(device=self.device)
        self._replacer_stop = False
        self.replacer_event.set()
        self._replacer_thread = threading.Thread(target=self._replacer_loop, name="hrf-pool-replacer", daemon=True)
        self._replacer_thread.start()

    def stop_replacer(self) -> None:
        t = self._replacer_thread
        if t is None:
            return
        self.replacer_event.set()
        self.drain_replacer()
        with self._tick:
            self._replacer_stop = True
            self._tick.notify_all()
        self.replacer_event.set()
        t.join()
        self._replacer_thread = None
        if self._replacer_stream is not None:
            torch.cuda.current_stream().wait_stream(self._replacer_stream)

    def tick(self) -> None:
        """A training step begins. Host bookkeeping of the frame window; with the replacer thread running, the next
        `replacements_per_tick` replacements of the schedule are handed to it."""
        with self._tick:
            self._step += 1
            self._present[self._step] = self.frames_in_pool()
            self._present.pop(self._step - 3, None)
            if self._replacer_thread is not None:
                if self._replacer_error is not None:
                    raise RuntimeError("the pool replacer thread died") from self._replacer_error
                self._queue.append(self._schedule(self.replacements_per_tick))
                self._tick.notify()

    def drain_replacer(self) -> None:
        """Block until every scheduled replacement has been enqueued on the replacer stream (tests; end of a region)."""
        if self._replacer_thread is None:
            return
        with self._tick:
            while (self._queue or self._busy) and not self._replacer_stop:
                if self._replacer_error is not None:
                    raise RuntimeError("the pool replacer thread died") from self._replacer_error
                self._tick.wait(0.05)

    def _replacer_loop(self) -> None:
        try:
            if self.device.type == "cuda" and self.device.index is not None:
                torch.cuda.set_device(self.device)
            self._replacer_body()
        except BaseException as e:   # surfaced by drain_replacer / tick in the training thread
            with self._tick:
                self._replacer_error = e
                self._busy = False
                self._tick.notify_all()

    def _replacer_body(self) -> None:
        while True:
            with self._tick:
                while not self._queue and not self._replacer_stop:
                    self._tick.wait()
                if self._replacer_stop:
                    return
                pairs, slots = self._queue.pop(0)
                self._busy = True
            self.replacer_event.wait()          # pause_replacing() / continue_replacing(), data_loader.py:519-523
            with self.data_lock:
                with torch.cuda.stream(self._replacer_stream):
                    if self.pool_read_done is not None:
                        self._replacer_stream.wait_event(self.pool_read_done)
                    self._load_many(pairs, slots)
                    self.replacements += len(pairs)
                    ev = torch.cuda.Event()
                    ev.record(self._replacer_stream)
                    self.pool_write_done = ev
            with self._tick:
                self._busy = False
                self._tick.notify_all()

    class _PoolReader:
        def __init__(self, loader):
            self.loader = loader

        def __enter__(self):
            ld = self.loader
            ld.data_lock.acquire()
            if ld.pool_write_done is not None and ld.pixel_colors.is_cuda:
                torch.cuda.current_stream().wait_event(ld.pool_write_done)
            return ld

        def __exit__(self, *exc):
            ld = self.loader
            if ld.pixel_colors.is_cuda:
                # The pool has readers on two streams (the collector's prefetch stream and the main stream) but the
                # replacer waits on ONE event: chain them -- this stream first waits for the previous reader's event, so
                # the event recorded now implies every earlier read of the pool, whatever stream it ran on.
                cur = torch.cuda.current_stream()
                if ld.pool_read_done is not None:
                    cur.wait_event(ld.pool_read_done)
                ev = torch.cuda.Event()
                ev.record(cur)
                ld.pool_read_done = ev
            ld.data_lock.release()
            return False

    def pool_reader(self):
        """Context for code that launches kernels reading the pool / per-slot tables on the current stream."""
        return SyntheticDataLoader._PoolReader(self)

    def pause_replacing(self):
        self._replacing = False
        self.replacer_event.clear()

    def continue_replacing(self):
        self._replacing = True
        self.replacer_event.set()

    def __iter__(self):
        self.iternum = 0
        self.continue_replacing()
        return self

    def draw_ray_indices(self, batch_size: Optional[int] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        n = batch_size or self.batch_size
        high = self.buffer_size * self.num_pixels_per_camera
        if out is not None:  # caller-owned buffer (no allocation on the step)
            return out[:n].random_(0, high)
        return torch.randint(0, high, size=(n,), dtype=torch.int64, device=self.device)  # data_loader.py:540-546

    def sample(self, ray_indices: torch.Tensor):
        width, height = self.resolution
        with self.pool_reader():   # data_loader.py:548
            return self._sample_locked(ray_indices, width, height)

    def _sample_locked(self, ray_indices: torch.Tensor, width: int, height: int):
        return self.ray_sampler_func(
            self.pixel_colors.view(-1, 4), self.light_mask.view(-1), self.frame_numbers_cuda, self.camera_numbers_cuda,
            self.grid_texture_objects_cuda, self.landscape_mode_cuda, ray_indices, self.inverse_krs_cuda,
            self.camera_origins_cuda, self.aabb, self.occupancy_grid_resolution, width, height, 4e-4, False)

    def validation_batches(self, camera_number: int, frame: int, batch_size: int = 8192):
        """Validation / test branch of DataLoader.__next__ (data_loader.py:576-624) for one (camera, frame) image:
        consecutive `arange` pixel ranges, one-slot camera tables, no light-bloom filtering; yields InputBatch objects
        whose `ray_masks` cover the whole range (rays that miss the occupancy grid are dropped by the sampler)."""
        dev = self.device
        width, height = self.resolution
        P = self.num_pixels_per_camera
        rgba = (self.capture.image(camera_number, frame).to(dev) if self.capture is not None and camera_number in self.capture.cam_slot
                else self.scene.render_rgba(camera_number, frame))
        cam = self.scene.cameras[camera_number]
        frames = torch.tensor([frame], dtype=torch.int32, device=dev)
        cams = torch.tensor([camera_number], dtype=torch.int32, device=dev)
        land = torch.tensor([cam.width >= cam.height], dtype=torch.bool, device=dev)
        if self.occupancy:
            tex = torch.tensor([self.frame_to_grid_texture[frame]], dtype=torch.int64, device=dev)
        else:
            tex = torch.zeros(1, dtype=torch.int64, device=dev)
        ikr = self.scene.all_inverse_krs[camera_number:camera_number + 1].contiguous()
        org = self.scene.all_camera_origins[camera_number:camera_number + 1].contiguous()
        light = torch.zeros(P, dtype=torch.bool, device=dev)
        for start in range(0, P, batch_size):
            idx = torch.arange(start, min(start + batch_size, P), dtype=torch.int64, device=dev)
            out = self.ray_sampler_func(rgba, light, frames, cams, tex, land, idx, ikr, org, self.aabb,
                                        self.occupancy_grid_resolution, width, height, 4e-4, False)
            yield InputBatch(
                ray_origins=out[0].view(-1, 3), ray_directions=out[1].view(-1, 3), rgba=out[2].view(-1, 4),
                frame_numbers=out[3].view(-1, 1), camera_numbers=out[4].view(-1, 1), minmaxes=out[5].view(-1, 2),
                ray_masks=out[6].view(-1, 1), unique_frame_numbers=frames.view(-1, 1),
                sample_distances=out[7].view(-1, 1), ray_indices=out[8].view(-1).long(), width=width, height=height)

    def __next__(self) -> InputBatch:
        width, height = self.resolution
        ray_indices = self.draw_ray_indices()
        (ray_origins, ray_directions, rgba, frame_numbers, camera_numbers, minmaxes, ray_masks, distance_per_sample,
         relative_ray_indices_per_sample) = self.sample(ray_indices)
        self.iternum += ray_indices.numel()
        return InputBatch(  # data_loader.py:633-660
            ray_origins=ray_origins.view(-1, 3), ray_directions=ray_directions.view(-1, 3),
            minmaxes=minmaxes.view(-1, 2), rgba=rgba.view(-1, 4), ray_masks=ray_masks.view(-1, 1),
            frame_numbers=frame_numbers.view(-1, 1), camera_numbers=camera_numbers.view(-1, 1),
            unique_frame_numbers=torch.unique(frame_numbers, sorted=False, return_inverse=False).view(-1, 1),
            sample_distances=distance_per_sample.view(-1, 1),
            ray_indices=relative_ray_indices_per_sample.view(-1).long(), width=width, height=height)
