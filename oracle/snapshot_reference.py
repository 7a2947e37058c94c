"""oracle/snapshot_reference.py -- TEST INFRASTRUCTURE ONLY.

Puts the reference's own Python for the hot path next to the oracle, under oracle/_ref/reference/, so that the `-m gpu`
tests can execute the REFERENCE'S SOURCE over libhrf_hip.so on the GPU box (tests/test_gpu_reference_dropin.py):
/root/reference only exists in the build container, which has no GPU; oracle/_ref/ is git-ignored (nothing of the
reference enters the repository's history) but not gpurun-ignored, so the snapshot travels with the built libraries.

`snapshot()` is called by `__graft_entry__.build()` whenever /root/reference is present. It copies the .py files listed
below byte for byte and records their SHA-256 in MANIFEST.txt. Nothing under humanrf_amd/, bench.py's timed region or
smoke()'s product path reads oracle/_ref (tests/test_cpu_golden_and_host.py checks the package never imports oracle/)."""
from __future__ import annotations

import hashlib
import os
import shutil

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCE_ROOT = "/root/reference"
SNAPSHOT_ROOT = os.path.join(HERE, "_ref", "reference")

# what oracle/ref_harness.py imports (and what those modules import from their own packages; the reference's
# directories are namespace packages: it ships no __init__.py files)
FILES = (
    "humanrf/input.py",
    "humanrf/trainer.py",
    "humanrf/volume_rendering.py",
    "humanrf/adaptive_temporal_partitioning.py",
    "humanrf/args/model_args.py",
    "humanrf/args/run_args.py",
    "humanrf/scene_representation/humanrf.py",
    "humanrf/scene_representation/decomposition4d.py",
    "humanrf/scene_representation/query_io.py",
    "humanrf/utils/activation.py",
    "humanrf/utils/loss.py",
    "humanrf/utils/memory.py",
    "actorshq/dataset/input_batch.py",
    "actorshq/dataset/data_loader.py",
    "actorshq/dataset/camera_data.py",
    "actorshq/dataset/aabb_data.py",
    "actorshq/dataset/volumetric_dataset.py",
    "actorshq/dataset/trajectory.py",
    "actorshq/evaluation/presets.py",
)


def snapshot() -> bool:
    """-> True when the snapshot was (re)written; False when /root/reference is absent (the GPU box: nothing to do)."""
    if not os.path.isdir(os.path.join(SOURCE_ROOT, "humanrf")):
        return False
    if os.path.isdir(SNAPSHOT_ROOT):
        shutil.rmtree(SNAPSHOT_ROOT)
    lines = []
    for rel in FILES:
        src = os.path.join(SOURCE_ROOT, rel)
        dst = os.path.join(SNAPSHOT_ROOT, rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(src, dst)
        lines.append(f"{hashlib.sha256(open(src, 'rb').read()).hexdigest()}  {rel}")
    with open(os.path.join(SNAPSHOT_ROOT, "MANIFEST.txt"), "w") as f:
        f.write("# byte copies of /root/reference files, made by oracle/snapshot_reference.py; git-ignored test input\n")
        f.write("\n".join(lines) + "\n")
    return True


if __name__ == "__main__":
    print("snapshot written" if snapshot() else "no /root/reference here")
