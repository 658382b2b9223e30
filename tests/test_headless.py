"""The headless C++ driver (constraint_solver_amd/host/xpbd_headless.cpp): built by build(), fails loudly
without a GPU, and on a GPU runs the solver alone and reports body*substeps/s as one JSON line."""
import json
import os
import subprocess

import pytest

from constraint_solver_amd import capi

EXE = os.path.join(capi.LIB_DIR, "xpbd_headless")


def test_headless_driver_is_built_and_rejects_bad_arguments():
    assert os.access(EXE, os.X_OK)
    p = subprocess.run([EXE, "--bogus"], capture_output=True, text=True, timeout=60)
    assert p.returncode == 2 and "unknown argument" in p.stderr


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="this check is for the GPU-less build container")
def test_headless_driver_fails_loudly_without_a_gpu():
    p = subprocess.run([EXE, "--bodies", "32", "--frames", "1"], capture_output=True, text=True, timeout=60)
    assert p.returncode == 1 and "xpbd error -5" in p.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--mode", "substep"], ["--mode", "contacts", "--scene", "stacks"]])
def test_headless_driver_runs_on_the_gpu(tmp_path, extra):
    dump = tmp_path / "poses.bin"
    p = subprocess.run([EXE, "--bodies", "4096", "--substeps", "20", "--frames", "3", "--warmup", "1", "--dump", str(dump)] + extra,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["bodies"] == 4096 and out["body_substeps_per_s"] > 1e6
    assert dump.stat().st_size == 4096 * 304


@pytest.mark.gpu
def test_headless_driver_sharded_world_equals_the_single_world(tmp_path):
    """--shards N: the compiled host side above the multi-GPU ABI (world::ShardedWorld over xpbd_multi_world_*), here
    with the shards sharing the box's one device.  Its dump equals the unsharded run's, byte for byte."""
    base = [EXE, "--bodies", "4096", "--substeps", "10", "--frames", "4", "--warmup", "1", "--mode", "contacts", "--scene", "stacks"]
    one, three = tmp_path / "one.bin", tmp_path / "three.bin"
    p = subprocess.run(base + ["--dump", str(one)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    p = subprocess.run(base + ["--shards", "3", "--dump", str(three)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["shards"] == 3 and out["ghosts"] > 0 and out["transport"] in ("local", "rccl")
    assert one.read_bytes() == three.read_bytes()
    p = subprocess.run(base[:-4] + ["--shards", "2"], capture_output=True, text=True, timeout=60)   # not in contacts mode
    assert p.returncode == 2
