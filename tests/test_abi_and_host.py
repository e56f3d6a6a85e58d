"""CPU: the C-ABI library loads and exports every symbol include/macvo_hip.h declares; host-side logic; gloo gather."""
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "macvo_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mv_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import macvo_amd._lib as L

    if not os.path.exists(L.LIB_PATH):
        import __graft_entry__ as g

        g.build()
    lib = L.load()
    decl = _declared_symbols()
    assert len(decl) >= 14
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in macvo_hip.h but not exported"
        assert name in L.SIGNATURES, f"{name} has no ctypes signature in _lib.py"
    assert sorted(L.SIGNATURES) == decl
    assert lib.mv_abi_version() == L.ABI_VERSION
    assert lib.mv_error_string(-4) == b"workspace too small"
    assert lib.mv_kp_select_workspace_bytes(480, 640) >= 2 * 480 * 640 * 4


def test_lm_defaults_match_reference_config():
    """init_context / _optimize constants (Optimizer.py:68-75,93)."""
    from macvo_amd import ops

    p = ops.lm_default_params()
    assert (p.huber_delta, p.radius, p.reject, p.max_steps, p.patience, p.decreasing) == (0.1, 1e3, 16, 10, 2, 1e-5)
    assert (p.diag_min, p.diag_max, p.pinv_rcond) == (1e-6, 1e32, 1e-15)


def test_ops_refuse_cpu_tensors():
    """no CPU fallback: the product path must fail loudly off-GPU"""
    from macvo_amd import _lib as L
    from macvo_amd import ops

    with pytest.raises(L.MacvoHipError):
        ops.corr_volume(torch.zeros(1, 16, 2, 2), torch.zeros(1, 16, 2, 2))


def test_product_package_does_not_import_oracle():
    out = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import macvo_amd, macvo_amd.ops, "
                          "macvo_amd.pipeline, macvo_amd.distributed, macvo_amd.plugins; "
                          "print(any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules))" % ROOT],
                         capture_output=True, text=True, check=True)
    assert out.stdout.strip() == "False"


def test_shard_sequences():
    from macvo_amd.distributed import shard_sequences

    owned = [shard_sequences(10, r, 4) for r in range(4)]
    assert sorted(sum(owned, [])) == list(range(10)) and owned[1] == [1, 5, 9]


WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from macvo_amd.distributed import gather_poses
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
T = 5 + rank                                   # ragged: rank r tracked 5 + r frames
poses = torch.full((T, 7), float(rank)) + torch.arange(T)[:, None]
out = gather_poses(poses, dist, lengths=torch.tensor([T]))
assert out.shape == (world, 5 + world - 1, 7), out.shape
for r in range(world):
    assert torch.equal(out[r, : 5 + r], torch.full((5 + r, 7), float(r)) + torch.arange(5 + r)[:, None])
    assert out[r, 5 + r:].abs().sum() == 0
same = gather_poses(torch.full((4, 7), float(rank)), dist)
assert same.shape == (world, 4, 7) and all(float(same[r].mean()) == r for r in range(world))
dist.destroy_process_group()
print("OK", rank)
'''


def test_gloo_world2_pose_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29571")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29571", str(script)],
                         capture_output=True, text=True, env=env, timeout=240)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("OK") == 2
