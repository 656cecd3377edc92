"""CPU tests (-m "not gpu"): the C-ABI library loads and exports what include/icet_hip.h declares, fails
loudly without a GPU, and the host-side pieces (synthetic scans, pair sharding over ranks) behave."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = set()
    for h in ("icet_hip.h", "icet_nodes.h", "icet_io.h"):
        txt = open(os.path.join(ROOT, "include", h)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names |= set(re.findall(r"\b(icet_[a-z_0-9]+)\s*\(", txt))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    import icet_amd
    from icet_amd import api
    lib = icet_amd.load_library()
    declared = _declared_symbols()
    assert set(declared) == set(api.EXPORTED_SYMBOLS), declared
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.icet_version().startswith(b"icet_hip")


def test_header_compiles_as_plain_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "icet_hip.h"\n#include "icet_nodes.h"\n#include "icet_io.h"\nint main(void){ icet_params p = {7,24,75,25,0.1f,0.1f,0}; icet_node_params q = {{7,24,75,25,0.1f,0.1f,0}, 2.0f, 1, 0.f, 0.f, 0, 0, 0}; '
                   'return p.runlen == 7 && q.seed_x0 == 1 ? 0 : 1; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(tmp_path / "t")])
    subprocess.check_call([str(tmp_path / "t")])


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_fails_loudly_without_gpu():
    import icet_amd
    with pytest.raises(icet_amd.IcetError) as e:
        icet_amd.Context(0)
    assert e.value.status == icet_amd.api.ICET_ERR_NO_DEVICE
    with pytest.raises(icet_amd.IcetError):
        icet_amd.ICET(np.zeros((4, 3), np.float32), np.zeros((4, 3), np.float32), 7, np.zeros(6), 24, 75)


def test_null_and_bad_arguments_do_not_crash():
    import icet_amd
    lib = icet_amd.load_library()
    assert lib.icet_create(None, 0, None) == icet_amd.api.ICET_ERR_BAD_ARG
    assert lib.icet_destroy(None) == icet_amd.api.ICET_ERR_BAD_ARG
    assert lib.icet_sync(None) == icet_amd.api.ICET_ERR_BAD_ARG
    assert lib.icet_last_error(None) == b"null context"
    assert lib.icet_node_create(None, None, None) == icet_amd.api.ICET_ERR_BAD_ARG
    nh = C.c_void_p()
    assert lib.icet_node_create(None, C.byref(icet_amd.api.node_params()), C.byref(nh)) == icet_amd.api.ICET_ERR_BAD_ARG and not nh.value
    assert lib.icet_node_destroy(None) == icet_amd.api.ICET_ERR_BAD_ARG
    assert lib.icet_node_push(None, None, 0, 0, None) == icet_amd.api.ICET_ERR_BAD_ARG
    assert lib.icet_stream(None) is None and lib.icet_device(None) == -1
    h = C.c_void_p()
    assert lib.icet_create(C.byref(h), -1, None) in (icet_amd.api.ICET_ERR_NO_DEVICE,)
    assert not h.value


def test_multi_device_entry_validates_arguments_without_gpu():
    """icet_multi_* (the native multi-GPU entry of the batched-pairs case): argument checks run before any device is touched, and
    without a device the handle cannot be created -- loudly, with the documented status."""
    import icet_amd
    lib = icet_amd.load_library()
    A = icet_amd.api
    h = C.c_void_p()
    ids = (C.c_int32 * 2)(0, 0)
    assert lib.icet_multi_create(None, ids, 1) == A.ICET_ERR_BAD_ARG
    assert lib.icet_multi_create(C.byref(h), None, 1) == A.ICET_ERR_BAD_ARG and not h.value
    assert lib.icet_multi_create(C.byref(h), ids, 0) == A.ICET_ERR_BAD_ARG and not h.value
    if not torch.cuda.is_available():
        assert lib.icet_multi_create(C.byref(h), ids, 1) == A.ICET_ERR_NO_DEVICE and not h.value
        with pytest.raises(icet_amd.IcetError) as e:
            icet_amd.MultiContext([0])
        assert e.value.status == A.ICET_ERR_NO_DEVICE
    assert lib.icet_multi_destroy(None) == A.ICET_ERR_BAD_ARG
    assert lib.icet_multi_last_error(None) == b"null handle" and lib.icet_multi_devices(None) == 0 and lib.icet_multi_context(None, 0) is None
    assert lib.icet_multi_solve_batch(None, None, 0, None, None, None, None, None, None, None, None) == A.ICET_ERR_BAD_ARG
    assert lib.icet_multi_solve_batch_device(None, None, 0, None, None, None, None) == A.ICET_ERR_BAD_ARG
    assert lib.icet_multi_solve_batch_device_after(None, None, 0, None, None, None, None, None) == A.ICET_ERR_BAD_ARG
    assert lib.icet_multi_set_option(None, b"gather", 1.0) == A.ICET_ERR_BAD_ARG
    assert lib.icet_set_option(None, b"lds_slots", 1.0) == A.ICET_ERR_BAD_ARG


# ------------------------------------------------------------------ bench.py: `--gpus N` measures N GPUs or refuses (missing #1)
def _bench(*argv, env=None, timeout=300):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=e, capture_output=True, text=True, timeout=timeout)


def test_bench_plans_its_processes_before_touching_the_gpu():
    import json
    # no launcher in the environment: N > 1 starts one -- the driver's own command line -- as a child
    out = _bench("--gpus", "8", "--steps", "5", "--warmup", "2", "--dry-run-launch")
    d = json.loads(out.stdout)
    cmd = d["command"]
    assert d["action"] == "spawn" and cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"] and "--dry-run-launch" not in cmd
    # under the driver's launcher the script is a rank; a launcher of another size is refused, never silently shrunk
    assert json.loads(_bench("--gpus", "8", "--dry-run-launch", env={"WORLD_SIZE": "8", "RANK": "3"}).stdout)["action"] == "run"
    assert json.loads(_bench("--gpus", "8", "--dry-run-launch", env={"WORLD_SIZE": "4"}).stdout)["action"] == "refuse"
    assert json.loads(_bench("--gpus", "8", "--dry-run-launch", env={"WORLD_SIZE": "1"}).stdout)["action"] == "refuse"
    assert json.loads(_bench("--gpus", "1", "--dry-run-launch").stdout)["action"] == "run"
    # the one-process form never runs under a multi-rank launcher
    assert json.loads(_bench("--gpus", "4", "--multi", "--dry-run-launch").stdout)["action"] == "run"
    assert json.loads(_bench("--gpus", "4", "--multi", "--dry-run-launch", env={"WORLD_SIZE": "4"}).stdout)["action"] == "refuse"
    r = _bench("--gpus", "8", env={"WORLD_SIZE": "4"})
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr and not r.stdout.strip()


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode of the self-launched ranks")
def test_bench_self_launch_fails_loudly_without_gpus():
    """`python bench.py --gpus 2` really starts two ranks through torch.distributed.run; on this box they find no GPU, and the parent
    relays the failure: non-zero exit, no JSON line -- never a silent one-GPU (or zero-GPU) measurement."""
    r = _bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", timeout=600)
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "starting" in r.stderr and "torch.distributed.run" in r.stderr and "needs a GPU" in r.stderr


def test_eigen_adapter_header_compiles_against_the_mock(tmp_path):
    """include/icet.h must at least COMPILE and link everywhere (the run against a device is in tests/test_gpu_parity.py); without a
    GPU the constructed object carries an error status instead of throwing."""
    exe = str(tmp_path / "adapter_demo")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "tests", "cpp", "mock_eigen"), "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "adapter_demo.cpp"), "-L", os.path.join(ROOT, "icet_amd", "lib"), "-licet_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "icet_amd", "lib"), "-o", exe])
    if not torch.cuda.is_available():
        np.zeros((3, 100), np.float32).tofile(str(tmp_path / "z.f32"))
        out = subprocess.run([exe, str(tmp_path / "z.f32"), str(tmp_path / "z.f32"), "100", "100"], capture_output=True, text=True, timeout=120)
        assert out.returncode == 1 and "status" in out.stderr


def test_no_oracle_in_product_path():
    """The shipped package must not import, link or call anything under oracle/ (or the reference)."""
    for dp, _, fs in os.walk(os.path.join(ROOT, "icet_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")) or f == "Makefile":
                txt = open(os.path.join(dp, f)).read()
                assert "pyoracle" not in txt and "icet_oracle" not in txt and "import oracle" not in txt and "from oracle" not in txt, os.path.join(dp, f)
    out = subprocess.run(["ldd", os.path.join(ROOT, "icet_amd", "lib", "libicet_hip.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out


# ------------------------------------------------------------------ synthetic scans
def test_lidar_sim_deterministic_and_plausible():
    from icet_amd import lidar_sim as ls
    s1, s2, xt = ls.make_pair(rings=16, steps=512)
    t1, t2, _ = ls.make_pair(rings=16, steps=512)
    assert torch.equal(s1, t1) and torch.equal(s2, t2)
    assert s1.dtype == torch.float32 and s1.shape[0] == 3 and s1.is_contiguous()
    assert 0.8 * 16 * 512 < s1.shape[1] <= 16 * 512
    r = torch.linalg.norm(s1, dim=0)
    assert float(r.min()) > 0.5 and float(r.max()) <= 121.0
    a, _, _ = ls.make_pair(rings=16, steps=512, order="azimuth")
    assert a.shape == s1.shape and not torch.equal(a, s1)
    assert np.allclose(np.sort(a.numpy()[0]), np.sort(s1.numpy()[0]), atol=0.2)     # same rays (noise drawn in storage order)
    assert np.allclose(xt, ls.DEFAULT_MOTION)
    m0, m1 = ls.batch_motion(0), ls.batch_motion(1)
    assert not np.allclose(m0, m1) and np.all(np.abs(m0) <= [0.6, 0.05, 0.02, 0.005, 0.005, 0.02])


def test_lidar_sim_pair_is_registrable_by_the_oracle():
    from icet_amd import lidar_sim as ls
    from oracle import pyoracle as po
    s1, s2, xt = ls.make_pair()           # BASELINE config 2: 64 x 2048 rays, seeds 1000/1001
    assert 110_000 < s1.shape[1] < 131_072
    o = po.solve(s1.T.numpy(), s2.T.numpy(), trace=True)
    assert np.abs(o["X"][:3] - xt[:3]).max() < 0.03 and np.abs(o["X"][3:] - xt[3:]).max() < 2e-3
    assert int(o["trace"]["used"][-1].sum()) > 100 and o["n_ub_voxels"] == 0


# ------------------------------------------------------------------ sharding pairs over ranks (gloo, world_size 2)
_WORKER = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from icet_amd.dist import solve_sharded, shard_indices, gather_results
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
n_pairs = int(sys.argv[2])
def fake_solver(ids):                      # result row of global pair k is a pure function of k
    return torch.tensor([[k * 100.0 + j for j in range(48)] for k in ids], dtype=torch.float32).reshape(-1, 48)
full = solve_sharded(n_pairs, fake_solver, torch.device("cpu"))
want = fake_solver(list(range(n_pairs)))
assert full.shape == (n_pairs, 48) and torch.equal(full, want), (rank, full[:, 0])
assert shard_indices(n_pairs, rank, world) == list(range(rank, n_pairs, world))
# real solver on CPU: the oracle stands in for the GPU on this box (test infrastructure only)
from oracle import pyoracle as po
d = np.load(os.path.join(sys.argv[1], "tests", "golden", "scans_frame_804_805.npz"))
a, b = d["scan1"][::4], d["scan2"][::4]
x0s = np.array([[0.01 * k, 0, 0, 0, 0, 0.001 * k] for k in range(3)], np.float32)
def oracle_solver(ids):
    rows = []
    for k in ids:
        o = po.solve(a, b, x0=x0s[k], runlen=2)
        rows.append(np.concatenate([o["X"], o["pred_stds"], o["cov"].ravel()]))
    return torch.tensor(np.array(rows), dtype=torch.float32).reshape(-1, 48)
full = solve_sharded(3, oracle_solver, torch.device("cpu"))
want = oracle_solver([0, 1, 2])
assert torch.equal(full, want)
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


@pytest.mark.parametrize("n_pairs", [6, 5])
def test_round_robin_shard_and_gather_gloo_world2(tmp_path, n_pairs):
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + n_pairs), WORLD_SIZE="2", OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(n_pairs)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


def test_gather_single_process_identity():
    from icet_amd.dist import gather_results, max_shard_size, shard_size
    x = torch.arange(3 * 48, dtype=torch.float32).reshape(3, 48)
    assert torch.equal(gather_results(x, 3, rank=0, world=1), x)
    assert max_shard_size(2048, 8) == 256 and shard_size(5, 1, 2) == 2
    with pytest.raises(ValueError):
        gather_results(x, 4, rank=0, world=1)


def test_ragged_batch_slot_order(tmp_path):
    """The slot order of a ragged throughput batch (icet_amd/csrc/icet_layout.h): a permutation, every XCD's share of the rows within a few per cent of the mean where
    the caller's order gives some XCDs twice the others', large and small pairs alternating down each XCD's column of slots (tests/cpp/test_layout.cpp; the GPU side --
    same bits per pair, the caller's order of X0 and results -- is tests/test_gpu_parity.py::test_ragged_batch_is_laid_out_xcd_balanced_with_the_callers_order_kept)."""
    import subprocess
    src = os.path.join(ROOT, "tests", "cpp", "test_layout.cpp")
    exe = str(tmp_path / "test_layout")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", src, "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith("OK"), (r.returncode, r.stdout, r.stderr)


def test_downsample_shuffle_head_equals_std_shuffle(tmp_path):
    """The map maker's down-sample indices (icet_amd/csrc/icet_shuffle.h): the first entries of iota + std::shuffle and the generator's state afterwards, tracked
    without the n-entry vector, with the library's distribution objects and with the generator and the bounded draw written out -- against std::shuffle itself, both of
    libstdc++'s branches, 60 chained frames of random sizes (tests/cpp/test_shuffle.cpp).  The node runs the same comparison when it is created and falls back to the
    plain form if it fails."""
    import subprocess
    src = os.path.join(ROOT, "tests", "cpp", "test_shuffle.cpp")
    exe = str(tmp_path / "test_shuffle")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", src, "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith("OK"), (r.returncode, r.stdout, r.stderr)


def test_multi_gpu_scheduler_with_fake_devices(tmp_path):
    """The host-side scheduler of icet_multi_* (icet_amd/csrc/icet_multi_sched.h: one thread per device, the two-phase protocol around a collective gather) driven
    on the CPU with 8 fake devices and a gather that blocks until every rank has entered it: a failure injected on one rank BEFORE the collective (by status and by
    exception) hangs nobody, every rank's failure is surfaced by the sync, a failing preparation queues nothing (tests/cpp/test_multi_sched.cpp; SURVEY 8(e): no run
    on N > 1 GPUs exists, so this is where the protocol is tested).  Also under ThreadSanitizer when the toolchain has it."""
    import subprocess, shutil
    src = os.path.join(ROOT, "tests", "cpp", "test_multi_sched.cpp")
    exe = str(tmp_path / "test_multi_sched")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-Wall", "-Werror", "-I", ROOT, src, "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.returncode, r.stdout, r.stderr)
    tsan = str(tmp_path / "test_multi_sched_tsan")
    if subprocess.run(["g++", "-std=c++17", "-O1", "-pthread", "-fsanitize=thread", "-I", ROOT, src, "-o", tsan], capture_output=True).returncode == 0:
        r = subprocess.run([tsan], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and r.stdout.strip().endswith("ok") and "ThreadSanitizer" not in r.stderr, (r.returncode, r.stdout, r.stderr[-2000:])
