"""REAL multi-device correctness (SURVEY.md section 4 "on the GPU box with 2/4/8 ranks"): one process per GPU,
backend "nccl" = RCCL over xGMI, fresh `spawn` children.  Skipped on a box with one GPU (the pool's test boxes): there the SAME worker
functions run as gloo ranks sharing the one device (tests/test_dp_training_gpu.py, tests/test_volume_ranks_gpu.py) and every collective
runs through RCCL in a world of one rank (tests/test_rccl_gpu.py).  What this file adds on a multi-GPU node is the part those cannot
reach: device-to-device transfers over xGMI, RCCL's stream ordering against the kernels on different devices, sub-communicators.

  (i)   data-parallel `train_step_2d` (BASELINE configs[4] in miniature): all-reduced per-rank gradients equal the single-process
        full-batch gradients, identical parameters on every rank after the step; the decoder-only and the 3-D (BPTT) steps likewise;
  (ii)  `segment_volume` on 2 ranks and on min(4, all) ranks: every slice and object bit-equal to the single-rank result -- key-split
        (one object), object-sharded (objects >= ranks) and, from 3 ranks on, the object x key hybrid;
  (iii) `bench.py --gpus N` (`--mode volume` and the default 2-D line) returns rc 0 with `n_gpus: N` on the line."""
import json
import os
import subprocess
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
N_GPUS = torch.cuda.device_count()          # (does not initialise the runtime)
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(N_GPUS < 2, reason="needs >= 2 GPUs: one RCCL rank per device")]


def _spawn(worker, world, port_base, extra=(), timeout=900):
    os.environ["MSAM2_TEST_REAL_DEVICES"] = "1"          # inherited by the spawned ranks: helpers.init_test_process_group -> nccl, device = rank
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = port_base + (os.getpid() % 90)
        procs = [ctx.Process(target=worker, args=(r, world, port, q) + tuple(extra)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout)
            assert p.exitcode == 0, f"rank process exited {p.exitcode}"
        return sorted(q.get(timeout=10) for _ in range(world))
    finally:
        os.environ.pop("MSAM2_TEST_REAL_DEVICES", None)


def test_rccl_data_parallel_decoder_step():
    import test_dp_training_gpu as t
    for rank, rel, same in _spawn(t._worker, 2, 32100):
        assert rel < 1e-2 and same, (rank, rel, same)


def test_rccl_data_parallel_full_train_step():
    import test_dp_training_gpu as t
    for rank, rels, same, finite in _spawn(t._full_worker, 2, 32200):
        assert rels["image_encoder"] < 0.1 and rels["decoder"] < 0.1 and rels["memory_attention"] < 0.1, (rank, rels)
        assert same and finite, rank


def test_rccl_data_parallel_joint_step_shares_the_loss_scale():
    import test_dp_training_gpu as t
    (_, loc0, sh0, same0, fin0), (_, loc1, sh1, same1, fin1) = _spawn(t._joint_worker, 2, 32300)
    assert sh0 == sh1 == min(loc0, loc1) and loc0 != loc1 and same0 and same1 and fin0 and fin1


def test_rccl_data_parallel_train_step_3d():
    import test_dp_training_gpu as t
    for rank, frac, same, finite in _spawn(t._bptt_worker, 2, 32400):
        assert same and finite and frac > 0.97, (rank, frac, same, finite)


def test_rccl_segment_volume_two_ranks_equal_one_rank():
    import test_volume_ranks_gpu as t
    for rank, res in _spawn(t._worker, 2, 32500):
        for n_obj, n_cond, same, kv_calls, lay in res:
            assert same, (rank, n_obj, n_cond, lay)
            assert (kv_calls > 0) == (n_obj < 2), (rank, n_obj, kv_calls)


@pytest.mark.skipif(N_GPUS < 3, reason="the object x key hybrid needs 1 < objects < ranks")
def test_rccl_segment_volume_hybrid_and_all_visible_gpus():
    import test_volume_ranks_gpu as t
    world = min(N_GPUS, 4)
    # objects 2 (hybrid: two groups), 1 (one key-split group of `world` ranks), world + 1 (object shards, ragged)
    cases = [(2, (0, 4)), (1, (0, 4)), (world + 1, (0, 4))]
    for rank, res in _spawn(t._worker, world, 32600, extra=(cases,)):
        for (n_obj, n_cond, same, kv_calls, lay), (n_want, _) in zip(res, cases):
            assert same, (rank, n_obj, lay)
            assert lay["groups"] == min(n_want, world), (rank, lay)
            assert (kv_calls > 0) == (lay["key_split_ranks"] > 1), (rank, kv_calls, lay)


@pytest.mark.parametrize("mode", ["volume", "2d"])
def test_bench_on_all_visible_gpus(mode):
    """the driver's SCALE command in miniature: `bench.py --gpus N` launches its own ranks; rc 0, one JSON line, n_gpus == N, no
    error object on it (a hung exchange would exit 3: bench.Watchdog)"""
    n = min(N_GPUS, 8)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1"]
    cmd += ["--mode", "volume", "--slices", "32"] if mode == "volume" else ["--no-volume", "--no-rooflines"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == n and line["value"] > 0 and line["scaling"] == ("strong" if mode == "volume" else "weak")
    assert "rehearsal" not in line["config"] and not any(isinstance(v, dict) and "error" in v for v in line.values())
