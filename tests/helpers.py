"""Shared test helpers (CPU side)."""
import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_npz(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def load_meta():
    with open(os.path.join(GOLDEN, "meta.json")) as f:
        return json.load(f)


def sub(t: torch.Tensor, n: int = 4096) -> np.ndarray:
    """Same deterministic strided subsample as tests/golden/make_golden.py:sub."""
    f = t.detach().float().contiguous().flatten()
    step = max(1, f.numel() // n)
    return f[::step][:n].numpy().copy()


def rel_err(a, b) -> float:
    a = torch.as_tensor(np.asarray(a), dtype=torch.float64)
    b = torch.as_tensor(np.asarray(b), dtype=torch.float64)
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_abs(a, b) -> float:
    a = torch.as_tensor(np.asarray(a), dtype=torch.float64)
    b = torch.as_tensor(np.asarray(b), dtype=torch.float64)
    return float((a - b).abs().max())


def op16_is_fp16() -> bool:
    """operand type of the loaded library (fp16 by default; bf16 with MSAM2_LIB_PATH=libmsam2_hip_bf16.so)"""
    import medical_sam2_amd.ops as ops
    return ops.OP16 == torch.float16


def btol(tol_fp16: float, factor: float = 4.0) -> float:
    """The STATED bf16 tolerance of a bound written for the default fp16 operands: bf16 keeps 8 significand bits against fp16's 11, an 8x
    coarser unit round-off per operand; through a few dozen independent roundings the measured error ratios are 2-3.5x (round 3:
    gpurun_out/bf16_suite.log), so the bf16 bound is 4x the fp16 bound unless a test says otherwise."""
    return tol_fp16 if op16_is_fp16() else tol_fp16 * factor


def mask_iou(a, b) -> float:
    a = torch.as_tensor(np.asarray(a)) > 0
    b = torch.as_tensor(np.asarray(b)) > 0
    u = (a | b).sum().item()
    return 1.0 if u == 0 else (a & b).sum().item() / u


def gradient_step_drops(loss_fn, groups: dict, grads: dict, loss0: float, frac: float = 0.02) -> dict:
    """The loss-goes-down statement that the arithmetic guarantees ("the loss is lower after n Adam steps at lr x" is
    not one: Adam's first update moves every parameter by +-lr whatever the gradient's size).  Per group {name: module} with TRUE
    gradients grads[name] = {parameter: tensor}: the plain step p <- p - eta g with eta = frac * loss0 / |g|^2 has a first-order drop of
    frac * loss0.  Returns {name: measured relative drop}; the parameters are restored (loss_fn must rebuild stale weight copies itself,
    which the modules' version-checked caches do)."""
    out = {}
    with torch.no_grad():
        for name, mod in groups.items():
            g = grads[name]
            params = dict(mod.named_parameters())
            g2 = sum(float(v.double().pow(2).sum()) for v in g.values())
            eta = frac * loss0 / g2
            for k in g:
                params[k].add_(g[k].to(params[k].dtype), alpha=-eta)
            out[name] = (loss0 - loss_fn()) / loss0
            for k in g:
                params[k].add_(g[k].to(params[k].dtype), alpha=eta)
    return out


# ---- multi-process GPU tests: the same workers run as gloo ranks sharing the one GPU of the test box (RCCL refuses two ranks on one
# device) and -- tests/test_multigpu_gpu.py, MSAM2_TEST_REAL_DEVICES=1 -- as RCCL ranks on one GPU each
def real_devices() -> bool:
    import os
    return os.environ.get("MSAM2_TEST_REAL_DEVICES") == "1"


def rank_device(rank: int) -> int:
    return rank if real_devices() else 0


def host_cpu_share() -> int:
    """CPUs this process may actually use: the cgroup quota when there is one (the GPU box gives 16 of its 256 hardware threads), else the
    affinity mask -- bench.host_cpu_share's rule.  torch sizes its pool by the host's hardware threads; on all 256 the oracle runs 3x
    slower than on 16 (tests/probes/cpu_threads_probe.py)."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n


def init_test_process_group(rank: int, world: int):
    """process group of a spawned test worker: backend nccl (= RCCL over xGMI) with one device per rank under MSAM2_TEST_REAL_DEVICES=1,
    otherwise gloo with every rank on device 0.  MASTER_ADDR / MASTER_PORT are in the environment."""
    import torch.distributed as dist
    torch.set_num_threads(max(1, host_cpu_share() // world))
    torch.cuda.set_device(rank_device(rank))
    if real_devices():
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank_device(rank)))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)


def all_gather_flat(flat: torch.Tensor, world: int):
    """all-gather of one flat tensor per rank for cross-rank comparisons (RCCL moves device tensors, gloo host tensors); returns CPU copies"""
    import torch.distributed as dist
    t = flat.cuda() if real_devices() else flat.cpu()
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [o.cpu() for o in out]
