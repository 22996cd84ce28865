"""§8e multi-GPU path: replicas + sharding, exercised with world_size 2 over gloo on CPU (kernels run under the emulator)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cosyvoice_amd.replica import run_sharded, shard_requests, timed_region


def test_shard_requests_is_balanced_and_deterministic():
    costs = [30, 5, 12, 12, 40, 7, 1, 22]
    s = shard_requests(costs, 3)
    assert sorted(i for sh in s for i in sh) == list(range(8))
    loads = [sum(costs[i] for i in sh) for sh in s]
    assert max(loads) - min(loads) <= max(costs) and s == shard_requests(costs, 3)
    assert shard_requests([], 2) == [[], []]
    assert shard_requests([3], 4) == [[0], [], [], []]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here); sys.path.insert(0, os.path.dirname(here))
    from emu.build_emu import build_emu
    from cosyvoice_amd._lib import Lib
    from cosyvoice_amd.llm import Qwen2LM
    from cosyvoice_amd import synthetic as W
    lib = Lib(build_emu(), allow_emulated=True)
    cfg = W.tiny()[0]
    lm = Qwen2LM(W.make_llm(cfg), cfg, lib=lib, max_len=96, sampling="greedy")
    reqs = [W.synthetic_utterance(cfg, W.tiny()[1], n_prompt_tok=6, n_prompt_text=3, n_text=n, seed=100 + n) for n in (3, 5, 2, 4)]
    t = lambda n: torch.tensor([n], dtype=torch.int32)

    def run(u):
        return list(lm.inference(text=u["text"], text_len=t(u["text"].shape[1]), prompt_text=u["prompt_text"], prompt_text_len=t(3),
                                 prompt_speech_token=u["llm_prompt_speech_token"], prompt_speech_token_len=t(6), embedding=None,
                                 max_token_text_ratio=3, min_token_text_ratio=3))
    res = run_sharded(reqs, [u["text"].shape[1] for u in reqs], run, rank, world, dist)
    el = timed_region(lambda: None, 2, dist)
    if rank == 0:
        torch.save({"res": res, "elapsed": el}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_replicas_match_single_process(tmp_path, emu_lib):
    from cosyvoice_amd.llm import Qwen2LM
    from cosyvoice_amd import synthetic as W
    out = str(tmp_path / "r.pt")
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    cfg = W.tiny()[0]
    lm = Qwen2LM(W.make_llm(cfg), cfg, lib=emu_lib, max_len=96, sampling="greedy")
    t = lambda n: torch.tensor([n], dtype=torch.int32)
    for n, toks in zip((3, 5, 2, 4), got["res"]):
        u = W.synthetic_utterance(cfg, W.tiny()[1], n_prompt_tok=6, n_prompt_text=3, n_text=n, seed=100 + n)
        want = list(lm.inference(text=u["text"], text_len=t(n), prompt_text=u["prompt_text"], prompt_text_len=t(3), prompt_speech_token=u["llm_prompt_speech_token"],
                                 prompt_speech_token_len=t(6), embedding=None, max_token_text_ratio=3, min_token_text_ratio=3))
        assert toks == want and len(toks) >= 1              # same tokens whichever replica ran the utterance
    assert got["elapsed"] >= 0


def test_bench_refuses_to_mislabel_gpu_count():
    """`python bench.py --gpus N` outside a launcher spawns N ranks itself and fails loudly when the node has fewer GPUs (this container has
    none); under a launcher a WORLD_SIZE that disagrees with --gpus is an error - never an N=1 run reported under an N-GPU label."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0 and "--gpus 2 requested but only" in r.stderr and not r.stdout.strip()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr and not r.stdout.strip()


def test_mixed64_workload_is_seeded_and_fully_assigned():
    """BASELINE.json configs[3] workload of bench.py: 64 seeded utterances, equal mix of 125/250/375/500 tokens; every rank count deals each
    utterance to exactly one rank with balanced audio."""
    import importlib.util, os
    import torch
    from cosyvoice_amd.configs import tiny
    from cosyvoice_amd.replica import shard_requests
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    cfgs = tiny()
    a, costs = bench.mixed_requests(cfgs, torch.device("cpu"))
    b, _ = bench.mixed_requests(cfgs, torch.device("cpu"))
    assert len(a) == 64 and sorted(set(costs)) == [125, 250, 375, 500] and all(costs.count(c) == 16 for c in set(costs))
    assert all(torch.equal(x["text"], y["text"]) and torch.equal(x["prompt_speech_feat"], y["prompt_speech_feat"]) for x, y in zip(a, b))
    assert not torch.equal(a[0]["text"], a[4]["text"])
    for world in (1, 2, 4, 8):
        shards = shard_requests(costs, world)
        assert sorted(i for s in shards for i in s) == list(range(64))
        loads = [sum(costs[i] for i in s) for s in shards]
        assert max(loads) - min(loads) <= 125


def test_flow_groups_bucket_by_length():
    """CosyVoice2Model._flow_groups (host logic, no device): finished sequences are cut into flow passes longest-first, a pass holds at most
    `flow_batch` sequences whose shortest has at least 1 / flow_pad of the longest's tokens (prompt + generated); equal lengths always fit."""
    import types
    import torch
    from cosyvoice_amd.model import CosyVoice2Model
    me = types.SimpleNamespace(flow_batch=4, flow_pad=1.25)
    req = lambda p: {"flow_prompt_speech_token": torch.zeros(1, p, dtype=torch.int32)}
    job = lambda i, n, p=87: (i, req(p), [0] * n)
    groups = lambda jobs: [[j[0] for j in g] for g in CosyVoice2Model._flow_groups(me, jobs)]
    # the mixed64 lengths: 500 / 375 are more than 25 % apart (587 vs 462 tokens with the prompt), so are 250 / 125
    jobs = [job(i, n) for i, n in enumerate([125, 500, 250, 375] * 3)]
    assert groups(jobs) == [[1, 5, 9], [3, 7, 11], [2, 6, 10], [0, 4, 8]]
    # similar lengths share a pass, at most flow_batch of them, longest first, ties by submission order
    jobs = [job(0, 250), job(1, 230), job(2, 215), job(3, 205), job(4, 200), job(5, 120)]
    assert groups(jobs) == [[0, 1, 2, 3], [4], [5]]
    me.flow_batch = 8
    assert groups(jobs) == [[0, 1, 2, 3, 4], [5]]                    # (120 + 87) * 1.25 < 250 + 87
    me.flow_pad = 1.0
    assert groups([job(0, 100), job(1, 100, 80), job(2, 93), job(3, 100)]) == [[0, 3], [1, 2]]     # equal token totals only (prompt included)
    me.flow_batch = 1
    assert groups([job(0, 10), job(1, 10)]) == [[0], [1]]


def _bench_line(gpus, workload, extra_env=None):
    """`python bench.py --gpus N` with CV_BENCH_DRYRUN=1: bench.py itself spawns the N ranks through torch.distributed.run (rendezvous on 127.0.0.1) and every rank
    runs the REAL rank body at emulator size - no copy of that code lives in the tests."""
    import json, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")}
    env.update(CV_BENCH_DRYRUN="1", OMP_NUM_THREADS="2", **(extra_env or {}))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(gpus), "--steps", "1", "--warmup", "0", "--workload", workload],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                               # ONE JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.timeout(1800)
def test_bench_rank_body_runs_at_world_size_2(emu_lib):
    """bench.py's N > 1 branch had never executed anywhere.  Here the real rank body runs with world size 2 (gloo, CPU, the emulator build of the
    kernels, emulator-size model) for BOTH workloads: one JSON line, n_gpus == 2, every rank pinned to its own device (HIP_VISIBLE_DEVICES = its local rank - or its
    entry of a list the launcher already restricts), every mixed utterance assigned to exactly one rank, and the hash over the per-utterance waveform hashes equal
    to the world-size-1 run's - the determinism contract of SURVEY.md section 8e.  (emu_lib: the emulator library is built before the ranks race for it.)"""
    one = _bench_line(1, "mixed64")
    two = _bench_line(2, "mixed64")
    assert one["dry_run"] and two["dry_run"] and one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["scaling"] == "strong" and two["utterance_hashes_sha1"] == one["utterance_hashes_sha1"]
    assign = two["config"]["assignment"]
    n_utt = sum(len(v) for v in one["config"]["assignment"].values())
    assert sorted(i for sh in assign.values() for i in sh) == list(range(n_utt)) and all(len(sh) > 0 for sh in assign.values())
    assert [d["hip_visible_devices"] for d in two["rank_devices"]] == ["0", "1"] and [d["rank"] for d in two["rank_devices"]] == [0, 1]
    assert one["rank_devices"][0]["hip_visible_devices"] is None        # world size 1: nothing is pinned, the process keeps the launcher's view
    assert "min(48, shard size)" in two["config"]["workload"]
    u10 = _bench_line(2, "u10", {"HIP_VISIBLE_DEVICES": "5,3"})           # a launcher that already restricts the node: rank i takes the i-th entry
    assert u10["n_gpus"] == 2 and u10["scaling"] == "weak" and [d["hip_visible_devices"] for d in u10["rank_devices"]] == ["5", "3"]
    assert u10["value"] > 0 and u10["self_check"]["n_tokens"] >= 1 and u10["config"]["parallelism"] == "replicas x2, no collective"
