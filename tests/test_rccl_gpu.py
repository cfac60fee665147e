"""The multi-GPU paths of bench.py over RCCL on the hardware that is there: one MI355X.

`pytest -m gpu` runs on a 1-GPU box, and RCCL refuses two ranks on one device, so the world is ONE rank — but it is the real thing
that runs: `torch.distributed` "nccl" (= RCCL) process group with `device_id`, the count all-reduce and the asynchronous record
all-gather next to the library's private stream (weak mode), the packed fragment set through the ring code (strong mode / c4),
the timing all-reduce and the per-rank all-gather, launched exactly as the driver launches N ranks (`python -m
torch.distributed.run --nproc-per-node 1 ... bench.py --gpus 1` with ANI_BENCH_FORCE_DIST=1).  The rows must equal the
single-process rows of the same genomes.  World sizes 2 and 4 of the same code run on CPU ranks over gloo
(tests/test_distributed_gloo.py); --simulate-world runs every rank's share of a 4-rank job on this GPU, one after the other."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, L, SEED = 40, 300000, 20260925


def _single_rows(gpu_engine):
    import torch
    from fastani_amd.api import DeviceGenomes, Sketch
    words = (L + 15) // 16
    buf = torch.zeros(N * words + 64, dtype=torch.int32, device="cuda:0")
    gpu_engine.synth_packed(SEED, 0, N, L, buf.data_ptr())
    dg = DeviceGenomes(buf.data_ptr(), N, L)
    sk = Sketch(gpu_engine, gpu_engine.params(), dg)
    rows = sk.map_cgi_batch(dg, 0)
    sk.close()
    return rows


def _bench(args, env=None, launcher=True, port=29650):
    cmd = ([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port)] if launcher else [sys.executable])
    cmd += [os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--genomes", str(N), "--genome-len", str(L), "--no-cpu-baseline", "--no-e2e"] + args
    r = subprocess.run(cmd, capture_output=True, env=dict(os.environ, **(env or {})), timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    return json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][-1])


def _sorted(rows):
    return rows[np.lexsort((rows["refGenomeId"], rows["qryGenomeId"]))]


@pytest.mark.gpu
@pytest.mark.parametrize("config,scaling", [("many-to-many", "auto"), ("many-to-many", "strong"), ("c4", "auto")])
def test_bench_over_rccl_one_rank(gpu_engine, tmp_path, config, scaling):
    single = _single_rows(gpu_engine)
    assert len(single) >= 2 * N
    dump = os.path.join(str(tmp_path), "rows")
    out = _bench(["--config", config, "--scaling", scaling, "--dump-rows", dump], env={"ANI_BENCH_FORCE_DIST": "1"}, port=29650 + (os.getpid() % 200))
    ranks = out["ranks"]
    assert ranks["ranks_seen_by_rccl"] == 1 == int(os.environ.get("WORLD_SIZE", "1")) and out["n_gpus"] == 1
    assert out["rows_identical_across_steps"] and out["data"] == "synthetic"
    strong = config == "c4" or scaling == "strong"
    assert out["scaling"] == ("strong" if strong else "weak") and out["config"]["mode"] == ("ring" if strong else "gather")
    assert ("reference-sharded" in ranks["mode"]) == strong
    got = _sorted(np.load(dump + ".rank0.npy"))
    assert np.array_equal(got, single)
    assert out["parity_timed_rows"]["ok"] and out["parity_timed_rows"]["vs_oracle"]["pairs_with_rows"] > 0        # and the oracle agrees, bit for bit
    if config == "many-to-many" and strong:
        wl = out["weak_scaling_leg"]
        assert wl["ranks"]["ranks_seen_by_rccl"] == 1 and wl["ranks"]["rows"] == [len(single)]
    elif config == "many-to-many":
        # the default at N > 1 is the weak job; the strong job is measured beside it and its rows over all ranks are the single-GPU rows
        sys.path.insert(0, ROOT)
        import bench
        sl = out["strong_scaling_leg"]
        assert "error" not in sl, sl
        assert sl["ranks"]["ranks_seen_by_rccl"] == 1 and sl["ranks"]["rows"] == [len(single)] and sl["rows_identical_across_steps"]
        assert sl["rows_multiset"]["hash"] == "%016x" % bench.rows_multiset_hash(single) == out["rows_multiset"]["hash"]


@pytest.mark.gpu
def test_simulated_ranks_add_up(gpu_engine, tmp_path):
    """--simulate-world 4: rank r's compute of the 4-GPU strong-scaling job alone on this GPU (its shard indexed, four fragment sets
    mapped); the four ranks' rows together are the single-process rows"""
    single = _single_rows(gpu_engine)
    parts = []
    for r in range(4):
        dump = os.path.join(str(tmp_path), "sim%d" % r)
        out = _bench(["--simulate-world", "4", "--simulate-rank", str(r), "--dump-rows", dump, "--no-verify"], launcher=False)
        assert out["config"]["mode"] == "simulate" and out["simulated"]["world"] == 4 and out["rows_identical_across_steps"]
        parts.append(np.load(dump + ".npy"))
    assert np.array_equal(_sorted(np.concatenate(parts)), single)


@pytest.mark.gpu
def test_dry_collectives_over_rccl_one_rank():
    """bench.py --dry-collectives over RCCL (one rank is what one GPU allows): the all-reduce, the in-place all-gather at the real slot
    size of the 1000-genome workload's shard, the barrier — the 30-second check an 8-GPU run starts with"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(29350 + (os.getpid() % 200)),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dry-collectives"]
    r = subprocess.run(cmd, capture_output=True, env=dict(os.environ, ANI_BENCH_FORCE_DIST="1"), timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    out = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][-1])
    assert out["dry_collectives"] and out["n_gpus"] == 1 and out["slots_arrived_intact"] == [True]
    assert out["slot_bytes"] > 10 ** 9                      # 1000 genomes x ~1.6 MB of packed fragment sketches
