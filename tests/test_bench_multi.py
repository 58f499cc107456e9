"""bench.py's N>1 path.  CPU: asking for more GPUs than the box has must fail loudly (non-zero, error JSON), and the
self-launch must not need a launcher.  GPU: two ranks of the PRODUCT path (HIP decode on cuda:0, gloo collectives through
the host) started by bench.py itself, exactly as the driver invokes it (`python bench.py --gpus N ...`)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=600):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout,
                       env=env, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_more_gpus_than_present_fails_loudly():
    r, j = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert j is not None and "error" in j and j["n_gpus"] == 2


@pytest.mark.gpu
def test_two_ranks_started_by_bench_itself_product_path():
    r, j = _run(["--gpus", "2", "--backend", "gloo", "--same-device", "--packets", "256", "--extra-packets", "64", "--steps", "2",
                 "--warmup", "1"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["status_ok"] is True and j["allgather_ok"] is True
    assert j["config"]["packets_per_gpu"] == 256 and j["value"] > 0
    assert j["allgather_ms"] > 0 and j["decode_allgather_overlapped_ms"] > 0
    for c in ("cfg4", "cfg5"):
        assert j["extra_configs"][c]["status_ok"] is True and j["extra_configs"][c]["value"] > 0
        assert j["extra_configs"][c]["allgather_ms"] > 0


@pytest.mark.gpu
def test_single_gpu_line_has_the_contract_keys_and_parity():
    r, j = _run(["--packets", "512", "--steps", "3", "--warmup", "1", "--cpu-seconds", "0.5", "--no-big-batch"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j
    assert j["parity_vs_oracle"] is True and j["status_ok"] is True
    assert j["roofline"]["bound"] == "hbm" and 0 < j["roofline"]["frac"] < 1
    assert j["cpu_baseline"]["cores"] == 1 and j["cpu_baseline"]["kind"] == "port"
    assert j["host_path"]["int32_pinned_ms"] > 0 and j["host_path"]["decode_frame_ms"] > 0
    # two batches in flight on two streams: the same outputs, reported beside `value`, never as it
    assert j["two_in_flight"]["same_output_as_value_run"] is True and j["two_in_flight"]["value"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [3, 4, 5])
def test_full_size_configs_equal_the_oracle(cfg):
    # BASELINE configs 3..5 at their full per-GPU size, every sample compared with the oracle inside bench.py
    r, j = _run(["--config", str(cfg), "--steps", "2", "--warmup", "1", "--cpu-seconds", "0.2", "--no-host-path"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert j["parity_vs_oracle"] is True and j["status_ok"] is True


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,per_gpu", [(4, 8192), (5, 4096)])
def test_every_rank_shard_of_the_eight_gpu_configs_equals_the_oracle(cfg, per_gpu, oracle, synth):
    """BASELINE configs 4 and 5 are whole-job batches of 65 536 / 32 768 packets over 8 GPUs.  bench.py gives rank r the
    packets [r * per_gpu, (r + 1) * per_gpu) of the job (make_config_batch(first_index=...)); rank 0's shard is compared with
    the oracle by test_full_size_configs_equal_the_oracle.  Here the shards of ranks 1..7 go through the same C-ABI entry, one
    after the other on the one GPU there is, every sample against the oracle: the whole job has then been decoded bit-exactly at
    its full size, rank by rank."""
    import numpy as np
    import alac.net_amd as pkg

    for rank in range(1, 8):
        b = synth.make_config_batch(cfg, n_packets=per_gpu, first_index=rank * per_gpu)
        with pkg.AlacGpuContext(b["stream_cfgs"], device=0) as ctx:
            gp, gob, gos, gst = ctx.decode_batch(b["blob"], b["offsets"], b["sizes"], b["cfg_idx"], b["slot_ints"])
        op, oob, oos, ost = oracle.decode_batch(oracle.make_cfgs(b["stream_cfgs"]), b["blob"], b["offsets"], b["sizes"],
                                                 b["cfg_idx"], b["slot_ints"], n_threads=16)
        assert np.array_equal(gst, ost) and np.array_equal(gob, oob) and np.array_equal(gos, oos), rank
        ci = np.zeros(per_gpu, dtype=int) if b["cfg_idx"] is None else b["cfg_idx"].astype(int)
        nc = np.array([b["stream_cfgs"][int(i)][5] for i in ci], dtype=np.int64)
        cnt = np.where(ost == 0, oos.astype(np.int64) * nc, 0)
        mask = np.arange(gp.shape[1])[None, :] < cnt[:, None]
        assert np.array_equal(gp[mask], op[mask]), rank
        assert (ost == 0).mean() > 0.9
