"""GPU: bench.py's N > 1 code path (rank-local shards, barrier, max-over-ranks, one JSON line on rank 0), exercised
with 2 ranks over gloo that share GPU 0 (the box has one GPU; the driver runs the real 2/4/8-GPU RCCL launch)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_share_device():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--chunks", "4000", "--dist-backend", "gloo", "--share-device", "--no-cpu-baseline"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().split("\n") if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["unit"] == "Mbp/s"
    assert d["value"] > 0 and abs(d["value"] - 2 * 4000 * 2120 * 2 / (d["ms_per_step"] * 2 / 1e3) / 1e6) < 1e-3 * d["value"]
    assert "cpu_baseline" not in d and d["roofline"]["launches"] == 2
    assert d["control_plane"] == "gloo" and [r["rank"] for r in d["per_rank"]] == [0, 1]
    assert all(r["pci_bus_id"] and r["hip_device"] == 0 for r in d["per_rank"])


def test_gpus_flag_starts_its_own_ranks():
    """`python bench.py --gpus 2` started PLAINLY (no torchrun, no WORLD_SIZE): the script launches its own two ranks and the line says
    n_gpus = 2 with two per-rank rows (the reference's --cores N forks its own pool, nucleoatac/run_occ.py:101-102)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-device", "--chunks", "4000",
                          "--no-cpu-baseline", "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().split("\n") if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and [r["rank"] for r in d["per_rank"]] == [0, 1] and d["control_plane"] == "gloo"
    assert d["config"]["chunks_total"] == 8000 and d["config"]["chunks_this_rank"] == 4000 and d["value"] > 0


def test_two_ranks_run_the_host_to_host_pipeline_side_by_side():
    """the host-to-host pipeline (pinned slots, six contexts per rank) with more than one rank on a host -- every rank
    runs it on its own shard at the same time (`--h2h-ranks`) and the line carries a rate and the host placement (CPUs allowed, their
    NUMA nodes, the GPU's NUMA node) per rank."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-device", "--chunks", "4000",
                          "--no-cpu-baseline", "--steps", "2", "--warmup", "1", "--h2h-ranks", "--h2h-rank-chunks", "3000", "--h2h-sub", "1000"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.strip().split("\n") if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and len(d["per_rank"]) == 2
    for r in d["per_rank"]:
        assert r["host_to_host_mbp_s"] > 0 and r["host_to_host_pcie_gbs_down"] > 0
        pl = r["placement"]
        assert pl["cpus_allowed"] >= 1 and pl["cpu_list"] and isinstance(pl["cpu_numa_nodes"], list)
    assert "placement" in d


def test_single_rank_default_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--chunks", "3000", "--cpu-chunks", "32",
                          "--cpu-literal-chunks", "16", "--h2h-sub", "1000", "--cli-chunks", "400"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.strip().split("\n") if l.startswith("{")][-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["dtype"] == "f64" and d["vs_baseline"] is None and d["higher_is_better"] is True
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])
    r = d["roofline"]               # round 4: fp64 view first, the HBM view beside it, step traffic, live clocks
    assert r["bound"] == "fp64_valu" and r["unit"] == "TFLOP/s" and 0 < r["frac"] < 1
    assert r["hbm"]["unit"] == "GB/s" and r["hbm"]["peak"] == 8000.0 and 0 < r["hbm"]["frac"] < 1
    assert r["clock_ghz"] and 1.0 < r["clock_ghz"]["background"] < 3.0
    assert d["value_boundary"].startswith("hbm_resident") and d["value_host_to_host"] == d["host_to_host"]["host_to_host_mbp_s"]
    cal = d["cpu_baseline"]["reference_calibration"]
    assert cal is None or cal["reference_s_per_chunk"] > cal["port_literal_s_per_chunk"] > cal["port_optimised_s_per_chunk"] > 0
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1 and "workload" in d["config"]
    assert d["cpu_baseline"]["literal"]["chunks"] == 16 and d["cpu_baseline"]["optimised"]["chunks"] == 32
    h = d["host_to_host"]            # pipelined host -> host rate: pinned buffers, 3 contexts, sub-batches of 1000 chunks
    assert h["host_to_host_mbp_s"] > 0 and h["sub_batches"] == 3 and h["contexts"] == 6 and h["gb_down_per_step"] > 0
    assert "PipelinedExecutor" in h["executor"]                 # the product's executor, not a bench-only harness
    e = d["cli_end_to_end"]          # `nucleoatac occ` / `nuc` from input files to .bedgraph.gz + .tbi
    assert e["chunks"] == 400 and e["occ_mbp_s"] > 0 and e["nuc_mbp_s"] > 0 and e["nucleosome_calls"] > 0 and e["cores"] >= 1
    assert e["run_seconds"] > 0 and e["run_mbp_s"] > 0 and "HBM" in e["resident_occ_tracks"]
    assert e["real_inputs"]["occ_mbp_s"] > 0 and e["real_inputs"]["bam_gb"] > 0      # the same windows from a real .bam + .fa
    ts = d["roofline"]["traffic_source"]
    assert ts is None or set(("file", "collected_at_source_sha16", "current_source_sha16", "stale")) <= set(ts)


def test_cfg4_strong_scaling_two_ranks_share_device():
    """--workload cfg4 (BASELINE configs[3] shape, scaled down): the chunk list is sharded by balanced_ranges, every rank
    generates only its shard, sub-batches, one JSON line with the per-rank imbalance; the sharded total equals 1 rank's"""
    common = ["--workload", "cfg4", "--chunks", "1500", "--sub-chunks", "400", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29534", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--share-device"] + common
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d2 = json.loads([l for l in out.stdout.strip().split("\n") if l.startswith("{")][-1])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d1 = json.loads([l for l in out.stdout.strip().split("\n") if l.startswith("{")][-1])
    assert d2["scaling"] == "strong" and d1["scaling"] == "strong" and d2["n_gpus"] == 2
    c1, c2 = d1["config"], d2["config"]
    assert c1["bp_total"] == c2["bp_total"] == 1500 * 10120 and c1["fragments_total"] == c2["fragments_total"]
    assert c1["candidates_per_step"] == c2["candidates_per_step"]          # same chunks, same results, however they are sharded
    imb = c2["shard_imbalance"]
    assert len(imb["bp_per_rank"]) == 2 and sum(imb["bp_per_rank"]) == c2["bp_total"] and imb["bp_max_over_mean"] < 1.05
    assert c1["sub_batches_this_rank"] == 4 and c2["sub_batches_this_rank"] == 2


def test_rccl_failure_falls_back_to_gloo():
    """RCCL asked for (--dist-backend nccl; the default is gloo since round 4) with two ranks on ONE GPU: RCCL cannot form a
    communicator on a duplicate device and the ranks agree on that; the bench's collectives (barrier, max / sum of three scalars) are
    not data, so it goes on over gloo and still prints its line"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29536", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--chunks", "2000", "--share-device", "--no-cpu-baseline", "--dist-backend", "nccl"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "RCCL group not used" in out.stderr
    d = json.loads([l for l in out.stdout.strip().split("\n") if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["control_plane"] == "gloo"


def test_ranks_that_share_a_gpu_are_refused_without_the_switch():
    """one rank per GPU is checked before any timing: two ranks that land on the same PCI device without --share-device end with a
    message that lists (rank, local_rank, hip device, pci bus id, visible-device variables) of every rank"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29537", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--chunks", "1000", "--no-cpu-baseline"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NATAC_DEVICE="0")      # both ranks on GPU 0, without --share-device
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode != 0
    assert "ranks share a GPU" in out.stderr and "--share-device" in out.stderr


def test_eight_ranks_share_device_cfg4():
    """the N = 8 shape the driver launches for the scaling run, with all eight ranks on GPU 0: configs[3] geometry sharded by
    balanced_ranges into eight shards, control plane over gloo (RCCL cannot form a communicator on one device and the ranks
    agree on that collectively), one JSON line with the eight per-rank timing rows; the sharded totals equal one rank's"""
    common = ["--workload", "cfg4", "--chunks", "2400", "--sub-chunks", "200", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", "29538", os.path.join(ROOT, "bench.py"), "--gpus", "8", "--share-device"] + common
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.strip().split("\n") if l.startswith("{")]
    assert len(lines) == 1
    d8 = json.loads(lines[0])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d1 = json.loads([l for l in out.stdout.strip().split("\n") if l.startswith("{")][-1])
    assert d8["n_gpus"] == 8 and d8["scaling"] == "strong" and d8["control_plane"] == "gloo"
    c1, c8 = d1["config"], d8["config"]
    assert c1["bp_total"] == c8["bp_total"] == 2400 * 10120 and c1["fragments_total"] == c8["fragments_total"]
    assert c1["candidates_per_step"] == c8["candidates_per_step"]
    rows = d8["per_rank"]
    assert [r["rank"] for r in rows] == list(range(8)) and sum(r["bp"] for r in rows) == c8["bp_total"]
    assert all(r["ms_per_step"] > 0 and r["generate_s"] >= 0 for r in rows)
    assert c8["shard_imbalance"]["bp_max_over_mean"] < 1.05 and len(c8["shard_imbalance"]["bp_per_rank"]) == 8


def test_cfg5_independent_samples_with_cov_sweep():
    """--workload cfg5 (BASELINE configs[4]): one independent sample per rank + the multinomial_cov tolerance sweep, 2 ranks on GPU 0"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29539", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "cfg5", "--chunks", "1500", "--steps", "1",
           "--warmup", "1", "--share-device", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.strip().split("\n") if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["chunks_total"] == 3000
    s = d["multinomial_cov_sweep"]
    assert [r["rank"] for r in s["samples"]] == [0, 1] and all(r["candidates"] > 2000 for r in s["samples"])
    assert s["worst_closed_fp64"] < 1e-9 and 1e-8 < s["worst_fp32"] < 1e-2
    assert s["samples"][0]["closed_fp64"] != s["samples"][1]["closed_fp64"]        # two different samples
