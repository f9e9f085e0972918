"""bench.py started the way a driver starts it: `python bench.py --gpus 2` with no launcher around it must start its own
two ranks (a child torch.distributed.run, before the parent touches the GPU), let each rank ingest the attestations of
its own attesters, and print ONE JSON line.  gloo stands in for RCCL so that both ranks can share the one GPU of the test
box; without ET_BENCH_BACKEND the same command takes the nccl branch."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def run_bench(*flags):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["ET_BENCH_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(flags), env=env, cwd=ROOT, timeout=600,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("chunks", [1, 2])
def test_two_ranks_started_by_bench_itself(chunks):
    line = run_bench("--gpus", "2", "--workload", "c3", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--chunks", str(chunks))
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["metric"] == "edge-iterations/sec"
    assert line["config"]["parallelism"] == "rows sharded x2" and line["config"]["exchange_pieces"] == chunks
    assert line["config"]["edges"] == 1 << 24
    assert line["checks"]["fr_sum_conserved"] is True                 # native.rs:331-334 over both shares
    assert line["checks"]["f64_sum_rel_err"] < 1e-12
    assert line["value"] > 0 and line["roofline"]["kernel"].startswith("k_step_f64_blocked")
    assert line["cpu_baseline"] is None and line["vs_baseline"] is None
    it = line["iteration_us"]                                         # where an iteration's time goes: launches / the rest
    assert it["total"] > 0 and abs(it["launches"] + it["exchange_and_waits"] - it["total"]) < 1e-6 * it["total"] + 1e-9
