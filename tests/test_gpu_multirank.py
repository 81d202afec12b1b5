"""The N > 1 bench path end to end on ONE GPU: two ranks (gloo, both on cuda:0) run the real kernels on their
shards, exchange through torch.distributed exactly as the RCCL run does, and rank 0 checks the result against an
unsharded engine bit for bit (bench.py --verify)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("nproc,batch", [(2, 8), (3, 8), (2, 0)])
def test_bench_two_ranks_on_one_gpu(nproc, batch):
    torch = pytest.importorskip("torch")
    assert torch.cuda.is_available()
    port = 29600 + os.getpid() % 300 + nproc
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", str(nproc), "--backend", "gloo", "--same-device", "--verify", "--docs", "60000", "--chunks", "250000",
           "--terms", "50000", "--queries-per-step", str(batch), "--steps", "2", "--warmup", "1", "--latency-queries", "2",
           "--no-cpu-baseline"]
    torch.cuda.empty_cache()                                     # hand cached blocks of earlier tests back before the ranks start
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, cwd=ROOT)
    except subprocess.TimeoutExpired as ex:                      # show what the ranks were doing instead of hanging the suite
        tail = lambda b: (b.decode("utf-8", "replace") if isinstance(b, bytes) else (b or ""))[-3000:]
        pytest.fail(f"bench.py with {nproc} ranks did not finish in 240 s\nstdout: {tail(ex.stdout)}\nstderr: {tail(ex.stderr)}")
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == nproc and line["outputs_sane"] and line["sharded_equals_unsharded"] is True
    # an explicit batch is strong scaling; the default (0) is 256 queries per GPU: weak scaling
    assert line["scaling"] == ("strong" if batch else "weak") and line["value"] > 0
    assert line["config"]["queries_per_step"] == (batch or 256 * nproc)
