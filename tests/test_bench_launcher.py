"""bench.py --gpus N without a launcher starts N rank processes by itself.
Rehearsed here without a GPU: `--rehearsal` runs everything but the GPU work (compile, witness
cache, gloo rendezvous, barriers, the per-step all_gather of proof records) and prints value null."""
import json
import os
import subprocess
import sys

from gnark_crypto_primitives_amd import backend

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--rehearsal", "--levels", "8", "--populated", "3", "--batch", "48", "--steps", "2"]


def _run(args, tmp_path, env=None):
    e = dict(os.environ, ZKMI_CACHE_DIR=str(tmp_path / "cache"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e,
                          capture_output=True, text=True, timeout=600)


def test_gpus_2_self_launch_world_2_one_json_line(tmp_path):
    r = _run(["--gpus", "2"] + SMALL, tmp_path)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rehearsal"] is True and out["value"] is None
    assert out["config"]["global_batch"] == 96 and out["gathered_on_every_rank"] is True
    assert "2 rank(s) as torch.distributed reports it" in out["config"]["parallelism"]
    # the launcher warmed the cache: both ranks loaded the compiled circuit and their witnesses
    assert out["startup_s"]["compile_cache_hit"] is True
    names = os.listdir(tmp_path / "cache")
    assert sum(n.startswith("wit-") and n.endswith(".npy") for n in names) == 4   # main + worst-case set of ranks 0, 1
    assert sum(n.startswith("cc-") and n.endswith(".pkl") for n in names) == 1
    # strong scaling through the same launcher: 3 ranks, shards 17 / 17 / 16 of one batch of 50
    r = _run(["--gpus", "3", "--scaling", "strong"] + SMALL[:5] + ["--batch", "50", "--steps", "2"], tmp_path)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip())
    assert out["n_gpus"] == 3 and out["config"]["global_batch"] == 50
    assert out["gathered_on_every_rank"] is True


def test_gpus_disagreeing_with_world_size_is_an_error(tmp_path):
    r = _run(["--gpus", "4"] + SMALL, tmp_path, {"WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0 and "disagrees" in r.stderr and not r.stdout.strip()


def test_launcher_reports_a_failing_rank(tmp_path):
    prog = ("import os, sys, time\n"
            "r = int(os.environ['RANK'])\n"
            "assert os.environ['WORLD_SIZE'] == '3' and os.environ['LOCAL_RANK'] == str(r)\n"
            "assert os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
            "print('rank', r, flush=True)\n"
            "if r == 1: sys.exit(7)\n"
            "time.sleep(60)\n")
    script = tmp_path / "child.py"
    script.write_text("import sys\nsys.path.insert(0, %r)\n"
                      "from gnark_crypto_primitives_amd import backend\n"
                      "sys.exit(backend.launch_local_ranks([sys.executable, '-c', %r], 3, 29999))\n"
                      % (ROOT, prog))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 7
    assert r.stdout.strip() == "rank 0"             # only rank 0's stdout is forwarded
    assert "rank 1 exited with 7" in r.stderr


def test_under_torch_distributed_run_as_the_driver_launches_it(tmp_path):
    """The driver's N > 1 command line: python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...  Each process is then one rank
    (RANK / WORLD_SIZE inherited); rank 0 prints the one JSON line."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    e = dict(os.environ, ZKMI_CACHE_DIR=str(tmp_path / "cache"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                        "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
                        str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL,
                       env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["gathered_on_every_rank"] is True
