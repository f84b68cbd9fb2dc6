"""bench.py --gpus N without a launcher must spawn N ranks itself (one process per GPU; here: a stand-in script on
CPU with the gloo backend) and relay rank 0's single JSON line."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_self_launch_spawns_n_ranks(tmp_path):
    worker = tmp_path / "worker.py"
    worker.write_text(textwrap.dedent("""
        import json, os, sys
        import torch, torch.distributed as dist
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        assert os.environ["MASTER_ADDR"] == "127.0.0.1"
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        if rank == 0:
            print(json.dumps({"n_gpus": world, "max": t.item(), "argv": sys.argv[1:]}), flush=True)
        dist.destroy_process_group()
    """))
    driver = tmp_path / "driver.py"
    driver.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        import bench
        sys.exit(bench._self_launch(2, script={str(worker)!r}, argv=["--gpus", "2", "--steps", "1"]))
    """))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(driver)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["max"] == 2.0 and out["argv"] == ["--gpus", "2", "--steps", "1"]


def test_gpus_flag_without_world_size_takes_the_launcher_path(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    called = {}
    monkeypatch.setattr(bench, "_self_launch", lambda n, **k: called.setdefault("n", n) and 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    try:
        bench.main()
    except SystemExit:
        pass
    assert called.get("n") == 4
