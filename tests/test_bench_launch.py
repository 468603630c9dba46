"""bench.py's own launcher for --gpus N > 1 (no torchrun): N children with the launcher
environment, rank 0's line relayed, worst return code reported, a dead rank ends the others."""
import json
import os
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _script(tmp_path, body):
    path = tmp_path / "child.py"
    path.write_text(textwrap.dedent(body))
    return str(path)


def test_self_launch_sets_rank_environment_and_relays_rank0(tmp_path):
    import bench
    script = _script(tmp_path, """
        import json, os, sys
        keys = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")
        print(json.dumps({k: os.environ[k] for k in keys} | {"argv": sys.argv[1:]}))
    """)
    rc, out = bench.self_launch(3, ["--gpus", "3", "--toy"], script=script, timeout=60)
    assert rc == 0
    line = json.loads(out.strip())
    assert line["RANK"] == "0" and line["LOCAL_RANK"] == "0" and line["WORLD_SIZE"] == "3"
    assert line["MASTER_ADDR"] == "127.0.0.1" and int(line["MASTER_PORT"]) > 0
    assert line["argv"] == ["--gpus", "3", "--toy"]


def test_self_launch_two_ranks_rendezvous_over_gloo(tmp_path):
    import bench
    script = _script(tmp_path, """
        import os, torch, torch.distributed as dist
        dist.init_process_group("gloo")
        t = torch.tensor([float(dist.get_rank() + 1)])
        dist.all_reduce(t)
        if dist.get_rank() == 0:
            print(int(t.item()), dist.get_world_size())
        dist.destroy_process_group()
    """)
    rc, out = bench.self_launch(2, [], script=script, timeout=120)
    assert rc == 0 and out.strip().splitlines()[-1].split() == ["3", "2"]   # (gloo logs to stdout)


def test_self_launch_reports_worst_code_and_ends_survivors(tmp_path):
    import bench
    script = _script(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(600)          # a rank stuck in a collective
    """)
    rc, _ = bench.self_launch(2, [], script=script, timeout=120)
    assert rc == 7           # the failing rank's own code, not the 137 of the survivor it killed


def test_bench_refuses_mismatched_world_size():
    import subprocess
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--toy"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE=4" in r.stderr
