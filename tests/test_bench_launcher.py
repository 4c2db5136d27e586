"""bench.py --gpus N without a launcher: the ranks are started as a CHILD process under torch.distributed.run (never an
exec), rank 0's JSON line is relayed and the child's exit code returned (missing #3)."""
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_launcher_command_and_relay(monkeypatch, tmp_path):
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    class R:
        returncode = 7
        stdout = b'{"metric": "x"}\n'

    def fake_run(cmd, stdout=None, env=None):
        seen["cmd"], seen["env"], seen["stdout"] = cmd, env, stdout
        return R()
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    out = tmp_path / "line.json"
    fd = os.open(out, os.O_WRONLY | os.O_CREAT)
    try:
        rc = bench._launch_ranks(4, fd)
    finally:
        os.close(fd)
    cmd = seen["cmd"]
    assert rc == 7                                              # the child's exit code is the launcher's
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]      # the ranks see the same arguments
    assert seen["stdout"] == subprocess.PIPE
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert out.read_bytes() == R.stdout                          # exactly the ranks' stdout: one JSON line


def test_mismatched_launcher_world_size_is_refused(monkeypatch):
    """Under a launcher (WORLD_SIZE set) a rank count that does not match --gpus is an error, not a nested launch."""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True,
                       text=True)
    assert r.returncode != 0 and "rank count must match" in r.stderr
