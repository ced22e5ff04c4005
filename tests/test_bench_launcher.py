"""bench.py's own launcher: `--gpus N` outside a torchrun environment must start N ranks (VERDICT r1:
it used to run one rank silently), and a mismatch between --gpus and WORLD_SIZE must fail loudly.
Runs without a GPU (--dry-run stops before any device call)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT,
                          capture_output=True, text=True, timeout=timeout)


def test_gpus_2_starts_two_ranks():
    p = _run(["--gpus", "2", "--dry-run"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    res = json.loads(lines[0])
    assert res == {"dry_run": True, "n_gpus": 2, "ranks_joined": 2}


def test_gpus_world_size_mismatch_is_an_error():
    p = _run(["--gpus", "4", "--dry-run"], {"WORLD_SIZE": "1", "RANK": "0"})
    assert p.returncode == 2
    assert "WORLD_SIZE" in p.stderr


def test_single_rank_dry_run():
    p = _run(["--dry-run"])
    assert p.returncode == 0 and json.loads(p.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_failing_rank_reaches_the_caller_as_a_nonzero_exit():
    """a rank that leaves the way a failed / stuck scatter leg does (exit code 3 after the line is flushed)
    while its peer waits in a collective must end the whole launch with a non-zero code -- not hang, not report 0."""
    for failing in (0, 1):
        p = _run(["--gpus", "2", "--dry-run", "--dry-run-fail-rank", str(failing)], timeout=240)
        assert p.returncode != 0, (failing, p.stdout[-500:], p.stderr[-500:])
        if failing == 0:
            assert '"scatter": {"error"' in p.stdout          # the headline line still came out
