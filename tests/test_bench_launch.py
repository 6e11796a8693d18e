"""CPU tests of bench.py's multi-GPU entry: `python bench.py --gpus N` (N > 1) with no launcher around it must start its
own N rank processes BEFORE anything in the parent touches HIP, relay rank 0's JSON line and the children's exit code;
and the timed-run helper of the strong-scaling bench must execute the same collectives on every rank whether or not a
rank's local loop failed (a failing rank used to leave the others in a barrier)."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import pytest

from conftest import PKG_NAME, ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


PARENT_PROBE = textwrap.dedent("""
    import json, os, subprocess, sys
    sys.argv = ["bench.py", "--gpus", "{gpus}", "--steps", "7", "--warmup", "2"]
    os.environ.pop("WORLD_SIZE", None)
    seen = {{}}
    class FakePopen:
        def __init__(self, cmd, env=None, stdout=None, stderr=None, text=None):
            seen["cmd"], seen["env"] = cmd, env
            self.stdout = iter(["some rank chatter\\n", json.dumps({{"metric": "m", "n_gpus": {gpus}, "value": 1.0}}) + "\\n"])
        def wait(self):
            return {rc}
    subprocess.Popen = FakePopen
    sys.path.insert(0, {root!r})
    import bench
    try:
        bench.main()
        code = 0
    except SystemExit as e:
        code = e.code
    maps = open("/proc/self/maps").read()
    print("PROBE " + json.dumps({{"code": code, "cmd": seen.get("cmd"), "ipc": seen.get("env", {{}}).get("HSA_ENABLE_IPC_MODE_LEGACY"),
                                 "torch_imported": "torch" in sys.modules,
                                 "hip_mapped": ("libamdhip64" in maps) or ("libhsa-runtime64" in maps)}}))
""")


@pytest.mark.parametrize("gpus,rc", [(2, 0), (8, 0), (4, 3)])
def test_parent_spawns_ranks_without_touching_the_gpu(gpus, rc):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "CG_FORCE_DIST")}
    out = subprocess.run([sys.executable, "-c", PARENT_PROBE.format(gpus=gpus, rc=rc, root=ROOT)], env=env,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    probe = json.loads([ln for ln in lines if ln.startswith("PROBE ")][0][6:])
    # the parent relays exactly one JSON line (rank 0's) and the children's exit code
    relayed = [ln for ln in lines if ln.startswith("{")]
    assert len(relayed) == 1 and json.loads(relayed[0])["n_gpus"] == gpus
    assert probe["code"] == rc
    assert "some rank chatter" in out.stderr
    # ... from fresh `torch.distributed.run` children on 127.0.0.1, one per GPU, same bench arguments
    cmd = probe["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and f"--nproc-per-node={gpus}" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", str(gpus), "--steps", "7", "--warmup", "2"] and cmd[-7].endswith("bench.py")
    assert probe["ipc"] == "0"
    # and never initialised HIP itself: no torch import, no HIP / HSA runtime mapped into the parent
    assert probe["torch_imported"] is False and probe["hip_mapped"] is False


def test_rank_process_does_not_respawn():
    """a rank started by torch.distributed.run (WORLD_SIZE set) must take the rank path, not spawn again"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"WORLD_SIZE" not in os.environ' in src
    # main() decides on spawning before the first torch import of the function body
    body = src[src.index("def main():"):]
    assert body.index("spawn_ranks(args") < body.index("import torch")


RUN_DIST_WORKER = textwrap.dedent("""
    import importlib, os, sys, time
    rank, world, port, fail_rank, fail_at = (int(v) for v in sys.argv[1:6])
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, {root!r})
    import torch
    import torch.distributed as dist
    torch.cuda.synchronize = lambda *a, **k: None          # CPU rehearsal: there is no device to wait for
    dmod = importlib.import_module({pkg!r} + ".dist")
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class FakeSolver:
        calls = 0
        def set_rhs(self, b, x0): pass
        def iterate(self, n):
            FakeSolver.calls += 1
            if rank == fail_rank and FakeSolver.calls == fail_at:
                raise RuntimeError("injected failure")
        def synchronize(self): pass

    t, ok, err = dmod._run_dist(FakeSolver(), None, 2, 5, dist, torch, "cpu")
    agreed = dmod._all_ok(ok, dist, torch, "cpu")
    print(f"RESULT {{rank}} {{int(ok)}} {{int(agreed)}} {{'-' if err is None else 'err'}}", flush=True)
    dist.barrier()
    dist.destroy_process_group()
""")


@pytest.mark.parametrize("fail_rank,fail_at", [(-1, 0), (1, 1), (0, 2)])
def test_run_dist_keeps_collectives_matched_when_a_rank_fails(tmp_path, fail_rank, fail_at):
    port = _free_port()
    script = tmp_path / "worker.py"
    script.write_text(RUN_DIST_WORKER.format(root=ROOT, pkg=PKG_NAME))
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", str(port), str(fail_rank), str(fail_at)],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=120)      # a mismatched collective would hang here
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("ranks hung: collectives did not match after a local failure")
        assert p.returncode == 0, e
        outs.append([ln for ln in o.splitlines() if ln.startswith("RESULT")][0].split())
    expect_ok = fail_rank < 0
    for r, (_, rank, ok, agreed, err) in enumerate(outs):
        assert int(rank) == r
        assert int(ok) == int(expect_ok) and int(agreed) == int(expect_ok)      # every rank learns the same verdict
        assert (err == "err") == (r == fail_rank)


def test_roofline_fraction_is_priced_on_moved_bytes():
    """VERDICT r2 weak #2 / ADVICE r2: `roofline.frac` and every `*_pct_of_8tbs` field are fractions of the bytes the kernels
    really move; the reference's CSR byte model appears only as an "effective" rate.  Round 2's own numbers (one-byte column
    codes: 827.5 MB moved per SpMV, 1 036.6 MB in the CSR model, 137.8 us per launch, 4 121 it/s) must read 0.75, not 0.94."""
    sys.path.insert(0, ROOT)
    import bench
    n, nnz, V = 10_000_000, 69_720_000, 8
    moved = nnz * (V + 1) + (n + 1) * 4 + 2 * n * V
    csr = nnz * (V + 4) + (n + 1) * 4 + 2 * n * V
    it_moved = nnz * (V + 1) + (n + 1) * 4 + 10 * n * V
    it_csr = nnz * (V + 4) + (n + 1) * 4 + 14 * n * V
    f = bench.rate_fields(spmv_moved=moved, spmv_csr=csr, spmv_ms=0.1378, spmv_alone_ms=0.132, iter_moved=it_moved, iter_csr=it_csr,
                          it_s=4121.0, traffic=889.6e6, traffic_source="test", n_offsets=7)
    r = f["roofline"]
    assert r["moved_bytes_per_launch"] == 827_480_004 and r["effective_csr_bytes_per_launch"] == 1_036_640_004
    assert abs(r["frac"] - moved / 0.1378e-3 / 8e12) < 1e-12 and abs(r["frac"] - 0.7507) < 1e-3
    assert abs(r["achieved"] - r["frac"] * 8000.0) < 1e-9
    assert abs(r["traffic_ratio"] - 889.6e6 / moved) < 1e-12
    # no printed fraction of moved bytes may exceed the peak (the CSR-priced iteration figure of round 2 read 102.8 %)
    for key in ("spmv_pct_of_8tbs", "spmv_back_to_back_pct_of_8tbs", "cg_iter_pct_of_8tbs"):
        assert 0.0 < f[key] <= 100.0, (key, f[key])
    assert abs(f["cg_iter_pct_of_8tbs"] - 100.0 * it_moved * 4121.0 / 8e12) < 1e-9
    assert f["effective_csr"]["cg_iter_gbs"] > 8000.0          # the effective figure may, and is labelled as such
    assert "frac_moved" not in r and "frac" not in f["effective_csr"]
