"""The C library's gather (include/zamd_gpu.h zgpu_comm_*, zgpu_deflate_gather_sizes, zgpu_deflate_gather) with MORE THAN ONE rank.  RCCL refuses two
ranks on one device and this box has one GPU, so the ranks (separate processes, all on GPU 0) load tests/tools/fake_rccl.cpp in RCCL's place
(ZAMD_RCCL_LIB): what is tested is everything the library does around the nine RCCL calls -- which ranks send, where rank 0 receives, header, trailer,
the Adler-32 of the whole input, exact sizing -- against the stream ONE engine writes for the same input (mode B: the chunks are independent, so the
sharded stream is the same bytes).  The real RCCL is behind the world-1 test of tests/test_gpu_fullsize.py and behind bench.py --gpus N."""
import os
import subprocess
import sys
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import corpus_py as CP  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fake_rccl(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("fake") / "libfakerccl.so")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-fPIC", "-shared", "-w", "-o", so, os.path.join(ROOT, "tests", "tools", "fake_rccl.cpp"), "-lrt"], check=True, timeout=300)
    return so


@pytest.mark.parametrize("world,nbytes,level", [(2, 40 * 65536 - 777, 6), (3, 100 * 65536, 1), (4, 2 * 65536 + 5, 9), (3, 0, 6)])
def test_sharded_stream_is_the_single_engine_stream(fake_rccl, tmp_path, world, nbytes, level):
    import zlib_amd
    data = CP.chunks(CP.KIND_SILESIA, 300 + world, max(1, (nbytes + 65535) // 65536)).reshape(-1)[:nbytes]
    in_file, id_file, out_file = (str(tmp_path / n) for n in ("in.bin", "id.bin", "out.z"))
    data.tofile(in_file)
    env = dict(os.environ, ZAMD_RCCL_LIB=fake_rccl, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "tools", "comm_rank.py"), str(world), str(r), str(level), in_file, id_file, out_file],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        outs.append((p.returncode, o.decode(errors="replace")[-2000:]))
    assert all(rc == 0 for rc, _ in outs), outs
    z = open(out_file, "rb").read()
    e = zlib_amd.Engine(0)
    try:
        whole = e.deflate_host(data, level)  # F_FINAL | F_ZLIB_WRAP: one engine, the whole input
    finally:
        e.close()
    assert z == whole
    assert zlib.decompress(z) == data.tobytes()
    assert int(open(out_file + ".adler").read()) == zlib.adler32(data.tobytes())


def test_bench_flow_with_two_ranks_on_one_gpu(fake_rccl):
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one process per rank), rehearsed on one GPU: ZAMD_BENCH_SHARE_GPU puts both
    ranks on GPU 0 with a gloo group, the C library's gather runs over the test double.  The line must carry the whole job (both ranks' input) and the
    gathered stream's length; the value is no measurement."""
    import json
    env = dict(os.environ, ZAMD_RCCL_LIB=fake_rccl, ZAMD_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29517",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--gib", "0.125"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["scaling"] == "weak" and j["value"] > 0
    cfg = j["config"]
    assert cfg["gather"].startswith("zgpu_deflate_gather") and "rehearsal" in cfg
    assert cfg["gathered_bytes"] > cfg["stream_bytes"] + 6 and cfg["chunks_checked_against_reference_hashes"] > 0
