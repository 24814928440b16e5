"""the GPU-direct halo transport against real RCCL on the one GPU a test box has (see rccl_self_worker.py)"""
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_device_transport_moves_rows_through_rccl():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_self_worker.py"), str(port)], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    assert "all_reduce ok 3.0" in p.stdout and "RCCL_SELF_TRANSPORT OK" in p.stdout, p.stdout + p.stderr[-2000:]
