"""pytest configuration: the `gpu` marker and one-time builds of the native pieces.

* oracle/liboracle.so (CPU restatement, test infrastructure) is built with make;
* pogs_amd/libpogs_amd.so (the HIP engine) is cross-compiled with hipcc if stale;
* oracle/_ref/libpogs_cpu.so (the real reference) is built only where
  /root/reference exists (the build container).
"""
import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.dirname(__file__)):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    # the GPU tests hand torch device pointers to the library: torch has to bring its HIP runtime
    # first (pogs_amd/_lib.py: the package does not import torch on its own)
    os.environ.setdefault("POGS_AMD_TORCH_PRELOAD", "1")
    # the suite's one-GPU communicators (ranks as threads / processes) are a transport plug-in of the library,
    # tests/transport/test_transport.hip, built by pogs_amd/build.py
    os.environ.setdefault("POGS_AMD_TRANSPORT_PLUGIN", os.path.join(ROOT, "tests", "transport", "libpogs_test_transport.so"))
    import oracle_binding

    oracle_binding.build_oracle()
    oracle_binding.build_ref()
    if os.path.exists("/opt/rocm/bin/hipcc"):
        import importlib.util

        # by path: importing the package itself requires the built library
        spec = importlib.util.spec_from_file_location("_pogs_amd_build", os.path.join(ROOT, "pogs_amd", "build.py"))
        _build = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(_build)
        _build.build()


def has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_available():
    return has_gpu()
