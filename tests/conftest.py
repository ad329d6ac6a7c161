import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MODEL_ROOT = os.path.join(ROOT, "soundswallower_amd", "model")

# The library reads its SSW_* tuning knobs once, at ssw_model_load (csrc/ssw_host_model.inc,
# Knobs); the knob tests flip them between calls on one session-wide model, so they ask for the
# knobs to be re-read on every call.  Must be set before libssw_amd.so is loaded.
os.environ.setdefault("SSW_KNOBS_DYNAMIC", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _has_gpu():
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def orc_en(oracle_mod):
    return oracle_mod.Model(os.path.join(MODEL_ROOT, "en-us"))


@pytest.fixture(scope="session")
def orc_fr(oracle_mod):
    return oracle_mod.Model(os.path.join(MODEL_ROOT, "fr-fr"))


@pytest.fixture(scope="session")
def gpu_en():
    import soundswallower_amd as ssw
    from soundswallower_amd import _lib
    _lib.build()
    return ssw.Model(os.path.join(MODEL_ROOT, "en-us"))


@pytest.fixture(scope="session")
def gpu_fr():
    import soundswallower_amd as ssw
    from soundswallower_amd import _lib
    _lib.build()
    return ssw.Model(os.path.join(MODEL_ROOT, "fr-fr"))


from soundswallower_amd.synth import read_raw_means as raw_means  # noqa: E402


@pytest.fixture(scope="session")
def means_en():
    return raw_means(os.path.join(MODEL_ROOT, "en-us"))


@pytest.fixture(scope="session")
def means_fr():
    return raw_means(os.path.join(MODEL_ROOT, "fr-fr"))
