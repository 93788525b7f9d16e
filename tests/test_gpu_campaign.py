"""A bounded sample of tools/parity_campaign.py under `pytest -m gpu` (the campaign was builder-run
evidence only): random batch shapes around every size threshold of the kernels, all levels, raw / zlib, with and without
dictionaries, every stream compared byte for byte with the oracle and round-tripped on the device.  Fixed seeds; the
runs are bounded to 30 s each (and to a batch count, whichever comes first)."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def campaign():
    spec = importlib.util.spec_from_file_location("ldh_parity_campaign", os.path.join(ROOT, "tools", "parity_campaign.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.campaign


def test_random_batches_match_the_oracle(campaign):
    batches, checked = campaign(30.0, False, max_batches=400, seed=31, verbose=False)
    print("%d batches, %d streams" % (batches, checked))
    assert batches >= 5 and checked >= batches


def test_random_batches_in_whole_stream_mode_match_the_oracle(campaign):
    # every batch padded to 1024..1100 streams, ONE *_dev call: the whole-stream runs of k_seg_links / k_match_lazy
    batches, checked = campaign(30.0, True, max_batches=200, seed=32, verbose=False)
    print("%d batches, %d streams" % (batches, checked))
    assert batches >= 2 and checked >= batches
