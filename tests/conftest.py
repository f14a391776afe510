import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'd-lsg-video-caption_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')


def pytest_collection_modifyitems(config, items):
    """gpu-marked tests are skipped (not errored) on a box without a GPU when somebody runs plain `pytest`."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason='needs a real MI355X (torch.cuda.is_available() is False)')
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)
